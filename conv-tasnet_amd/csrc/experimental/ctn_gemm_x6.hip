// EXPERIMENTAL, not part of the default build (CTN_BUILD_X6=1 python -m ... _build): split-bf16 ("x6", "p6", "x6ws")
// forms of the 1x1-convolution GEMMs.  Round-1 conclusion (profiles/README.md): they do not beat the lean fp32-MFMA
// kernels at the paper shapes; kept for reference behind include/ctn_hip_experimental.h.
#include "../ctn_gemm_common.h"

namespace {

// ===========================================================================================
// Split-bf16 ("x6") GEMMs: fp32-accurate products on the bf16 matrix cores.
//
// Every fp32 operand is split exactly into three bf16 pieces  a = a1 + a2 + a3  (8 significand bits each,
// the subtractions are exact in fp32), and a.b is formed from the six piece-products whose weight is
// >= 2^-16:  a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1);  the dropped terms are <= 2^-24 |ab|, i.e. below
// fp32 rounding.  Each piece-product is exact in the fp32 accumulator (8x8-bit significands).  The leading term
// and the five small ones go to separate accumulators that are added once at the end.
// v_mfma_f32_32x32x16_bf16 retires 16 contraction steps in 32 cycles, v_mfma_f32_32x32x2_f32 2 steps in 64:
// six bf16 MFMAs replace eight fp32 MFMAs at a quarter of their cycles each -> 2.7x the fp32-MFMA rate.
//
// Weights arrive pre-split (ctn_split_bf16: [3][R][Cnp] bf16, contraction contiguous, Cnp = Cn padded to 32
// with zeros); activations are split while they are staged global -> LDS (after the optional PReLU+gLN
// prologue).  A fragments are 16-byte row reads; B fragments (contraction = channels, strided in memory) come
// out of ds_read_b64_tr_b16, the hardware transpose read, from channel-major LDS planes.
// ===========================================================================================
#ifndef CTN_XK
#define CTN_XK 32
#endif
constexpr int XK = CTN_XK;        // channels per k-tile (XK/16 MFMA steps of depth 16)
constexpr int XAT = XK / 8;       // threads per 16-byte-chunked weight row
#ifndef CTN_X6_PF
#define CTN_X6_PF 2
#endif
constexpr int X6_PF = CTN_X6_PF;
#ifdef CTN_X3      // energy experiment: two pieces per operand, three products (a1b1 + a1b2 + a2b1), error <= 2^-16 |ab|
constexpr int NPL = 2;
#else
constexpr int NPL = 3;
#endif
constexpr int XPA = XK + 8;       // A-plane row pitch in bf16 (80 B / 144 B: conflict-free ds_read_b128 over 16 rows)
constexpr int WXK = 32, WXPA = 40;   // weight-gradient kernel: 32 frames per k-tile

template <typename TL>
struct X6 {
    static constexpr int PB = TL::TN + 32;                          // B-plane row pitch in bf16 (TN*2 + 64 B)
    static constexpr int A_ELEMS = NPL * TL::TM * XPA;                // bf16 elements
    static constexpr int B_ELEMS = NPL * XK * PB;
    static constexpr int MAIN_BYTES = (A_ELEMS + B_ELEMS) * 2;
    static constexpr int STAGE_BYTES = TL::STAGE_FLOATS * 4;
    static constexpr int SMEM_BYTES = MAIN_BYTES > STAGE_BYTES ? MAIN_BYTES : STAGE_BYTES;
};

struct X6Args {
    PwArgs p;              // W unused
    const __bf16* Wp;      // [3][R][Cnp]
    int Cnp;
};

template <typename TL, int PRO, int EPI>
__global__ __launch_bounds__(TL::NTH, TL::NW == 8 ? 2 : ((TL::TM * TL::TN <= 4096) ? 4 : ((TL::TM * TL::TN <= 8192) ? 2 : 1)))
void pw_gemm_x6_kernel(X6Args xa) {
    const PwArgs& a = xa.p;
    constexpr int TM = TL::TM, TN = TL::TN, MT = TL::MT, NTL = TL::NTL, WM = TL::WM, WN = TL::WN;
    constexpr int PB = X6<TL>::PB;
    constexpr int NTH = TL::NTH;
    constexpr int A_L = TM * XAT / NTH;           // 16-byte loads per thread per plane (XAT threads per weight row)
    constexpr int B_L = XK * TN / 4 / NTH;        // float4 loads per thread (TN/4 threads per channel row)
    constexpr int AR = NTH / XAT, BR = NTH / (TN / 4);   // rows covered per pass
    static_assert(A_L >= 1 && B_L >= 1, "tile too small for the workgroup");
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[X6<TL>::SMEM_BYTES];
    __shared__ double red[NTH / 64];
    __bf16* const Ap = reinterpret_cast<__bf16*>(smem_raw);                  // [3][TM][XPA]
    __bf16* const Bp = Ap + X6<TL>::A_ELEMS;                                 // [3][XK][PB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / TL::WGN, wn = wave % TL::WGN;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int rt = bid % a.tiles_r; bid /= a.tiles_r;
    const int ct = bid % a.tiles_c;
    const int m = bid / a.tiles_c;
    const int r0 = rt * TM, c0 = ct * TN;
    const float* __restrict__ Xm = a.X + (size_t)m * a.Cn * a.Kp;

    float p_mean = 0.f, p_rstd = 1.f, p_alpha = 0.f;
    if constexpr (PRO == PRO_PRELU_NORM) {
        finalize_stats<NTH>(a.pro_part + (size_t)m * a.pro_nparts * 2, a.pro_nparts,
                           (double)a.Cn * (double)a.K, red, p_mean, p_rstd);
        p_alpha = a.pro_alpha[0];
        if (a.pro_ms_out != nullptr && rt == 0 && ct == 0 && tid == 0) {
            a.pro_ms_out[2 * m] = p_mean;
            a.pro_ms_out[2 * m + 1] = p_rstd;
        }
    }
    const int nk = xa.Cnp / XK;
    const size_t plane = (size_t)a.R * xa.Cnp;

    // loop-invariant per-thread source pointers / predicates; only the k-tile offset changes per iteration
    const __bf16* a_src[A_L];
    bool a_ok[A_L];
#pragma unroll
    for (int j = 0; j < A_L; ++j) {
        const int r = r0 + tid / XAT + AR * j;
        a_ok[j] = r < a.R;
        a_src[j] = xa.Wp + (size_t)(a_ok[j] ? r : 0) * xa.Cnp + (tid % XAT) * 8;
    }
    const float* b_src[B_L];
    int b_ch[B_L];
    const int b_k = c0 + (tid % (TN / 4)) * 4;
    const bool b_kok = b_k < a.Kp;
#pragma unroll
    for (int j = 0; j < B_L; ++j) {
        b_ch[j] = tid / (TN / 4) + BR * j;
        b_src[j] = Xm + (size_t)b_ch[j] * a.Kp + (b_kok ? b_k : 0);
    }
    const size_t b_step = (size_t)XK * a.Kp;
    auto load_regs = [&](int kt, uint4 (&ra)[3][A_L], float4 (&rb)[B_L], float2 (&rp)[B_L]) {
        const int kc = kt * XK;
#pragma unroll
        for (int j = 0; j < A_L; ++j) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (a_ok[j]) v = *reinterpret_cast<const uint4*>(a_src[j] + p * plane + kc);
                ra[p][j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < B_L; ++j) {
            const int i = kc + b_ch[j];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 gb = make_float2(0.f, 0.f);
            if (i < a.Cn && b_kok) {
                v = ld4(b_src[j] + (size_t)kt * b_step);
                if constexpr (PRO == PRO_PRELU_NORM) gb = make_float2(a.pro_gamma[i], a.pro_beta[i]);
            }
            rb[j] = v;
            rp[j] = gb;
        }
    };
    auto write_lds = [&](const uint4 (&ra)[3][A_L], const float4 (&rb)[B_L], const float2 (&rp)[B_L]) {
#pragma unroll
        for (int j = 0; j < A_L; ++j) {
            const int r = tid / XAT + AR * j, c = (tid % XAT) * 8;
#pragma unroll
            for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint4*>(Ap + (p * TM + r) * XPA + c) = ra[p][j];
        }
#pragma unroll
        for (int j = 0; j < B_L; ++j) {
            const int i = tid / (TN / 4) + BR * j, k = (tid % (TN / 4)) * 4;
            float4 v = rb[j];
            if constexpr (PRO == PRO_PRELU_NORM) v = pro_apply(v, c0 + k, a.K, rp[j].x, rp[j].y, p_alpha, p_mean, p_rstd);
            bf16x4 q1, q2, q3;
            split3x4(v, q1, q2, q3);
            *reinterpret_cast<bf16x4*>(Bp + (0 * XK + i) * PB + k) = q1;
            *reinterpret_cast<bf16x4*>(Bp + (1 * XK + i) * PB + k) = q2;
            if constexpr (NPL == 3) *reinterpret_cast<bf16x4*>(Bp + (2 * XK + i) * PB + k) = q3;
        }
    };

    f32x16 hi[MT][NTL], lo[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { hi[i][j][e] = 0.f; lo[i][j][e] = 0.f; }

    const int l31 = lane & 31, lhi = lane >> 5;
    // transpose-read lane map: 16-lane group g = lane>>4 covers frames 16*(g&1).. of the 32-frame MFMA tile and
    // channels 8*(g>>1)..; lane 4q+p of a group addresses channel row q, frames 4p..4p+3.
    const int tr_q = (lane >> 2) & 3, tr_f = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    typedef bf16x4 __attribute__((address_space(3))) * lds_b4;

    auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < XK / 16; ++ks) {
        bf16x8 af[MT][3], bfr[NTL][3];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                af[i][p] = *reinterpret_cast<const bf16x8*>(Ap + (p * TM + wm * WM + i * 32 + l31) * XPA + ks * 16 + lhi * 8);
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                const __bf16* base = Bp + (p * XK + ks * 16 + lhi * 8 + tr_q) * PB + wn * WN + j * 32 + tr_f;
                const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(base));
                const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(base + 4 * PB));
                bfr[j][p] = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                if constexpr (NPL == 3) {
                lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bfr[j][0], lo[i][j], 0, 0, 0);
                lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][1], lo[i][j], 0, 0, 0);
                lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][2], lo[i][j], 0, 0, 0);
                }
                lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][0], lo[i][j], 0, 0, 0);
                lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][1], lo[i][j], 0, 0, 0);
                hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][0], hi[i][j], 0, 0, 0);
            }
    }
    };

    // One LDS buffer, two barriers per k-tile; the global loads of the next tile(s) wait in registers.
    // X6_PF = 2 keeps two staging sets in flight (prefetch distance 2).
    uint4 pa[3][A_L];
    float4 pb[B_L];
    float2 pp[B_L];
    load_regs(0, pa, pb, pp);
    if constexpr (X6_PF == 1 || TL::NW == 8) {      // 8-wave tiles: one staging set keeps two workgroups per CU
        for (int kt = 0; kt < nk; ++kt) {
            write_lds(pa, pb, pp);
            __syncthreads();
            if (kt + 1 < nk) load_regs(kt + 1, pa, pb, pp);
            compute();
            __syncthreads();
        }
    } else {
        uint4 qa[3][A_L];
        float4 qb[B_L];
        float2 qp[B_L];
        if (nk > 1) load_regs(1, qa, qb, qp);
        for (int kt = 0; kt < nk; kt += 2) {
            write_lds(pa, pb, pp);
            __syncthreads();
            if (kt + 2 < nk) load_regs(kt + 2, pa, pb, pp);
            compute();
            __syncthreads();
            if (kt + 1 < nk) {
                write_lds(qa, qb, qp);
                __syncthreads();
                if (kt + 3 < nk) load_regs(kt + 3, qa, qb, qp);
                compute();
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) hi[i][j][e] += lo[i][j][e];
    gemm_epilogue<TL, EPI>(a, hi, reinterpret_cast<float*>(smem_raw), red, m, rt, ct);
}

// -------------------------------------------------------------------------------------------
// Wave-specialised split-bf16 GEMM ("x6ws"): 128x128 tile, 8 waves, two-stage LDS ring, one barrier per k-tile.
// Waves 0..3 (one per SIMD) are consumers: fragment reads + MFMAs only, a 64x64 sub-tile each (2x2 MFMA tiles:
// 12 fragment reads feed 24 MFMAs, half the LDS bytes per MFMA of a 32x32 wave tile).  Waves 4..7 are producers:
// global -> registers one k-tile ahead -> PReLU+gLN prologue -> exact 3-way bf16 split -> the other LDS stage.
// The conversion VALU work and the copy latency sit in different waves from the MFMA chains and co-issue with them.
// -------------------------------------------------------------------------------------------
#ifdef CTN_WS_STAMPS
// Diagnostic build only (-DCTN_WS_STAMPS): s_memtime brackets around the phases of the wave-specialised kernel.
__device__ unsigned long long ctn_ws_dbg[8192];
__device__ __forceinline__ unsigned long long ws_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define WS_T(var) const unsigned long long var = ws_stamp()
#define WS_ACC(sum, t1, t0) sum += (t1) - (t0)
#else
#define WS_T(var)
#define WS_ACC(sum, t1, t0)
#endif

struct X6WS {
    using TL = T128x128;                                            // consumer layout: 2x2 waves of 64x64
    static constexpr int TM = 128, TN = 128, NTHB = 512;
    static constexpr int PB = TN + 32;
    static constexpr int A_STAGE = 3 * TM * XPA, B_STAGE = 3 * XK * PB;   // bf16 elements
    static constexpr int STAGE = A_STAGE + B_STAGE;
    static constexpr int MAIN_BYTES = 2 * STAGE * 2;
    static constexpr int EPI_BYTES = TL::STAGE_FLOATS * 4;
    static constexpr int SMEM_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
};

template <int PRO, int EPI>
__global__ __launch_bounds__(X6WS::NTHB, 1) void pw_gemm_x6ws_kernel(X6Args xa) {
    static_assert(XK == 32, "x6ws is laid out for 32-channel k-tiles");
    using TL = X6WS::TL;
    const PwArgs& a = xa.p;
    constexpr int TM = X6WS::TM, TN = X6WS::TN, PB = X6WS::PB, NTHB = X6WS::NTHB;
    constexpr int NP = 256;                        // producer threads
    constexpr int A_L = TM * XAT / NP;             // 2: 16-byte loads per producer thread per plane
    constexpr int B_L = XK * TN / 4 / NP;          // 4: float4 loads per producer thread
    constexpr int AR = NP / XAT, BR = NP / (TN / 4);
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[X6WS::SMEM_BYTES];
    __shared__ double red[NTHB / 64];
    __bf16* const S = reinterpret_cast<__bf16*>(smem_raw);          // [2][A planes | B planes]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int rt = bid % a.tiles_r; bid /= a.tiles_r;
    const int ct = bid % a.tiles_c;
    const int m = bid / a.tiles_c;
    const int r0 = rt * TM, c0 = ct * TN;
    const float* __restrict__ Xm = a.X + (size_t)m * a.Cn * a.Kp;

    float p_mean = 0.f, p_rstd = 1.f, p_alpha = 0.f;
    if constexpr (PRO == PRO_PRELU_NORM) {
        finalize_stats<NTHB>(a.pro_part + (size_t)m * a.pro_nparts * 2, a.pro_nparts,
                             (double)a.Cn * (double)a.K, red, p_mean, p_rstd);
        p_alpha = a.pro_alpha[0];
        if (a.pro_ms_out != nullptr && rt == 0 && ct == 0 && tid == 0) {
            a.pro_ms_out[2 * m] = p_mean;
            a.pro_ms_out[2 * m + 1] = p_rstd;
        }
    }
    const int nk = xa.Cnp / XK;

    f32x16 hi[2][2], lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { hi[i][j][e] = 0.f; lo[i][j][e] = 0.f; }

    if (wave >= 4) {
        // ------------------------------ producers ------------------------------
        const int ptid = tid - 256;
        const size_t plane = (size_t)a.R * xa.Cnp;
        const __bf16* a_src[A_L];
        bool a_ok[A_L];
#pragma unroll
        for (int j = 0; j < A_L; ++j) {
            const int r = r0 + ptid / XAT + AR * j;
            a_ok[j] = r < a.R;
            a_src[j] = xa.Wp + (size_t)(a_ok[j] ? r : 0) * xa.Cnp + (ptid % XAT) * 8;
        }
        const int b_k = c0 + (ptid % (TN / 4)) * 4;
        const bool b_kok = b_k < a.Kp;
        const int b_ch0 = ptid / (TN / 4);
        const float* const b_col = Xm + (b_kok ? b_k : 0);
        struct Regs {
            uint4 ra[3][A_L];
            float4 rb[B_L];
            float2 rp[B_L];
        };
        auto load_regs = [&](int kt, Regs& g) {
            const int kc = kt * XK;
            // unconditional loads from clamped addresses (rows / channels / columns out of range are zeroed when the
            // tile is written to LDS): predicated loads would make hipcc wait for ALL outstanding loads at the first use
#pragma unroll
            for (int j = 0; j < A_L; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) g.ra[p][j] = *reinterpret_cast<const uint4*>(a_src[j] + p * plane + kc);
#pragma unroll
            for (int j = 0; j < B_L; ++j) {
                const int i = min(kc + b_ch0 + BR * j, a.Cn - 1);
                g.rb[j] = ld4(b_col + (size_t)i * a.Kp);
                if constexpr (PRO == PRO_PRELU_NORM) g.rp[j] = make_float2(a.pro_gamma[i], a.pro_beta[i]);
            }
        };
        auto write_lds = [&](int stage, int kt, const Regs& g) {
            __bf16* const Ap = S + stage * X6WS::STAGE;
            __bf16* const Bp = Ap + X6WS::A_STAGE;
#pragma unroll
            for (int j = 0; j < A_L; ++j) {
                const int r = ptid / XAT + AR * j, c = (ptid % XAT) * 8;
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    *reinterpret_cast<uint4*>(Ap + (p * TM + r) * XPA + c) = a_ok[j] ? g.ra[p][j] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int j = 0; j < B_L; ++j) {
                const int i = b_ch0 + BR * j, k = (ptid % (TN / 4)) * 4;
                float4 v = g.rb[j];
                if constexpr (PRO == PRO_PRELU_NORM) v = pro_apply(v, c0 + k, a.K, g.rp[j].x, g.rp[j].y, p_alpha, p_mean, p_rstd);
                if (!(b_kok && kt * XK + i < a.Cn)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                bf16x4 q1, q2, q3;
                split3x4(v, q1, q2, q3);
                *reinterpret_cast<bf16x4*>(Bp + (0 * XK + i) * PB + k) = q1;
                *reinterpret_cast<bf16x4*>(Bp + (1 * XK + i) * PB + k) = q2;
                *reinterpret_cast<bf16x4*>(Bp + (2 * XK + i) * PB + k) = q3;
            }
        };
        // Three register sets rotate: tile t lives in set t % 3, is written to LDS stage t & 1 during the consumers'
        // pass over tile t-1, and its set is refilled with tile t+3 right away -- every global load has three
        // k-tiles of MFMA time to land.
        Regs g0, g1, g2;
        const int last = nk - 1;        // loads past the end re-read the last tile (never written to LDS): no branches,
        load_regs(0, g0);               // so hipcc keeps counted vmcnt waits and the prefetch depth survives
        load_regs(min(1, last), g1);
        load_regs(min(2, last), g2);
        write_lds(0, 0, g0);
        load_regs(min(3, last), g0);
        __syncthreads();
        int kt = 0;
#ifdef CTN_WS_STAMPS
        unsigned long long p_write = 0, p_load = 0, p_bar = 0;
        const unsigned long long p_t0 = ws_stamp();
#endif
        for (; kt + 3 < nk; kt += 3) {
            WS_T(s0);
            write_lds((kt + 1) & 1, kt + 1, g1);
            WS_T(s1);
            load_regs(min(kt + 4, last), g1);
            WS_T(s2);
            __syncthreads();
            WS_T(s3);
            write_lds((kt + 2) & 1, kt + 2, g2);
            WS_T(s4);
            load_regs(min(kt + 5, last), g2);
            WS_T(s5);
            __syncthreads();
            WS_T(s6);
            write_lds((kt + 3) & 1, kt + 3, g0);
            WS_T(s7);
            load_regs(min(kt + 6, last), g0);
            WS_T(s8);
            __syncthreads();
            WS_T(s9);
            WS_ACC(p_write, s1, s0); WS_ACC(p_write, s4, s3); WS_ACC(p_write, s7, s6);
            WS_ACC(p_load, s2, s1); WS_ACC(p_load, s5, s4); WS_ACC(p_load, s8, s7);
            WS_ACC(p_bar, s3, s2); WS_ACC(p_bar, s6, s5); WS_ACC(p_bar, s9, s8);
        }
#ifdef CTN_WS_STAMPS
        if (tid == 256 && blockIdx.x < 512) {
            unsigned long long* d = ctn_ws_dbg + blockIdx.x * 16 + 8;
            d[0] = ws_stamp() - p_t0; d[1] = p_write; d[2] = p_load; d[3] = p_bar;
        }
#endif
        // tail: consumer passes kt .. nk-1 (one to three of them), tiles kt+1 and kt+2 if they exist
        if (kt + 1 < nk) write_lds((kt + 1) & 1, kt + 1, g1);
        __syncthreads();
        if (kt + 1 < nk) {
            if (kt + 2 < nk) write_lds((kt + 2) & 1, kt + 2, g2);
            __syncthreads();
        }
        if (kt + 2 < nk) __syncthreads();
    } else {
        // ------------------------------ consumers ------------------------------
        const int wm = wave >> 1, wn = wave & 1;
        const int l31 = lane & 31, lhi = lane >> 5;
        const int tr_q = (lane >> 2) & 3, tr_f = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
        typedef bf16x4 __attribute__((address_space(3))) * lds_b4;
#ifdef CTN_WS_STAMPS
        unsigned long long c_comp = 0, c_bar = 0;
        const unsigned long long c_t00 = ws_stamp();
#endif
        __syncthreads();
#ifdef CTN_WS_STAMPS
        const unsigned long long c_t0 = ws_stamp();
#endif
        for (int kt = 0; kt < nk; ++kt) {
            WS_T(c0);
            const __bf16* const Ap = S + (kt & 1) * X6WS::STAGE;
            const __bf16* const Bp = Ap + X6WS::A_STAGE;
#pragma unroll
            for (int ks = 0; ks < XK / 16; ++ks) {
                bf16x8 af[2][3], bfr[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        af[i][p] = *reinterpret_cast<const bf16x8*>(Ap + (p * TM + wm * 64 + i * 32 + l31) * XPA + ks * 16 + lhi * 8);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const __bf16* base = Bp + (p * XK + ks * 16 + lhi * 8 + tr_q) * PB + wn * 64 + j * 32 + tr_f;
                        const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(base));
                        const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(base + 4 * PB));
                        bfr[j][p] = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bfr[j][0], lo[i][j], 0, 0, 0);
                        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][1], lo[i][j], 0, 0, 0);
                        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][2], lo[i][j], 0, 0, 0);
                        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][0], lo[i][j], 0, 0, 0);
                        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][1], lo[i][j], 0, 0, 0);
                        hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][0], hi[i][j], 0, 0, 0);
                    }
            }
            WS_T(c1);
            __syncthreads();
            WS_T(c2);
            WS_ACC(c_comp, c1, c0); WS_ACC(c_bar, c2, c1);
        }
#ifdef CTN_WS_STAMPS
        if (tid == 0 && blockIdx.x < 512) {
            unsigned long long* d = ctn_ws_dbg + blockIdx.x * 16;
            d[0] = ws_stamp() - c_t0; d[1] = c_comp; d[2] = c_bar; d[3] = c_t0 - c_t00; d[4] = c_t00;
        }
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) hi[i][j][e] += lo[i][j][e];
    }
#ifdef CTN_WS_STAMPS
    const unsigned long long e_t0 = ws_stamp();
#endif
    gemm_epilogue<TL, EPI, NTHB>(a, hi, reinterpret_cast<float*>(smem_raw), red, m, rt, ct, wave < 4);
#ifdef CTN_WS_STAMPS
    if (tid == 0 && blockIdx.x < 512) { ctn_ws_dbg[blockIdx.x * 16 + 5] = ws_stamp() - e_t0; ctn_ws_dbg[blockIdx.x * 16 + 6] = ws_stamp(); }
#endif
}

// -------------------------------------------------------------------------------------------
// "p6": the split-bf16 GEMM with BOTH operands already split in HBM (weights by ctn_split_bf16, activations by the
// epilogue / elementwise kernel that produced them).  The main loop is then 16-byte copies global -> LDS plus
// fragment reads and six MFMAs per 16-deep step: no conversion VALU work between the MFMAs.
// -------------------------------------------------------------------------------------------
struct P6Args {
    PwArgs p;                 // X / W unused
    const __bf16* Wp;         // [3][R][Cnp]  (+ m * w_m_stride when the weights are per utterance)
    size_t w_plane_stride;    // elements between weight planes
    size_t w_m_stride;        // 0 or elements between utterances' weight sets
    const __bf16* Xp;         // [3][M, Cn, Kp]
    size_t x_plane_stride;
    int Cnp;
};

template <typename TL, int EPI>
__global__ __launch_bounds__(NT, (TL::TM * TL::TN <= 4096) ? 4 : ((TL::TM * TL::TN <= 8192) ? 2 : 1))
void pw_gemm_p6_kernel(P6Args xa) {
    const PwArgs& a = xa.p;
    constexpr int TM = TL::TM, TN = TL::TN, MT = TL::MT, NTL = TL::NTL, WM = TL::WM, WN = TL::WN;
    constexpr int PB = X6<TL>::PB;
    constexpr int A_L = TM * XAT / NT;            // 16-byte loads per thread per plane (XAT threads per weight row)
    constexpr int B_L = XK * (TN / 8) / NT;       // 16-byte loads per thread per plane (TN/8 threads per channel row)
    constexpr int AR = NT / XAT;
    constexpr int BT = TN / 8;                    // threads per channel row
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[X6<TL>::SMEM_BYTES];
    __shared__ double red[NT / 64];
    __bf16* const Ap = reinterpret_cast<__bf16*>(smem_raw);                  // [3][TM][XPA]
    __bf16* const Bp = Ap + X6<TL>::A_ELEMS;                                 // [3][XK][PB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / TL::WGN, wn = wave % TL::WGN;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int rt = bid % a.tiles_r; bid /= a.tiles_r;
    const int ct = bid % a.tiles_c;
    const int m = bid / a.tiles_c;
    const int r0 = rt * TM, c0 = ct * TN;
#ifdef CTN_SKIP_MAINLOOP
    const int nk = 0;      // diagnostic build: fixed cost of launch + prologue + epilogue only
#else
    const int nk = xa.Cnp / XK;
#endif

    const __bf16* a_src[A_L];
    bool a_ok[A_L];
#pragma unroll
    for (int j = 0; j < A_L; ++j) {
        const int r = r0 + tid / XAT + AR * j;
        a_ok[j] = r < a.R;
        a_src[j] = xa.Wp + (size_t)m * xa.w_m_stride + (size_t)(a_ok[j] ? r : 0) * xa.Cnp + (tid % XAT) * 8;
    }
    const int b_k = c0 + (tid % BT) * 8;
    const bool b_kok = b_k < a.Kp;
    int b_ch[B_L];
    const __bf16* b_src[B_L];
#pragma unroll
    for (int j = 0; j < B_L; ++j) {
        b_ch[j] = tid / BT + (NT / BT) * j;
        b_src[j] = xa.Xp + ((size_t)m * a.Cn + b_ch[j]) * a.Kp + (b_kok ? b_k : 0);
    }
    const size_t b_step = (size_t)XK * a.Kp;

    uint4 ra[3][A_L], rb[3][B_L];
    auto load_regs = [&](int kt) {
        const int kc = kt * XK;
#pragma unroll
        for (int j = 0; j < A_L; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (a_ok[j]) v = *reinterpret_cast<const uint4*>(a_src[j] + p * xa.w_plane_stride + kc);
                ra[p][j] = v;
            }
#pragma unroll
        for (int j = 0; j < B_L; ++j) {
            const bool ok = b_kok && (kc + b_ch[j] < a.Cn);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (ok) v = *reinterpret_cast<const uint4*>(b_src[j] + p * xa.x_plane_stride + (size_t)kt * b_step);
                rb[p][j] = v;
            }
        }
    };
    auto write_lds = [&]() {
#pragma unroll
        for (int j = 0; j < A_L; ++j) {
            const int r = tid / XAT + AR * j, c = (tid % XAT) * 8;
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(Ap + (p * TM + r) * XPA + c) = ra[p][j];
        }
#pragma unroll
        for (int j = 0; j < B_L; ++j) {
            const int i = tid / BT + (NT / BT) * j, k = (tid % BT) * 8;
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(Bp + (p * XK + i) * PB + k) = rb[p][j];
        }
    };

    f32x16 hi[MT][NTL], lo[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { hi[i][j][e] = 0.f; lo[i][j][e] = 0.f; }

    const int l31 = lane & 31, lhi = lane >> 5;
    const int tr_q = (lane >> 2) & 3, tr_f = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    typedef bf16x4 __attribute__((address_space(3))) * lds_b4;

    load_regs(0);
    for (int kt = 0; kt < nk; ++kt) {
        write_lds();
        __syncthreads();
        if (kt + 1 < nk) load_regs(kt + 1);
#pragma unroll
        for (int ks = 0; ks < XK / 16; ++ks) {
            bf16x8 af[MT][3], bfr[NTL][3];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    af[i][p] = *reinterpret_cast<const bf16x8*>(Ap + (p * TM + wm * WM + i * 32 + l31) * XPA + ks * 16 + lhi * 8);
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const __bf16* base = Bp + (p * XK + ks * 16 + lhi * 8 + tr_q) * PB + wn * WN + j * 32 + tr_f;
                    const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(base));
                    const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(base + 4 * PB));
                    bfr[j][p] = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bfr[j][0], lo[i][j], 0, 0, 0);
                    lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][1], lo[i][j], 0, 0, 0);
                    lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][2], lo[i][j], 0, 0, 0);
                    lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][0], lo[i][j], 0, 0, 0);
                    lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][1], lo[i][j], 0, 0, 0);
                    hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][0], hi[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) hi[i][j][e] += lo[i][j][e];
    gemm_epilogue<TL, EPI, TL::NTH, true>(a, hi, reinterpret_cast<float*>(smem_raw), red, m, rt, ct);
}

// X [n] fp32 -> planes [3][n] bf16 (n multiple of 4); stand-alone form of what the producers' epilogues emit
__global__ __launch_bounds__(NT) void split_act_kernel(const float* __restrict__ X, __bf16* __restrict__ P, long long n) {
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        bf16x4 q1, q2, q3;
        split3x4(ld4(X + 4 * i), q1, q2, q3);
        *reinterpret_cast<bf16x4*>(P + 4 * i) = q1;
        *reinterpret_cast<bf16x4*>(P + n + 4 * i) = q2;
        *reinterpret_cast<bf16x4*>(P + 2 * n + 4 * i) = q3;
    }
}

// W [rows, cols] fp32 -> planes [3][R][Cnp] bf16 with R x Cn = (transpose ? cols x rows : rows x cols); zero pad to Cnp
__global__ __launch_bounds__(NT) void split_bf16_kernel(const float* __restrict__ W, __bf16* __restrict__ P,
                                                        int cols, int transpose, int R, int Cn, int Cnp) {
    const long long n = (long long)R * Cnp;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const int r = (int)(i / Cnp), c = (int)(i % Cnp);
        float v = 0.f;
        if (c < Cn) v = transpose ? W[(size_t)c * cols + r] : W[(size_t)r * cols + c];
        const Bf3 q = split3(v);
        P[i] = q.a;
        P[n + i] = q.b;
        P[2 * n + i] = q.c;
    }
}

template <int PRO>
__global__ __launch_bounds__(NT) void pw_wgrad_x6_kernel(WgArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 Ap[NPL * BM * WXPA];
    __shared__ __attribute__((aligned(16))) __bf16 Bp[NPL * BN * WXPA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bid = blockIdx.x;
    const int rt = bid % a.tiles_r; bid /= a.tiles_r;
    const int ct = bid % a.tiles_c; bid /= a.tiles_c;
    const int sp = bid;
    const int m = sp / a.chunks_per_m, ch = sp % a.chunks_per_m;
    const int kb = ch * a.chunk;
    const int ke = min(kb + a.chunk, a.Kp);
    const int r0 = rt * BM, c0 = ct * BN;
    const float* __restrict__ Gm = a.dOut + (size_t)m * a.R * a.Kp;
    const float* __restrict__ Xm = a.X + (size_t)m * a.Cn * a.Kp;

    float p_mean = 0.f, p_rstd = 1.f, p_alpha = 0.f;
    if constexpr (PRO == PRO_PRELU_NORM) {
        p_mean = a.pro_ms[2 * m];
        p_rstd = a.pro_ms[2 * m + 1];
        p_alpha = a.pro_alpha[0];
    }
    // staging map: 8 threads per row (32 frames = 8 float4), 32 rows per pass, 4 passes for 128 rows
    float4 ra[4], rb[4];
    float2 rg[4];
    const int nk = (ke - kb + WXK - 1) / WXK;
    auto load_regs = [&](int kt) {
        const int k = kb + kt * WXK + (tid & 7) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (tid >> 3) + 32 * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + row < a.R && k < ke) v = ld4(Gm + (size_t)(r0 + row) * a.Kp + k);
            ra[j] = v;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c = c0 + row;
            if (c < a.Cn && k < ke) {
                x = ld4(Xm + (size_t)c * a.Kp + k);
                if constexpr (PRO == PRO_PRELU_NORM) rg[j] = make_float2(a.pro_gamma[c], a.pro_beta[c]);
            } else if constexpr (PRO == PRO_PRELU_NORM) {
                rg[j] = make_float2(0.f, 0.f);
            }
            rb[j] = x;
        }
    };
    auto write_one = [&](__bf16* P, int row, int kq, const float4& v) {
        bf16x4 q1, q2, q3;
        split3x4(v, q1, q2, q3);
        *reinterpret_cast<bf16x4*>(P + (0 * BM + row) * WXPA + kq) = q1;
        *reinterpret_cast<bf16x4*>(P + (1 * BM + row) * WXPA + kq) = q2;
        if constexpr (NPL == 3) *reinterpret_cast<bf16x4*>(P + (2 * BM + row) * WXPA + kq) = q3;
    };
    auto write_lds = [&](int kt) {
        const int kq = (tid & 7) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (tid >> 3) + 32 * j;
            write_one(Ap, row, kq, ra[j]);
            if constexpr (PRO == PRO_PRELU_NORM)
                rb[j] = pro_apply(rb[j], kb + kt * WXK + kq, a.K, rg[j].x, rg[j].y, p_alpha, p_mean, p_rstd);
            write_one(Bp, row, kq, rb[j]);
        }
    };

    f32x16 hi[2][2], lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { hi[i][j][e] = 0.f; lo[i][j][e] = 0.f; }

    const int l31 = lane & 31, lhi = lane >> 5;
    if (nk > 0) load_regs(0);
    for (int kt = 0; kt < nk; ++kt) {
        write_lds(kt);
        __syncthreads();
        if (kt + 1 < nk) load_regs(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2][3], bfr[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
                    af[i][p] = *reinterpret_cast<const bf16x8*>(Ap + (p * BM + wm * 64 + i * 32 + l31) * WXPA + ks * 16 + lhi * 8);
                    bfr[i][p] = *reinterpret_cast<const bf16x8*>(Bp + (p * BN + wn * 64 + i * 32 + l31) * WXPA + ks * 16 + lhi * 8);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (NPL == 3) {
                    lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bfr[j][0], lo[i][j], 0, 0, 0);
                    lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][1], lo[i][j], 0, 0, 0);
                    lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][2], lo[i][j], 0, 0, 0);
                    }
                    lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][0], lo[i][j], 0, 0, 0);
                    lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][1], lo[i][j], 0, 0, 0);
                    hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][0], hi[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    float* __restrict__ S = a.slab + (size_t)sp * a.R * a.Cn;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = r0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int c = c0 + wn * 64 + nt * 32 + l31;
                if (r < a.R && c < a.Cn) S[(size_t)r * a.Cn + c] = hi[mt][nt][e] + lo[mt][nt][e];
            }
        }
}

}  // namespace

extern "C" {
// ---- split-bf16 entry points ---------------------------------------------------------------------
int ctn_split_cols(int Cn) { return (Cn + XK - 1) / XK * XK; }

// planes: [3][R][ctn_split_cols(Cn)] bf16, (R, Cn) = transpose ? (cols, rows) : (rows, cols)
int ctn_split_bf16(const float* W, void* planes, int rows, int cols, int transpose, void* stream) {
    CTN_REQUIRE(W && planes && rows > 0 && cols > 0, "ctn_split_bf16: bad arguments");
    CTN_REQUIRE(aligned16(planes), "ctn_split_bf16: planes must be 16-byte aligned");
    const int R = transpose ? cols : rows, Cn = transpose ? rows : cols, Cnp = ctn_split_cols(Cn);
    long long nb = ctn_cdivll((long long)R * Cnp, NT);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)nb), dim3(NT), 0, (hipStream_t)stream, W, (__bf16*)planes, cols,
                       transpose, R, Cn, Cnp);
    CTN_CHECK_LAUNCH("ctn_split_bf16");
    return CTN_OK;
}

}  // extern "C"

template <typename TL>
static void launch_x6(const X6Args& xa, bool pro, bool residual, bool stats, bool relu, bool gln_bwd, hipStream_t st) {
    const PwArgs& a = xa.p;
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * a.M)), block(TL::NTH);
    if (gln_bwd) hipLaunchKernelGGL((pw_gemm_x6_kernel<TL, PRO_NONE, EPI_GLN_BWD>), grid, block, 0, st, xa);
    else if (pro) {
        if (residual) hipLaunchKernelGGL((pw_gemm_x6_kernel<TL, PRO_PRELU_NORM, EPI_RESIDUAL>), grid, block, 0, st, xa);
        else hipLaunchKernelGGL((pw_gemm_x6_kernel<TL, PRO_PRELU_NORM, EPI_NONE>), grid, block, 0, st, xa);
    } else if (stats) hipLaunchKernelGGL((pw_gemm_x6_kernel<TL, PRO_NONE, EPI_PRELU_STATS>), grid, block, 0, st, xa);
    else if (residual) hipLaunchKernelGGL((pw_gemm_x6_kernel<TL, PRO_NONE, EPI_RESIDUAL>), grid, block, 0, st, xa);
    else if (relu) hipLaunchKernelGGL((pw_gemm_x6_kernel<TL, PRO_NONE, EPI_RELU>), grid, block, 0, st, xa);
    else hipLaunchKernelGGL((pw_gemm_x6_kernel<TL, PRO_NONE, EPI_NONE>), grid, block, 0, st, xa);
}

static void launch_x6ws(const X6Args& xa, bool pro, bool residual, bool stats, bool relu, bool gln_bwd, hipStream_t st) {
    const PwArgs& a = xa.p;
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * a.M)), block(X6WS::NTHB);
    if (gln_bwd) hipLaunchKernelGGL((pw_gemm_x6ws_kernel<PRO_NONE, EPI_GLN_BWD>), grid, block, 0, st, xa);
    else if (pro) {
        if (residual) hipLaunchKernelGGL((pw_gemm_x6ws_kernel<PRO_PRELU_NORM, EPI_RESIDUAL>), grid, block, 0, st, xa);
        else hipLaunchKernelGGL((pw_gemm_x6ws_kernel<PRO_PRELU_NORM, EPI_NONE>), grid, block, 0, st, xa);
    } else if (stats) hipLaunchKernelGGL((pw_gemm_x6ws_kernel<PRO_NONE, EPI_PRELU_STATS>), grid, block, 0, st, xa);
    else if (residual) hipLaunchKernelGGL((pw_gemm_x6ws_kernel<PRO_NONE, EPI_RESIDUAL>), grid, block, 0, st, xa);
    else if (relu) hipLaunchKernelGGL((pw_gemm_x6ws_kernel<PRO_NONE, EPI_RELU>), grid, block, 0, st, xa);
    else hipLaunchKernelGGL((pw_gemm_x6ws_kernel<PRO_NONE, EPI_NONE>), grid, block, 0, st, xa);
}

static void dispatch_x6(X6Args& xa, bool pro, bool residual, bool stats, bool relu, bool gln_bwd, hipStream_t st) {
    PwArgs& a = xa.p;
    const int id = pick_tile(a.M, a.R, a.Kp);
    int tm, tn;
    tile_dims(id, &tm, &tn);
    a.tiles_r = ctn_cdiv(a.R, tm);
    a.tiles_c = ctn_cdiv(a.Kp, tn);
    switch (id) {
        case 10: launch_x6ws(xa, pro, residual, stats, relu, gln_bwd, st); break;
        case 8: launch_x6<T128x128w8>(xa, pro, residual, stats, relu, gln_bwd, st); break;
        case 9: launch_x6<T128x64w8>(xa, pro, residual, stats, relu, gln_bwd, st); break;
        case 0: case 7: launch_x6<T128x128>(xa, pro, residual, stats, relu, gln_bwd, st); break;
        case 1: case 4: case 6: launch_x6<T128x64>(xa, pro, residual, stats, relu, gln_bwd, st); break;
        case 2: launch_x6<T64x128>(xa, pro, residual, stats, relu, gln_bwd, st); break;
        default: launch_x6<T64x64>(xa, pro, residual, stats, relu, gln_bwd, st); break;
    }
}

#ifdef CTN_WS_STAMPS
extern "C" int ctn_ws_debug_read(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ctn_ws_dbg), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

extern "C" {

// same contract as ctn_pw_gemm, with the weights given as pre-split planes [3][R][ctn_split_cols(Cn)] (already
// oriented rows = output channels: use transpose=1 in ctn_split_bf16 for the input-gradient form).
int ctn_pw_gemm_x6(const void* Wp, const float* X, float* Out, int M, int R, int Cn, int K, int Kp,
                   const double* pro_part, int pro_nparts, const float* pro_gamma, const float* pro_beta,
                   const float* pro_alpha, float* pro_ms_out,
                   const float* residual, const float* epi_alpha, double* epi_part, int relu_out, void* stream) {
    int rc = check_common("ctn_pw_gemm_x6", (const float*)Wp, X, Out, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(!relu_out || !(residual || epi_part || pro_part), "ctn_pw_gemm_x6: relu_out only on the plain GEMM");
    CTN_REQUIRE(!(residual && epi_part), "ctn_pw_gemm_x6: residual and stats epilogues are exclusive");
    CTN_REQUIRE(!pro_part || (pro_gamma && pro_beta && pro_alpha && pro_nparts > 0), "ctn_pw_gemm_x6: incomplete prologue arguments");
    CTN_REQUIRE(!epi_part || epi_alpha, "ctn_pw_gemm_x6: stats epilogue needs alpha");
    CTN_REQUIRE(!residual || aligned16(residual), "ctn_pw_gemm_x6: residual must be 16-byte aligned");
    X6Args xa{};
    PwArgs& a = xa.p;
    a.store_f32 = 1;
    a.X = X; a.Out = Out; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.pro_part = pro_part; a.pro_nparts = pro_nparts; a.pro_gamma = pro_gamma; a.pro_beta = pro_beta;
    a.pro_alpha = pro_alpha; a.pro_ms_out = pro_ms_out;
    a.residual = residual; a.epi_alpha = epi_alpha; a.epi_part = epi_part;
    xa.Wp = (const __bf16*)Wp; xa.Cnp = ctn_split_cols(Cn);
    dispatch_x6(xa, pro_part != nullptr, residual != nullptr, epi_part != nullptr, relu_out != 0, false, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_pw_gemm_x6");
    return CTN_OK;
}

int ctn_pw_dgrad_gln_x6(const void* Wp, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                        const float* y, const float* gamma, const float* alpha, const float* ms, double* sums_part,
                        void* stream) {
    int rc = check_common("ctn_pw_dgrad_gln_x6", (const float*)Wp, dOut, dN, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(y && gamma && alpha && ms && sums_part && aligned16(y), "ctn_pw_dgrad_gln_x6: bad arguments");
    X6Args xa{};
    PwArgs& a = xa.p;
    a.store_f32 = 1;
    a.X = dOut; a.Out = dN; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.bwd_y = y; a.bwd_gamma = gamma; a.bwd_alpha = alpha; a.bwd_ms = ms; a.bwd_part = sums_part;
    xa.Wp = (const __bf16*)Wp; xa.Cnp = ctn_split_cols(Cn);
    dispatch_x6(xa, false, false, false, false, true, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_pw_dgrad_gln_x6");
    return CTN_OK;
}

static void wgrad_x6_plan(int M, int R, int Cn, int Kp, int* chunk, int* chunks_per_m) {
    const int tiles = ctn_cdiv(R, BM) * ctn_cdiv(Cn, BN);
    int cpm = ctn_cdiv(512, tiles * M);
    const int max_cpm = ctn_cdiv(Kp, 256);
    if (cpm > max_cpm) cpm = max_cpm;
    if (cpm < 1) cpm = 1;
    int c = ctn_cdiv(ctn_cdiv(Kp, cpm), WXK) * WXK;
    *chunk = c;
    *chunks_per_m = ctn_cdiv(Kp, c);
}

size_t ctn_pw_wgrad_x6_workspace(int M, int R, int Cn, int Kp) {
    int chunk, cpm;
    wgrad_x6_plan(M, R, Cn, Kp, &chunk, &cpm);
    return (size_t)M * cpm * R * Cn * sizeof(float);
}

int ctn_pw_wgrad_x6(const float* dOut, const float* X, float* dW, int M, int R, int Cn, int K, int Kp,
                    const float* pro_gamma, const float* pro_beta, const float* pro_alpha, const float* pro_ms,
                    void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common("ctn_pw_wgrad_x6", dW, X, (const float*)dOut, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(!pro_ms || (pro_gamma && pro_beta && pro_alpha), "ctn_pw_wgrad_x6: incomplete prologue arguments");
    WgArgs a{};
    a.dOut = dOut; a.X = X; a.slab = (float*)workspace; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.tiles_r = ctn_cdiv(R, BM); a.tiles_c = ctn_cdiv(Cn, BN);
    wgrad_x6_plan(M, R, Cn, Kp, &a.chunk, &a.chunks_per_m);
    const int nsplit = M * a.chunks_per_m;
    if (workspace == nullptr || workspace_bytes < (size_t)nsplit * R * Cn * sizeof(float)) {
        ctn_set_error("ctn_pw_wgrad_x6: workspace too small");
        return CTN_ERR_WORKSPACE;
    }
    a.pro_gamma = pro_gamma; a.pro_beta = pro_beta; a.pro_alpha = pro_alpha; a.pro_ms = pro_ms;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * nsplit)), block(NT);
    if (pro_ms) hipLaunchKernelGGL((pw_wgrad_x6_kernel<PRO_PRELU_NORM>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pw_wgrad_x6_kernel<PRO_NONE>), grid, block, 0, st, a);
    CTN_CHECK_LAUNCH("ctn_pw_wgrad_x6");
    const long long n = (long long)R * Cn;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ctn_cdivll(n / 4, NT)), block, 0, st, a.slab, nsplit, n, dW);
    CTN_CHECK_LAUNCH("ctn_pw_wgrad_x6/reduce");
    return CTN_OK;
}

// ---- p6: both operands pre-split --------------------------------------------------------------------
// X planes of a whole activation tensor: planes [3][n] bf16
int ctn_split_act(const float* X, void* planes, long long n, void* stream) {
    CTN_REQUIRE(X && planes && n > 0 && n % 4 == 0, "ctn_split_act: bad arguments");
    CTN_REQUIRE(aligned16(X) && aligned16(planes), "ctn_split_act: alignment");
    long long nb = ctn_cdivll(n / 4, NT);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(split_act_kernel, dim3((unsigned)nb), dim3(NT), 0, (hipStream_t)stream, X, (__bf16*)planes, n);
    CTN_CHECK_LAUNCH("ctn_split_act");
    return CTN_OK;
}

}  // extern "C"

template <typename TL>
static void launch_p6(const P6Args& xa, bool residual, bool stats, bool relu, bool gln_bwd, hipStream_t st) {
    const PwArgs& a = xa.p;
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * a.M)), block(NT);
    if (gln_bwd) hipLaunchKernelGGL((pw_gemm_p6_kernel<TL, EPI_GLN_BWD>), grid, block, 0, st, xa);
    else if (stats) hipLaunchKernelGGL((pw_gemm_p6_kernel<TL, EPI_PRELU_STATS>), grid, block, 0, st, xa);
    else if (residual) hipLaunchKernelGGL((pw_gemm_p6_kernel<TL, EPI_RESIDUAL>), grid, block, 0, st, xa);
    else if (relu) hipLaunchKernelGGL((pw_gemm_p6_kernel<TL, EPI_RELU>), grid, block, 0, st, xa);
    else hipLaunchKernelGGL((pw_gemm_p6_kernel<TL, EPI_NONE>), grid, block, 0, st, xa);
}

extern "C" {

// Out[m] = Wp(m) . Xp[m] (+ row_bias[m] for k < K) (+ residual[m]); result stored as fp32 (Out != NULL) and / or as
// three bf16 planes (out_planes != NULL).  Wp: [3][R][ctn_split_cols(Cn)] bf16, per utterance when w_per_m != 0
// (then [M][3][R][Cnp]).  Xp: [3][M,Cn,Kp] bf16.  y/gamma/alpha/ms/sums_part non-NULL selects the gLN-backward epilogue.
int ctn_pw_gemm_p6(const void* Wp, int w_per_m, const void* Xp, float* Out, void* out_planes, int M, int R, int Cn,
                   int K, int Kp, const float* row_bias, const float* residual, const float* epi_alpha, double* epi_part,
                   int relu_out, const float* bwd_y, const float* bwd_gamma, const float* bwd_alpha, const float* bwd_ms,
                   double* bwd_part, void* stream) {
    CTN_REQUIRE(Wp && Xp && (Out || out_planes), "ctn_pw_gemm_p6: null pointer");
    CTN_REQUIRE(M > 0 && R > 0 && Cn > 0 && K > 0 && Kp >= K && Kp % 8 == 0 && R % 4 == 0, "ctn_pw_gemm_p6: bad sizes");
    CTN_REQUIRE(aligned16(Wp) && aligned16(Xp) && aligned16(Out) && aligned16(out_planes) && aligned16(residual),
                "ctn_pw_gemm_p6: alignment");
    const bool gln = bwd_part != nullptr;
    CTN_REQUIRE(!gln || (bwd_y && bwd_gamma && bwd_alpha && bwd_ms), "ctn_pw_gemm_p6: incomplete gLN-backward arguments");
    CTN_REQUIRE((int)(residual != nullptr) + (int)(epi_part != nullptr) + (int)(relu_out != 0) + (int)gln <= 1,
                "ctn_pw_gemm_p6: at most one of residual / stats / relu / gln-backward epilogues");
    P6Args xa{};
    PwArgs& a = xa.p;
    a.Out = Out; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.residual = residual; a.epi_alpha = epi_alpha; a.epi_part = epi_part;
    a.bwd_y = bwd_y; a.bwd_gamma = bwd_gamma; a.bwd_alpha = bwd_alpha; a.bwd_ms = bwd_ms; a.bwd_part = bwd_part;
    a.row_bias = row_bias; a.out_planes = out_planes; a.out_plane_stride = (size_t)M * R * Kp; a.store_f32 = Out != nullptr;
    xa.Cnp = ctn_split_cols(Cn);
    xa.Wp = (const __bf16*)Wp; xa.w_plane_stride = (size_t)R * xa.Cnp; xa.w_m_stride = w_per_m ? 3 * xa.w_plane_stride : 0;
    xa.Xp = (const __bf16*)Xp; xa.x_plane_stride = (size_t)M * Cn * Kp;
    int id = pick_tile(M, R, Kp);
    if (id == 8 || id == 10) id = 0;          // the 8-wave tiles exist for the on-the-fly kernel only
    if (id == 9) id = 1;
    int tm, tn;
    tile_dims(id, &tm, &tn);
    a.tiles_r = ctn_cdiv(R, tm);
    a.tiles_c = ctn_cdiv(Kp, tn);
    hipStream_t st = (hipStream_t)stream;
    switch (id) {
        case 0: case 7: launch_p6<T128x128>(xa, residual != nullptr, epi_part != nullptr, relu_out != 0, gln, st); break;
        case 1: case 4: case 6: launch_p6<T128x64>(xa, residual != nullptr, epi_part != nullptr, relu_out != 0, gln, st); break;
        case 2: launch_p6<T64x128>(xa, residual != nullptr, epi_part != nullptr, relu_out != 0, gln, st); break;
        default: launch_p6<T64x64>(xa, residual != nullptr, epi_part != nullptr, relu_out != 0, gln, st); break;
    }
    CTN_CHECK_LAUNCH("ctn_pw_gemm_p6");
    return CTN_OK;
}

}  // extern "C"

