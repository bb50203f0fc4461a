// Wave-specialised forward / input-gradient GEMM on pre-split weights (round 4): the product kernel of the h3 stacks.
//
// Why: the one-role kernel (pw_gemm_b3p_kernel) runs main loop and epilogue on the same waves.  gfx950 has ONE in-order vmcnt for
// loads and stores, so a wave that has issued its output stores waits for them at its next load; every resident workgroup
// reaches its epilogue in the same phase (they share the matrix pipe), the chip alternates between an MFMA-bound phase with little
// HBM traffic and an HBM-write phase with idle matrix cores, and 1600 tiles over 1024 slots take two such rounds: K1 30 us against
// 13 us of bytes and 10-13 us of MFMA (profiles/README.md, r04_a).  Storing straight from the accumulator layout (no LDS
// transposition, no barrier) did not change that (30.3 vs 29.9 us), nor did looping over column tiles in the same waves.
//
// Here a 512-thread workgroup has two roles and loops over tiles (persistent, 2 workgroups per CU):
//   waves 0-3 "MFMA": weight fragments global -> registers (pre-split pieces in MFMA operand order, as before), activation
//       fragments from LDS (ds_read_b64_tr_b16), 12 MFMAs per 32-deep k-tile, and at the end of a tile the finished fp32 values
//       (h3: (acc + 2^-11 acc2) 2^-(ew + ex)) go to a per-wave LDS patch.  These waves never store to global memory: their
//       vmcnt waits see weight-fragment loads only.
//   waves 4-7 "IO": activation tile global -> registers (four k-tiles ahead) -> PReLU+gLN prologue -> fp16 pieces -> LDS; and while
//       the MFMA waves multiply tile t, they drain tile t-1's patches: LDS -> float4, residual / PReLU statistics / gLN-backward
//       sums / max |out| on the float4s, 16-byte global stores.  The stores of a tile are spread over the next tile's main loop.
// One s_barrier per k-tile orders both hand-offs (stage kt & 1 written before it, read after it; a patch is written behind the
// last k-tile's barrier of tile t and drained before the last k-tile's barrier of tile t + 1).
// Tiles are dealt round-robin (tile = L + i G, L = XCD-contiguous logical workgroup id): the row tiles of one column tile run at
// the same time on the same XCD, so the activation re-reads hit its L2.  Per-utterance constants (gLN statistics of the operand
// prologue, h3 scales) are computed once per workgroup into an LDS table.
// Same values as pw_gemm_b3p_kernel for the GEMM itself (same MFMA order per accumulator); the statistics partials are summed in
// a different (fixed) order.  Constraints (host falls back to pw_gemm_b3p_kernel otherwise): Cn = 256 or 512 (the k loop is
// unrolled so that every wait count is static), M <= WS_MAXM.
// Included by ctn_gemm.hip behind ctn_gemm_b3.h.
#pragma once
#include "ctn_gemm_b3.h"

namespace {

constexpr int WS_TM = 128, WS_TN = 64, WS_NTH = 512, WS_MAXM = 64;
constexpr int WS_PB = WS_TN + 32;                 // pitch of a channel-major plane row (bf16), as B3P
constexpr int WS_LST = WS_TN + 4;                 // pitch of a patch row (floats), as Tile::LDS_ST
constexpr int WS_PASSES = 8;                      // drain passes per tile: 4 rows x 16 float4 per wave and pass

constexpr int WS_D = 4;                           // k-tiles of activation loads in flight per IO thread (register ring)
constexpr int WS_INVALID = 0x7fffffff;            // per-lane buffer offset that fails the range check: loads 0, stores dropped
// `off` for a tile index t >= 0, WS_INVALID for t < 0 -- as arithmetic on the sign bits: a select on a uniform condition becomes
// a branch around the load, and the join of the two paths costs an s_waitcnt vmcnt(0)
__device__ __forceinline__ int ws_off(int off, int t) { return (int)((unsigned)off | ((unsigned)(t >> 31) & 0x7fffffffu)); }

// diagnostic builds (-DCTN_EXP_B3_TIMELINE, benchmarks/ws_timeline.py): cycles each role spends in total / waiting at barriers
#ifdef CTN_EXP_B3_TIMELINE
#define WS_BARRIER() do { const unsigned long long b0_ = __builtin_amdgcn_s_memtime(); __syncthreads(); tl_wait += __builtin_amdgcn_s_memtime() - b0_; } while (0)
#define WS_TL(x) x
#else
#define WS_BARRIER() __syncthreads()
#define WS_TL(x)
#endif

template <int AR, int PRO, int EPI, int NK>
__global__ __launch_bounds__(WS_NTH, 4) void pw_gemm_ws_kernel(PwArgs a) {
    constexpr int NP = Ar<AR>::NP;
    constexpr int TM = WS_TM, TN = WS_TN, PB = WS_PB, LST = WS_LST;
    constexpr int STAGE_ELEMS = NP * XK * PB;
    constexpr int B_L = XK * TN / 4 / 256;                                 // float4 loads per IO thread and k-tile (2)
    constexpr int CN = NK * XK;                                            // contraction length
    static_assert(NK % WS_D == 0 && NK >= 8, "the k loop is unrolled; ring slots and stages are compile-time");
    __shared__ __attribute__((aligned(16))) __bf16 Bp[2 * STAGE_ELEMS];    // [stage][piece][XK][PB]
    __shared__ __attribute__((aligned(16))) float patch[4 * 32 * LST];     // [MFMA wave][32 rows][LST]
    __shared__ float tab_mean[WS_MAXM], tab_rstd[WS_MAXM];
    __shared__ int tab_ex[WS_MAXM];
    __shared__ float2 pro_gb[PRO == PRO_PRELU_NORM ? CN : 1];              // (gamma, beta) of the operand prologue
    __shared__ double red[8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    const int L = xcd_remap(blockIdx.x, G);
    const int tiles_m = a.tiles_r * a.tiles_c, T = tiles_m * a.M;
    const int Rp = (a.R + 31) / 32 * 32;
    WS_TL(unsigned long long tl_wait = 0; unsigned long long tl_part = 0; int tl_tiles = 0;
          const unsigned long long tl_t0 = __builtin_amdgcn_s_memtime(); const unsigned long long tl_r0 = __builtin_amdgcn_s_memrealtime();)
    // full rounds are dealt by the XCD-contiguous logical id; the tiles of the last, partial round go to raw block ids (the
    // dispatcher deals consecutive ids to different CUs, so the extra tiles land on different CUs)
    const int full = T / G * G;
    const int my_tail = full + (int)blockIdx.x < T ? full + (int)blockIdx.x : -1;
    auto tile_at = [&](int i) -> int {                                    // i-th tile of this workgroup, -1 past the end
        const int t = L + i * G;
        if (t < full) return t;
        return t - L == full ? my_tail : -1;
    };

    // ---- per-utterance table: wave w finalises utterances w, w + 8, ... (fixed summation order) ----
    float p_alpha = 0.f;
    if constexpr (PRO == PRO_PRELU_NORM) {
        p_alpha = a.pro_alpha[0];
        for (int i = tid; i < CN; i += WS_NTH) pro_gb[i] = make_float2(a.pro_gamma[i], a.pro_beta[i]);
    }
    for (int m = wave; m < a.M; m += 8) {
        float mean = 0.f, rstd = 1.f;
        if constexpr (PRO == PRO_PRELU_NORM) {
            const double* __restrict__ part = a.pro_part + (size_t)m * a.pro_nparts * 2;
            double s = 0.0, q = 0.0;
            for (int i = lane; i < a.pro_nparts; i += 64) { s += part[2 * i]; q += part[2 * i + 1]; }
            s = wave_sum(s);
            q = wave_sum(q);
            const double count = (double)a.Cn * (double)a.K, mu = s / count;
            double var = q / count - mu * mu;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            rstd = (float)(1.0 / sqrt(var + (double)CTN_EPS));
            if (a.pro_ms_out != nullptr && blockIdx.x == 0 && lane == 0) {
                a.pro_ms_out[2 * m] = mean;
                a.pro_ms_out[2 * m + 1] = rstd;
            }
        }
        int ex = 0;
        if constexpr (Ar<AR>::F16) {
            const float xm = amax_read(a.x_amax + (size_t)m * CTN_AMAX_SLOTS);
            ex = h3_exp(PRO == PRO_PRELU_NORM ? h3_pro_bound(xm, p_alpha, mean, rstd, a.pro_gbmax) : xm);
        }
        if (lane == 0) { tab_mean[m] = mean; tab_rstd[m] = rstd; tab_ex[m] = ex; }
    }
    __syncthreads();
    WS_TL(const unsigned long long tl_t1 = __builtin_amdgcn_s_memtime();)

    if (wave < 4) {
        // =========================================== MFMA role ===========================================
        const int wm = wave;
        const int l31 = lane & 31, lhi = lane >> 5;
        int ew = 0;
        if constexpr (Ar<AR>::F16)
            ew = h3_exp(__uint_as_float(*reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(a.W) + (size_t)Rp * CN * (2 * NP))));
        const __amdgpu_buffer_rsrc_t rsW = make_rsrc(a.W, (unsigned)Rp * (unsigned)CN * (2u * NP));
        auto load_a = [&](int voA, int kt, bf16x8 (&fa)[2][NP]) {          // [k step][piece]
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
#ifdef CTN_EXP_B3_NOA
                for (int p = 0; p < NP; ++p) { float4 z = make_float4((float)lane, 1.f, 2.f, (float)(kt + voA)); fa[ks][p] = __builtin_bit_cast(bf16x8, z); }
#else
                for (int p = 0; p < NP; ++p) fa[ks][p] = buf_ld_frag(rsW, voA + kt * (2 * NP * 1024), (ks * NP + p) * 1024);
#endif
        };
        f32x16 acc[2], acc2[Ar<AR>::W2 ? 2 : 1];
        // One k-tile: four groups g = (k step ks = g / 2, column half j = g % 2) of NP activation fragments and Prods<NP>::N MFMAs each.
        // This role has ONE wave per SIMD and workgroup, so the LDS latency is hidden by issue order, not by other waves: the reads
        // of two groups are in flight before the first MFMA, each further group's follow the MFMAs of the group before it (three
        // groups in flight cost 5 spilled registers at the 128-register budget of two workgroups per CU).
        auto compute = [&](int stage, const bf16x8 (&fa)[2][NP]) {
            const __bf16* const S = Bp + stage * STAGE_ELEMS;
            bf16x8 bq[4][NP];
            auto rd = [&](int g) {
#pragma unroll
#ifdef CTN_EXP_B3_NOLDSRD
                for (int p = 0; p < NP; ++p) { float4 z = make_float4((float)lane, 1.f, (float)stage, (float)g); bq[g][p] = __builtin_bit_cast(bf16x8, z); }
#else
                for (int p = 0; p < NP; ++p) bq[g][p] = frag_tr(S + (p * XK + (g >> 1) * 16) * PB + (g & 1) * 32, PB, lane);
#endif
            };
#ifdef CTN_EXP_B3_NOMFMA
            auto mm = [&](int g) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const float4 x = __builtin_bit_cast(float4, fa[g >> 1][p]), y = __builtin_bit_cast(float4, bq[g][p]);
                    acc[g & 1][p] += x.x * y.x; acc[g & 1][p + 4] += x.y * y.y; acc[g & 1][p + 8] += x.z * y.z; acc[g & 1][p + 12] += x.w * y.w;
                }
            };
#else
            auto mm = [&](int g) { mfma_pieces<AR>(acc[g & 1], acc2[Ar<AR>::W2 ? (g & 1) : 0], fa[g >> 1], bq[g]); };
#endif
            rd(0); rd(1);
            __builtin_amdgcn_sched_barrier(0);
            mm(0);
            __builtin_amdgcn_sched_barrier(0);
            rd(2);
            __builtin_amdgcn_sched_barrier(0);
            mm(1);
            __builtin_amdgcn_sched_barrier(0);
            rd(3);
            __builtin_amdgcn_sched_barrier(0);
            mm(2); mm(3);
        };
        bf16x8 fa0[2][NP], fa1[2][NP];
        int t = tile_at(0);
        int voA = ws_off(((t < 0 ? 0 : t % a.tiles_r) * TM / 32 + wm) * (CN / 16) * (NP * 1024) + lane * 16, t);
        load_a(voA, 0, fa0);
        __builtin_amdgcn_sched_barrier(0);          // (k order: the first k-tile's fragments must not wait behind the second's)
        load_a(voA, 1, fa1);
        __builtin_amdgcn_sched_barrier(0);
        float* const my_patch = patch + wm * 32 * LST;
        for (int i = 0; t >= 0; ++i) {
            const int m = t / tiles_m;
            // the next tile's first weight fragments are issued behind the last two k-tiles of this one (no load gap at the seam)
            const int tn = tile_at(i + 1);
            const int voN = ws_off(((tn < 0 ? 0 : tn % a.tiles_r) * TM / 32 + wm) * (CN / 16) * (NP * 1024) + lane * 16, tn);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc[j][e] = 0.f; acc2[Ar<AR>::W2 ? j : 0][e] = 0.f; }
#pragma unroll
            for (int kt = 0; kt < NK; kt += 2) {
                WS_BARRIER();                                   // stage 0 holds k-tile kt
                compute(0, fa0);
                if (kt + 2 < NK) load_a(voA, kt + 2, fa0);
                else load_a(voN, 0, fa0);
                WS_BARRIER();                                   // stage 1 holds k-tile kt + 1
                compute(1, fa1);
                if (kt + 3 < NK) load_a(voA, kt + 3, fa1);
                else load_a(voN, 1, fa1);
            }
            voA = voN;
            const int h3_t = -(ew + tab_ex[m]);
            WS_TL(const unsigned long long tl_p0 = __builtin_amdgcn_s_memtime(); ++tl_tiles;)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[j][e];
                    if constexpr (Ar<AR>::W2) v = fmaf(acc2[Ar<AR>::W2 ? j : 0][e], 1.f / H3_LOW, v);
                    if constexpr (Ar<AR>::F16) v = __builtin_amdgcn_ldexpf(v, h3_t);
                    my_patch[((e & 3) + 8 * (e >> 2) + 4 * lhi) * LST + j * 32 + l31] = v;
                }
            // The patch must be IN LDS before this wave arrives at the next barrier (the IO waves read it behind that barrier).
            // hipcc (ROCm 7.2) emits no wait for the release fence here -- the barrier sits in the header of the tile loop:
            // ds_write_b32 ; s_cbranch ; s_barrier in the listing -- so it is written out.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            WS_TL(tl_part += __builtin_amdgcn_s_memtime() - tl_p0;)
            t = tn;
        }
        WS_BARRIER();        // the last patch is complete
        WS_BARRIER();        // (the IO waves' last partial sums)
        WS_TL(if (tid == 0 && blockIdx.x < 8192) {
            unsigned long long* d = ctn_dbg_tl + blockIdx.x * 12;
            d[0] = __builtin_amdgcn_s_memtime() - tl_t0; d[1] = tl_wait; d[2] = tl_part; d[7] = tl_tiles; d[8] = tl_t1 - tl_t0;
        })
    } else {
        // ============================================ IO role ============================================
        // One instruction stream per k-tile whatever the state (nothing to drain yet, no tile left to load: the per-lane offset is
        // pushed out of range instead of branching), so that the compiler's vmcnt counts are exact: a wait for the activation
        // loads of k-tile kt + 4 leaves every younger store in flight.
        const int it = tid - 256, iw = wave - 4;
        const int bi = it / (TN / 4), bk = (it % (TN / 4)) * 4;          // B tile: channel row bi (+ 16 j), frames bk .. + 3
        const int rl0 = lane >> 4, cl = (lane & 15) * 4;                  // drain: row rl0 (+ 4 p) of this wave's patch, columns cl .. + 3
        float e_alpha = 0.f;
        if constexpr (EPI == EPI_PRELU_STATS) e_alpha = a.epi_alpha[0];
        if constexpr (EPI == EPI_GLN_BWD) e_alpha = a.bwd_alpha[0];
        const unsigned xbytes = (unsigned)a.Cn * (unsigned)a.Kp * 4u, obytes = (unsigned)a.R * (unsigned)a.Kp * 4u;
        const int sB = XK * a.Kp * 4;

        // cursors over this workgroup's (tile, k-tile) sequence: `st` = the k-tile being split into LDS, `ld` = WS_D ahead (loads)
        struct Cur { int i, t, m, c0; };
        auto open = [&](Cur& c, int i) {
            c.i = i; c.t = tile_at(i);
            c.m = c.t >= 0 ? c.t / tiles_m : 0;
            c.c0 = c.t >= 0 ? ((c.t % tiles_m) / a.tiles_r) * TN : 0;
        };
        auto load_b = [&](const Cur& c, int kt, float4 (&rb)[B_L]) {
            const __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.X + (size_t)c.m * a.Cn * a.Kp, xbytes);
#pragma unroll
            for (int j = 0; j < B_L; ++j)
#ifdef CTN_EXP_B3_NOBLOAD
                rb[j] = make_float4((float)(kt + c.c0), 1.f, (float)lane, 2.f);
#else
                rb[j] = buf_ld4(rsX, ws_off(((bi + 16 * j) * a.Kp + c.c0 + bk) * 4 + kt * sB, c.t), 0);
#endif
        };
        float sx = 1.f, p_mean = 0.f, p_rstd = 1.f;                     // constants of the tile being split
        auto store_b = [&](const Cur& c, int kt, const float4 (&rb)[B_L]) {
            __bf16* const S = Bp + (kt & 1) * STAGE_ELEMS;
#pragma unroll
            for (int j = 0; j < B_L; ++j) {
                float4 v = rb[j];
                if constexpr (PRO == PRO_PRELU_NORM) {
                    const float2 gb = pro_gb[kt * XK + bi + 16 * j];
                    if constexpr (Ar<AR>::F16) v = pro_apply(v, c.c0 + bk, a.K, gb.x * sx, gb.y * sx, p_alpha, p_mean, p_rstd);
                    else v = pro_apply(v, c.c0 + bk, a.K, gb.x, gb.y, p_alpha, p_mean, p_rstd);
                }
                bf16x4 q[NP];
#ifdef CTN_EXP_B3_NOSPLIT
                q[0] = __builtin_bit_cast(bf16x4, make_float2(v.x, v.y));
                for (int p = 1; p < NP; ++p) q[p] = __builtin_bit_cast(bf16x4, make_float2(v.z, v.w));
#else
                split_x4<AR, PRO == PRO_PRELU_NORM>(v, q, sx);
#endif
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(S + (p * XK + bi + 16 * j) * PB + bk) = q[p];
            }
        };

        // ---- drain state of the previous tile ----
        int d_t = -1, d_m = 0, d_vo = WS_INVALID;   // tile being drained (-1: none, every access out of range), utterance, offset of (row rl0, column cl)
        bool d_ok = false;
        float s1 = 0.f, s2 = 0.f, amax = 0.f;
        float b_mean = 0.f, b_rstd = 1.f;
        float4 aux[WS_PASSES];
        float gam[WS_PASSES];
        const float* const my_patch = patch + iw * 32 * LST;
        auto drain_open = [&](int t) {              // tile t's patches are complete: fetch its auxiliary operand (all passes in flight)
            d_t = t;
            d_m = t >= 0 ? t / tiles_m : 0;
            const int tt = t < 0 ? 0 : t;
            const int rt = tt % a.tiles_r, c0 = ((tt % tiles_m) / a.tiles_r) * TN;
            const int row = rt * TM + iw * 32 + rl0, col = c0 + cl;
            d_ok = col < a.Kp;
            d_vo = ws_off(d_ok ? (row * a.Kp + col) * 4 : WS_INVALID, t);
            d_ok = d_ok && t >= 0;
            s1 = s2 = amax = 0.f;
            if constexpr (EPI == EPI_RESIDUAL || EPI == EPI_GLN_BWD) {
                const float* base = (EPI == EPI_RESIDUAL ? a.residual : a.bwd_y) + (size_t)d_m * a.R * a.Kp;
                const __amdgpu_buffer_rsrc_t rsAux = make_rsrc(base, obytes);
#pragma unroll
                for (int p = 0; p < WS_PASSES; ++p) aux[p] = buf_ld4(rsAux, d_vo, p * 4 * a.Kp * 4);
            }
            if constexpr (EPI == EPI_GLN_BWD) {
                const __amdgpu_buffer_rsrc_t rsGam = make_rsrc(a.bwd_gamma, (unsigned)a.R * 4u);
#pragma unroll
                for (int p = 0; p < WS_PASSES; ++p) gam[p] = buf_ld1(rsGam, ws_off(row * 4, t), p * 16);
                b_mean = a.bwd_ms[2 * d_m];
                b_rstd = a.bwd_ms[2 * d_m + 1];
            }
        };
        auto drain_read = [&](int p) -> float4 {    // rows rl0 + 4 p of this wave's patch
            return *reinterpret_cast<const float4*>(my_patch + (p * 4 + rl0) * LST + cl);
        };
        auto drain_pass = [&](int p, float4 v) {
#ifdef CTN_EXP_B3_NOEPI
            if (a.K < 0) a.Out[tid] = v.x + v.y + v.z + v.w;      // never taken
            return;
#endif
            if constexpr (EPI == EPI_RESIDUAL) {
                const float4 q = aux[p];
                v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                const float w = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                amax = fmaxf(amax, d_ok ? w : 0.f);
            }
            if constexpr (EPI == EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if constexpr (EPI == EPI_PRELU_STATS) {         // (rows >= R: exact zeros)
                const float p0 = prelu_f(v.x, e_alpha), p1 = prelu_f(v.y, e_alpha), p2 = prelu_f(v.z, e_alpha), p3 = prelu_f(v.w, e_alpha);
                const float t1 = (p0 + p1) + (p2 + p3), t2 = (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
                s1 += d_ok ? t1 : 0.f;
                s2 += d_ok ? t2 : 0.f;
            }
            if constexpr (EPI == EPI_GLN_BWD) {
                const float4 y = aux[p];
                const float g = gam[p];
                const float t0 = g * v.x, t1 = g * v.y, t2 = g * v.z, t3 = g * v.w;
                const float x0 = (prelu_f(y.x, e_alpha) - b_mean) * b_rstd, x1 = (prelu_f(y.y, e_alpha) - b_mean) * b_rstd;
                const float x2 = (prelu_f(y.z, e_alpha) - b_mean) * b_rstd, x3 = (prelu_f(y.w, e_alpha) - b_mean) * b_rstd;
                const float u1 = (t0 + t1) + (t2 + t3), u2 = (t0 * x0 + t1 * x1) + (t2 * x2 + t3 * x3);
                s1 += d_ok ? u1 : 0.f;
                s2 += d_ok ? u2 : 0.f;
            }
            const __amdgpu_buffer_rsrc_t rsOut = make_rsrc(a.Out + (size_t)d_m * a.R * a.Kp, obytes);
            // The row offset goes into the per-lane offset, NOT into an SGPR soffset: hipcc (ROCm 7.2) takes a 16-byte buffer store
            // with a register soffset to need no wait state before a VALU write of its data registers and schedules one right
            // behind it (buffer_store_dwordx4 v[42:45] .. s21 offen ; v_mul_f32 v44, ..) -- on gfx950 the store then picks up the
            // NEW value in lanes 12-15 of every 16 when the CU is busy (16 of 8192 outputs of a tile, 2 workgroups per CU only;
            // benchmarks/ws_check.py).  With a zero soffset the compiler's hazard recogniser inserts the wait states.
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, f32x4v{v.x, v.y, v.z, v.w}), rsOut, d_vo + p * 4 * a.Kp * 4, 0, 0);
        };
        auto drain_close = [&]() {                  // this wave's sums -> LDS (read by one thread behind the next barrier)
            if constexpr (EPI == EPI_PRELU_STATS || EPI == EPI_GLN_BWD) {
                const double d1 = wave_sum((double)s1), d2 = wave_sum((double)s2);
                if (lane == 0) { red[2 * iw] = d1; red[2 * iw + 1] = d2; }
            }
            if constexpr (EPI == EPI_RESIDUAL) {
                const float w = wave_max(amax);
                if (lane == 0) reinterpret_cast<unsigned*>(red)[iw] = __float_as_uint(w);
            }
        };
        auto drain_publish = [&]() {                // behind the barrier that follows drain_close
            if (it != 0 || d_t < 0) return;
            const int rt = d_t % a.tiles_r, ct = (d_t % tiles_m) / a.tiles_r;
            if constexpr (EPI == EPI_PRELU_STATS || EPI == EPI_GLN_BWD) {
                double* dst = (EPI == EPI_PRELU_STATS ? a.epi_part : a.bwd_part) + ((size_t)d_m * tiles_m + (size_t)ct * a.tiles_r + rt) * 2;
                dst[0] = ((red[0] + red[2]) + red[4]) + red[6];
                dst[1] = ((red[1] + red[3]) + red[5]) + red[7];
            }
            if constexpr (EPI == EPI_RESIDUAL) {
                if (a.out_amax != nullptr) {
                    const unsigned* r = reinterpret_cast<const unsigned*>(red);
                    unsigned b = r[0];
                    b = b > r[1] ? b : r[1]; b = b > r[2] ? b : r[2]; b = b > r[3] ? b : r[3];
                    if (b != 0u) atomicMax(a.out_amax + (size_t)d_m * CTN_AMAX_SLOTS + ((ct * a.tiles_r + rt) & (CTN_AMAX_SLOTS - 1)), b);
                }
            }
        };

        // drain schedule: WS_PASSES passes over the k-tiles 0 .. NK - 2 of the next tile (the patch is rewritten behind k-tile NK - 1)
        constexpr int SLOTS = NK - 1;
        Cur st, ld;
        open(st, 0);
        open(ld, 0);
        float4 rb[WS_D][B_L];
#pragma unroll
        for (int k = 0; k < WS_D; ++k) {                               // NK >= WS_D: all of the first tile
            load_b(ld, k, rb[k]);
            __builtin_amdgcn_sched_barrier(0);      // in k order: the scheduler put k-tile 0's loads LAST, and the loop header then waits for vmcnt(0)
        }
        if constexpr (Ar<AR>::F16) sx = h3_pow2(tab_ex[st.m]);
        if constexpr (PRO == PRO_PRELU_NORM) { p_mean = tab_mean[st.m]; p_rstd = tab_rstd[st.m]; }
        // One k-tile of this role: [barrier] patch read(s) of the drain + the activation loads of k-tile kt + WS_D are issued, then the
        // split of k-tile kt + 1 (its registers arrived long ago) and its LDS writes, then the drain's arithmetic and store on the
        // patch data that has arrived meanwhile, [barrier].  One wave per SIMD: the order of issue hides the latencies.
        store_b(st, 0, rb[0]);                      // (before the first barrier: k-tile 0 of the first tile)
        while (st.t >= 0) {
            Cur nx;
            open(nx, st.i + 1);
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                if (kt == NK - 1) drain_close();
                WS_BARRIER();                       // stage kt & 1 holds k-tile kt
                if (kt == NK - 1) drain_publish();
                if (kt == 0) drain_open(st.i > 0 ? tile_at(st.i - 1) : -1);
                float4 pv[2];
#pragma unroll
                for (int p = 0; p < WS_PASSES; ++p)
                    if (p >= kt * WS_PASSES / SLOTS && p < (kt + 1) * WS_PASSES / SLOTS && kt < SLOTS) pv[p - kt * WS_PASSES / SLOTS] = drain_read(p);
                if (kt + WS_D < NK) load_b(st, kt + WS_D, rb[kt % WS_D]);
                else load_b(nx, kt + WS_D - NK, rb[kt % WS_D]);
                __builtin_amdgcn_sched_barrier(0);
                // the split of the next k-tile (of the next tile behind the last one; nothing behind the last tile's last)
                if (kt + 1 < NK) store_b(st, kt + 1, rb[(kt + 1) % WS_D]);
                else {
                    if constexpr (Ar<AR>::F16) sx = h3_pow2(tab_ex[nx.m]);
                    if constexpr (PRO == PRO_PRELU_NORM) { p_mean = tab_mean[nx.m]; p_rstd = tab_rstd[nx.m]; }
                    store_b(nx, 0, rb[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < WS_PASSES; ++p)
                    if (p >= kt * WS_PASSES / SLOTS && p < (kt + 1) * WS_PASSES / SLOTS && kt < SLOTS) drain_pass(p, pv[p - kt * WS_PASSES / SLOTS]);
            }
            st = nx;
        }
        // the last tile of this workgroup: its patches are complete behind the next barrier
        WS_BARRIER();
        drain_open(st.i > 0 ? tile_at(st.i - 1) : -1);
#pragma unroll
        for (int p = 0; p < WS_PASSES; ++p) drain_pass(p, drain_read(p));
        drain_close();
        WS_BARRIER();
        drain_publish();
        WS_TL(if (it == 0 && blockIdx.x < 8192) {
            unsigned long long* d = ctn_dbg_tl + blockIdx.x * 12;
            d[3] = __builtin_amdgcn_s_memtime() - tl_t0; d[4] = tl_wait; d[5] = tl_r0; d[6] = __builtin_amdgcn_s_memrealtime();
            d[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        })
    }
}

}  // namespace

static int g_ctn_b3_ws = 0;                 // 1: the h3 forward / input-gradient GEMMs run on pw_gemm_ws_kernel where it applies (ctn_tune("b3_ws", 0|1)); OFF by default:
                                            // measured 31-37 us alone against 29-35 for the one-role kernel, 11.18 vs 10.34 ms per step (profiles/README.md, r04_a)
static int g_ctn_b3_ws_blocks = 512;        // persistent workgroups per launch (two per CU)   (ctn_tune("b3_ws_blocks", n))

// true when the launch was taken (h3 on pre-split weights only)
static bool ctn_ws_launch(PwArgs& a, bool pro, bool residual, bool stats, bool relu, bool gln_bwd, hipStream_t st) {
    if (!g_ctn_b3_ws || (a.Cn != 256 && a.Cn != 512) || a.M > WS_MAXM || relu) return false;     // the k loop is unrolled: the two layer widths of the stacks
    if ((stats || gln_bwd) && g_ctn_b3_tile != 1) return false;            // the statistics partials are counted in 128 x 64 tiles
    a.tiles_r = ctn_cdiv(a.R, WS_TM);
    a.tiles_c = ctn_cdiv(a.Kp, WS_TN);
    const long long T = (long long)a.tiles_r * a.tiles_c * a.M;
    if (T >= (1ll << 30)) return false;
    const int G = T < g_ctn_b3_ws_blocks ? (int)T : g_ctn_b3_ws_blocks;
    const dim3 grid(G), block(WS_NTH);
#define CTN_WS_LAUNCH(NK)                                                                                                              \
    do {                                                                                                                               \
        if (gln_bwd) hipLaunchKernelGGL((pw_gemm_ws_kernel<H3AR, PRO_NONE, EPI_GLN_BWD, NK>), grid, block, 0, st, a);                  \
        else if (pro && residual) hipLaunchKernelGGL((pw_gemm_ws_kernel<H3AR, PRO_PRELU_NORM, EPI_RESIDUAL, NK>), grid, block, 0, st, a); \
        else if (pro) hipLaunchKernelGGL((pw_gemm_ws_kernel<H3AR, PRO_PRELU_NORM, EPI_NONE, NK>), grid, block, 0, st, a);              \
        else if (stats) hipLaunchKernelGGL((pw_gemm_ws_kernel<H3AR, PRO_NONE, EPI_PRELU_STATS, NK>), grid, block, 0, st, a);           \
        else if (residual) hipLaunchKernelGGL((pw_gemm_ws_kernel<H3AR, PRO_NONE, EPI_RESIDUAL, NK>), grid, block, 0, st, a);           \
        else hipLaunchKernelGGL((pw_gemm_ws_kernel<H3AR, PRO_NONE, EPI_NONE, NK>), grid, block, 0, st, a);                             \
    } while (0)
    if (a.Cn == 256) CTN_WS_LAUNCH(8);
    else CTN_WS_LAUNCH(16);
#undef CTN_WS_LAUNCH
    return true;
}
