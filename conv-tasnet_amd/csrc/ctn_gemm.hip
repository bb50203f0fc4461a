// Pointwise (1x1) convolution GEMMs on fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces every nn.Conv1d(..., 1) of the reference (src/conv_tasnet.py:174,191,223,262)
// in forward, input-gradient and weight-gradient form.  Exact fp32 products and
// accumulation (bf16 inputs break the 1e-3 dB budget).
//
// Data layout: activations [M, Ch, Kp] fp32, frames fastest, Kp = K rounded up
// (multiple of 4); columns k in [K, Kp) hold exact zeros in every activation
// and gradient tensor (invariant kept by every kernel's store path).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each
// wave 64x64 = 2x2 MFMA tiles of 32x32, 64 accumulator VGPRs), contraction
// step 16 staged through LDS with register double-buffering.  One MFMA
// (64 cycles) consumes one A and one B dword per lane, so LDS bandwidth is
// never the bound; the fused prologue/epilogue work rides in the VALU shadow.
#include "ctn_gemm_common.h"
#include <string.h>

namespace {

template <typename TL, int TRANS_W, int PRO, int EPI>
__global__ __launch_bounds__(TL::NTH) void pw_gemm_kernel(PwArgs a) {
    constexpr int TM = TL::TM, TN = TL::TN, LDA = TL::LDA, LDB = TL::LDB, MT = TL::MT, NTL = TL::NTL;
    constexpr int WM = TL::WM, WN = TL::WN, BK = TL::TK;
    constexpr int NTH = TL::NTH;
    constexpr int A_L = TM * BK / 4 / NTH, B_L = TN * BK / 4 / NTH;    // float4 loads per thread per k-tile
    constexpr int AT = BK / 4;                                  // threads per weight row (TRANS_W = 0)
    __shared__ __attribute__((aligned(16))) float smem[TL::SMEM_FLOATS];
    __shared__ double red[NTH / 64];
    static_assert(A_L >= 1 && B_L >= 1, "k-tile too small for the workgroup");
    float* const As = smem;                       // [2][BK][LDA]
    float* const Bs = smem + 2 * BK * LDA;        // [2][BK][LDB]

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

    // ---- global -> register staging maps (float4 each), as buffer loads ------------------------
    // A (weights): TRANS_W=0 reads W[r][c..c+3]: c4 = tid % AT, r = tid / AT (+ (NTH/AT) j)
    //              TRANS_W=1 reads W[c][r..r+3]: r4 = tid % (TM/4), c = tid / (TM/4) (+ (4 NTH/TM) j)
    // B (activations): X[i][k..k+3]:            k4 = tid % (TN/4), i = tid / (TN/4) (+ (4 NTH/TN) j)
    // Out-of-range rows of W (TRANS_W=0) / channels of W^T and of X fall past the end of their buffer and read 0.
    // A tile that overhangs the contraction (Cn % BK != 0, TRANS_W=0) reads the next weight row instead: finite
    // values that meet all-zero activation rows (their gamma/beta read 0 too), so the products vanish; overhanging
    // output rows / columns are never stored.
    const int nk = (a.Cn + BK - 1) / BK;
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(a.W, (unsigned)a.R * (unsigned)a.Cn * 4u);
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(Xm, (unsigned)a.Cn * (unsigned)a.Kp * 4u);
    __amdgpu_buffer_rsrc_t rsG = rsX, rsBt = rsX;
    if constexpr (PRO == PRO_PRELU_NORM) {
        rsG = make_rsrc(a.pro_gamma, (unsigned)a.Cn * 4u);
        rsBt = make_rsrc(a.pro_beta, (unsigned)a.Cn * 4u);
    }
    int voA[A_L], voB[B_L], voP[B_L];
#pragma unroll
    for (int j = 0; j < A_L; ++j) {
        if constexpr (TRANS_W == 0) voA[j] = ((r0 + tid / AT + (NTH / AT) * j) * a.Cn + (tid % AT) * 4) * 4;
        else voA[j] = ((tid / (TM / 4) + (4 * NTH / TM) * j) * a.R + r0 + (tid % (TM / 4)) * 4) * 4;
    }
#pragma unroll
    for (int j = 0; j < B_L; ++j) {
        const int i = tid / (TN / 4) + (4 * NTH / TN) * j;
        voB[j] = (i * a.Kp + c0 + (tid % (TN / 4)) * 4) * 4;
        voP[j] = i * 4;
    }
    const int sA = (TRANS_W == 0 ? BK : BK * a.R) * 4, sB = BK * a.Kp * 4;      // scalar byte steps per k-tile

    // PRO: the raw tile and its (gamma, beta) stay in registers across the MFMA phase; the norm is applied when the
    // tile is written to LDS, so the global loads never have a consumer before the compute they overlap with.
    auto load_tile = [&](int kt, float4 (&ra)[A_L], float4 (&rb)[B_L], float2 (&rp)[B_L]) {
#pragma unroll
        for (int j = 0; j < A_L; ++j) ra[j] = buf_ld4(rsW, voA[j], kt * sA);
#pragma unroll
        for (int j = 0; j < B_L; ++j) {
            rb[j] = buf_ld4(rsX, voB[j], kt * sB);
            if constexpr (PRO == PRO_PRELU_NORM)
                rp[j] = make_float2(buf_ld1(rsG, voP[j], kt * BK * 4), buf_ld1(rsBt, voP[j], kt * BK * 4));
            else
                rp[j] = make_float2(0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf, const float4 (&ra)[A_L], const float4 (&rb)[B_L], const float2 (&rp)[B_L]) {
        float* const Ab = As + buf * BK * LDA;
        float* const Bb = Bs + buf * BK * LDB;
#pragma unroll
        for (int j = 0; j < A_L; ++j) {
            if constexpr (TRANS_W == 0) {
                const int r = tid / AT + (NTH / AT) * j, c = (tid % AT) * 4;
                Ab[(c + 0) * LDA + r] = ra[j].x;
                Ab[(c + 1) * LDA + r] = ra[j].y;
                Ab[(c + 2) * LDA + r] = ra[j].z;
                Ab[(c + 3) * LDA + r] = ra[j].w;
            } else {
                const int c = tid / (TM / 4) + (4 * NTH / TM) * j, r = (tid % (TM / 4)) * 4;
                *reinterpret_cast<float4*>(Ab + c * LDA + r) = ra[j];
            }
        }
#pragma unroll
        for (int j = 0; j < B_L; ++j) {
            const int i = tid / (TN / 4) + (4 * NTH / TN) * j, k = (tid % (TN / 4)) * 4;
            float4 v = rb[j];
            if constexpr (PRO == PRO_PRELU_NORM) v = pro_apply(v, c0 + k, a.K, rp[j].x, rp[j].y, p_alpha, p_mean, p_rstd);
            *reinterpret_cast<float4*>(Bb + i * LDB + k) = v;
        }
    };

    f32x16 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    auto compute = [&](int buf) {
        const float* const Ab = As + buf * BK * LDA + wm * WM + l31;
        const float* const Bb = Bs + buf * BK * LDB + wn * WN + l31;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int kk = 2 * s + lhi;
            float av[MT], bv[NTL];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = Ab[kk * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < NTL; ++j) bv[j] = Bb[kk * LDB + 32 * j];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }

    };

    // Software pipeline, prefetch distance 2: while tile kt is multiplied out of LDS, tile kt+1 sits in one register
    // set (written to the other LDS buffer after the MFMAs) and tile kt+2 is in flight into the second set -- the
    // global-load latency is covered by two k-tiles of MFMA work instead of one.
    float4 pa[A_L], pb[B_L], qa[A_L], qb[B_L];
    float2 pp[B_L], qp[B_L];
    load_tile(0, pa, pb, pp);
    store_tile(0, pa, pb, pp);
    if (nk > 1) load_tile(1, pa, pb, pp);
    if (nk > 2) load_tile(2, qa, qb, qp);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        compute(0);
        if (kt + 1 < nk) store_tile(1, pa, pb, pp);
        __syncthreads();
        if (kt + 3 < nk) load_tile(kt + 3, pa, pb, pp);
        if (kt + 1 < nk) {
            compute(1);
            if (kt + 2 < nk) store_tile(0, qa, qb, qp);
            __syncthreads();
            if (kt + 4 < nk) load_tile(kt + 4, qa, qb, qp);
        }
    }

    gemm_epilogue<TL, EPI>(a, acc, smem, red, m, rt, ct);
}


template <int PRO, int WTM, int WTN>     // WTM x WTN output tile (multiples of 64), waves 2x2, (WTM/64)*(WTN/64) accumulator chains per wave
__global__ __launch_bounds__(NT) void pw_wgrad_kernel(WgArgs a) {
    constexpr int HTM = WTM / 2, HTN = WTN / 2;             // wave tile
    constexpr int MTM = WTM / 64, MTN = WTN / 64;           // MFMA tiles per wave edge = float4 loads per thread and k-tile
    __shared__ float As[2][WTM][LDW];
    __shared__ float Bs[2][WTN][LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // the tiles of one split read the same frames of both operands: keep them on one XCD (one L2) -- without the
    // remap the fabric fetch of a launch was 205 MB against 79 MB of operands
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int rt = bid % a.tiles_r; bid /= a.tiles_r;
    const int ct = bid % a.tiles_c; bid /= a.tiles_c;
    const int sp = bid;
    const int m = sp / a.chunks_per_m, ch = sp % a.chunks_per_m;
    const int kb = ch * a.chunk;
    const int ke = min(kb + a.chunk, a.Kp);
    const int r0 = rt * WTM, c0 = ct * WTN;
    const float* __restrict__ Gm = a.dOut + (size_t)m * a.R * a.Kp;
    const float* __restrict__ Xm = a.X + (size_t)m * a.Cn * a.Kp;

    float p_mean = 0.f, p_rstd = 1.f, p_alpha = 0.f;
    if constexpr (PRO == PRO_PRELU_NORM) {
        p_mean = a.pro_ms[2 * m];
        p_rstd = a.pro_ms[2 * m + 1];
        p_alpha = a.pro_alpha[0];
    }

    float2 rg[MTN];
    const int nk = (ke - kb + WK - 1) / WK;
    // Buffer loads (see buf_ld4): rows past R / channels past Cn fall off the end of their per-utterance buffer and read
    // 0 (as do their gamma / beta); the frame offset of a k-tile is a scalar.  Only a k-tile that straddles the end
    // of the chunk (Kp not a multiple of 16: never with ctn_padded_frames) needs a per-lane mask, under a uniform branch.
    const __amdgpu_buffer_rsrc_t rsG = make_rsrc(Gm, (unsigned)a.R * (unsigned)a.Kp * 4u);
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(Xm, (unsigned)a.Cn * (unsigned)a.Kp * 4u);
    __amdgpu_buffer_rsrc_t rsGa = rsX, rsBe = rsX;
    if constexpr (PRO == PRO_PRELU_NORM) {
        rsGa = make_rsrc(a.pro_gamma, (unsigned)a.Cn * 4u);
        rsBe = make_rsrc(a.pro_beta, (unsigned)a.Cn * 4u);
    }
    int voG[MTM], voX[MTN];
#pragma unroll
    for (int j = 0; j < MTM; ++j) voG[j] = ((r0 + (tid >> 2) + 64 * j) * a.Kp + (tid & 3) * 4) * 4;
#pragma unroll
    for (int j = 0; j < MTN; ++j) {
        const int row = (tid >> 2) + 64 * j;
        voX[j] = ((c0 + row) * a.Kp + (tid & 3) * 4) * 4;
        if constexpr (PRO == PRO_PRELU_NORM)      // per-channel constants: loaded once, not per k-tile
            rg[j] = make_float2(buf_ld1(rsGa, (c0 + row) * 4, 0), buf_ld1(rsBe, (c0 + row) * 4, 0));
    }
    auto load_tile = [&](int kt, float4 (&ra)[MTM], float4 (&rb)[MTN]) {
        const int so = (kb + kt * WK) * 4;
#pragma unroll
        for (int j = 0; j < MTM; ++j) ra[j] = buf_ld4(rsG, voG[j], so);
#pragma unroll
        for (int j = 0; j < MTN; ++j) rb[j] = buf_ld4(rsX, voX[j], so);
    };
    auto store_tile = [&](int buf, int kt, float4 (&ra)[MTM], float4 (&rb)[MTN]) {
        const int kq = (tid & 3) * 4;
        if (kb + (kt + 1) * WK > ke) {            // uniform: the chunk's ragged last k-tile
            if (kb + kt * WK + kq >= ke) {
#pragma unroll
                for (int j = 0; j < MTM; ++j) ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < MTN; ++j) rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int j = 0; j < MTM; ++j) {
            const int row = (tid >> 2) + 64 * j;
            As[buf][row][kq + 0] = ra[j].x; As[buf][row][kq + 1] = ra[j].y;
            As[buf][row][kq + 2] = ra[j].z; As[buf][row][kq + 3] = ra[j].w;
        }
#pragma unroll
        for (int j = 0; j < MTN; ++j) {
            const int row = (tid >> 2) + 64 * j;
            if constexpr (PRO == PRO_PRELU_NORM)      // applied here, after the MFMA phase the loads overlapped with
                rb[j] = pro_apply(rb[j], kb + kt * WK + kq, a.K, rg[j].x, rg[j].y, p_alpha, p_mean, p_rstd);
            Bs[buf][row][kq + 0] = rb[j].x; Bs[buf][row][kq + 1] = rb[j].y;
            Bs[buf][row][kq + 2] = rb[j].z; Bs[buf][row][kq + 3] = rb[j].w;
        }
    };

    f32x16 acc[MTM][MTN];
#pragma unroll
    for (int i = 0; i < MTM; ++i)
#pragma unroll
        for (int j = 0; j < MTN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    auto compute = [&](int buf) {
#pragma unroll
        for (int s = 0; s < WK / 2; ++s) {
            const int kk = 2 * s + lhi;
            float av[MTM], bv[MTN];
#pragma unroll
            for (int i = 0; i < MTM; ++i) av[i] = As[buf][wm * HTM + 32 * i + l31][kk];
#pragma unroll
            for (int j = 0; j < MTN; ++j) bv[j] = Bs[buf][wn * HTN + 32 * j + l31][kk];
#pragma unroll
            for (int i = 0; i < MTM; ++i)
#pragma unroll
                for (int j = 0; j < MTN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };
    // prefetch distance 2, as in pw_gemm_kernel: tile kt+1 waits in one register set, tile kt+2 is in flight into the other
    float4 pa[MTM], pb[MTN], qa[MTM], qb[MTN];
    if (nk > 0) {
        load_tile(0, pa, pb);
        store_tile(0, 0, pa, pb);
        if (nk > 1) load_tile(1, pa, pb);
        if (nk > 2) load_tile(2, qa, qb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        compute(0);
        if (kt + 1 < nk) store_tile(1, kt + 1, pa, pb);
        __syncthreads();
        if (kt + 3 < nk) load_tile(kt + 3, pa, pb);
        if (kt + 1 < nk) {
            compute(1);
            if (kt + 2 < nk) store_tile(0, kt + 2, qa, qb);
            __syncthreads();
            if (kt + 4 < nk) load_tile(kt + 4, qa, qb);
        }
    }
    float* __restrict__ S = a.slab + (size_t)sp * a.R * a.Cn;
#pragma unroll
    for (int mt = 0; mt < MTM; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = r0 + wm * HTM + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
#pragma unroll
            for (int nt = 0; nt < MTN; ++nt) {
                const int c = c0 + wn * HTN + nt * 32 + l31;
                if (r < a.R && c < a.Cn) S[(size_t)r * a.Cn + c] = acc[mt][nt][e];
            }
        }
}

// ---- weight gradient, second form ("w4"): 16-byte LDS traffic and no VALU in the main loop -----------------------------
// Both operands are [channel][frame] with the contraction (frames) contiguous, so a lane can fetch four consecutive
// contraction steps of its row with ONE ds_read_b128 -- if the k index of the MFMA chain is permuted: in the 8-frame
// group j, MFMA i (i = 0..3) multiplies frame 8j + i in lanes 0-31 and frame 8j + 4 + i in lanes 32-63 (the two k
// slots of v_mfma_f32_32x32x2_f32).  Any bijection of the contraction index is a valid GEMM; A and B use the same one.
// LDS rows are 20 floats (80 B): 16-byte aligned for ds_write_b128 / ds_read_b128 and conflict-free over the 16-lane
// groups of a b128 read.  Per 16-frame k-tile and wave: 8 MFMAs, 4 ds_read_b128, 2 ds_write_b128, 2 buffer loads.
constexpr int W4LD = 20;

template <int PRO>
__global__ __launch_bounds__(NT) void pw_wgrad4_kernel(WgArgs a) {
    __shared__ __attribute__((aligned(16))) float As[2][64][W4LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][64][W4LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bid = xcd_remap(blockIdx.x, gridDim.x);          // the tiles of one split share their frames: one XCD, one L2
    const int rt = bid % a.tiles_r; bid /= a.tiles_r;
    const int ct = bid % a.tiles_c; bid /= a.tiles_c;
    const int sp = bid;
    const int m = sp / a.chunks_per_m, ch = sp % a.chunks_per_m;
    const int kb = ch * a.chunk;
    const int ke = min(kb + a.chunk, a.Kp);
    const int r0 = rt * 64, c0 = ct * 64;

    float p_mean = 0.f, p_rstd = 1.f, p_alpha = 0.f;
    if constexpr (PRO == PRO_PRELU_NORM) {
        p_mean = a.pro_ms[2 * m];
        p_rstd = a.pro_ms[2 * m + 1];
        p_alpha = a.pro_alpha[0];
    }
    const int nk = (ke - kb + WK - 1) / WK;
    const __amdgpu_buffer_rsrc_t rsG = make_rsrc(a.dOut + (size_t)m * a.R * a.Kp, (unsigned)a.R * (unsigned)a.Kp * 4u);
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.X + (size_t)m * a.Cn * a.Kp, (unsigned)a.Cn * (unsigned)a.Kp * 4u);
    const int row = tid >> 2, kq = (tid & 3) * 4;
    const int voG = ((r0 + row) * a.Kp + kq) * 4, voX = ((c0 + row) * a.Kp + kq) * 4;      // rows past R / Cn read 0 (range check)
    float2 rg = make_float2(0.f, 0.f);
    if constexpr (PRO == PRO_PRELU_NORM) {
        const __amdgpu_buffer_rsrc_t rsGa = make_rsrc(a.pro_gamma, (unsigned)a.Cn * 4u);
        const __amdgpu_buffer_rsrc_t rsBe = make_rsrc(a.pro_beta, (unsigned)a.Cn * 4u);
        rg = make_float2(buf_ld1(rsGa, (c0 + row) * 4, 0), buf_ld1(rsBe, (c0 + row) * 4, 0));
    }
    auto load_tile = [&](int kt, float4& ra, float4& rb) {
        const int so = (kb + kt * WK) * 4;
        ra = buf_ld4(rsG, voG, so);
        rb = buf_ld4(rsX, voX, so);
    };
    auto store_tile = [&](int buf, int kt, float4 ra, float4 rb) {      // (whole k-tiles only: the host requires Kp % 16 == 0)
        if constexpr (PRO == PRO_PRELU_NORM) rb = pro_apply(rb, kb + kt * WK + kq, a.K, rg.x, rg.y, p_alpha, p_mean, p_rstd);
        *reinterpret_cast<float4*>(&As[buf][row][kq]) = ra;
        *reinterpret_cast<float4*>(&Bs[buf][row][kq]) = rb;
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int l31 = lane & 31, lhi = lane >> 5;
    // lane (row l % 32, k slot l / 32) reads frames 8j + 4 (l / 32) .. +3 of its row, twice per 16-frame k-tile
    const float* const fA = &As[0][wm * 32 + l31][4 * lhi];
    const float* const fB = &Bs[0][wn * 32 + l31][4 * lhi];
    auto compute = [&](int buf) {
#pragma unroll
        for (int j = 0; j < WK / 8; ++j) {
            const float4 av = *reinterpret_cast<const float4*>(fA + buf * 64 * W4LD + 8 * j);
            const float4 bv = *reinterpret_cast<const float4*>(fB + buf * 64 * W4LD + 8 * j);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
        }

    };
    // prefetch distance 2: k-tile kt+1 waits in one register set, kt+2 is in flight into the other
    float4 pa, pb, qa, qb;
    if (nk > 0) {
        load_tile(0, pa, pb);
        store_tile(0, 0, pa, pb);
        if (nk > 1) load_tile(1, pa, pb);
        if (nk > 2) load_tile(2, qa, qb);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        compute(0);
        if (kt + 1 < nk) store_tile(1, kt + 1, pa, pb);
        __syncthreads();
        if (kt + 3 < nk) load_tile(kt + 3, pa, pb);
        if (kt + 1 < nk) {
            compute(1);
            if (kt + 2 < nk) store_tile(0, kt + 2, qa, qb);
            __syncthreads();
            if (kt + 4 < nk) load_tile(kt + 4, qa, qb);
        }
    }
    // slab [split][R][Cn]: two 128-byte row segments per store instruction, rows / columns past the matrix dropped
    const __amdgpu_buffer_rsrc_t rsS = make_rsrc(a.slab + (size_t)sp * a.R * a.Cn, (unsigned)a.R * (unsigned)a.Cn * 4u);
    const int c = c0 + wn * 32 + l31;
    const int voS = c < a.Cn ? ((r0 + wm * 32 + 4 * lhi) * a.Cn + c) * 4 : 0x7fffffff;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float v = acc[e];       // (clang lowers __builtin_bit_cast of a vector ELEMENT to element 0: go through a scalar)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsS, voS, ((e & 3) + 8 * (e >> 2)) * a.Cn * 4, 0);
    }

}

}  // namespace

#include "ctn_gemm_b3.h"            // the split-bf16 ("b3") arithmetic of the same GEMMs
#include "ctn_gemm_ws.h"            // the wave-specialised h3 forward / input-gradient kernel

int g_ctn_tile_override = -2;
extern int g_ctn_cln_lean;
extern int g_ctn_bwd_events;                // csrc/ctn_block.hip: cross-stream forks per block of the composite backward passes
#ifdef CTN_EXP_SKIP
extern int g_ctn_exp_skip;                  // csrc/ctn_block.hip: timing-experiment switch of lab builds
#endif
extern int g_ctn_gln_fuse;                  // csrc/ctn_tcn.hip: gLN stacks without the gLN-1' / PReLU-1' pass
extern int g_ctn_cln_fuse;                  // csrc/ctn_tcn.hip: cLN stacks with the second norm's backward fused into its neighbours
extern int g_ctn_cln_fr;                    // csrc/ctn_tcn.hip: frames per workgroup of the channel-wise LayerNorm backward kernel

// GEMM arithmetic (ctn_gemm_b3.h): 3 = "h3" (default: the composite stacks run their GEMMs on two fp16 pieces per operand under
// tracked power-of-two scales, three f16 MFMAs -- the ctn_*_h3 entry points; every other GEMM as b6), 2 = "b6" (three bf16 pieces
// per operand, six bf16 MFMAs, fp32 accumulation), 0 = fp32 MFMA (bit-exact fp32 FMA chains).  (1 was round 2's ~16-bit "b3":
// removed.)  CTN_GEMM_ARITH=h3|b6|fp32, ctn_tune("arith", 3|2|0).  Layers with fewer than 64 output rows (the decoder's basis
// GEMM) and weight gradients with a side below 32 stay on the fp32 kernels.
static int g_arith = -1;
static int arith_id() {
    if (g_arith < 0) {
        const char* e = getenv("CTN_GEMM_ARITH");
        if (!e || !*e || !strcmp(e, "h3")) g_arith = 3;
        else if (!strcmp(e, "b6")) g_arith = 2;
        else if (!strcmp(e, "fp32")) g_arith = 0;
        else {
            // an unknown value (round 2's "b3", a typo such as "FP32") must not silently select an arithmetic the caller did not
            // ask for: say so once and run the bit-exact one
            fprintf(stderr, "libctn_hip: CTN_GEMM_ARITH=%s is not one of h3|b6|fp32 -- using fp32 (bit-exact fp32 MFMA)\n", e);
            ctn_set_error("CTN_GEMM_ARITH=%s is not one of h3|b6|fp32: fp32 selected", e);
            g_arith = 0;
        }
    }
    return g_arith;
}
// kernel arithmetic id (template parameter AR of ctn_gemm_b3.h) of the plain entry points: 0 fp32 MFMA, 3 b6 (also under h3)
static int arith_np() { return arith_id() == 0 ? 0 : 3; }
static bool b3_fwd(int R) { return arith_id() != 0 && R >= 64; }
static bool b3_wgrad(int R, int Cn) { return arith_id() != 0 && R >= 32 && Cn >= 32; }

template <typename TL>
static void launch_tile(const PwArgs& a, int trans_w, bool pro, bool residual, int stats, bool relu, int gln_bwd,
                        hipStream_t st) {
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * a.M)), block(TL::NTH);
    if (gln_bwd == 3) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_NONE, EPI_GLN_BWD2>), grid, block, 0, st, a);
    else if (gln_bwd == 2) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_NONE, EPI_CLN_BWD>), grid, block, 0, st, a);
    else if (gln_bwd) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_NONE, EPI_GLN_BWD>), grid, block, 0, st, a);
    else if (trans_w) {
        if (pro && residual) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_PRELU_NORM, EPI_RESIDUAL>), grid, block, 0, st, a);
        else if (pro) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_PRELU_NORM, EPI_NONE>), grid, block, 0, st, a);
        else if (stats == 2) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_NONE, EPI_CLN_STATS>), grid, block, 0, st, a);
        else if (stats) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_NONE, EPI_PRELU_STATS>), grid, block, 0, st, a);
        else if (residual) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_NONE, EPI_RESIDUAL>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_NONE, EPI_NONE>), grid, block, 0, st, a);
    } else if (pro) {
        if (residual) hipLaunchKernelGGL((pw_gemm_kernel<TL, 0, PRO_PRELU_NORM, EPI_RESIDUAL>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_gemm_kernel<TL, 0, PRO_PRELU_NORM, EPI_NONE>), grid, block, 0, st, a);
    } else if (stats == 2) hipLaunchKernelGGL((pw_gemm_kernel<TL, 0, PRO_NONE, EPI_CLN_STATS>), grid, block, 0, st, a);
    else if (stats) hipLaunchKernelGGL((pw_gemm_kernel<TL, 0, PRO_NONE, EPI_PRELU_STATS>), grid, block, 0, st, a);
    else if (residual) hipLaunchKernelGGL((pw_gemm_kernel<TL, 0, PRO_NONE, EPI_RESIDUAL>), grid, block, 0, st, a);
    else if (relu) hipLaunchKernelGGL((pw_gemm_kernel<TL, 0, PRO_NONE, EPI_RELU>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pw_gemm_kernel<TL, 0, PRO_NONE, EPI_NONE>), grid, block, 0, st, a);
}

// ---- weight transposes for the forward pass --------------------------------------------------------------------
// The persistent GEMM wants weights as [contraction][row] (16-byte row writes into unpadded LDS, no transposing
// scatter).  The input-gradient GEMMs read the stored [O, I] matrices that way as they are; the forward GEMMs get a
// transposed copy, refreshed once per step: up to 64 equally shaped matrices per launch, pointers by value.
namespace {
constexpr int TR_MAX = 64;
struct TrArgs {
    const float* src[TR_MAX];
    float* dst[TR_MAX];
    int rows, cols;
};
__global__ __launch_bounds__(256) void transpose_batch_kernel(TrArgs a) {
    __shared__ float t[32][33];
    const float* __restrict__ S = a.src[blockIdx.z];
    float* __restrict__ D = a.dst[blockIdx.z];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        if (r < a.rows && c < a.cols) t[ty + 8 * j][tx] = S[(size_t)r * a.cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, r = r0 + tx;
        if (r < a.rows && c < a.cols) D[(size_t)c * a.rows + r] = t[tx][ty + 8 * j];
    }
}
}  // namespace

extern "C" int ctn_transpose_batch(const void* const* src, void* const* dst, int n, int rows, int cols, void* stream) {
    CTN_REQUIRE(src && dst && n > 0 && rows > 0 && cols > 0, "ctn_transpose_batch: bad arguments");
    for (int o = 0; o < n; o += TR_MAX) {
        TrArgs a{};
        const int cnt = n - o < TR_MAX ? n - o : TR_MAX;
        for (int i = 0; i < cnt; ++i) {
            CTN_REQUIRE(src[o + i] && dst[o + i], "ctn_transpose_batch: null matrix %d", o + i);
            a.src[i] = (const float*)src[o + i];
            a.dst[i] = (float*)dst[o + i];
        }
        a.rows = rows; a.cols = cols;
        hipLaunchKernelGGL(transpose_batch_kernel, dim3(ctn_cdiv(cols, 32), ctn_cdiv(rows, 32), cnt), dim3(256), 0,
                           (hipStream_t)stream, a);
    }
    CTN_CHECK_LAUNCH("ctn_transpose_batch");
    return CTN_OK;
}

#ifdef CTN_EXP_B3_TIMELINE
extern "C" int ctn_debug_timeline(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ctn_dbg_tl), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

static void launch_fwd(PwArgs& a, int trans_w, bool pro, bool residual, int stats, bool relu, int gln_bwd, hipStream_t st) {
    const int id = pick_tile(a.M, a.R, a.Kp);
    int tm, tn;
    tile_dims(id, &tm, &tn);
    a.tiles_r = ctn_cdiv(a.R, tm);
    a.tiles_c = ctn_cdiv(a.Kp, tn);
    switch (id) {
        case 3: launch_tile<T64x64>(a, trans_w, pro, residual, stats, relu, gln_bwd, st); break;
        case 1: launch_tile<T128x64>(a, trans_w, pro, residual, stats, relu, gln_bwd, st); break;
        case 2: launch_tile<T64x128>(a, trans_w, pro, residual, stats, relu, gln_bwd, st); break;
        default: launch_tile<T128x128>(a, trans_w, pro, residual, stats, relu, gln_bwd, st); break;
    }
}

extern "C" {

int ctn_gemm_arith(void) { return arith_id(); }

int ctn_pw_stats_parts(int M, int R, int Kp) {
    if (b3_fwd(R)) {
        int tm, tn;
        ctn_b3_tile_dims(&tm, &tn);
        return ctn_cdiv(R, tm) * ctn_cdiv(Kp, tn);
    }
    int tm, tn;
    tile_dims(pick_tile(M, R, Kp), &tm, &tn);
    return ctn_cdiv(R, tm) * ctn_cdiv(Kp, tn);
}

// Out[m] = op(W) . f(X[m]) (+ residual) ; see include/ctn_hip.h
int ctn_pw_gemm(const float* W, const float* X, float* Out, int M, int R, int Cn, int K, int Kp, int trans_w,
                const double* pro_part, int pro_nparts, const float* pro_gamma, const float* pro_beta,
                const float* pro_alpha, float* pro_ms_out,
                const float* residual, const float* epi_alpha, double* epi_part, int relu_out, void* stream) {
    int rc = check_common("ctn_pw_gemm", W, X, Out, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(!relu_out || !(residual || epi_part || pro_part || trans_w), "ctn_pw_gemm: relu_out only on the plain forward GEMM");
    CTN_REQUIRE(!(residual && epi_part), "ctn_pw_gemm: residual and stats epilogues are exclusive");
    CTN_REQUIRE(!pro_part || (pro_gamma && pro_beta && pro_alpha && pro_nparts > 0), "ctn_pw_gemm: incomplete prologue arguments");
    CTN_REQUIRE(!epi_part || epi_alpha, "ctn_pw_gemm: stats epilogue needs alpha");
    CTN_REQUIRE(!residual || aligned16(residual), "ctn_pw_gemm: residual must be 16-byte aligned");
    PwArgs a{};
    a.W = W; a.X = X; a.Out = Out; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.pro_part = pro_part; a.pro_nparts = pro_nparts; a.pro_gamma = pro_gamma; a.pro_beta = pro_beta;
    a.pro_alpha = pro_alpha; a.pro_ms_out = pro_ms_out;
    a.residual = residual; a.epi_alpha = epi_alpha; a.epi_part = epi_part;
    CTN_REQUIRE(trans_w != 2 || b3_fwd(R), "ctn_pw_gemm: trans_w = 2 (pre-split weight pieces) needs the b3 arithmetic and R >= 64");
    if (b3_fwd(R)) {
        ctn_b3_launch_fwd(arith_np(), a, trans_w, pro_part != nullptr, residual != nullptr, epi_part != nullptr, relu_out != 0, false,
                          (hipStream_t)stream);
    } else {
        launch_fwd(a, trans_w, pro_part != nullptr, residual != nullptr, epi_part != nullptr, relu_out != 0, false,
                   (hipStream_t)stream);
    }
    CTN_CHECK_LAUNCH("ctn_pw_gemm");
    return CTN_OK;
}

// dN[m] = W^T . dOut[m]  (W stored [Cn=O_fwd, R=I_fwd]) plus the two per-utterance sums that
// gLN backward needs:  S1 = sum gamma*dN, S2 = sum gamma*dN*xhat, xhat = (prelu(y)-mean)*rstd.
static int dgrad_gln(const char* fn, const float* W, int planes, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                     const float* y, const float* gamma, const float* alpha, const float* ms, double* sums_part, void* stream) {
    int rc = check_common(fn, W, dOut, dN, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(y && gamma && alpha && ms && sums_part, "%s: null pointer", fn);
    CTN_REQUIRE(aligned16(y), "%s: y must be 16-byte aligned", fn);
    CTN_REQUIRE(!planes || b3_fwd(R), "%s: pre-split weight pieces need the b3 arithmetic and R >= 64", fn);
    PwArgs a{};
    a.W = W; a.X = dOut; a.Out = dN; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.bwd_y = y; a.bwd_gamma = gamma; a.bwd_alpha = alpha; a.bwd_ms = ms; a.bwd_part = sums_part;
    if (b3_fwd(R)) {
        ctn_b3_launch_fwd(arith_np(), a, planes ? 2 : 1, false, false, false, false, true, (hipStream_t)stream);
    } else {
        launch_fwd(a, 1, false, false, false, false, true, (hipStream_t)stream);
    }
    CTN_CHECK_LAUNCH(fn);
    return CTN_OK;
}

int ctn_pw_dgrad_gln(const float* W, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                     const float* y, const float* gamma, const float* alpha, const float* ms, double* sums_part,
                     void* stream) {
    return dgrad_gln("ctn_pw_dgrad_gln", W, 0, dOut, dN, M, R, Cn, K, Kp, y, gamma, alpha, ms, sums_part, stream);
}

// the same on pre-split weight pieces (ctn_split_b3_batch with k_major = 1 on the stored [Cn, R] matrix)
int ctn_pw_dgrad_gln_planes(const void* Wp, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                            const float* y, const float* gamma, const float* alpha, const float* ms, double* sums_part,
                            void* stream) {
    return dgrad_gln("ctn_pw_dgrad_gln_planes", (const float*)Wp, 1, dOut, dN, M, R, Cn, K, Kp, y, gamma, alpha, ms, sums_part, stream);
}

size_t ctn_split_b3_bytes(int R, int Cn) { return ctn_b3_planes_bytes(arith_np() ? arith_np() : 3, R, Cn); }

// dst[i] = bf16 piece fragments of the GEMM weight operand A [R, Cn] taken from src[i]: k_major = 0: src is stored [R, Cn];
// k_major = 1: src is stored [Cn, R] (its transpose is the operand).  HOST arrays of device pointers; see include/ctn_hip.h.
int ctn_split_b3_batch(const void* const* src, void* const* dst, int n, int R, int Cn, int k_major, void* stream) {
    CTN_REQUIRE(src && dst && n > 0 && R > 0 && Cn > 0, "ctn_split_b3_batch: bad arguments");
    for (int i = 0; i < n; ++i) CTN_REQUIRE(src[i] && dst[i] && aligned16(dst[i]), "ctn_split_b3_batch: matrix %d: null or unaligned pointer", i);
    CTN_REQUIRE(arith_np() != 0, "ctn_split_b3_batch: the fp32-MFMA arithmetic has no piece form");
    ctn_b3_launch_split(arith_np(), src, dst, n, R, Cn, k_major, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_split_b3_batch");
    return CTN_OK;
}

}  // extern "C"

extern "C" {

static int g_wgrad_blocks = 512;   // target workgroups per launch of the fp32-MFMA weight gradient (ctn_tune("wgrad_blocks", n))

// fp32-MFMA weight gradient: 64x64 output tiles on the "w4" kernel; layers with a side below 64 (the encoder / decoder bases,
// under every arithmetic) one 128x128 tile on the scalar-LDS kernel.  Small output tiles keep the number of split-K slabs down
// (slab traffic = splits x R x Cn x 4 B, written once and read once by the reduce kernel).
static void wgrad_plan(int M, int R, int Cn, int Kp, int* tile, int* chunk, int* chunks_per_m) {
    const int wt = (R >= 64 && Cn >= 64) ? 64 : 128;
    const int tiles = ctn_cdiv(R, wt) * ctn_cdiv(Cn, wt);
    int cpm = ctn_cdiv(g_wgrad_blocks, tiles * M);
    const int max_cpm = ctn_cdiv(Kp, 256);         // keep >= 256 frames of contraction per slab
    if (cpm > max_cpm) cpm = max_cpm;
    if (cpm < 1) cpm = 1;
    int c = ctn_cdiv(ctn_cdiv(Kp, cpm), WK) * WK;
    *tile = wt;
    *chunk = c;
    *chunks_per_m = ctn_cdiv(Kp, c);
}

// Library switches for in-process A/B runs and the test-suite (process-global: call them from the thread that issues the work,
// between steps).  Keys: "arith" (3 h3, 2 b6, 0 fp32 MFMA), "b3_tile" (0 128x128, 1 128x64, 2 256x64: tile of the split-bf16
// forward / input-gradient kernels), "b3_tile_k3" (the same for the prologue + residual form), "b3_wgrad_blocks" / "wgrad_blocks"
// (target workgroups per weight-gradient launch, split-bf16 / fp32), "pw_tile" (fp32 forward tile id 0..3, -1 = default),
// "wgrad_chain" (below).
static int g_ctn_wgrad_chain = 0;        // ctn_tune("wgrad_chain", 1): a weight gradient's slabs are summed inside the next launch of the chain (measured equal: off)
int ctn_tune(const char* key, int value) {
    if (!key) return CTN_ERR_ARG;
    if (!strcmp(key, "pw_tile") && value >= -1 && value <= 3) g_ctn_tile_override = value;
    else if (!strcmp(key, "wgrad_blocks") && value >= 1) g_wgrad_blocks = value;
    else if (!strcmp(key, "arith") && (value == 0 || value == 2 || value == 3)) g_arith = value;
    else if (!strcmp(key, "b3_tile") && value >= 0 && value <= 3) g_ctn_b3_tile = value;
    else if (!strcmp(key, "b3_tile_k3") && value >= 0 && value <= 3) g_ctn_b3_tile_k3 = value;
    else if (!strcmp(key, "b3_wgrad_blocks") && value >= 1) g_ctn_b3_wgrad_blocks = value;
    else if (!strcmp(key, "cln_fr") && (value == 16 || value == 32)) g_ctn_cln_fr = value;
    else if (!strcmp(key, "b3_ws") && (value == 0 || value == 1)) g_ctn_b3_ws = value;
    else if (!strcmp(key, "b3_ws_blocks") && value >= 1) g_ctn_b3_ws_blocks = value;
    else if (!strcmp(key, "wgrad_chain") && (value == 0 || value == 1)) g_ctn_wgrad_chain = value;
    else if (!strcmp(key, "cln_lean") && (value == 0 || value == 1)) g_ctn_cln_lean = value;
    else if (!strcmp(key, "cln_fuse") && value >= 0 && value <= 2) g_ctn_cln_fuse = value;
    else if (!strcmp(key, "gln_fuse") && (value == 0 || value == 1)) g_ctn_gln_fuse = value;
#ifdef CTN_EXP_SKIP
    else if (!strcmp(key, "exp_skip") && value >= 0) g_ctn_exp_skip = value;
#endif
    else if (!strcmp(key, "bwd_events") && value >= 0 && value <= 2) g_ctn_bwd_events = value;
    else { ctn_set_error("ctn_tune: unknown key or bad value: %s=%d", key, value); return CTN_ERR_ARG; }
    return CTN_OK;
}

size_t ctn_pw_wgrad_workspace(int M, int R, int Cn, int Kp) {
    int tile, chunk, cpm;
    if (b3_wgrad(R, Cn)) {
        ctn_b3_wgrad_plan(M, R, Cn, Kp, &chunk, &cpm);
        return (size_t)M * cpm * R * Cn * sizeof(float);
    }
    wgrad_plan(M, R, Cn, Kp, &tile, &chunk, &cpm);
    return (size_t)M * cpm * R * Cn * sizeof(float);
}

// dW[R,Cn] = sum_{m,k} dOut[m,r,k] * f(X[m,c,k])
int ctn_pw_wgrad(const float* dOut, const float* X, float* dW, int M, int R, int Cn, int K, int Kp,
                 const float* pro_gamma, const float* pro_beta, const float* pro_alpha, const float* pro_ms,
                 void* workspace, size_t workspace_bytes, void* stream) {
    return ctn_pw_wgrad_chained(dOut, X, dW, M, R, Cn, K, Kp, pro_gamma, pro_beta, pro_alpha, pro_ms, workspace, workspace_bytes, stream, nullptr);
}
}  // extern "C"

// ---- chained weight gradients (ctn_common.h) -------------------------------------------------------------------------------
// The split-K slabs of one launch are summed by the NEXT weight-gradient launch of the same stream (its workgroups do it while
// their first tiles are in flight) instead of a slab_reduce launch of their own: one launch less per weight gradient and the
// reduction off the stream's critical path.  ctn_wgrad_chain_flush() sums what is still pending.
static void chain_attach(WgArgs& a, const CtnWgradChain* chain) {
    if (chain && chain->slab) { a.prev_slab = chain->slab; a.prev_out = chain->out; a.prev_n = chain->n; a.prev_nsplit = chain->nsplit; }
}

static int chain_finish(const WgArgs& a, int ns, float* dW, CtnWgradChain* chain, hipStream_t st, const char* fn) {
    const long long nn = (long long)a.R * a.Cn;
    if (chain) {
        chain->slab = a.slab; chain->out = dW; chain->n = nn; chain->nsplit = ns;
        return CTN_OK;
    }
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ctn_cdivll(nn / 4, NT)), dim3(NT), 0, st, a.slab, ns, nn, dW);
    CTN_CHECK_LAUNCH(fn);
    return CTN_OK;
}

int ctn_wgrad_chain_flush(CtnWgradChain* chain, void* stream) {
    if (!chain || !chain->slab) return CTN_OK;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ctn_cdivll(chain->n / 4, NT)), dim3(NT), 0, (hipStream_t)stream, chain->slab, chain->nsplit,
                       chain->n, chain->out);
    *chain = CtnWgradChain{};
    CTN_CHECK_LAUNCH("ctn_wgrad_chain_flush");
    return CTN_OK;
}

int ctn_pw_wgrad_chained(const float* dOut, const float* X, float* dW, int M, int R, int Cn, int K, int Kp,
                         const float* pro_gamma, const float* pro_beta, const float* pro_alpha, const float* pro_ms,
                         void* workspace, size_t workspace_bytes, void* stream, CtnWgradChain* chain) {
    int rc = check_common("ctn_pw_wgrad", dW, X, (const float*)dOut, M, R, Cn, K, Kp);
    if (rc) return rc;
    if (!g_ctn_wgrad_chain && chain) { if ((rc = ctn_wgrad_chain_flush(chain, stream))) return rc; chain = nullptr; }
    CTN_REQUIRE(!pro_ms || (pro_gamma && pro_beta && pro_alpha), "ctn_pw_wgrad: incomplete prologue arguments");
    WgArgs a{};
    a.dOut = dOut; a.X = X; a.slab = (float*)workspace; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.pro_gamma = pro_gamma; a.pro_beta = pro_beta; a.pro_alpha = pro_alpha; a.pro_ms = pro_ms;
    if (b3_wgrad(R, Cn)) {
        ctn_b3_wgrad_plan(M, R, Cn, Kp, &a.chunk, &a.chunks_per_m);
        const size_t need = (size_t)M * a.chunks_per_m * R * Cn * sizeof(float);
        if (workspace == nullptr || workspace_bytes < need) {
            ctn_set_error("ctn_pw_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
            return CTN_ERR_WORKSPACE;
        }
        CTN_REQUIRE(!chain || chain->slab != a.slab, "ctn_pw_wgrad: a chained launch needs a slab buffer of its own");
        chain_attach(a, chain);
        const int ns = ctn_b3_launch_wgrad(arith_np(), a, pro_ms != nullptr, (hipStream_t)stream);
        CTN_CHECK_LAUNCH("ctn_pw_wgrad");
        return chain_finish(a, ns, dW, chain, (hipStream_t)stream, "ctn_pw_wgrad/reduce");
    }
    if ((rc = ctn_wgrad_chain_flush(chain, stream))) return rc;        // the fp32-MFMA kernels do not chain
    int wt;
    wgrad_plan(M, R, Cn, Kp, &wt, &a.chunk, &a.chunks_per_m);
    a.tiles_r = ctn_cdiv(R, wt); a.tiles_c = ctn_cdiv(Cn, wt);
    const int nsplit = M * a.chunks_per_m;
    if (workspace == nullptr || workspace_bytes < (size_t)nsplit * R * Cn * sizeof(float)) {
        ctn_set_error("ctn_pw_wgrad: workspace too small (%zu < %zu)", workspace_bytes, (size_t)nsplit * R * Cn * sizeof(float));
        return CTN_ERR_WORKSPACE;
    }
    a.pro_gamma = pro_gamma; a.pro_beta = pro_beta; a.pro_alpha = pro_alpha; a.pro_ms = pro_ms;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * nsplit)), block(NT);
    if (wt == 64) {
        if (pro_ms) hipLaunchKernelGGL((pw_wgrad4_kernel<PRO_PRELU_NORM>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_wgrad4_kernel<PRO_NONE>), grid, block, 0, st, a);
    } else {
        if (pro_ms) hipLaunchKernelGGL((pw_wgrad_kernel<PRO_PRELU_NORM, 128, 128>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_wgrad_kernel<PRO_NONE, 128, 128>), grid, block, 0, st, a);
    }
    CTN_CHECK_LAUNCH("ctn_pw_wgrad");
    const long long n = (long long)R * Cn;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ctn_cdivll(n / 4, NT)), block, 0, st, a.slab, nsplit, n, dW);   // R, Cn multiples of 4
    CTN_CHECK_LAUNCH("ctn_pw_wgrad/reduce");
    return CTN_OK;
}

extern "C" {
// ---- h3 arithmetic: explicit entry points (independent of ctn_tune("arith")); see include/ctn_hip.h ------------------------------
size_t ctn_split_h3_bytes(int R, int Cn) { return ctn_b3_planes_bytes(4, R, Cn); }

int ctn_split_h3_batch(const void* const* src, void* const* dst, int n, int R, int Cn, int k_major, void* stream) {
    CTN_REQUIRE(src && dst && n > 0 && R > 0 && Cn > 0 && (R * Cn) % 4 == 0, "ctn_split_h3_batch: bad arguments");
    for (int i = 0; i < n; ++i) CTN_REQUIRE(src[i] && dst[i] && aligned16(src[i]) && aligned16(dst[i]), "ctn_split_h3_batch: matrix %d: null or unaligned pointer", i);
    ctn_b3_launch_split(4, src, dst, n, R, Cn, k_major, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_split_h3_batch");
    return CTN_OK;
}

int ctn_absmax_batch(const void* const* src, void* const* dst, int n, int len, void* stream) {
    CTN_REQUIRE(src && dst && n > 0 && len > 0, "ctn_absmax_batch: bad arguments");
    for (int i = 0; i < n; ++i) CTN_REQUIRE(src[i] && dst[i] && aligned16(src[i]), "ctn_absmax_batch: array %d: null or unaligned pointer", i);
    ctn_b3_launch_absmax(src, dst, 0, n, len, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_absmax_batch");
    return CTN_OK;
}

int ctn_absmax_rows(const float* x, int M, long long n, unsigned* amax, void* stream) {
    CTN_REQUIRE(x && amax && M > 0 && n > 0 && n % 4 == 0 && aligned16(x), "ctn_absmax_rows: bad arguments");
    int nb = (int)ctn_cdivll(n, 1024 * 8);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(absmax_rows_kernel, dim3(nb, M), dim3(256), 0, (hipStream_t)stream, x, n, amax);
    CTN_CHECK_LAUNCH("ctn_absmax_rows");
    return CTN_OK;
}

int ctn_pw_gemm_h3(const void* Wp, const float* X, float* Out, int M, int R, int Cn, int K, int Kp,
                   const double* pro_part, int pro_nparts, const float* pro_gamma, const float* pro_beta,
                   const float* pro_alpha, float* pro_ms_out, const float* residual, const float* epi_alpha, double* epi_part,
                   const unsigned* x_amax, const float* pro_gbmax, unsigned* out_amax, void* stream) {
    int rc = check_common("ctn_pw_gemm_h3", (const float*)Wp, X, Out, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(R >= 64, "ctn_pw_gemm_h3: R >= 64");
    CTN_REQUIRE(x_amax, "ctn_pw_gemm_h3: the operand's maximum is required");
    CTN_REQUIRE(!(residual && epi_part), "ctn_pw_gemm_h3: residual and stats epilogues are exclusive");
    CTN_REQUIRE(!pro_part || (pro_gamma && pro_beta && pro_alpha && pro_nparts > 0 && pro_gbmax), "ctn_pw_gemm_h3: incomplete prologue arguments");
    CTN_REQUIRE(!epi_part || epi_alpha, "ctn_pw_gemm_h3: stats epilogue needs alpha");
    CTN_REQUIRE(!residual || aligned16(residual), "ctn_pw_gemm_h3: residual must be 16-byte aligned");
    CTN_REQUIRE(!out_amax || residual, "ctn_pw_gemm_h3: out_amax comes with the residual epilogue");
    PwArgs a{};
    a.W = (const float*)Wp; a.X = X; a.Out = Out; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.pro_part = pro_part; a.pro_nparts = pro_nparts; a.pro_gamma = pro_gamma; a.pro_beta = pro_beta;
    a.pro_alpha = pro_alpha; a.pro_ms_out = pro_ms_out;
    a.residual = residual; a.epi_alpha = epi_alpha; a.epi_part = epi_part;
    a.x_amax = x_amax; a.pro_gbmax = pro_gbmax; a.out_amax = out_amax;
    if (!ctn_ws_launch(a, pro_part != nullptr, residual != nullptr, epi_part != nullptr, false, false, (hipStream_t)stream))
        ctn_b3_launch_fwd(4, a, 2, pro_part != nullptr, residual != nullptr, epi_part != nullptr, false, false, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_pw_gemm_h3");
    return CTN_OK;
}

int ctn_pw_dgrad_gln_h3(const void* Wp, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                        const float* y, const float* gamma, const float* alpha, const float* ms, double* sums_part,
                        const unsigned* g_amax, void* stream) {
    int rc = check_common("ctn_pw_dgrad_gln_h3", (const float*)Wp, dOut, dN, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(y && gamma && alpha && ms && sums_part && g_amax, "ctn_pw_dgrad_gln_h3: null pointer");
    CTN_REQUIRE(aligned16(y) && R >= 64, "ctn_pw_dgrad_gln_h3: y must be 16-byte aligned, R >= 64");
    PwArgs a{};
    a.W = (const float*)Wp; a.X = dOut; a.Out = dN; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.bwd_y = y; a.bwd_gamma = gamma; a.bwd_alpha = alpha; a.bwd_ms = ms; a.bwd_part = sums_part;
    a.x_amax = g_amax;
    if (!ctn_ws_launch(a, false, false, false, false, true, (hipStream_t)stream))
        ctn_b3_launch_fwd(4, a, 2, false, false, false, false, true, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_pw_dgrad_gln_h3");
    return CTN_OK;
}

// ---- channel-wise LayerNorm backward fused into the input-gradient GEMM (round 4): dN[m] = W^T . dOut[m] plus, per FRAME, the two
// sums over channels that cLN backward needs (S1[k] = sum_c gamma_c dN[c,k], S2[k] = sum_c gamma_c dN[c,k] xhat[c,k]) as per-row-tile
// column partials.  w_form: 1 = W stored fp32 [Cn, R] (arithmetic by ctn_tune("arith")), 2 = b6 pieces (ctn_split_b3_batch, k_major = 1),
// 3 = h3 pieces (ctn_split_h3_batch; g_amax = tracked maximum of dOut).  See include/ctn_hip.h.
static bool col_b3(int R, int w_form) { return w_form == 3 || (w_form == 2) || (w_form <= 1 && b3_fwd(R)); }

int ctn_pw_col_parts(int M, int R, int Kp, int w_form) {
    int tm, tn;
    if (col_b3(R, w_form)) ctn_b3_tile_dims(&tm, &tn);
    else tile_dims(pick_tile(M, R, Kp), &tm, &tn);
    return ctn_cdiv(R, tm);
}

int ctn_pw_dgrad_cln(const void* W, int w_form, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                     const float* y, const float* gamma, const float* alpha, const float* mean, const float* rstd,
                     double* col_part, const unsigned* g_amax, void* stream) {
    int rc = check_common("ctn_pw_dgrad_cln", (const float*)W, dOut, dN, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(w_form >= 1 && w_form <= 3, "ctn_pw_dgrad_cln: w_form must be 1 (fp32 [Cn, R]), 2 (b6 pieces) or 3 (h3 pieces)");
    CTN_REQUIRE(y && gamma && alpha && mean && rstd && col_part, "ctn_pw_dgrad_cln: null pointer");
    CTN_REQUIRE(aligned16(y) && aligned16(mean) && aligned16(rstd) && aligned16(col_part), "ctn_pw_dgrad_cln: y, mean, rstd, col_part must be 16-byte aligned");
    CTN_REQUIRE(w_form != 3 || (g_amax && R >= 64), "ctn_pw_dgrad_cln: h3 pieces need the operand's maximum and R >= 64");
    CTN_REQUIRE(w_form != 2 || b3_fwd(R), "ctn_pw_dgrad_cln: b6 pieces need a split-bf16 arithmetic and R >= 64");
    PwArgs a{};
    a.W = (const float*)W; a.X = dOut; a.Out = dN; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.bwd_y = y; a.bwd_gamma = gamma; a.bwd_alpha = alpha; a.cln_mean = mean; a.cln_rstd = rstd; a.col_part = col_part;
    a.x_amax = g_amax;
    if (w_form == 3) ctn_b3_launch_fwd(4, a, 2, false, false, false, false, 2, (hipStream_t)stream);
    else if (col_b3(R, w_form)) ctn_b3_launch_fwd(arith_np(), a, w_form == 2 ? 2 : 1, false, false, false, false, 2, (hipStream_t)stream);
    else launch_fwd(a, 1, false, false, false, false, 2, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_pw_dgrad_cln");
    return CTN_OK;
}

// gLN blocks without the gLN-1' / PReLU-1' pass (round 4; include/ctn_hip.h): ctn_pw_dgrad_gln plus six more per-utterance sums from which
// the FIRST norm's backward sums follow before the depthwise backward has run -- the depthwise conv's adjoint moves them onto dd:
//   S1' = sum gamma1 dn1      = sum_{c,k} dd[c,k] gamma1[c] V[c,k]
//   S2' = sum gamma1 dn1 xh1  = sum_{c,k} dd[c,k] (d[c,k] - beta1[c] V[c,k]),     V[c,k] = sum of the taps of frame k that stay inside [0, K)
// and dd = u rstd2 (t - c1 - xh2 c2) is affine in (c1, c2) = (S1, S2) / n, so with u = prelu'(d), t = gamma2 dN:
//   sums_part [M, parts, 8] = S1, S2, sum u t g1V, sum u g1V, sum u xh2 g1V, sum u t e, sum u e, sum u xh2 e     (g1V = gamma1 V, e = d - beta1 V).
// w_form as ctn_pw_dgrad_cln.
int ctn_pw_dgrad_gln2(const void* W, int w_form, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                      const float* y, const float* gamma, const float* alpha, const float* ms,
                      const float* gamma1, const float* beta1, const float* D, int P, int dilation, int causal,
                      double* sums_part, const unsigned* g_amax, void* stream) {
    int rc = check_common("ctn_pw_dgrad_gln2", (const float*)W, dOut, dN, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(w_form >= 1 && w_form <= 3, "ctn_pw_dgrad_gln2: w_form must be 1 (fp32 [Cn, R]), 2 (b6 pieces) or 3 (h3 pieces)");
    CTN_REQUIRE(y && gamma && alpha && ms && gamma1 && beta1 && D && sums_part, "ctn_pw_dgrad_gln2: null pointer");
    CTN_REQUIRE(aligned16(y) && P >= 1 && P <= 8 && dilation >= 1, "ctn_pw_dgrad_gln2: y must be 16-byte aligned, 1 <= P <= 8");
    CTN_REQUIRE(w_form != 3 || (g_amax && R >= 64), "ctn_pw_dgrad_gln2: h3 pieces need the operand's maximum and R >= 64");
    CTN_REQUIRE(w_form != 2 || b3_fwd(R), "ctn_pw_dgrad_gln2: b6 pieces need a split-bf16 arithmetic and R >= 64");
    const int halo = (P - 1) * dilation;
    CTN_REQUIRE(causal || halo % 2 == 0, "ctn_pw_dgrad_gln2: non-causal 'same' padding needs (P-1)*dilation even");
    PwArgs a{};
    a.W = (const float*)W; a.X = dOut; a.Out = dN; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.bwd_y = y; a.bwd_gamma = gamma; a.bwd_alpha = alpha; a.bwd_ms = ms; a.bwd_part = sums_part;
    a.g1 = gamma1; a.b1 = beta1; a.dw_D = D; a.dw_P = P; a.dw_dil = dilation; a.dw_padl = causal ? halo : halo / 2;
    a.x_amax = g_amax;
    if (w_form == 3) ctn_b3_launch_fwd(4, a, 2, false, false, false, false, 3, (hipStream_t)stream);
    else if (col_b3(R, w_form)) ctn_b3_launch_fwd(arith_np(), a, w_form == 2 ? 2 : 1, false, false, false, false, 3, (hipStream_t)stream);
    else launch_fwd(a, 1, false, false, false, false, 3, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_pw_dgrad_gln2");
    return CTN_OK;
}

// Forward 1x1 conv whose output feeds PReLU + channel-wise LayerNorm: Out[m] = op(W) . X[m], and per FRAME the sums over channels of
// p = prelu(Out, alpha) and p^2 as per-row-tile column partials (layout of ctn_pw_dgrad_cln) -- the norm's statistics come out of
// the producing GEMM instead of a pass over its output.  w_form: 0 = W stored fp32 [R, Cn], 1 = stored [Cn, R] (used transposed),
// 2 = b6 pieces, 3 = h3 pieces (x_amax = tracked maximum of X).
int ctn_pw_gemm_cln(const void* W, int w_form, const float* X, float* Out, int M, int R, int Cn, int K, int Kp,
                    const float* alpha, double* col_part, const unsigned* x_amax, void* stream) {
    int rc = check_common("ctn_pw_gemm_cln", (const float*)W, X, Out, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(w_form >= 0 && w_form <= 3, "ctn_pw_gemm_cln: w_form must be 0 (fp32 [R, Cn]), 1 (fp32 [Cn, R]), 2 (b6 pieces) or 3 (h3 pieces)");
    CTN_REQUIRE(alpha && col_part && aligned16(col_part), "ctn_pw_gemm_cln: null or unaligned pointer");
    CTN_REQUIRE(w_form != 3 || (x_amax && R >= 64), "ctn_pw_gemm_cln: h3 pieces need the operand's maximum and R >= 64");
    CTN_REQUIRE(w_form != 2 || b3_fwd(R), "ctn_pw_gemm_cln: b6 pieces need a split-bf16 arithmetic and R >= 64");
    PwArgs a{};
    a.W = (const float*)W; a.X = X; a.Out = Out; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.epi_alpha = alpha; a.col_part = col_part; a.x_amax = x_amax;
    if (w_form == 3) ctn_b3_launch_fwd(4, a, 2, false, false, 2, false, 0, (hipStream_t)stream);
    else if (col_b3(R, w_form)) ctn_b3_launch_fwd(arith_np(), a, w_form, false, false, 2, false, 0, (hipStream_t)stream);
    else launch_fwd(a, w_form, false, false, 2, false, 0, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_pw_gemm_cln");
    return CTN_OK;
}

size_t ctn_pw_wgrad_h3_workspace(int M, int R, int Cn, int Kp) {
    int chunk, cpm;
    ctn_b3_wgrad_plan(M, R, Cn, Kp, &chunk, &cpm);
    return (size_t)M * cpm * R * Cn * sizeof(float);
}

int ctn_pw_wgrad_h3(const float* dOut, const float* X, float* dW, int M, int R, int Cn, int K, int Kp,
                    const float* pro_gamma, const float* pro_beta, const float* pro_alpha, const float* pro_ms,
                    const unsigned* g_amax, const unsigned* x_amax, const float* pro_gbmax,
                    void* workspace, size_t workspace_bytes, void* stream) {
    return ctn_pw_wgrad_h3_chained(dOut, X, dW, M, R, Cn, K, Kp, pro_gamma, pro_beta, pro_alpha, pro_ms, g_amax, x_amax, pro_gbmax, workspace,
                                   workspace_bytes, stream, nullptr);
}
}  // extern "C"

int ctn_pw_wgrad_h3_chained(const float* dOut, const float* X, float* dW, int M, int R, int Cn, int K, int Kp,
                            const float* pro_gamma, const float* pro_beta, const float* pro_alpha, const float* pro_ms,
                            const unsigned* g_amax, const unsigned* x_amax, const float* pro_gbmax,
                            void* workspace, size_t workspace_bytes, void* stream, CtnWgradChain* chain) {
    int rc = check_common("ctn_pw_wgrad_h3", dW, X, (const float*)dOut, M, R, Cn, K, Kp);
    if (rc) return rc;
    if (!g_ctn_wgrad_chain && chain) { if ((rc = ctn_wgrad_chain_flush(chain, stream))) return rc; chain = nullptr; }
    CTN_REQUIRE(R >= 32 && Cn >= 32, "ctn_pw_wgrad_h3: both sides >= 32");
    CTN_REQUIRE(g_amax && x_amax, "ctn_pw_wgrad_h3: the operands' maxima are required");
    CTN_REQUIRE(!pro_ms || (pro_gamma && pro_beta && pro_alpha && pro_gbmax), "ctn_pw_wgrad_h3: incomplete prologue arguments");
    WgArgs a{};
    a.dOut = dOut; a.X = X; a.slab = (float*)workspace; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.pro_gamma = pro_gamma; a.pro_beta = pro_beta; a.pro_alpha = pro_alpha; a.pro_ms = pro_ms;
    a.g_amax = g_amax; a.x_amax = x_amax; a.pro_gbmax = pro_gbmax;
    ctn_b3_wgrad_plan(M, R, Cn, Kp, &a.chunk, &a.chunks_per_m);
    const size_t need = (size_t)M * a.chunks_per_m * R * Cn * sizeof(float);
    if (workspace == nullptr || workspace_bytes < need) {
        ctn_set_error("ctn_pw_wgrad_h3: workspace too small (%zu < %zu)", workspace_bytes, need);
        return CTN_ERR_WORKSPACE;
    }
    CTN_REQUIRE(!chain || chain->slab != a.slab, "ctn_pw_wgrad_h3: a chained launch needs a slab buffer of its own");
    chain_attach(a, chain);
    const int ns = ctn_b3_launch_wgrad(4, a, pro_ms != nullptr, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_pw_wgrad_h3");
    return chain_finish(a, ns, dW, chain, (hipStream_t)stream, "ctn_pw_wgrad_h3/reduce");
}
