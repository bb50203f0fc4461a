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
#ifdef CTN_EXP_SKIP_MAIN          // experiment builds (benchmarks/gemm_lab.py): launch + prologue + epilogue only
    const int nk = 1;
#else
    const int nk = (a.Cn + BK - 1) / BK;
#endif
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
        if constexpr (TL::MF == 16) {
            // v_mfma_f32_16x16x4_f32: lane l supplies A[row l % 16][k l / 16] and B[k l / 16][column l % 16]; a 32x32 sub-tile
            // is 2x2 instruction tiles (accumulator elements 4q .. 4q+3 of sub-tile q = 2 si + sj), contraction step 4
            const float* const Ab = As + buf * BK * LDA + (lane >> 4) * LDA + wm * WM + (lane & 15);
            const float* const Bb = Bs + buf * BK * LDB + (lane >> 4) * LDB + wn * WN + (lane & 15);
#pragma unroll
            for (int s = 0; s < BK / 4; ++s) {
                float av[MT][2], bv[NTL][2];
#pragma unroll
                for (int i = 0; i < MT; ++i) { av[i][0] = Ab[4 * s * LDA + 32 * i]; av[i][1] = Ab[4 * s * LDA + 32 * i + 16]; }
#pragma unroll
                for (int j = 0; j < NTL; ++j) { bv[j][0] = Bb[4 * s * LDB + 32 * j]; bv[j][1] = Bb[4 * s * LDB + 32 * j + 16]; }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4v c = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                            c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][q >> 1], bv[j][q & 1], c, 0, 0, 0);
                            acc[i][j][4 * q] = c[0]; acc[i][j][4 * q + 1] = c[1]; acc[i][j][4 * q + 2] = c[2]; acc[i][j][4 * q + 3] = c[3];
                        }
            }
        } else {
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

#ifdef CTN_EXP_SKIP_EPI           // experiment builds: main loop only (one never-taken store keeps the accumulators live)
    if (acc[0][0][0] == 12345.678f) a.Out[tid] = acc[0][0][1] + acc[MT - 1][NTL - 1][15];
#else
    gemm_epilogue<TL, EPI>(a, acc, smem, red, m, rt, ct);
#endif
}


// ===========================================================================================================
// Persistent fp32-MFMA GEMM ("pk"): the product kernel of every 1x1 convolution in forward / input-gradient form.
//
// Cost model it is built on (benchmarks/mfma_probe.hip, profiles/r02_a_mfma_probe.txt): v_mfma_f32_32x32x2_f32 issues
// at 64.9 cycles whatever the occupancy or chain count (154.5 TF), but on a SIMD it SERIALISES with every VALU
// instruction of every resident wave (fp32 MFMA and fp32 VALU share the datapath: two waves, one MFMA-only and one
// v_fma-only, take 0.87-1.0 of the SUM of their times; the bf16 MFMA overlaps the same VALU stream completely),
// while LDS and vector-memory instructions overlap it.  A VALU burst between two MFMAs of one wave costs ~14 cycles
// + ~4.7 per instruction.  So the kernel spends VALU instructions nowhere it can avoid them:
//   * persistent workgroups: 64x64 output tiles dealt to gridDim.x resident workgroups, per-lane address registers
//     are computed once per workgroup, per-tile offsets are scalar (SALU is free);
//   * the k-tile pipeline (LDS double buffer + two register staging sets, prefetch distance 2) runs ACROSS tile
//     boundaries: the first k-tiles of the next tile are in flight while the current tile finishes;
//   * weights are read in [contraction][row] form (TRANS_W = 1: the forward pass gets a transposed copy), so both
//     operands land in LDS as 16-byte row writes into UNPADDED 64-dword rows and every fragment read is a
//     ds_read2st64_b32 with immediate offsets: the main loop has no VALU instruction;
//   * the epilogue stores straight from the accumulators (two 128-byte row segments per store instruction), with
//     residual / PReLU-statistics / gLN-backward sums computed on the accumulator registers; statistics leave as one
//     (sum, sum of squares) partial per WAVE, reduced by DPP in fp32 -- no LDS round trip, no barrier.
// ===========================================================================================================
#ifdef CTN_EXP_CLOCK      // experiment builds only: in-kernel clock of the persistent GEMM (s_memtime / s_memrealtime)
__device__ unsigned long long ctn_dbg[8];
#endif

struct PkTile {
    int m, r0, c0, pidx;     // utterance, first row, first column, index of the tile within the utterance (ct * tiles_r + rt)
};

// (mean, rstd) of one utterance from [nparts][2] fp64 partials, computed redundantly by every wave (no LDS, no barrier)
__device__ __forceinline__ void finalize_stats_wave(const double* __restrict__ part, int nparts, double count, float& mean,
                                                    float& rstd) {
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x & 63; i < nparts; i += 64) {
        s += part[2 * i];
        q += part[2 * i + 1];
    }
    s = wave_sum(s);
    q = wave_sum(q);
    const double mu = s / count;
    double var = q / count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + (double)CTN_EPS));
}

// experiment hooks (benchmarks/gemm_lab.py builds; never defined in the product build)
#ifdef CTN_EXP_NO_BARRIER
#define PK_SYNC() __builtin_amdgcn_s_waitcnt(0xc07f)   /* lgkmcnt(0) only */
#else
#define PK_SYNC() __syncthreads()
#endif
#ifdef CTN_EXP_NO_LDS_READ
#define PK_A(off) xa0
#define PK_B(off) xb0
#else
#define PK_A(off) fA[off]
#define PK_B(off) fB[off]
#endif

// WT = MFMA tiles per wave edge: WT = 1 -> 64x64 workgroup tile (32x32 per wave), WT = 2 -> 128x128 (64x64 per wave: half
// the LDS-read and global-load bytes per MFMA -- the kernel is power-limited, energy per MFMA sets the clock).
template <int TRANS_W, int PRO, int EPI, int WT>
__global__ __launch_bounds__(256) void pw_gemm_pk_kernel(PwArgs a, unsigned long long magic_r, unsigned long long magic_c) {
    constexpr int TM = 64 * WT, TN = 64 * WT, BK = 16, WS = 32 * WT;           // WS: wave sub-tile edge
    constexpr int LDA = TRANS_W ? TM : TM + 4, LDB = TN;
    constexpr int TPR = TN / 4;                  // threads per 16-byte-chunked row of a [k][TM | TN] tile
    constexpr int RPP = 256 / TPR;               // k-rows per pass; WT passes cover the BK = 16 rows
    static_assert(RPP * WT == BK, "tile / thread map");
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDA + 2 * BK * LDB];
    float* const As = smem;                       // [2][BK][LDA]
    float* const Bs = smem + 2 * BK * LDA;        // [2][BK][LDB]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- this workgroup's tiles: XCD x owns a contiguous range of tile indices (rows fastest, then column tiles, then
    // utterances), dealt round-robin to its workgroups -- at any time an XCD works on consecutive tiles, whose row tiles
    // re-read the same activation columns from its own L2 ------------------------------------------------------------
    const int ntm = a.tiles_r * a.tiles_c, ntiles = ntm * a.M;
    const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int lo = xcd * q8 + (xcd < r8 ? xcd : r8), cnt = q8 + (xcd < r8 ? 1 : 0);
    const int gx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);                   // workgroups on this XCD
    const int ntl = slot < cnt ? (cnt - slot + gx - 1) / gx : 0;          // tiles of this workgroup
    if (ntl == 0) return;
#ifdef CTN_EXP_CLOCK
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto decode = [&](int j) {
        PkTile t;
        const unsigned id = (unsigned)(lo + slot + j * gx);
        const unsigned u = (unsigned)(((unsigned long long)id * magic_r) >> 32);    // id / tiles_r  (magic = ceil(2^32 / d):
        const unsigned rt = id - u * (unsigned)a.tiles_r;                            //  exact while id * d < 2^32, checked by the host)
        const unsigned mm = (unsigned)(((unsigned long long)u * magic_c) >> 32);     // u / tiles_c
        const unsigned ct = u - mm * (unsigned)a.tiles_c;
        t.m = (int)mm; t.r0 = (int)rt * TM; t.c0 = (int)ct * TN; t.pidx = (int)(ct * (unsigned)a.tiles_r + rt);
        return t;
    };

    // ---- per-lane address registers, once per workgroup -----------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(a.W, (unsigned)a.R * (unsigned)a.Cn * 4u);
    const unsigned xbytes = (unsigned)a.Cn * (unsigned)a.Kp * 4u, obytes = (unsigned)a.R * (unsigned)a.Kp * 4u;
    // A: TRANS_W=1 reads W[c + RPP j][r0 + r4 ..+3]: c = tid / TPR, r4 = (tid % TPR) * 4;
    //    TRANS_W=0 reads W[r0 + r + 64 j][c4 ..+3]: r = tid / 4, c4 = (tid % 4) * 4
    // B: X[i + RPP j][c0 + k4 ..+3]: i = tid / TPR, k4 = (tid % TPR) * 4
    const int a_r4 = (tid % TPR) * 4, rowk = tid / TPR;
    const int voA = TRANS_W ? (rowk * a.R + a_r4) * 4 : ((tid >> 2) * a.Cn + (tid & 3) * 4) * 4;
    const int jA = TRANS_W ? RPP * a.R * 4 : 64 * a.Cn * 4;               // byte step between the WT loads of a thread
    const int voB = (rowk * a.Kp + a_r4) * 4, jB = RPP * a.Kp * 4;
    const int voP = rowk * 4;                                              // gamma / beta of channel i (+ RPP j)
    const int sAk = (TRANS_W ? BK * a.R : BK) * 4, sBk = BK * a.Kp * 4;    // scalar byte steps per k-tile
    __amdgpu_buffer_rsrc_t rsG = rsW, rsBt = rsW;
    if constexpr (PRO == PRO_PRELU_NORM) {
        rsG = make_rsrc(a.pro_gamma, (unsigned)a.Cn * 4u);
        rsBt = make_rsrc(a.pro_beta, (unsigned)a.Cn * 4u);
    }
    // LDS: store slots (A row write / transposing scatter, B row write) and fragment bases
    float* const stA = TRANS_W ? As + rowk * LDA + a_r4 : As + ((tid & 3) * 4) * LDA + (tid >> 2);
    float* const stB = Bs + rowk * LDB + a_r4;
    const float* const fA = As + lhi * LDA + wm * WS + l31;
    const float* const fB = Bs + lhi * LDB + wn * WS + l31;
    // epilogue: element e of a 32x32 C/D map sits at row (e&3) + 8*(e>>2) + 4*lhi, column l31 of the sub-tile
    const int voE = ((wm * WS + 4 * lhi) * a.Kp + wn * WS + l31) * 4;

    float p_alpha = 0.f, e_alpha = 0.f;
    if constexpr (PRO == PRO_PRELU_NORM) p_alpha = a.pro_alpha[0];
    if constexpr (EPI == EPI_PRELU_STATS) e_alpha = a.epi_alpha[0];
    if constexpr (EPI == EPI_GLN_BWD) e_alpha = a.bwd_alpha[0];

    // k-tiles per tile, rounded up to even so that tile boundaries coincide with the two-k-tile software pipeline (an odd
    // count gets one k-tile past the contraction: both operands read zeros there -- buffer range check)
    const int nk = (((a.Cn + BK - 1) / BK) + 1) & ~1;

    struct Stage { float4 a[WT], b[WT]; float2 p[WT]; };

    // ---- load cursor (two k-tiles ahead of the compute cursor; crosses tile boundaries) ------------------------------
    PkTile lt = decode(0);
    int lj = 0, lkt = 0;
    __amdgpu_buffer_rsrc_t rsXl = make_rsrc(a.X + (size_t)lt.m * a.Cn * a.Kp, xbytes);
    auto load_next = [&](Stage& r) {     // loads k-tile (lj, lkt), then advances
        if (lj < ntl) {
            const int sA = (TRANS_W ? lt.r0 : lt.r0 * a.Cn) * 4 + lkt * sAk;
#ifndef CTN_EXP_NO_GLOBAL
#pragma unroll
            for (int j = 0; j < WT; ++j) {
                r.a[j] = buf_ld4(rsW, voA, sA + j * jA);
                r.b[j] = buf_ld4(rsXl, voB, lt.c0 * 4 + lkt * sBk + j * jB);
            }
#else
            (void)sA;
#endif
            if constexpr (PRO == PRO_PRELU_NORM) {
#pragma unroll
                for (int j = 0; j < WT; ++j)
                    r.p[j] = make_float2(buf_ld1(rsG, voP, (lkt * BK + j * RPP) * 4), buf_ld1(rsBt, voP, (lkt * BK + j * RPP) * 4));
            }
            if (++lkt == nk) {
                lkt = 0;
                if (++lj < ntl) {
                    const int m_old = lt.m;
                    lt = decode(lj);
                    if (lt.m != m_old) rsXl = make_rsrc(a.X + (size_t)lt.m * a.Cn * a.Kp, xbytes);
                }
            }
        }
    };

    // ---- store cursor = compute cursor + 1 k-tile: the k-tile being written to LDS (prologue constants follow it) ----
    int sm = -1, skt = 0, sj = 0;
    PkTile stt = lt;
    float p_mean = 0.f, p_rstd = 1.f;
    auto store_tile = [&](int buf, const Stage& r) {       // writes k-tile (sj, skt), then advances
        if (sj >= ntl) return;
        if constexpr (PRO == PRO_PRELU_NORM) {
            if (stt.m != sm) {                       // a new utterance: its (mean, rstd) from the producer's partials
                sm = stt.m;
                finalize_stats_wave(a.pro_part + (size_t)sm * a.pro_nparts * 2, a.pro_nparts, (double)a.Cn * (double)a.K,
                                    p_mean, p_rstd);
            }
            if (a.pro_ms_out != nullptr && skt == 0 && stt.pidx == 0 && tid == 0) {     // the utterance's first tile publishes them
                a.pro_ms_out[2 * sm] = p_mean;
                a.pro_ms_out[2 * sm + 1] = p_rstd;
            }
        }
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            if constexpr (TRANS_W == 1) {
                *reinterpret_cast<float4*>(stA + buf * BK * LDA + j * RPP * LDA) = r.a[j];
            } else {
                float* const d = stA + buf * BK * LDA + j * 64;
                d[0] = r.a[j].x; d[LDA] = r.a[j].y; d[2 * LDA] = r.a[j].z; d[3 * LDA] = r.a[j].w;
            }
            float4 rb = r.b[j];
            if constexpr (PRO == PRO_PRELU_NORM) rb = pro_apply(rb, stt.c0 + a_r4, a.K, r.p[j].x, r.p[j].y, p_alpha, p_mean, p_rstd);
            *reinterpret_cast<float4*>(stB + buf * BK * LDB + j * RPP * LDB) = rb;
        }
        if (++skt == nk) {
            skt = 0;
            if (++sj < ntl) stt = decode(sj);
        }
    };

    // ---- pipeline: LDS buffer g & 1 holds k-tile g of the flattened (tile, k-tile) sequence, register set P the next odd
    // one, Q the next even one ------------------------------------------------------------------------------------------
    Stage P, Q;
#ifdef CTN_EXP_NO_GLOBAL
#pragma unroll
    for (int j = 0; j < WT; ++j) P.a[j] = P.b[j] = Q.a[j] = Q.b[j] = make_float4(1.f, 2.f, 3.f, 4.f);
#endif
#pragma unroll
    for (int j = 0; j < WT; ++j) P.p[j] = Q.p[j] = make_float2(0.f, 0.f);
    load_next(P);                 // g = 0
    store_tile(0, P);
    load_next(P);                 // g = 1 -> P
    load_next(Q);                 // g = 2 -> Q
    PK_SYNC();

    f32x16 acc[WT][WT];
    auto mma = [&](int off_a, int off_b, bool first) {       // one k-tile out of LDS (offsets of the buffer in floats)
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float av[WT], bv[WT];
#pragma unroll
            for (int i = 0; i < WT; ++i) av[i] = PK_A(off_a + 2 * s * LDA + 32 * i);
#pragma unroll
            for (int j = 0; j < WT; ++j) bv[j] = PK_B(off_b + 2 * s * LDB + 32 * j);
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j) {
                    if (first && s == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], zero, 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
                    }
                }
        }
    };
#ifdef CTN_EXP_NO_LDS_READ
    const float xa0 = fA[0], xb0 = fB[0];
#endif

    for (int cj = 0; cj < ntl; ++cj) {
        const PkTile cur = decode(cj);
        // first k-tile pair peeled: the accumulators start from the zero C operand of the first MFMA (no register clears),
        // and the steady-state loop below stays a plain counted loop (the accumulators then live in one register range)
        mma(0, 0, true);
        store_tile(1, P);
        PK_SYNC();
        load_next(P);
        mma(BK * LDA, BK * LDB, false);
        store_tile(0, Q);
        PK_SYNC();
        load_next(Q);
        for (int kt = 2; kt < nk; kt += 2) {
            mma(0, 0, false);
            store_tile(1, P);
            PK_SYNC();
            load_next(P);
            mma(BK * LDA, BK * LDB, false);
            store_tile(0, Q);
            PK_SYNC();
            load_next(Q);
        }

        // ---- epilogue, straight from the accumulators (the next tile's first k-tile is already in LDS, the two after it
        // are in flight) -----------------------------------------------------------------------------------------------
        const size_t mbase = (size_t)cur.m * a.R * a.Kp;
        const __amdgpu_buffer_rsrc_t rsOut = make_rsrc(a.Out + mbase, obytes);
        __amdgpu_buffer_rsrc_t rsAux = rsOut, rsGam = rsOut;
        float b_mean = 0.f, b_rstd = 1.f;
        if constexpr (EPI == EPI_RESIDUAL) rsAux = make_rsrc(a.residual + mbase, obytes);
        if constexpr (EPI == EPI_GLN_BWD) {
            rsAux = make_rsrc(a.bwd_y + mbase, obytes);
            rsGam = make_rsrc(a.bwd_gamma, (unsigned)a.R * 4u);
            b_mean = a.bwd_ms[2 * cur.m];
            b_rstd = a.bwd_ms[2 * cur.m + 1];
        }
        float s1 = 0.f, s2 = 0.f;
        const bool ovr = TRANS_W == 1 && cur.r0 + TM > a.R;       // uniform: rows >= R hold finite garbage (next contraction row)
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int j = 0; j < WT; ++j) {
                const int sE = ((cur.r0 + 32 * i) * a.Kp + cur.c0 + 32 * j) * 4;             // scalar byte offset of the sub-tile
                float aux[16], gam[16];
                if constexpr (EPI == EPI_RESIDUAL || EPI == EPI_GLN_BWD) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) aux[e] = buf_ld1(rsAux, voE, sE + ((e & 3) + 8 * (e >> 2)) * a.Kp * 4);
                }
                if constexpr (EPI == EPI_GLN_BWD) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        gam[e] = buf_ld1(rsGam, (wm * WS + 4 * lhi) * 4, (cur.r0 + 32 * i + (e & 3) + 8 * (e >> 2)) * 4);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[i][j][e];
                    if constexpr (EPI == EPI_RESIDUAL) v += aux[e];
                    if constexpr (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                    if constexpr (EPI == EPI_GLN_BWD) {
                        const float t = gam[e] * v;
                        const float xh = (prelu_f(aux[e], e_alpha) - b_mean) * b_rstd;
                        s1 += t;
                        s2 = fmaf(t, xh, s2);
                    }
#ifndef CTN_EXP_NO_STORE
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsOut, voE, sE + ((e & 3) + 8 * (e >> 2)) * a.Kp * 4, 0);
#else
                    if (v == 12345.678f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsOut, voE, sE, 0);
#endif
                }
                if constexpr (EPI == EPI_PRELU_STATS) {
                    // only the statistics must not see the garbage rows of a row-overhang tile (their stores are dropped by the
                    // range check).  A real branch, kept out of the common path (the asm stops if-conversion into selects).
                    if (ovr) {
                        asm volatile("; row-overhang tile" ::: "memory");
                        const int rl = cur.r0 + wm * WS + 32 * i + 4 * lhi;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float p = (rl + (e & 3) + 8 * (e >> 2) < a.R) ? prelu_f(acc[i][j][e], e_alpha) : 0.f;
                            s1 += p;
                            s2 = fmaf(p, p, s2);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float p = prelu_f(acc[i][j][e], e_alpha);
                            s1 += p;
                            s2 = fmaf(p, p, s2);
                        }
                    }
                }
            }
        if constexpr (EPI == EPI_PRELU_STATS || EPI == EPI_GLN_BWD) {
            s1 = wave_sum(s1);
            s2 = wave_sum(s2);
            if (lane == 0) {
                double* dst = (EPI == EPI_PRELU_STATS ? a.epi_part : a.bwd_part) +
                              (((size_t)cur.m * ntm + cur.pidx) * 4 + wave) * 2;
                dst[0] = (double)s1;
                dst[1] = (double)s2;
            }
        }
    }
#ifdef CTN_EXP_CLOCK
    if (blockIdx.x == 0 && tid == 0) {
        ctn_dbg[0] = __builtin_amdgcn_s_memtime() - dbg_t0;
        ctn_dbg[1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    }
#endif
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

template <int PRO, int MF>      // MF = 32: v_mfma_f32_32x32x2_f32;  16: v_mfma_f32_16x16x4_f32 (see Tile)
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
    // MF = 32: lane (row l % 32, k slot l / 32) reads frames 8j + 4 (l / 32) .. +3 of its row, twice per 16-frame k-tile;
    // MF = 16: lane (row l % 16, k slot l / 16) reads frames 4 (l / 16) .. +3 of its row in each 16-row half of the sub-tile
    const float* const fA = MF == 16 ? &As[0][wm * 32 + (lane & 15)][4 * (lane >> 4)] : &As[0][wm * 32 + l31][4 * lhi];
    const float* const fB = MF == 16 ? &Bs[0][wn * 32 + (lane & 15)][4 * (lane >> 4)] : &Bs[0][wn * 32 + l31][4 * lhi];
    auto compute = [&](int buf) {
        if constexpr (MF == 16) {
            float4 av[2], bv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                av[h] = *reinterpret_cast<const float4*>(fA + buf * 64 * W4LD + h * 16 * W4LD);
                bv[h] = *reinterpret_cast<const float4*>(fB + buf * 64 * W4LD + h * 16 * W4LD);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)          // contraction step: frames 4 g + i of the k-tile, g = lane / 16
#pragma unroll
                for (int q = 0; q < 4; ++q) {    // sub-tile q = 2 si + sj of the wave's 32x32 tile
                    const float4 x = av[q >> 1], y = bv[q & 1];
                    const float xa = i == 0 ? x.x : i == 1 ? x.y : i == 2 ? x.z : x.w;
                    const float yb = i == 0 ? y.x : i == 1 ? y.y : i == 2 ? y.z : y.w;
                    f32x4v c = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, yb, c, 0, 0, 0);
                    acc[4 * q] = c[0]; acc[4 * q + 1] = c[1]; acc[4 * q + 2] = c[2]; acc[4 * q + 3] = c[3];
                }
        } else {
#pragma unroll
            for (int j = 0; j < WK / 8; ++j) {
                const float4 av = *reinterpret_cast<const float4*>(fA + buf * 64 * W4LD + 8 * j);
                const float4 bv = *reinterpret_cast<const float4*>(fB + buf * 64 * W4LD + 8 * j);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
            }
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
    if constexpr (MF == 16) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {          // element 4q + r: row 16 si + 4 (lane / 16) + r, column 16 sj + lane % 16
            const int c = c0 + wn * 32 + 16 * ((e >> 2) & 1) + (lane & 15);
            const int vo = c < a.Cn ? ((r0 + wm * 32 + 4 * (lane >> 4)) * a.Cn + c) * 4 : 0x7fffffff;
            const float v = acc[e];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsS, vo, (16 * (e >> 3) + (e & 3)) * a.Cn * 4, 0);
        }
    } else {
        const int c = c0 + wn * 32 + l31;
        const int voS = c < a.Cn ? ((r0 + wm * 32 + 4 * lhi) * a.Cn + c) * 4 : 0x7fffffff;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = acc[e];       // (clang lowers __builtin_bit_cast of a vector ELEMENT to element 0: go through a scalar)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsS, voS, ((e & 3) + 8 * (e >> 2)) * a.Cn * 4, 0);
        }
    }
}

}  // namespace

#include "ctn_gemm_b3.h"            // the split-bf16 ("b3") arithmetic of the same GEMMs

int g_ctn_tile_override = -2;
extern int g_ctn_block_wt, g_ctn_block_fin_side, g_ctn_block_fuse_b4;          // ctn_block.hip

// GEMM arithmetic (ctn_gemm_b3.h): 2 = "b6" (default: three bf16 pieces per operand, six bf16 MFMAs, fp32 accumulation --
// fp32-faithful products), 1 = "b3" (two pieces, three MFMAs: ~16-bit products, opt-in), 0 = fp32 MFMA (bit-exact fp32 FMA
// chains).  CTN_GEMM_ARITH=b6|b3|fp32, ctn_tune("arith", 2|1|0).  Layers with fewer than 64 output rows (the decoder's basis
// GEMM) and weight gradients with a side below 32 stay on the fp32 kernels.
static int g_arith = -1;
static int arith_id() {
    if (g_arith < 0) {
        const char* e = getenv("CTN_GEMM_ARITH");
        g_arith = (e && !strcmp(e, "fp32")) ? 0 : (e && !strcmp(e, "b3")) ? 1 : 2;
    }
    return g_arith;
}
static int arith_np() { return arith_id() == 0 ? 0 : arith_id() + 1; }       // pieces per operand (0: fp32 MFMA)
static bool b3_fwd(int R) { return arith_id() != 0 && R >= 64; }
static bool b3_wgrad(int R, int Cn) { return arith_id() != 0 && R >= 32 && Cn >= 32; }

template <typename TL>
static void launch_tile(const PwArgs& a, int trans_w, bool pro, bool residual, bool stats, bool relu, bool gln_bwd,
                        hipStream_t st) {
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * a.M)), block(TL::NTH);
    if (gln_bwd) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_NONE, EPI_GLN_BWD>), grid, block, 0, st, a);
    else if (trans_w) {
        if (pro && residual) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_PRELU_NORM, EPI_RESIDUAL>), grid, block, 0, st, a);
        else if (pro) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_PRELU_NORM, EPI_NONE>), grid, block, 0, st, a);
        else if (stats) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_NONE, EPI_PRELU_STATS>), grid, block, 0, st, a);
        else if (residual) hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_NONE, EPI_RESIDUAL>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_gemm_kernel<TL, 1, PRO_NONE, EPI_NONE>), grid, block, 0, st, a);
    } else if (pro) {
        if (residual) hipLaunchKernelGGL((pw_gemm_kernel<TL, 0, PRO_PRELU_NORM, EPI_RESIDUAL>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_gemm_kernel<TL, 0, PRO_PRELU_NORM, EPI_NONE>), grid, block, 0, st, a);
    } else if (stats) hipLaunchKernelGGL((pw_gemm_kernel<TL, 0, PRO_NONE, EPI_PRELU_STATS>), grid, block, 0, st, a);
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

#ifdef CTN_EXP_CLOCK
extern "C" int ctn_debug_read(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ctn_dbg), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif
#ifdef CTN_EXP_B3_TIMELINE
extern "C" int ctn_debug_timeline(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ctn_dbg_tl), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

// ---- persistent kernel: selection and launch ---------------------------------------------------------------
// Default: the one-tile-per-workgroup kernels (pw_gemm_kernel).  The persistent kernels are 5-13 % faster launched alone
// (K1 73 -> 63 us, plain 64 -> 57 us, profiles/r02_*) but do NOT win inside the training step: in-process A/B
// (benchmarks/ab_step.py) 15.60 ms/step with pw_gemm_kernel against 16.0-16.4 ms with the persistent kernels at 3-8
// resident workgroups per CU or one tile per workgroup -- the step is at the 1400 W package power cap (1346-1350 W
// measured), the backward pass is the sum of its four GEMMs, and hardware-dispatched short workgroups interleave better
// with the concurrent weight-gradient stream.  CTN_PW_KERNEL=2 selects them (kept with their tests for the next round).
static int g_pk = -1, g_pk_wgs = 4, g_pk_wt = 1;     // 4 resident workgroups per CU leave room for the concurrent weight-gradient stream
static bool use_pk() {
    if (g_pk < 0) {
        const char* e = getenv("CTN_PW_KERNEL");
        g_pk = (e && *e && atoi(e) == 2) ? 1 : 0;
        const char* w = getenv("CTN_PK_WGS");
        if (w && *w && atoi(w) >= 0 && atoi(w) <= 16) g_pk_wgs = atoi(w);
        const char* t = getenv("CTN_PK_WT");
        if (t && *t && atoi(t) == 2) g_pk_wt = 2;
    }
    return g_pk == 1;
}

template <int TW, int WT>
static void launch_pk_t(const PwArgs& a, bool pro, bool residual, bool stats, bool relu, bool gln_bwd, dim3 grid, hipStream_t st,
                        unsigned long long mr, unsigned long long mc) {
    const dim3 block(256);
    if (gln_bwd) hipLaunchKernelGGL((pw_gemm_pk_kernel<1, PRO_NONE, EPI_GLN_BWD, WT>), grid, block, 0, st, a, mr, mc);
    else if (pro) {
        if (residual) hipLaunchKernelGGL((pw_gemm_pk_kernel<TW, PRO_PRELU_NORM, EPI_RESIDUAL, WT>), grid, block, 0, st, a, mr, mc);
        else hipLaunchKernelGGL((pw_gemm_pk_kernel<TW, PRO_PRELU_NORM, EPI_NONE, WT>), grid, block, 0, st, a, mr, mc);
    } else if (stats) hipLaunchKernelGGL((pw_gemm_pk_kernel<TW, PRO_NONE, EPI_PRELU_STATS, WT>), grid, block, 0, st, a, mr, mc);
    else if (residual) hipLaunchKernelGGL((pw_gemm_pk_kernel<TW, PRO_NONE, EPI_RESIDUAL, WT>), grid, block, 0, st, a, mr, mc);
    else if (relu) hipLaunchKernelGGL((pw_gemm_pk_kernel<TW, PRO_NONE, EPI_RELU, WT>), grid, block, 0, st, a, mr, mc);
    else hipLaunchKernelGGL((pw_gemm_pk_kernel<TW, PRO_NONE, EPI_NONE, WT>), grid, block, 0, st, a, mr, mc);
}

static int launch_pk(PwArgs& a, int trans_w, bool pro, bool residual, bool stats, bool relu, bool gln_bwd, hipStream_t st) {
    const int wt = (trans_w && g_pk_wt == 2 && a.Kp % 128 == 0) ? 2 : 1;      // 128x128 tiles: W^T form, whole column tiles only
    a.tiles_r = ctn_cdiv(a.R, 64 * wt);
    a.tiles_c = ctn_cdiv(a.Kp, 64 * wt);
    const long long ntiles = (long long)a.tiles_r * a.tiles_c * a.M;
    const int tmax = a.tiles_r > a.tiles_c ? a.tiles_r : a.tiles_c;
    CTN_REQUIRE(ntiles * tmax < (1ll << 32), "ctn_pw_gemm: too many tiles for the 32-bit tile decode (%lld x %d)", ntiles, tmax);
    const unsigned long long mr = ((1ull << 32) + (unsigned)a.tiles_r - 1) / (unsigned)a.tiles_r;
    const unsigned long long mc = ((1ull << 32) + (unsigned)a.tiles_c - 1) / (unsigned)a.tiles_c;
    long long g8 = (ntiles + 7) / 8;
    if (g_pk_wgs > 0 && g8 > 32ll * g_pk_wgs) g8 = 32ll * g_pk_wgs;     // 32 CUs per XCD, g_pk_wgs resident workgroups per CU
                                                                        // (g_pk_wgs = 0: one tile per workgroup, hardware dispatch)
    const dim3 grid((unsigned)(8 * g8));
    if (wt == 2) launch_pk_t<1, 2>(a, pro, residual, stats, relu, gln_bwd, grid, st, mr, mc);
    else if (trans_w) launch_pk_t<1, 1>(a, pro, residual, stats, relu, gln_bwd, grid, st, mr, mc);
    else launch_pk_t<0, 1>(a, pro, residual, stats, relu, gln_bwd, grid, st, mr, mc);
    return CTN_OK;
}

static void launch_fwd(PwArgs& a, int trans_w, bool pro, bool residual, bool stats, bool relu, bool gln_bwd, hipStream_t st) {
    const int id = pick_tile(a.M, a.R, a.Kp);
    int tm, tn;
    tile_dims(id, &tm, &tn);
    a.tiles_r = ctn_cdiv(a.R, tm);
    a.tiles_c = ctn_cdiv(a.Kp, tn);
    switch (id) {      // (ids 2 and 4..9, round-1 experiments that lost every A/B, share the instantiations of their nearest shape)
        case 11: launch_tile<T64x64m16>(a, trans_w, pro, residual, stats, relu, gln_bwd, st); break;
        case 3: case 5: launch_tile<T64x64>(a, trans_w, pro, residual, stats, relu, gln_bwd, st); break;
        case 1: case 4: case 6: case 9: launch_tile<T128x64>(a, trans_w, pro, residual, stats, relu, gln_bwd, st); break;
        case 2: launch_tile<T64x128>(a, trans_w, pro, residual, stats, relu, gln_bwd, st); break;
        default: launch_tile<T128x128>(a, trans_w, pro, residual, stats, relu, gln_bwd, st); break;
    }
}

extern "C" {

// experiment / autotune hook: force a tile id (0..4) for every ctn_pw_gemm / ctn_pw_dgrad_gln, -1 = heuristic
int ctn_tune_pw_tile(int id) {
    if (id < -1 || id > 11) return CTN_ERR_ARG;
    g_ctn_tile_override = id;
    return CTN_OK;
}

// internal (ctn_block.hip): 1 when the persistent kernels are active, i.e. fused prologue / statistics work with trans_w = 1
int ctn_pw_uses_pk(void) { return use_pk() ? 1 : 0; }

int ctn_gemm_arith(void) { return arith_id(); }

int ctn_pw_stats_parts(int M, int R, int Kp) {
    if (b3_fwd(R)) {
        int tm, tn;
        ctn_b3_tile_dims(&tm, &tn);
        return ctn_cdiv(R, tm) * ctn_cdiv(Kp, tn);
    }
    if (use_pk()) return ctn_cdiv(R, 64) * ctn_cdiv(Kp, 64) * 4;      // >= one partial per wave of every tile (64x64 tiles: exactly)
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
    a.store_f32 = 1;
    a.W = W; a.X = X; a.Out = Out; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.pro_part = pro_part; a.pro_nparts = pro_nparts; a.pro_gamma = pro_gamma; a.pro_beta = pro_beta;
    a.pro_alpha = pro_alpha; a.pro_ms_out = pro_ms_out;
    a.residual = residual; a.epi_alpha = epi_alpha; a.epi_part = epi_part;
    CTN_REQUIRE(trans_w != 2 || b3_fwd(R), "ctn_pw_gemm: trans_w = 2 (pre-split weight pieces) needs the b3 arithmetic and R >= 64");
    if (b3_fwd(R)) {
        ctn_b3_launch_fwd(arith_np(), a, trans_w, pro_part != nullptr, residual != nullptr, epi_part != nullptr, relu_out != 0, false,
                          (hipStream_t)stream);
    } else if (use_pk()) {
        rc = launch_pk(a, trans_w, pro_part != nullptr, residual != nullptr, epi_part != nullptr, relu_out != 0, false,
                       (hipStream_t)stream);
        if (rc) return rc;
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
    a.store_f32 = 1;
    a.W = W; a.X = dOut; a.Out = dN; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.bwd_y = y; a.bwd_gamma = gamma; a.bwd_alpha = alpha; a.bwd_ms = ms; a.bwd_part = sums_part;
    if (b3_fwd(R)) {
        ctn_b3_launch_fwd(arith_np(), a, planes ? 2 : 1, false, false, false, false, true, (hipStream_t)stream);
    } else if (use_pk()) {
        rc = launch_pk(a, 1, false, false, false, false, true, (hipStream_t)stream);
        if (rc) return rc;
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

static int g_w4_mf = 32;          // MFMA tile of the w4 kernel: 32 (32x32x2) or 16 (16x16x4); ctn_tune("wgrad_mf", ...)
static int g_w4 = -1;             // weight-gradient kernel: 1 = "w4" (16-byte LDS traffic), 0 = the round-1 kernel
static int g_wgrad_tile = 0;      // 0: heuristic; 64, 128: square tiles; 12864: 128 x 64 (ctn_tune_wgrad)
static int g_wgrad_blocks = 512;   // target workgroups per launch

static void wgrad_tile_dims(int code, int* tm, int* tn) {
    *tm = code == 12864 ? 128 : code;
    *tn = code == 12864 ? 64 : code;
}

static void wgrad_plan(int M, int R, int Cn, int Kp, int* tile, int* chunk, int* chunks_per_m) {
    // Small output tiles keep the number of split-K slabs down (slab traffic = splits x R x Cn x 4 B, written once and
    // read once by the reduce kernel).  Measured inside the training step (CTN_WGRAD_TILE / CTN_WGRAD_BLOCKS, one box,
    // alternating): 64x64 / 512 workgroups 497 utt/s, 128x64 (two accumulator chains per wave) 492, 64x64 / 1024 494.
    int wt = g_wgrad_tile ? g_wgrad_tile : ((R >= 64 && Cn >= 64) ? 64 : 128);
    int tm, tn;
    wgrad_tile_dims(wt, &tm, &tn);
    const int tiles = ctn_cdiv(R, tm) * ctn_cdiv(Cn, tn);
    int cpm = ctn_cdiv(g_wgrad_blocks, tiles * M);
    const int max_cpm = ctn_cdiv(Kp, 256);         // keep >= 256 frames of contraction per slab
    if (cpm > max_cpm) cpm = max_cpm;
    if (cpm < 1) cpm = 1;
    int c = ctn_cdiv(ctn_cdiv(Kp, cpm), WK) * WK;
    *tile = wt;
    *chunk = c;
    *chunks_per_m = ctn_cdiv(Kp, c);
}

// experiment hook for in-process A/B runs (benchmarks/ab_step.py): the same switches the CTN_* environment variables set
// once at first use.  Keys: "pk" (1 persistent GEMMs, 0 round-1 kernels), "pk_wgs" (resident workgroups per CU),
// "wgrad_kernel" (1 w4, 0 round-1), "wgrad_blocks" (target workgroups per weight-gradient launch).
int ctn_tune(const char* key, int value) {
    if (!key) return CTN_ERR_ARG;
    use_pk();                                   // read the environment defaults first
    if (!strcmp(key, "pk")) g_pk = value ? 1 : 0;
    else if (!strcmp(key, "pk_wgs") && value >= 0 && value <= 16) g_pk_wgs = value;
    else if (!strcmp(key, "wgrad_kernel")) g_w4 = value ? 1 : 0;
    else if (!strcmp(key, "pw_tile") && value >= -1 && value <= 11) g_ctn_tile_override = value;
    else if (!strcmp(key, "block_wt")) g_ctn_block_wt = value ? 1 : 0;
    else if (!strcmp(key, "fin_side")) g_ctn_block_fin_side = value ? 1 : 0;
    else if (!strcmp(key, "fuse_b4")) g_ctn_block_fuse_b4 = value ? 1 : 0;
    else if (!strcmp(key, "wgrad_mf") && (value == 16 || value == 32)) g_w4_mf = value;
    else if (!strcmp(key, "wgrad_blocks") && value >= 1) g_wgrad_blocks = value;
    else if (!strcmp(key, "arith") && value >= 0 && value <= 2) g_arith = value;
    else if (!strcmp(key, "b3_tile") && value >= 0 && value <= 2) g_ctn_b3_tile = value;
    else if (!strcmp(key, "b3_tile_k3") && value >= 0 && value <= 2) g_ctn_b3_tile_k3 = value;
    else if (!strcmp(key, "b3_wgrad_blocks") && value >= 1) g_ctn_b3_wgrad_blocks = value;
    else { ctn_set_error("ctn_tune: unknown key or bad value: %s=%d", key, value); return CTN_ERR_ARG; }
    return CTN_OK;
}

int ctn_tune_wgrad(int tile, int blocks) {
    if (!(tile == 0 || tile == 64 || tile == 128 || tile == 12864) || blocks < 1) return CTN_ERR_ARG;
    g_wgrad_tile = tile;
    g_wgrad_blocks = blocks;
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
    int rc = check_common("ctn_pw_wgrad", dW, X, (const float*)dOut, M, R, Cn, K, Kp);
    if (rc) return rc;
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
        const int ns = ctn_b3_launch_wgrad(arith_np(), a, pro_ms != nullptr, (hipStream_t)stream);
        CTN_CHECK_LAUNCH("ctn_pw_wgrad");
        const long long nn = (long long)R * Cn;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ctn_cdivll(nn / 4, NT)), dim3(NT), 0, (hipStream_t)stream, a.slab, ns, nn, dW);
        CTN_CHECK_LAUNCH("ctn_pw_wgrad/reduce");
        return CTN_OK;
    }
    int wt;
    wgrad_plan(M, R, Cn, Kp, &wt, &a.chunk, &a.chunks_per_m);
    int wtm, wtn;
    wgrad_tile_dims(wt, &wtm, &wtn);
    a.tiles_r = ctn_cdiv(R, wtm); a.tiles_c = ctn_cdiv(Cn, wtn);
    const int nsplit = M * a.chunks_per_m;
    if (workspace == nullptr || workspace_bytes < (size_t)nsplit * R * Cn * sizeof(float)) {
        ctn_set_error("ctn_pw_wgrad: workspace too small (%zu < %zu)", workspace_bytes, (size_t)nsplit * R * Cn * sizeof(float));
        return CTN_ERR_WORKSPACE;
    }
    a.pro_gamma = pro_gamma; a.pro_beta = pro_beta; a.pro_alpha = pro_alpha; a.pro_ms = pro_ms;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * nsplit)), block(NT);
    if (g_w4 < 0) { const char* e = getenv("CTN_WGRAD_KERNEL"); g_w4 = (e && *e && atoi(e) == 1) ? 0 : 1; }   // 1 = the round-1 kernel (A/B runs)
    const int w4 = g_w4;
    if (wt == 64 && w4 && Kp % WK == 0) {
        if (g_w4_mf == 16) {
            if (pro_ms) hipLaunchKernelGGL((pw_wgrad4_kernel<PRO_PRELU_NORM, 16>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((pw_wgrad4_kernel<PRO_NONE, 16>), grid, block, 0, st, a);
        } else {
            if (pro_ms) hipLaunchKernelGGL((pw_wgrad4_kernel<PRO_PRELU_NORM, 32>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((pw_wgrad4_kernel<PRO_NONE, 32>), grid, block, 0, st, a);
        }
    } else if (wt == 64) {
        if (pro_ms) hipLaunchKernelGGL((pw_wgrad_kernel<PRO_PRELU_NORM, 64, 64>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_wgrad_kernel<PRO_NONE, 64, 64>), grid, block, 0, st, a);
    } else if (wt == 12864) {
        if (pro_ms) hipLaunchKernelGGL((pw_wgrad_kernel<PRO_PRELU_NORM, 128, 64>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_wgrad_kernel<PRO_NONE, 128, 64>), grid, block, 0, st, a);
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
// ---- the gLN'/PReLU' backward pass folded into its two consumers (b3 arithmetic, pre-split weights): see include/ctn_hip.h
int ctn_pw_gemm_glnbwd(const void* Wp, const float* dN, const float* y, float* Out, int M, int R, int Cn, int K, int Kp,
                       const double* sums_part, int nparts, const float* gamma, const float* alpha, const float* ms,
                       const float* residual, void* stream) {
    int rc = check_common("ctn_pw_gemm_glnbwd", (const float*)Wp, dN, Out, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(y && sums_part && nparts > 0 && gamma && alpha && ms && residual, "ctn_pw_gemm_glnbwd: null pointer");
    CTN_REQUIRE(aligned16(y) && aligned16(residual), "ctn_pw_gemm_glnbwd: alignment");
    CTN_REQUIRE(b3_fwd(R), "ctn_pw_gemm_glnbwd: needs the b3 arithmetic and R >= 64");
    PwArgs a{};
    a.store_f32 = 1;
    a.W = (const float*)Wp; a.X = dN; a.Out = Out; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.pro_part = sums_part; a.pro_nparts = nparts; a.pro_gamma = gamma; a.pro_alpha = alpha; a.pro_ms = ms; a.pro_y = y;
    a.residual = residual;
    ctn_b3_launch_fwd(arith_np(), a, 2, false, true, false, false, false, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_pw_gemm_glnbwd");
    return CTN_OK;
}

int ctn_pw_wgrad_glnbwd_parts(int M, int R, int Cn, int Kp) {
    int chunk, cpm;
    ctn_b3_wgrad_plan(M, R, Cn, Kp, &chunk, &cpm);
    return M * cpm * ctn_cdiv(R, BM);
}

int ctn_pw_wgrad_glnbwd(const float* dN, const float* y, const float* X, float* dW, int M, int R, int Cn, int K, int Kp,
                        const double* sums_part, int nparts, const float* gamma, const float* alpha, const float* ms,
                        float* dalpha_part, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common("ctn_pw_wgrad_glnbwd", dW, X, dN, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(y && sums_part && nparts > 0 && gamma && alpha && ms && dalpha_part && aligned16(y), "ctn_pw_wgrad_glnbwd: bad arguments");
    CTN_REQUIRE(b3_wgrad(R, Cn), "ctn_pw_wgrad_glnbwd: needs the b3 arithmetic and R, Cn >= 32");
    WgArgs a{};
    a.dOut = dN; a.X = X; a.slab = (float*)workspace; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.a_y = y; a.a_gamma = gamma; a.a_alpha = alpha; a.a_ms = ms; a.a_part = sums_part; a.a_nparts = nparts; a.dalpha_part = dalpha_part;
    ctn_b3_wgrad_plan(M, R, Cn, Kp, &a.chunk, &a.chunks_per_m);
    const size_t need = (size_t)M * a.chunks_per_m * R * Cn * sizeof(float);
    if (workspace == nullptr || workspace_bytes < need) {
        ctn_set_error("ctn_pw_wgrad_glnbwd: workspace too small (%zu < %zu)", workspace_bytes, need);
        return CTN_ERR_WORKSPACE;
    }
    const int ns = ctn_b3_launch_wgrad(arith_np(), a, false, (hipStream_t)stream);
    CTN_CHECK_LAUNCH("ctn_pw_wgrad_glnbwd");
    const long long nn = (long long)R * Cn;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ctn_cdivll(nn / 4, NT)), dim3(NT), 0, (hipStream_t)stream, a.slab, ns, nn, dW);
    CTN_CHECK_LAUNCH("ctn_pw_wgrad_glnbwd/reduce");
    return CTN_OK;
}
}  // extern "C"
