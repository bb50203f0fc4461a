// Shared device / host pieces of the 1x1-convolution GEMM kernels (fp32-MFMA kernels in ctn_gemm.hip, split-bf16
// kernels in ctn_gemm_b3.h).  gfx950 only.
#pragma once
#include "ctn_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NT = 256;
constexpr int BM = 128, BN = 128;     // weight-gradient tile (below)

enum { PRO_NONE = 0, PRO_PRELU_NORM = 1 };
enum { EPI_NONE = 0, EPI_RESIDUAL = 1, EPI_PRELU_STATS = 2, EPI_GLN_BWD = 3, EPI_RELU = 4, EPI_CLN_BWD = 5, EPI_CLN_STATS = 6, EPI_GLN_BWD2 = 7 };

// Output tile BMxBN per 256-thread workgroup, waves arranged WGM x WGN, each wave (BM/WGM)x(BN/WGN)
// in 32x32 MFMA tiles.  Smaller tiles trade operand reuse (plentiful: one fp32 MFMA = 64 cycles for one
// A and one B dword per lane) for finer load balance over the 256 CUs.
// MF_ = edge of the MFMA instruction tile: 32 -> v_mfma_f32_32x32x2_f32, 16 -> v_mfma_f32_16x16x4_f32 (each 32x32 sub-tile
// of a wave is then 2x2 MFMA tiles).  Same FLOP rate (64 FLOP/clk/SIMD), same operand bytes per FLOP at this blocking,
// but the 16x16x4 form moves half the accumulator bytes per FLOP: a bare loop at 155 TFLOP/s draws 990 W against 1087 W
// (profiles/r02_b_power_lab_mfma.txt) -- and the training step runs AT the 1400 W package power cap, where energy per
// step, not issue rate, sets the step time.
template <int BM_, int BN_, int WGM_, int WGN_, int BK_ = 16, int MF_ = 32>
struct Tile {
    static constexpr int MF = MF_;
    static constexpr int TM = BM_, TN = BN_, WGM = WGM_, WGN = WGN_, TK = BK_;
    static constexpr int WM = BM_ / WGM_, WN = BN_ / WGN_;
    static constexpr int MT = WM / 32, NTL = WN / 32;
    // LDS row pitches: 32x32x2 fragments read 32 consecutive floats of ONE k row per half-wave (pad 4: the transposing
    // scatter of TRANS_W = 0); 16x16x4 fragments read 16 floats of TWO consecutive k rows per half-wave, which must sit
    // 16 banks apart: pitch = 16 mod 32
    static constexpr int LDA = MF_ == 16 ? BM_ + 16 : BM_ + 4, LDB = MF_ == 16 ? BN_ + 16 : BN_ + 4, LDS_ST = WN + 4;
    static constexpr int MAIN_FLOATS = 2 * BK_ * (LDA + LDB);
    static constexpr int NW = WGM_ * WGN_, NTH = 64 * NW;          // waves / threads per workgroup
    static constexpr int STAGE_FLOATS = NW * 32 * LDS_ST;
    static constexpr int SMEM_FLOATS = MAIN_FLOATS > STAGE_FLOATS ? MAIN_FLOATS : STAGE_FLOATS;
    static_assert((NW == 4 || NW == 8) && WM % 32 == 0 && WN % 32 == 0, "4 or 8 waves of 32x32 MFMA tiles");
};
using T128x128 = Tile<128, 128, 2, 2>;
using T128x64 = Tile<128, 64, 2, 2>;
using T64x128 = Tile<64, 128, 2, 2>;
using T64x64 = Tile<64, 64, 2, 2>;

struct PwArgs {
    const float* W;      // TRANS_W=0: [R, Cn]   TRANS_W=1: [Cn, R]
    const float* X;      // [M, Cn, Kp]
    float* Out;          // [M, R, Kp]
    int M, R, Cn, K, Kp;
    int tiles_r, tiles_c;
    // operand prologue: x' = gamma[i]*((prelu(x,alpha)-mean_m)*rstd_m)+beta[i], 0 for k>=K
    const double* pro_part; int pro_nparts;
    const float* pro_gamma; const float* pro_beta; const float* pro_alpha;
    float* pro_ms_out;   // [M,2] (mean, rstd) for the backward pass, optional
    // epilogues
    const float* residual;                         // EPI_RESIDUAL: [M,R,Kp]
    const float* epi_alpha; double* epi_part;      // EPI_PRELU_STATS: [M, tiles_r*tiles_c, 2]
    const float* bwd_y; const float* bwd_gamma; const float* bwd_alpha;
    const float* bwd_ms; double* bwd_part;         // EPI_GLN_BWD
    // EPI_CLN_BWD (channel-wise LayerNorm backward, per-FRAME sums over the rows = channels): bwd_y / bwd_gamma / bwd_alpha as above,
    // the norm's saved per-frame statistics, and the per-row-tile column partials col_part [M][tiles_r][Kp][2] =
    // (sum_c gamma_c dN[c,k], sum_c gamma_c dN[c,k] xhat[c,k]) over the tile's rows (summed over tiles_r by ctn_cln_bwd_frame)
    const float* cln_mean; const float* cln_rstd; double* col_part;
    // EPI_GLN_BWD2 (round 4): EPI_GLN_BWD plus the six sums from which the FIRST norm's backward sums follow by adjointness of the
    // depthwise conv (ctn_pw_dgrad_gln2, include/ctn_hip.h): bwd_part is [M, tiles, 8]; gamma / beta of the first norm, the depthwise
    // taps D [R, dw_P] and its geometry
    const float* g1; const float* b1; const float* dw_D; int dw_P, dw_dil, dw_padl;
    // EPI_CLN_STATS (channel-wise LayerNorm forward statistics from the producing GEMM): epi_alpha, and col_part receives
    // (sum_c p, sum_c p^2), p = prelu(Out[c,k], alpha), over the tile's rows (ctn_cln_stats_frame turns them into mean / rstd)
    // h3 arithmetic (ctn_gemm_b3.h): range information of the operands, all optional elsewhere
    const unsigned* x_amax;   // [M][CTN_AMAX_SLOTS] max |X[m]| as stored (before the prologue): scale of the B operand
    const float* pro_gbmax;   // {max |gamma|, max |beta|} of the prologue's norm
    unsigned* out_amax;       // EPI_RESIDUAL: [M][CTN_AMAX_SLOTS] max |Out[m]| (caller zeroes it), for the GEMMs that read Out next
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Buffer (SRSRC) loads: 32-bit per-lane byte offset + scalar offset, and the hardware range check returns 0 for any
// 16-byte access that ends past `bytes` -- no per-load predicates, zero fills or 64-bit address arithmetic in the
// main loops (VALU instructions do not overlap the MFMAs of the other waves on a SIMD, so every one of them is
// paid in full; see profiles/README.md).
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    // (bit_cast the whole vector: clang lowers __builtin_bit_cast(float, v[i]) of a vector ELEMENT to element 0)
    const f32x4v f = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    return make_float4(f.x, f.y, f.z, f.w);
}
__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// Operand prologue: gLN(prelu(x)) = gamma*((prelu(x)-mean)*rstd)+beta, folded to one select + one FMA per element:
//   gs = gamma*rstd, cc = beta - gs*mean  ->  x' = x * (x >= 0 ? gs : gs*alpha) + cc ;  0 for frames k >= K.
__device__ __forceinline__ float4 pro_apply(float4 v, int k, int K, float g, float b, float alpha, float mean, float rstd) {
    const float gs = g * rstd, cc = b - gs * mean, gn = gs * alpha;
    v.x = fmaf(v.x, v.x >= 0.f ? gs : gn, cc);
    v.y = fmaf(v.y, v.y >= 0.f ? gs : gn, cc);
    v.z = fmaf(v.z, v.z >= 0.f ? gs : gn, cc);
    v.w = fmaf(v.w, v.w >= 0.f ? gs : gn, cc);
    if (k + 3 >= K) {                       // only the last column tile of an utterance
        if (k + 0 >= K) v.x = 0.f;
        if (k + 1 >= K) v.y = 0.f;
        if (k + 2 >= K) v.z = 0.f;
        if (k + 3 >= K) v.w = 0.f;
    }
    return v;
}

// ---- shared epilogue (fp32-MFMA and split-bf16 kernels: the 32x32 C/D register map is dtype-independent) ----
// Each wave transposes its accumulators through a private LDS patch (32 rows at a time) so that global traffic
// is 16 bytes per lane along frames instead of 64 dword accesses; residual / ReLU / PReLU-statistics / gLN-backward
// sums are applied on the float4s.  C/D map: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
// Global traffic goes through buffer descriptors of this utterance's [R, Kp] matrices: rows >= R are dropped /
// read as 0 by the hardware range check (their accumulators are exact zeros, so the statistics need no row
// predicate either); only a tile that overhangs Kp -- a uniform condition -- masks its columns per lane.
template <typename TL, int EPI>
__device__ __forceinline__ void gemm_epilogue(const PwArgs& a, f32x16 (&acc)[TL::MT][TL::NTL], float* smem, double* red,
                                              int m, int rt, int ct) {
    constexpr int MT = TL::MT, NTL = TL::NTL, WM = TL::WM, WN = TL::WN, TM = TL::TM, TN = TL::TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / TL::WGN, wn = wave % TL::WGN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int r0 = rt * TM, c0 = ct * TN;
    float e_alpha = 0.f, b_mean = 0.f, b_rstd = 1.f;
    if constexpr (EPI == EPI_PRELU_STATS) e_alpha = a.epi_alpha[0];
    if constexpr (EPI == EPI_GLN_BWD || EPI == EPI_GLN_BWD2) {
        e_alpha = a.bwd_alpha[0];
        b_mean = a.bwd_ms[2 * m];
        b_rstd = a.bwd_ms[2 * m + 1];
    }
    if constexpr (EPI == EPI_CLN_BWD) e_alpha = a.bwd_alpha[0];
    if constexpr (EPI == EPI_CLN_STATS) e_alpha = a.epi_alpha[0];
    constexpr int LST = TL::LDS_ST;
    constexpr int C4 = WN / 4;              // lanes per staged row
    constexpr int RPP = 64 / C4;            // rows per pass
    float* const stage = smem + wave * 32 * LST;
    float s1 = 0.f, s2 = 0.f, amax = 0.f;
    const size_t mbase = (size_t)m * a.R * a.Kp;
    const unsigned mat_bytes = (unsigned)a.R * (unsigned)a.Kp * 4u;
    const __amdgpu_buffer_rsrc_t rsOut = make_rsrc(a.Out + mbase, mat_bytes);
    __amdgpu_buffer_rsrc_t rsAux = rsOut, rsGam = rsOut;
    if constexpr (EPI == EPI_RESIDUAL) rsAux = make_rsrc(a.residual + mbase, mat_bytes);
    if constexpr (EPI == EPI_GLN_BWD || EPI == EPI_CLN_BWD || EPI == EPI_GLN_BWD2) {
        rsAux = make_rsrc(a.bwd_y + mbase, mat_bytes);
        rsGam = make_rsrc(a.bwd_gamma, (unsigned)a.R * 4u);
    }
    // EPI_GLN_BWD2: per-row constants of the first norm and the depthwise taps through range-checked descriptors (rows >= R read 0)
    __amdgpu_buffer_rsrc_t rsG1 = rsOut, rsB1 = rsOut, rsD = rsOut;
    float q3 = 0.f, q4 = 0.f, q5 = 0.f, q6 = 0.f, q7 = 0.f, q8 = 0.f;
    bool interior = false;
    // per-row constants of the tile in LDS, loaded once per workgroup: (gamma2, gamma1, beta1, sum of the taps) and the taps themselves --
    // per-pass global loads of them (five more VMEM instructions per pass) made this epilogue 14-19 us slower per launch
    __shared__ float rowc[EPI == EPI_GLN_BWD2 ? TM : 1][4];
    __shared__ float rowt[EPI == EPI_GLN_BWD2 ? TM : 1][8];
    if constexpr (EPI == EPI_GLN_BWD2) {
        rsG1 = make_rsrc(a.g1, (unsigned)a.R * 4u);
        rsB1 = make_rsrc(a.b1, (unsigned)a.R * 4u);
        rsD = make_rsrc(a.dw_D, (unsigned)a.R * (unsigned)a.dw_P * 4u);
        // every tap of every frame of this tile inside [0, K): V = sum of the taps (uniform over the workgroup)
        interior = c0 - a.dw_padl >= 0 && c0 + TN - 1 + (a.dw_P - 1) * a.dw_dil - a.dw_padl < a.K;
        for (int r = tid; r < TM; r += TL::NTH) {
            float ts = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float tj = j < a.dw_P ? buf_ld1(rsD, ((r0 + r) * a.dw_P + j) * 4, 0) : 0.f;
                rowt[r][j] = tj;
                ts += tj;
            }
            rowc[r][0] = buf_ld1(rsGam, (r0 + r) * 4, 0);
            rowc[r][1] = buf_ld1(rsG1, (r0 + r) * 4, 0);
            rowc[r][2] = buf_ld1(rsB1, (r0 + r) * 4, 0);
            rowc[r][3] = ts;
        }
        __syncthreads();
    }
    const bool ragged = c0 + TN > a.Kp;     // uniform
    const int rl0 = lane / C4, cl = (lane % C4) * 4;
    const int kcol = c0 + wn * WN + cl;
    // EPI_CLN_BWD: this thread's four frames keep their columns through every pass: per-frame statistics once, column sums in registers
    float4 c_mu = make_float4(0.f, 0.f, 0.f, 0.f), c_rs = c_mu;
    float cs1[4] = {0.f, 0.f, 0.f, 0.f}, cs2[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_CLN_BWD) {
        if (!ragged || kcol < a.Kp) {
            c_mu = ld4(a.cln_mean + (size_t)m * a.Kp + kcol);
            c_rs = ld4(a.cln_rstd + (size_t)m * a.Kp + kcol);
        }
    }
    const int vo0 = ((r0 + wm * WM + rl0) * a.Kp + kcol) * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if constexpr (TL::MF == 16)      // element 4q + r of sub-tile q = 2*si + sj: row 16 si + 4 (lane / 16) + r, column 16 sj + lane % 16
                    stage[(16 * (e >> 3) + 4 * (lane >> 4) + (e & 3)) * LST + nt * 32 + 16 * ((e >> 2) & 1) + (lane & 15)] = acc[mt][nt][e];
                else
                    stage[((e & 3) + 8 * (e >> 2) + 4 * lhi) * LST + nt * 32 + l31] = acc[mt][nt][e];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < 32 / RPP; ++p) {
            const int rl = p * RPP + rl0;
            const int so = (mt * 32 + p * RPP) * a.Kp * 4;                  // scalar byte offset of this pass
            float4 v = *reinterpret_cast<const float4*>(stage + rl * LST + cl);
            if (!ragged || kcol < a.Kp) {
                if constexpr (EPI == EPI_RESIDUAL) {
                    const float4 q = buf_ld4(rsAux, vo0, so);
                    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
                if constexpr (EPI == EPI_RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                if constexpr (EPI == EPI_PRELU_STATS) {
                    // rows >= R of an overhanging tile: exact zeros with TRANS_W = 0 (range-checked weight rows), but the NEXT
                    // contraction row's values with TRANS_W = 1 -- their stores are dropped, the statistics must skip them
                    if (r0 + wm * WM + mt * 32 + rl < a.R) {
                        const float p0 = prelu_f(v.x, e_alpha), p1 = prelu_f(v.y, e_alpha);
                        const float p2 = prelu_f(v.z, e_alpha), p3 = prelu_f(v.w, e_alpha);
                        s1 += (p0 + p1) + (p2 + p3);
                        s2 += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
                    }
                }
                if constexpr (EPI == EPI_GLN_BWD) {
                    const float4 y = buf_ld4(rsAux, vo0, so);
                    const float g = buf_ld1(rsGam, (r0 + wm * WM + rl0) * 4, (mt * 32 + p * RPP) * 4);
                    const float t0 = g * v.x, t1 = g * v.y, t2 = g * v.z, t3 = g * v.w;
                    const float x0 = (prelu_f(y.x, e_alpha) - b_mean) * b_rstd, x1 = (prelu_f(y.y, e_alpha) - b_mean) * b_rstd;
                    const float x2 = (prelu_f(y.z, e_alpha) - b_mean) * b_rstd, x3 = (prelu_f(y.w, e_alpha) - b_mean) * b_rstd;
                    s1 += (t0 + t1) + (t2 + t3);
                    s2 += (t0 * x0 + t1 * x1) + (t2 * x2 + t3 * x3);
                }
                if constexpr (EPI == EPI_GLN_BWD2) {
                    const float4 y = buf_ld4(rsAux, vo0, so);
                    const int rloc = wm * WM + mt * 32 + rl;                    // row of this pass inside the tile
                    const float4 rc = *reinterpret_cast<const float4*>(&rowc[rloc][0]);
                    const float g = rc.x, g1v = rc.y, b1v = rc.z;
                    const float vv[4] = {v.x, v.y, v.z, v.w}, yy[4] = {y.x, y.y, y.z, y.w};
                    // V of this thread's four frames: the sum of the taps whose source frame lies in [0, K)
                    float V[4] = {rc.w, rc.w, rc.w, rc.w};
                    if (!interior) {                                            // (uniform; 2-5 of 50 column tiles)
                        V[0] = V[1] = V[2] = V[3] = 0.f;
#pragma unroll 1
                        for (int j = 0; j < a.dw_P; ++j) {
                            const float tj = rowt[rloc][j];
                            const int k0 = kcol + j * a.dw_dil - a.dw_padl;
#pragma unroll
                            for (int e = 0; e < 4; ++e) V[e] += (k0 + e >= 0 && k0 + e < a.K) ? tj : 0.f;
                        }
                    }
                    // with u = prelu'(y): u y = prelu(y) = pp, so  sum u (y - b1 V) f = sum pp f - b1 sum u V f  for f in {t, 1, xh}
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = g * vv[e], pp = prelu_f(yy[e], e_alpha);
                        const float xh = (pp - b_mean) * b_rstd;
                        s1 += t;
                        s2 += t * xh;
                        const float ok = kcol + e < a.K ? 1.f : 0.f;            // (frames >= K: t = 0, but the terms without t are not)
                        const float u = yy[e] >= 0.f ? ok : ok * e_alpha, uV = u * V[e];
                        const float w1 = g1v * uV, w2 = ok * pp - b1v * uV;     // u g1 V  and  u (y - b1 V)
                        q3 += w1 * t; q4 += w1; q5 += w1 * xh;
                        q6 += w2 * t; q7 += w2; q8 += w2 * xh;
                    }
                }
                if constexpr (EPI == EPI_CLN_BWD) {
                    // (rows >= R: gamma reads 0 through the range check and y reads 0: both sums get exact zeros)
                    const float4 y = buf_ld4(rsAux, vo0, so);
                    const float g = buf_ld1(rsGam, (r0 + wm * WM + rl0) * 4, (mt * 32 + p * RPP) * 4);
                    const float t0 = g * v.x, t1 = g * v.y, t2 = g * v.z, t3 = g * v.w;
                    cs1[0] += t0; cs1[1] += t1; cs1[2] += t2; cs1[3] += t3;
                    cs2[0] += t0 * ((prelu_f(y.x, e_alpha) - c_mu.x) * c_rs.x);
                    cs2[1] += t1 * ((prelu_f(y.y, e_alpha) - c_mu.y) * c_rs.y);
                    cs2[2] += t2 * ((prelu_f(y.z, e_alpha) - c_mu.z) * c_rs.z);
                    cs2[3] += t3 * ((prelu_f(y.w, e_alpha) - c_mu.w) * c_rs.w);
                }
                if constexpr (EPI == EPI_CLN_STATS) {
                    if (r0 + wm * WM + mt * 32 + rl < a.R) {        // (rows >= R of an overhanging tile: see EPI_PRELU_STATS)
                        const float p0 = prelu_f(v.x, e_alpha), p1 = prelu_f(v.y, e_alpha);
                        const float p2 = prelu_f(v.z, e_alpha), p3 = prelu_f(v.w, e_alpha);
                        cs1[0] += p0; cs1[1] += p1; cs1[2] += p2; cs1[3] += p3;
                        cs2[0] += p0 * p0; cs2[1] += p1 * p1; cs2[2] += p2 * p2; cs2[3] += p3 * p3;
                    }
                }
                // The pass offset rides in the per-lane offset, not in an SGPR soffset.  hipcc (ROCm 7.2) takes a 16-byte buffer
                // store with a REGISTER soffset to need no wait state before a VALU write of its data registers and may schedule
                // one right behind it (40 such pairs in the round-3 listings of these epilogues, e.g. buffer_store_dwordx4 v[2:5]
                // .. s4 offen ; v_cvt_f64_f32 v[2:3]); on gfx950 the store can then pick up the NEW value in part of its lanes when
                // the CU is busy (observed in round 4 on pw_gemm_ws_kernel: 16 of 8192 outputs of a tile, profiles/README.md
                // r04_a).  With a zero soffset the compiler's hazard recogniser inserts the wait states itself.
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, f32x4v{v.x, v.y, v.z, v.w}), rsOut, vo0 + so, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (EPI == EPI_RESIDUAL) {
        // (rows >= R / columns >= Kp never reach `amax`: the loop skips them or they are exact zeros)
        if (a.out_amax != nullptr) block_amax_atomic<TL::NTH>(amax, red, a.out_amax + (size_t)m * CTN_AMAX_SLOTS, ct * a.tiles_r + rt);
    }
    if constexpr (EPI == EPI_CLN_BWD || EPI == EPI_CLN_STATS) {
        // column sums of this tile's rows: the thread's rows in fp32 (8-16 terms), then fp64 over the row groups of the wave
        // (lanes C4 apart), then over the row waves through LDS in wave order -- a fixed order: bitwise reproducible
        double dv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { dv[e] = (double)cs1[e]; dv[4 + e] = (double)cs2[e]; }
#pragma unroll
        for (int o = C4; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) dv[e] += __shfl_xor(dv[e], o, 64);
        double* const cs = reinterpret_cast<double*>(smem);         // [WGM][TN][2]; the staging patches are free after the barrier
        __syncthreads();
        if (rl0 == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cs[((wm * TN) + wn * WN + cl + e) * 2] = dv[e];
                cs[((wm * TN) + wn * WN + cl + e) * 2 + 1] = dv[4 + e];
            }
        }
        __syncthreads();
        if (tid < TN && c0 + tid < a.Kp) {
            double q1 = cs[tid * 2], q2 = cs[tid * 2 + 1];
#pragma unroll
            for (int w = 1; w < TL::WGM; ++w) { q1 += cs[(w * TN + tid) * 2]; q2 += cs[(w * TN + tid) * 2 + 1]; }
            double* const dst = a.col_part + (((size_t)m * a.tiles_r + rt) * a.Kp + c0 + tid) * 2;
            dst[0] = q1;
            dst[1] = q2;
        }
    }
    if constexpr (EPI == EPI_GLN_BWD2) {
        // (rows >= R: gamma2 and gamma1 read 0 and y reads 0 -> t = 0, w1 = 0, w2 = -b1 V = 0 as beta1 reads 0 too: exact zeros)
        // the eight sums in ONE block reduction (two barriers instead of sixteen): fp64 over the lanes, then over the waves in wave order
        const float qs[8] = {s1, s2, q3, q4, q5, q6, q7, q8};
        double* const sc8 = reinterpret_cast<double*>(smem);         // [8][NW]; the staging patches are free after the barrier
        double w8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w8[i] = wave_sum((double)qs[i]);
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sc8[i * TL::NW + wave] = w8[i];
        }
        __syncthreads();
        if (tid < 8) {
            double r = sc8[tid * TL::NW];
#pragma unroll
            for (int w = 1; w < TL::NW; ++w) r += sc8[tid * TL::NW + w];
            a.bwd_part[((size_t)m * (a.tiles_r * a.tiles_c) + (size_t)ct * a.tiles_r + rt) * 8 + tid] = r;
        }
    }
    if constexpr (EPI == EPI_PRELU_STATS || EPI == EPI_GLN_BWD) {
        const double d1 = block_sum<double, TL::NTH>((double)s1, red);
        const double d2 = block_sum<double, TL::NTH>((double)s2, red);
        if (tid == 0) {
            double* dst = (EPI == EPI_PRELU_STATS ? a.epi_part : a.bwd_part) +
                          ((size_t)m * (a.tiles_r * a.tiles_c) + (size_t)ct * a.tiles_r + rt) * 2;
            dst[0] = d1;
            dst[1] = d2;
        }
    }
}

// ---------------------------------------------------------------------------
// weight gradient: dW[r,c] = sum_{m,k} dOut[m,r,k] * f(X[m,c,k]); split over (m, k-chunks)
// into fp32 slabs that a second kernel sums in a fixed order (bitwise reproducible).
// ---------------------------------------------------------------------------
constexpr int WK = 16, LDW = 17;

struct WgArgs {
    const float* dOut;   // [M, R, Kp]
    const float* X;      // [M, Cn, Kp]
    float* slab;         // [nsplit, R, Cn]
    int M, R, Cn, K, Kp;
    int tiles_r, tiles_c, chunk, chunks_per_m;
    const float* pro_gamma; const float* pro_beta; const float* pro_alpha; const float* pro_ms;  // [M,2]
    // h3 arithmetic: [M][CTN_AMAX_SLOTS] max |dOut[m]| / max |X[m]| (as stored), {max |gamma|, max |beta|} of the prologue
    const unsigned* g_amax; const unsigned* x_amax; const float* pro_gbmax;
    // chained launches (ctn_wgrad_chain, ctn_common.h): the slabs of the PREVIOUS weight gradient on this stream, summed by this
    // launch's workgroups while their first tiles are in flight (nullptr: nothing pending)
    const float* prev_slab; float* prev_out; long long prev_n; int prev_nsplit;
};


// out[i] = ((slab[0][i] + slab[1][i]) + slab[2][i]) + ...  (fixed order -> bitwise reproducible).  One float4 per thread,
// eight slab loads in flight per thread: 8 MiB of slabs are summed in ~5 us instead of the 22 us of a scalar loop.
__global__ __launch_bounds__(NT) void slab_reduce_kernel(const float* __restrict__ slab, int nsplit, long long n,
                                                         float* __restrict__ out) {
    const long long i = ((long long)blockIdx.x * NT + threadIdx.x) * 4;
    if (i >= n) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(slab + (size_t)(k + j) * n + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
    }
    for (; k < nsplit; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)k * n + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i) = s;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_common(const char* fn, const float* W, const float* X, const float* Out, int M, int R, int Cn, int K, int Kp) {
    CTN_REQUIRE(W && X && Out, "%s: null pointer", fn);
    CTN_REQUIRE(M > 0 && R > 0 && Cn > 0 && K > 0 && Kp >= K, "%s: bad sizes M=%d R=%d Cn=%d K=%d Kp=%d", fn, M, R, Cn, K, Kp);
    CTN_REQUIRE(Kp % 4 == 0 && R % 4 == 0 && Cn % 4 == 0, "%s: Kp, rows and contraction must be multiples of 4 (Kp=%d R=%d Cn=%d)", fn, Kp, R, Cn);
    CTN_REQUIRE(aligned16(W) && aligned16(X) && aligned16(Out), "%s: pointers must be 16-byte aligned", fn);
    CTN_REQUIRE((long long)R * Cn * 4 < (1ll << 31) && (long long)(Cn > R ? Cn : R) * Kp * 4 < (1ll << 31),
                "%s: one weight matrix / one utterance's activations must stay below 2 GiB (32-bit buffer offsets)", fn);
    return CTN_OK;
}


}  // namespace

// ---- tile selection of the fp32-MFMA forward / input-gradient kernel (ctn_tune("pw_tile", id) / CTN_PW_TILE override it) ----
// id: 0 = 128x128, 1 = 128x64, 2 = 64x128, 3 = 64x64
extern int g_ctn_tile_override;   // -2: not read yet, -1: heuristic (defined in ctn_gemm.hip)

static void tile_dims(int id, int* tm, int* tn) {
    static const int d[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    *tm = d[id][0];
    *tn = d[id][1];
}

static int pick_tile(int M, int R, int Kp) {
    if (g_ctn_tile_override == -2) {
        const char* e = getenv("CTN_PW_TILE");
        g_ctn_tile_override = (e && *e) ? atoi(e) : -1;
        if (g_ctn_tile_override < -1 || g_ctn_tile_override > 3) g_ctn_tile_override = -1;
    }
    if (g_ctn_tile_override >= 0) return g_ctn_tile_override;
    // Measured on MI355X (benchmarks/gemm_sweep.py, paper shapes): 64x64 tiles win every variant -- 3200 / 1600
    // workgroups balance over the 256 CUs far better than 800 / 400 tiles of 128x128, and one fp32 MFMA (64 cycles)
    // needs so little operand bandwidth that the smaller tile's lower reuse costs nothing.
    (void)M; (void)R; (void)Kp;
    return 3;
}
