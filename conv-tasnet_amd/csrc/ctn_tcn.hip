// HBM-bound kernels of the temporal-conv blocks, gfx950.
//
//  * depthwise dilated conv forward/backward (src/conv_tasnet.py:247-295) with the
//    neighbouring PReLU + global-LayerNorm (:224-225, :259-260, :338-361) fused in:
//    one wave owns one (utterance, channel) row, frames on the lanes (coalesced
//    256-B / 1-KiB wave accesses), the P-tap dilated window is served from an
//    LDS copy of the row segment (+halo) that already holds the normalised values.
//  * element-wise gLN+PReLU backward.
//  * channel-wise LayerNorm (:313-335) forward / backward for the causal variant
//    and the input norm (:172).
//  * small fixed-order reductions for per-channel parameter gradients.
//
// All cross-lane / cross-block sums have a fixed order (no float atomics), so a
// step is bitwise reproducible run to run.
#include "ctn_common.h"
#include <stdlib.h>

namespace {

constexpr int NT = 256;
constexpr int ROWS = 4;          // one wave per row
constexpr int MAXP = 8;          // max depthwise kernel size supported
// LDS floats per wave for the row segment (+halo).  Small buffers when the receptive field is short (more
// workgroups per CU in flight = more HBM requests outstanding), large ones when the halo would dominate.
constexpr int FWD_BUF_S = 1024, FWD_BUF_L = 3584;    // forward: 16 / 56 KiB per workgroup
constexpr int BWD_BUF_S = 768, BWD_BUF_M = 1280, BWD_BUF_L = 1792;     // backward (two arrays): 24 / 40 / 56 KiB per workgroup

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ int floor4(int v) { return (v >> 2) << 2; }  // arithmetic shift: floors negatives

// ---------------------------------------------------------------------------
// depthwise forward:  z[k] = sum_j D[c,j] * n[k + j*dil - padl],  n = (PRO ? gLN(prelu(y)) : y)
// ---------------------------------------------------------------------------
struct DwFwdArgs {
    const float* Y; float* Z; const float* D;
    int M, H, K, Kp, P, dil, padl, seg;
    const double* pro_part; int pro_nparts;
    const float* pro_gamma; const float* pro_beta; const float* pro_alpha; float* pro_ms_out;
    const float* epi_alpha; double* epi_part;   // [M, H, 2]
    unsigned* amax_out;                         // EPI: [M][CTN_AMAX_SLOTS] max |Z[m]| (h3 arithmetic of the GEMM that reads Z), optional
    const float* cln_mean; const float* cln_rstd;   // PRO = 2: [M][Kp] per-frame statistics of the channel-wise LayerNorm
};

// PRO: 0 n = y, 1 n = gLN(prelu(y)) (per-utterance statistics), 2 n = cLN(prelu(y)) (per-frame statistics, round 4)
template <int PRO, bool EPI, int FWD_BUF, bool VEC4, int PT>      // PT: compile-time kernel size (3) or 0 = a.P at run time
__global__ __launch_bounds__(NT) void dw_fwd_kernel(DwFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float buf[ROWS][FWD_BUF];
    __shared__ double red[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hb = (a.H + ROWS - 1) / ROWS;
    const int m = blockIdx.x / hb;
    const int c = (blockIdx.x % hb) * ROWS + wave;
    const bool live = c < a.H;
    const size_t row = ((size_t)m * a.H + (live ? c : 0)) * a.Kp;
    const float* __restrict__ y = a.Y + row;
    float* __restrict__ z = a.Z + row;
    float* __restrict__ L = buf[wave];

    float mean = 0.f, rstd = 1.f, alpha = 0.f, g = 1.f, b = 0.f;
    const float* __restrict__ cmu = PRO == 2 ? a.cln_mean + (size_t)m * a.Kp : nullptr;
    const float* __restrict__ crs = PRO == 2 ? a.cln_rstd + (size_t)m * a.Kp : nullptr;
    if constexpr (PRO == 2) {
        alpha = a.pro_alpha[0];
        if (live) { g = a.pro_gamma[c]; b = a.pro_beta[c]; }
    }
    if constexpr (PRO == 1) {
        finalize_stats<NT>(a.pro_part + (size_t)m * a.pro_nparts * 2, a.pro_nparts, (double)a.H * (double)a.K, red,
                           mean, rstd);
        alpha = a.pro_alpha[0];
        if (live) { g = a.pro_gamma[c]; b = a.pro_beta[c]; }
        if (a.pro_ms_out != nullptr && (blockIdx.x % hb) == 0 && tid == 0) {
            a.pro_ms_out[2 * m] = mean;
            a.pro_ms_out[2 * m + 1] = rstd;
        }
    }
    const float gs = g * rstd, cc = b - gs * mean, gn = gs * alpha;
    constexpr int NP = PT ? PT : MAXP;
    const int P_ = PT ? PT : a.P;
    float taps[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) taps[j] = (live && j < P_) ? a.D[(size_t)c * P_ + j] : 0.f;
    float e_alpha = 0.f;
    if constexpr (EPI) e_alpha = a.epi_alpha[0];
    const int halo = (P_ - 1) * a.dil;
    float s1 = 0.f, s2 = 0.f, amax = 0.f;

    for (int k0 = 0; k0 < a.Kp; k0 += a.seg) {
        const int kend = min(k0 + a.seg, a.Kp);
        const int base = floor4(k0 - a.padl);
        const int nfill = (kend - k0) + halo + 4;           // covers idx up to (kend-1-base-padl)+halo
        for (int j = lane * 4; j < nfill; j += 256) {
            const int k = base + j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live && k >= 0 && k < a.Kp) {
                v = ld4(y + k);
                if constexpr (PRO == 2) {  // gamma ((prelu(x) - mean[k]) rstd[k]) + beta: the order of cln_fwd_v4_kernel
                    const float4 mu = ld4(cmu + k), rs = ld4(crs + k);
                    v.x = k + 0 < a.K ? g * ((prelu_f(v.x, alpha) - mu.x) * rs.x) + b : 0.f;
                    v.y = k + 1 < a.K ? g * ((prelu_f(v.y, alpha) - mu.y) * rs.y) + b : 0.f;
                    v.z = k + 2 < a.K ? g * ((prelu_f(v.z, alpha) - mu.z) * rs.z) + b : 0.f;
                    v.w = k + 3 < a.K ? g * ((prelu_f(v.w, alpha) - mu.w) * rs.w) + b : 0.f;
                }
                if constexpr (PRO == 1) {  // gamma*((prelu(x)-mean)*rstd)+beta as one select + one FMA per element
                    v.x = fmaf(v.x, v.x >= 0.f ? gs : gn, cc);
                    v.y = fmaf(v.y, v.y >= 0.f ? gs : gn, cc);
                    v.z = fmaf(v.z, v.z >= 0.f ? gs : gn, cc);
                    v.w = fmaf(v.w, v.w >= 0.f ? gs : gn, cc);
                    if (k + 3 >= a.K) {
                        if (k + 0 >= a.K) v.x = 0.f;
                        if (k + 1 >= a.K) v.y = 0.f;
                        if (k + 2 >= a.K) v.z = 0.f;
                        if (k + 3 >= a.K) v.w = 0.f;
                    }
                }
            }
            *reinterpret_cast<float4*>(L + j) = v;
        }
        __builtin_amdgcn_wave_barrier();   // the LDS patch is private to this wave; DS ops of one wave retire in order
        if constexpr (VEC4) {
            // dilation and pad are multiples of 4: every tap of 4 consecutive frames is one aligned 16-byte LDS read
            // and the result leaves as a float4 (1 KiB per wave store)
            for (int k = k0 + lane * 4; k < kend; k += 256) {
                const int idx = k - base - a.padl;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    if (PT || j < a.P) {
                        const float4 t = *reinterpret_cast<const float4*>(L + idx + j * a.dil);
                        acc.x += taps[j] * t.x; acc.y += taps[j] * t.y; acc.z += taps[j] * t.z; acc.w += taps[j] * t.w;
                    }
                if (k + 3 >= a.K) {
                    if (k + 0 >= a.K) acc.x = 0.f;
                    if (k + 1 >= a.K) acc.y = 0.f;
                    if (k + 2 >= a.K) acc.z = 0.f;
                    if (k + 3 >= a.K) acc.w = 0.f;
                }
                if constexpr (EPI) {
                    const float p0 = prelu_f(acc.x, e_alpha), p1 = prelu_f(acc.y, e_alpha);
                    const float p2 = prelu_f(acc.z, e_alpha), p3 = prelu_f(acc.w, e_alpha);
                    s1 += (p0 + p1) + (p2 + p3);
                    s2 += (p0 * p0 + p1 * p1) + (p2 * p2 + p3 * p3);
                    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
                }
                if (live) *reinterpret_cast<float4*>(z + k) = acc;
            }
        } else {
        for (int k = k0 + lane; k < kend; k += 64) {
            const int idx = k - base - a.padl;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < NP; ++j)
                if (PT || j < a.P) acc += taps[j] * L[idx + j * a.dil];
            if (k >= a.K) acc = 0.f;
            if constexpr (EPI) {
                const float p = prelu_f(acc, e_alpha);
                s1 += p;
                s2 += p * p;
                amax = fmaxf(amax, fabsf(acc));
            }
            if (live) z[k] = acc;
        }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (EPI) {
        const double d1 = wave_sum((double)s1), d2 = wave_sum((double)s2);
        if (live && lane == 0) {
            double* dst = a.epi_part + ((size_t)m * a.H + c) * 2;
            dst[0] = d1;
            dst[1] = d2;
        }
        if (a.amax_out != nullptr) block_amax_atomic<NT>(amax, red, a.amax_out + (size_t)m * CTN_AMAX_SLOTS, blockIdx.x % hb);      // (dead waves hold 0)
    }
}

// ---------------------------------------------------------------------------
// depthwise backward.
//   FUSED: given dN2 (grad of the 2nd norm's output), d (= dw output, pre-PReLU) and h1
//          (= first 1x1 output, pre-PReLU):
//            xh2 = (prelu(d)-mean2)*rstd2 ; da2 = rstd2*(g2*dN2 - S1/n - xh2*S2/n) ; dd = da2*prelu'(d)
//            n1  = g1*xh1+b1, xh1 = (prelu(h1)-mean1)*rstd1
//            dN1[k] = sum_j D[j]*dd[k - j*dil + padl] ;  dD[j] = sum_k dd[k]*n1[k + j*dil - padl]
//          plus every per-channel / per-utterance sum the two norms and PReLUs need.
//   PLAIN: dd = dZ, n1 = X as stored.
//   CLN (round 4; channel-wise LayerNorm, the causal config): dd as in FUSED but with PER-FRAME constants of the second norm --
//          fc [M][4][Kp] = (rstd2, mean2 rstd2, rstd2 S1/H, rstd2 S2/H)[k] from ctn_cln_bwd_frame (S1, S2: the per-frame sums over
//          channels that the input-gradient GEMM's epilogue produced) -- and n1 = X as stored (the first norm's output):
//            xh2 = prelu(d) fc0 - fc1 ; da2 = g2 fc0 dN2 - fc2 - xh2 fc3 ; dd = da2 * prelu'(d)
//          i.e. the whole stand-alone cLN-backward pass of the second norm (three tensor passes) rides in this kernel's dd image.
// Template: DDM = how dd is formed (0 plain, 1 gLN, 2 cLN), XM = how the x image is formed (0 as stored, 1 gLN-1 recomputed from h1,
// 2 cLN-1 recomputed from h1 with its per-frame statistics: the first norm's output is never stored; 3 (with DDM = 1, round 4) as 1 AND
// the first norm's own backward applied to the result: the kernel writes dh1 = gLN1' . PReLU1'(dn1) instead of dn1 -- its two sums
// S1', S2' are known BEFORE this kernel runs (ctn_pw_dgrad_gln2: sums2_part is [M, parts, 8]), so the gln_prelu_bwd pass is gone;
// pc gets a row P+5 with the dalpha1 partials).
// per-row float outputs pc[f][m][c]:  f = 0..P-1: dD ; (DDM, XM) = (1, 1) adds P: dgamma2, P+1: dbeta2,
//   P+2: dgamma1, P+3: dbeta1, P+4: dalpha2 ; (2, 0) adds P: dgamma2, P+1: dbeta2, P+2: dalpha2
// ---------------------------------------------------------------------------
struct DwBwdArgs {
    const float* dN2; const float* Dz; const float* Y1; float* dN1; const float* D;
    int M, H, K, Kp, P, dil, padl, seg;
    const float* g1; const float* b1; const float* a1; const float* ms1;
    const float* g2; const float* a2; const float* ms2;
    const double* sums2_part; int sums2_nparts;
    float* pc;             // [F, M, H]
    double* sums1_part;    // [M, H, 2]
    const float* fc2;      // DDM = 2: [M][4][Kp] per-frame constants of the second norm's backward
    const float* mean1f; const float* rstd1f;      // XM = 2: [M][Kp] per-frame statistics of the first (channel-wise) norm
    unsigned* amax_out;    // XM = 3: [M][CTN_AMAX_SLOTS] max |dY1[m]| (h3 arithmetic of the GEMMs that read it), optional
};

template <int DDM, int XM, int BWD_BUF, bool VEC4, int PT>
__global__ __launch_bounds__(NT) void dw_bwd_kernel(DwBwdArgs a) {
    static_assert((DDM == 0 && XM == 0) || (DDM == 1 && (XM == 1 || XM == 3)) || (DDM == 2 && (XM == 0 || XM == 2)), "supported forms");
    constexpr bool APPLY = XM == 3;         // write dh1 (first norm's backward applied) instead of dn1
    constexpr bool FUSED = DDM == 1;        // (the gLN form couples both images)
    constexpr bool XHAT = XM != 0;          // the x image holds xhat1: gamma1 / beta1 are applied where it is read
    __shared__ __attribute__((aligned(16))) float bufA[ROWS][BWD_BUF];  // dd
    __shared__ __attribute__((aligned(16))) float bufB[ROWS][BWD_BUF];  // xh1 (FUSED) or x (PLAIN)
    __shared__ double red[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hb = (a.H + ROWS - 1) / ROWS;
    const int m = blockIdx.x / hb;
    const int c = (blockIdx.x % hb) * ROWS + wave;
    const bool live = c < a.H;
    const size_t row = ((size_t)m * a.H + (live ? c : 0)) * a.Kp;
    const float* __restrict__ dn2 = a.dN2 + row;
    const float* __restrict__ dz = DDM != 0 ? a.Dz + row : nullptr;
    const float* __restrict__ fcm = DDM == 2 ? a.fc2 + (size_t)m * 4 * a.Kp : nullptr;
    const float* __restrict__ y1 = a.Y1 + row;
    float* __restrict__ dn1 = a.dN1 + row;
    float* __restrict__ LA = bufA[wave];
    float* __restrict__ LB = bufB[wave];

    float mean1 = 0.f, rstd1 = 1.f, mean2 = 0.f, rstd2 = 1.f, al1 = 0.f, al2 = 0.f;
    float g1 = 1.f, b1 = 0.f, g2 = 1.f, c1 = 0.f, c2 = 0.f;
    float c1p = 0.f, c2p = 0.f;            // APPLY: S1' / n, S2' / n of the first norm
    if constexpr (FUSED) {
        constexpr int NS = APPLY ? 8 : 2;
        double S[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) S[q] = 0.0;
        for (int i = tid; i < a.sums2_nparts; i += NT)
#pragma unroll
            for (int q = 0; q < NS; ++q) S[q] += a.sums2_part[((size_t)m * a.sums2_nparts + i) * NS + q];
        if constexpr (APPLY) {          // the eight sums in one reduction (two barriers): fp64 over the lanes, then over the waves in order
            __shared__ double red8[8][NT / 64];
#pragma unroll
            for (int q = 0; q < NS; ++q) S[q] = wave_sum(S[q]);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < NS; ++q) red8[q][wave] = S[q];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                double r = red8[q][0];
#pragma unroll
                for (int w2 = 1; w2 < NT / 64; ++w2) r += red8[q][w2];
                S[q] = r;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NS; ++q) S[q] = block_sum<double, NT>(S[q], red);
        }
        const double n = (double)a.H * (double)a.K;
        const double d1 = S[0] / n, d2 = S[1] / n;
        c1 = (float)d1;
        c2 = (float)d2;
        if constexpr (APPLY) {          // S1' = rstd2 (A1 - c1 B1 - c2 C1), S2' = rstd2 (A2 - c1 B2 - c2 C2)   (ctn_pw_dgrad_gln2)
            const double r2 = (double)a.ms2[2 * m + 1];
            c1p = (float)(r2 * (S[2] - d1 * S[3] - d2 * S[4]) / n);
            c2p = (float)(r2 * (S[5] - d1 * S[6] - d2 * S[7]) / n);
        }
        mean1 = a.ms1[2 * m]; rstd1 = a.ms1[2 * m + 1];
        mean2 = a.ms2[2 * m]; rstd2 = a.ms2[2 * m + 1];
        al1 = a.a1[0]; al2 = a.a2[0];
        if (live) { g1 = a.g1[c]; b1 = a.b1[c]; g2 = a.g2[c]; }
    }
    if constexpr (DDM == 2) {
        al2 = a.a2[0];
        if (live) g2 = a.g2[c];
    }
    const float* __restrict__ mu1f = XM == 2 ? a.mean1f + (size_t)m * a.Kp : nullptr;
    const float* __restrict__ rs1f = XM == 2 ? a.rstd1f + (size_t)m * a.Kp : nullptr;
    if constexpr (XM == 2) {
        al1 = a.a1[0];
        if (live) { g1 = a.g1[c]; b1 = a.b1[c]; }
    }
    constexpr int NP = PT ? PT : MAXP;
    const int P_ = PT ? PT : a.P;
    float taps[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) taps[j] = (live && j < P_) ? a.D[(size_t)c * P_ + j] : 0.f;
    const int halo = (P_ - 1) * a.dil;

    // folded constants of the two norms: xh = x*(x>=0 ? rstd : alpha*rstd) - mean*rstd ;  da = rg2*dn2 - rc1 - xh*rc2
    const float ar1 = al1 * rstd1, mr1 = mean1 * rstd1, ar2 = al2 * rstd2, mr2 = mean2 * rstd2;
    const float rg2 = rstd2 * g2, rc1 = rstd2 * c1, rc2 = rstd2 * c2;
    float dD[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) dD[j] = 0.f;
    float dg2 = 0.f, db2 = 0.f, dal2 = 0.f, dg1 = 0.f, db1 = 0.f, t1 = 0.f, t2 = 0.f;
    float dal1 = 0.f, amax1 = 0.f;
    const float rg1 = rstd1 * g1, rc1p = rstd1 * c1p, rc2p = rstd1 * c2p;       // APPLY: da1 = rstd1 (g1 dn1 - c1' - xh1 c2')
    // dh1 of one element: dn = dn1[k], xh = xhat1[k], h = h1[k] (its sign selects the PReLU branch); k < K
    auto apply1 = [&](float dn, float xh, float h) -> float {
        const float da = fmaf(-xh, rc2p, fmaf(rg1, dn, -rc1p));
        if (h < 0.f) dal1 += da * h;
        const float o = h >= 0.f ? da : al1 * da;
        amax1 = fmaxf(amax1, fabsf(o));
        return o;
    };

    for (int k0 = 0; k0 < a.Kp; k0 += a.seg) {
        const int kend = min(k0 + a.seg, a.Kp);
        // dd is needed on [k0 + padl - halo, kend-1 + padl]; x on [k0 - padl, kend-1 - padl + halo]
        const int baseA = floor4(k0 + a.padl - halo);
        const int baseB = floor4(k0 - a.padl);
        const int nfill = (kend - k0) + halo + 4;
        // dd image.  Frames of [k0, kend) belong to this segment: their parameter-gradient sums are taken here, and
        // when the segment lies inside [0, K) -- a uniform condition -- nothing in the loop is predicated.  The halo
        // frames on either side are transformed only (the neighbouring segment owns their sums).
        auto fill_dd = [&](int kb, int ke, const bool own, const bool all_valid) {
            for (int k = kb + lane * 4; k < ke; k += 256) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live && k >= 0 && k < a.Kp) {
                    v = ld4(dn2 + k);
                    if constexpr (FUSED) {
                        const float4 d = ld4(dz + k);
                        float vv[4] = {v.x, v.y, v.z, v.w};
                        const float dd_[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool valid = all_valid || (k + e) < a.K;
                            const float xh = fmaf(dd_[e], dd_[e] >= 0.f ? rstd2 : ar2, -mr2);    // (prelu(d)-mean2)*rstd2
                            const float da = fmaf(-xh, rc2, fmaf(rg2, vv[e], -rc1));            // rstd2*(g2*dn2 - c1 - xh*c2)
                            if (own && valid) {
                                dg2 += vv[e] * xh;
                                db2 += vv[e];
                                dal2 += dd_[e] < 0.f ? da * dd_[e] : 0.f;
                            }
                            vv[e] = valid ? (dd_[e] >= 0.f ? da : al2 * da) : 0.f;
                        }
                        v = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    }
                    if constexpr (DDM == 2) {
                        const float4 d = ld4(dz + k);
                        const float4 f0 = ld4(fcm + k), f1 = ld4(fcm + a.Kp + k), f2 = ld4(fcm + 2 * a.Kp + k), f3 = ld4(fcm + 3 * a.Kp + k);
                        float vv[4] = {v.x, v.y, v.z, v.w};
                        const float dd_[4] = {d.x, d.y, d.z, d.w};
                        const float q0[4] = {f0.x, f0.y, f0.z, f0.w}, q1[4] = {f1.x, f1.y, f1.z, f1.w};
                        const float q2[4] = {f2.x, f2.y, f2.z, f2.w}, q3[4] = {f3.x, f3.y, f3.z, f3.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool valid = all_valid || (k + e) < a.K;
                            const float xh = fmaf(dd_[e], dd_[e] >= 0.f ? q0[e] : al2 * q0[e], -q1[e]);      // (prelu(d) - mean2[k]) rstd2[k]
                            const float da = fmaf(-xh, q3[e], fmaf(q0[e] * g2, vv[e], -q2[e]));               // rstd2 (g2 dn2 - S1/H - xh S2/H)
                            if (own && valid) {
                                dg2 += vv[e] * xh;
                                db2 += vv[e];
                                dal2 += dd_[e] < 0.f ? da * dd_[e] : 0.f;
                            }
                            vv[e] = valid ? (dd_[e] >= 0.f ? da : al2 * da) : 0.f;
                        }
                        v = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    }
                }
                *reinterpret_cast<float4*>(LA + (k - baseA)) = v;
            }
        };
        const int endA = baseA + ((nfill + 3) & ~3);
        if (baseA < k0) fill_dd(baseA, k0, false, false);
        if (kend <= a.K) fill_dd(max(k0, baseA), kend, true, true);
        else fill_dd(max(k0, baseA), kend, true, false);
        if (endA > kend) fill_dd(kend, endA, false, false);
        for (int j = lane * 4; j < nfill; j += 256) {
            {
                const int k = baseB + j;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live && k >= 0 && k < a.Kp) {
                    v = ld4(y1 + k);
                    if constexpr (XM == 2) {
                        const float4 mu = ld4(mu1f + k), rs = ld4(rs1f + k);
                        v.x = (prelu_f(v.x, al1) - mu.x) * rs.x;
                        v.y = (prelu_f(v.y, al1) - mu.y) * rs.y;
                        v.z = (prelu_f(v.z, al1) - mu.z) * rs.z;
                        v.w = (prelu_f(v.w, al1) - mu.w) * rs.w;
                    }
                    if constexpr (FUSED) {
                        v.x = fmaf(v.x, v.x >= 0.f ? rstd1 : ar1, -mr1);
                        v.y = fmaf(v.y, v.y >= 0.f ? rstd1 : ar1, -mr1);
                        v.z = fmaf(v.z, v.z >= 0.f ? rstd1 : ar1, -mr1);
                        v.w = fmaf(v.w, v.w >= 0.f ? rstd1 : ar1, -mr1);
                    }
                }
                *reinterpret_cast<float4*>(LB + j) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();   // wave-private LDS patches
        if constexpr (VEC4) {
            for (int k = k0 + lane * 4; k < kend; k += 256) {
                const int ia = k + a.padl - baseA;
                float accv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    if (PT || j < a.P) {
                        const float4 t = *reinterpret_cast<const float4*>(LA + ia - j * a.dil);
                        accv[0] += taps[j] * t.x; accv[1] += taps[j] * t.y; accv[2] += taps[j] * t.z; accv[3] += taps[j] * t.w;
                    }
                const float4 dq = *reinterpret_cast<const float4*>(LA + (k - baseA));
                const float ddv[4] = {dq.x, dq.y, dq.z, dq.w};
                const int ib = k - a.padl - baseB;
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    if (PT || j < a.P) {
                        const float4 xq = *reinterpret_cast<const float4*>(LB + ib + j * a.dil);
                        float xv[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if constexpr (XHAT) {
                                const int kk = k + e - a.padl + j * a.dil;
                                xv[e] = (kk >= 0 && kk < a.K) ? g1 * xv[e] + b1 : 0.f;
                            }
                            dD[j] += ddv[e] * xv[e];
                        }
                    }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e >= a.K) accv[e] = 0.f;
                if constexpr (FUSED) {
                    const float4 hq = *reinterpret_cast<const float4*>(LB + (k - baseB));
                    const float xh[4] = {hq.x, hq.y, hq.z, hq.w};
                    float4 hr = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (APPLY) { if (live) hr = ld4(y1 + k); }       // (the raw h1: its sign; the lines were just read for the image)
                    const float hv[4] = {hr.x, hr.y, hr.z, hr.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < a.K) {
                            dg1 += accv[e] * xh[e];
                            db1 += accv[e];
                            if constexpr (APPLY) accv[e] = apply1(accv[e], xh[e], hv[e]);
                            else {
                                const float t = g1 * accv[e];
                                t1 += t;
                                t2 += t * xh[e];
                            }
                        }
                }
                if (live) *reinterpret_cast<float4*>(dn1 + k) = make_float4(accv[0], accv[1], accv[2], accv[3]);
            }
        } else {
        for (int kg = k0; kg < kend; kg += 64) {
            const int k = kg + lane;
            // a 64-frame group whose taps all stay inside [0, K) (uniform test) needs no per-lane predicates
            if (kg - a.padl >= 0 && kg + 63 - a.padl + halo < a.K && kg + 63 < kend) {
                float acc = 0.f;
                const int ia = k + a.padl - baseA, ib = k - a.padl - baseB;
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    if (PT || j < a.P) {
                        acc += taps[j] * LA[ia - j * a.dil];
                        float xv = LB[ib + j * a.dil];
                        if constexpr (XHAT) xv = g1 * xv + b1;
                        dD[j] += LA[k - baseA] * xv;
                    }
                if constexpr (FUSED) {
                    const float xh1 = LB[k - baseB];
                    dg1 += acc * xh1;
                    db1 += acc;
                    if constexpr (APPLY) acc = apply1(acc, xh1, live ? y1[k] : 0.f);
                    else {
                        const float t = g1 * acc;
                        t1 += t;
                        t2 += t * xh1;
                    }
                }
                if (live) dn1[k] = acc;
                continue;
            }
            if (k >= kend) continue;
            // input gradient: transposed taps
            float acc = 0.f;
            const int ia = k + a.padl - baseA;
#pragma unroll
            for (int j = 0; j < NP; ++j)
                if (PT || j < a.P) acc += taps[j] * LA[ia - j * a.dil];
            if (k >= a.K) acc = 0.f;
            // tap gradients
            const float ddk = LA[k - baseA];
            const int ib = k - a.padl - baseB;
#pragma unroll
            for (int j = 0; j < NP; ++j)
                if (PT || j < a.P) {
                    const int kk = k - a.padl + j * a.dil;
                    float xv = LB[ib + j * a.dil];
                    if constexpr (XHAT) xv = (kk >= 0 && kk < a.K) ? g1 * xv + b1 : 0.f;
                    dD[j] += ddk * xv;
                }
            if constexpr (FUSED) {
                if (k < a.K) {
                    const float xh1 = LB[k - baseB];
                    dg1 += acc * xh1;
                    db1 += acc;
                    if constexpr (APPLY) acc = apply1(acc, xh1, live ? y1[k] : 0.f);
                    else {
                        const float t = g1 * acc;
                        t1 += t;
                        t2 += t * xh1;
                    }
                }
            }
            if (live) dn1[k] = acc;
        }
        }
        __builtin_amdgcn_wave_barrier();
    }
    const size_t MH = (size_t)a.M * a.H, rc = (size_t)m * a.H + c;
#pragma unroll
    for (int j = 0; j < NP; ++j)
        if (PT || j < a.P) {
            const float v = wave_sum(dD[j]);
            if (live && lane == 0) a.pc[(size_t)j * MH + rc] = v;
        }
    if constexpr (DDM == 2) {
        const float v0 = wave_sum(dg2), v1 = wave_sum(db2), v4 = wave_sum(dal2);
        if (live && lane == 0) {
            a.pc[(size_t)(P_ + 0) * MH + rc] = v0;
            a.pc[(size_t)(P_ + 1) * MH + rc] = v1;
            a.pc[(size_t)(P_ + 2) * MH + rc] = v4;
        }
    }
    if constexpr (FUSED) {
        const float v0 = wave_sum(dg2), v1 = wave_sum(db2), v2 = wave_sum(dg1), v3 = wave_sum(db1), v4 = wave_sum(dal2);
        const double w1 = wave_sum((double)t1), w2 = wave_sum((double)t2);
        if (live && lane == 0) {
            a.pc[(size_t)(P_ + 0) * MH + rc] = v0;
            a.pc[(size_t)(P_ + 1) * MH + rc] = v1;
            a.pc[(size_t)(P_ + 2) * MH + rc] = v2;
            a.pc[(size_t)(P_ + 3) * MH + rc] = v3;
            a.pc[(size_t)(P_ + 4) * MH + rc] = v4;
            if constexpr (!APPLY) {
                a.sums1_part[rc * 2] = w1;
                a.sums1_part[rc * 2 + 1] = w2;
            }
        }
        if constexpr (APPLY) {
            const float v5 = wave_sum(dal1);
            if (live && lane == 0) a.pc[(size_t)(P_ + 5) * MH + rc] = v5;
            if (a.amax_out != nullptr) block_amax_atomic<NT>(amax1, red, a.amax_out + (size_t)m * CTN_AMAX_SLOTS, blockIdx.x % hb);
        }
    }
}

// ---------------------------------------------------------------------------
// dY = rstd*(g*dN - S1/n - xh*S2/n) * prelu'(y),  xh = (prelu(y)-mean)*rstd ; dalpha partial per row
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gln_prelu_bwd_kernel(const float* __restrict__ dN, const float* __restrict__ Y,
                                                           float* __restrict__ dY, int M, int H, int K, int Kp,
                                                           const float* __restrict__ gamma, const float* __restrict__ alpha_p,
                                                           const float* __restrict__ ms, const double* __restrict__ sums_part,
                                                           int nparts, float* __restrict__ dalpha_part,
                                                           unsigned* __restrict__ amax_out) {
    __shared__ double red[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hb = (H + ROWS - 1) / ROWS;
    const int m = blockIdx.x / hb;
    const int c = (blockIdx.x % hb) * ROWS + wave;
    double S1 = 0.0, S2 = 0.0;
    for (int i = tid; i < nparts; i += NT) {
        S1 += sums_part[((size_t)m * nparts + i) * 2];
        S2 += sums_part[((size_t)m * nparts + i) * 2 + 1];
    }
    S1 = block_sum<double, NT>(S1, red);
    S2 = block_sum<double, NT>(S2, red);
    const bool live = c < H;
    const double n = (double)H * (double)K;
    const float c1 = (float)(S1 / n), c2 = (float)(S2 / n);
    const float mean = ms[2 * m], rstd = ms[2 * m + 1], al = alpha_p[0], g = live ? gamma[c] : 0.f;
    const float ar = al * rstd, mr = mean * rstd, rg = rstd * g, rc1 = rstd * c1, rc2 = rstd * c2;
    const size_t row = ((size_t)m * H + c) * Kp;
    float dal = 0.f, amax = 0.f;
    for (int k = lane * 4; live && k < Kp; k += 256) {
        const float4 dn = ld4(dN + row + k);
        const float4 y = ld4(Y + row + k);
        const float dv[4] = {dn.x, dn.y, dn.z, dn.w};
        const float yv[4] = {y.x, y.y, y.z, y.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = fmaf(yv[e], yv[e] >= 0.f ? rstd : ar, -mr);
            const float da = fmaf(-xh, rc2, fmaf(rg, dv[e], -rc1));
            const bool valid = (k + e) < K;
            if (valid && yv[e] < 0.f) dal += da * yv[e];
            o[e] = valid ? (yv[e] >= 0.f ? da : al * da) : 0.f;
        }
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
        *reinterpret_cast<float4*>(dY + row + k) = make_float4(o[0], o[1], o[2], o[3]);
    }
    dal = wave_sum(dal);
    if (live && lane == 0) dalpha_part[(size_t)m * H + c] = dal;
    if (amax_out != nullptr) block_amax_atomic<NT>(amax, red, amax_out + (size_t)m * CTN_AMAX_SLOTS, blockIdx.x % hb);
}

// ---------------------------------------------------------------------------
// Stand-alone gLN(prelu(Y)) backward, first pass (the fused path gets these sums from the GEMM / depthwise epilogues):
// per (utterance, channel) row  S1 = sum_k g*dN, S2 = sum_k g*dN*xh  -> sums_part [M, H, 2] fp64, and the parameter-
// gradient partials  pc[0] = sum_k dN*xh (dgamma), pc[1] = sum_k dN (dbeta)  -> pc [2, M, H].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gln_bwd_sums_kernel(const float* __restrict__ dN, const float* __restrict__ Y,
                                                          int M, int H, int K, int Kp, const float* __restrict__ gamma,
                                                          const float* __restrict__ alpha_p, const float* __restrict__ ms,
                                                          double* __restrict__ sums_part, float* __restrict__ pc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hb = (H + ROWS - 1) / ROWS;
    const int m = blockIdx.x / hb;
    const int c = (blockIdx.x % hb) * ROWS + wave;
    if (c >= H) return;
    const float mean = ms[2 * m], rstd = ms[2 * m + 1], al = alpha_p[0], g = gamma[c];
    const size_t row = ((size_t)m * H + c) * Kp;
    float sdn = 0.f, sdx = 0.f;
    for (int k = lane * 4; k < Kp; k += 256) {
        const float4 dn = ld4(dN + row + k);
        const float4 y = ld4(Y + row + k);
        const float dv[4] = {dn.x, dn.y, dn.z, dn.w};
        const float yv[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k + e < K) {
                const float xh = (prelu_f(yv[e], al) - mean) * rstd;
                sdn += dv[e];
                sdx = fmaf(dv[e], xh, sdx);
            }
    }
    const double d1 = wave_sum((double)sdn), d2 = wave_sum((double)sdx);
    if (lane == 0) {
        sums_part[((size_t)m * H + c) * 2] = (double)g * d1;
        sums_part[((size_t)m * H + c) * 2 + 1] = (double)g * d2;
        pc[(size_t)m * H + c] = (float)d2;
        pc[(size_t)(M + m) * H + c] = (float)d1;
    }
}

// ---------------------------------------------------------------------------
// channel-wise LayerNorm (optionally after PReLU), per (m, frame) over channels.
// block = 64 frames x 4 channel-groups (one wave each).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void cln_fwd_kernel(const float* __restrict__ Y, float* __restrict__ Out,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                     int M, int Ch, int K, int Kp, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ alpha_p) {
    __shared__ float sh[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kb = (Kp + 63) / 64;
    const int m = blockIdx.x / kb, k = (blockIdx.x % kb) * 64 + lane;
    const bool in = k < Kp;
    const bool has_a = alpha_p != nullptr;
    const float al = has_a ? alpha_p[0] : 1.f;
    const float* __restrict__ y = Y + (size_t)m * Ch * Kp + (in ? k : 0);
    float s = 0.f;
    for (int c = wave; c < Ch; c += 4) {
        float v = in ? y[(size_t)c * Kp] : 0.f;
        if (has_a) v = prelu_f(v, al);
        s += v;
    }
    sh[wave][lane] = s;
    __syncthreads();
    const float mu = (sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane]) / (float)Ch;
    __syncthreads();
    float q = 0.f;
    for (int c = wave; c < Ch; c += 4) {
        float v = in ? y[(size_t)c * Kp] : 0.f;
        if (has_a) v = prelu_f(v, al);
        q += (v - mu) * (v - mu);
    }
    sh[wave][lane] = q;
    __syncthreads();
    const float var = (sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane]) / (float)Ch;
    const float rs = 1.0f / sqrtf(var + CTN_EPS);
    if (wave == 0 && in) {
        mean_o[(size_t)m * Kp + k] = mu;
        rstd_o[(size_t)m * Kp + k] = rs;
    }
    if (!in) return;
    float* __restrict__ o = Out + (size_t)m * Ch * Kp + k;
    const bool valid = k < K;
    for (int c = wave; c < Ch; c += 4) {
        float v = y[(size_t)c * Kp];
        if (has_a) v = prelu_f(v, al);
        o[(size_t)c * Kp] = valid ? gamma[c] * ((v - mu) * rs) + beta[c] : 0.f;
    }
}

// dY = [ rstd*(t - mean_c(t) - xh*mean_c(t*xh)) * prelu'(y) + add ] * (relu_ref > 0)
__global__ __launch_bounds__(NT) void cln_bwd_dx_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                        float* __restrict__ dY, const float* __restrict__ mean_i,
                                                        const float* __restrict__ rstd_i, int M, int Ch, int K, int Kp,
                                                        const float* __restrict__ gamma, const float* __restrict__ alpha_p,
                                                        const float* __restrict__ add, const float* __restrict__ relu_ref,
                                                        float* __restrict__ dalpha_part) {
    __shared__ float sh[2][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kb = (Kp + 63) / 64;
    const int m = blockIdx.x / kb, k = (blockIdx.x % kb) * 64 + lane;
    const bool in = k < Kp, valid = k < K;
    const bool has_a = alpha_p != nullptr;
    const float al = has_a ? alpha_p[0] : 1.f;
    const size_t off = (size_t)m * Ch * Kp + (in ? k : 0);
    const float mu = in ? mean_i[(size_t)m * Kp + k] : 0.f, rs = in ? rstd_i[(size_t)m * Kp + k] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    for (int c = wave; c < Ch; c += 4) {
        if (valid) {
            float v = Y[off + (size_t)c * Kp];
            if (has_a) v = prelu_f(v, al);
            const float t = gamma[c] * dOut[off + (size_t)c * Kp];
            s1 += t;
            s2 += t * ((v - mu) * rs);
        }
    }
    sh[0][wave][lane] = s1;
    sh[1][wave][lane] = s2;
    __syncthreads();
    const float m1 = (sh[0][0][lane] + sh[0][1][lane] + sh[0][2][lane] + sh[0][3][lane]) / (float)Ch;
    const float m2 = (sh[1][0][lane] + sh[1][1][lane] + sh[1][2][lane] + sh[1][3][lane]) / (float)Ch;
    float dal = 0.f;
    if (in) {
        for (int c = wave; c < Ch; c += 4) {
            const size_t o = off + (size_t)c * Kp;
            float r = 0.f;
            if (valid) {
                const float yv = Y[o];
                const float v = has_a ? prelu_f(yv, al) : yv;
                const float xh = (v - mu) * rs;
                const float da = rs * (gamma[c] * dOut[o] - m1 - xh * m2);
                if (has_a && yv < 0.f) dal += da * yv;
                r = (has_a && yv < 0.f) ? al * da : da;
                if (add != nullptr) r += add[o];
                if (relu_ref != nullptr && !(relu_ref[o] > 0.f)) r = 0.f;
            }
            dY[o] = r;
        }
    }
    if (dalpha_part != nullptr) {
        __syncthreads();
        dal = wave_sum(dal);
        if (lane == 0) sh[0][wave][0] = dal;
        __syncthreads();
        if (tid == 0) dalpha_part[blockIdx.x] = sh[0][0][0] + sh[0][1][0] + sh[0][2][0] + sh[0][3][0];
    }
}

// ---------------------------------------------------------------------------
// Register-resident channel-wise LayerNorm: a 1024-thread workgroup owns FR frames x all channels.  Lane l of wave w
// holds frame l % FR for channel group g = w * (64/FR) + l / FR, i.e. channels g, g + NG, ... (NG = 1024/FR groups,
// CPT channels per thread), so the tensor is read ONCE for the two-pass statistics and the normalisation (the generic
// kernels above re-read it per pass with 4 waves per 64 frames: 1.6 TB/s).  FR = 32 keeps 128-byte row segments.
// ---------------------------------------------------------------------------
constexpr int CLN_NT = 1024, CLN_FR = 32;

template <int FR, int CPT>
__global__ __launch_bounds__(CLN_NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void cln_fwd_reg_kernel(const float* __restrict__ Y, float* __restrict__ Out,
                                                             float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                             int M, int Ch, int K, int Kp, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ alpha_p) {
    constexpr int NG = CLN_NT / FR;
    __shared__ float sh[NG][FR];
    const int fr = threadIdx.x % FR, g = threadIdx.x / FR;
    const int kb = (Kp + FR - 1) / FR;
    const int m = blockIdx.x / kb, k = (blockIdx.x % kb) * FR + fr;
    const bool in = k < Kp;
    const bool has_a = alpha_p != nullptr;
    const float al = has_a ? alpha_p[0] : 1.f;
    const float* __restrict__ y = Y + (size_t)m * Ch * Kp + (in ? k : 0);
    float v[CPT];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = g + NG * j;
        float t = (in && c < Ch) ? y[(size_t)c * Kp] : 0.f;
        if (has_a) t = prelu_f(t, al);
        v[j] = t;
        s += t;
    }
    sh[g][fr] = s;
    __syncthreads();
    float mu = 0.f;
    for (int w = 0; w < NG; ++w) mu += sh[w][fr];
    mu /= (float)Ch;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j)
        if (g + NG * j < Ch) q += (v[j] - mu) * (v[j] - mu);
    sh[g][fr] = q;
    __syncthreads();
    float var = 0.f;
    for (int w = 0; w < NG; ++w) var += sh[w][fr];
    var /= (float)Ch;
    const float rs = 1.0f / sqrtf(var + CTN_EPS);
    if (!in) return;
    if (g == 0) {
        mean_o[(size_t)m * Kp + k] = mu;
        rstd_o[(size_t)m * Kp + k] = rs;
    }
    float* __restrict__ o = Out + (size_t)m * Ch * Kp + k;
    const bool valid = k < K;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = g + NG * j;
        if (c < Ch) o[(size_t)c * Kp] = valid ? gamma[c] * ((v[j] - mu) * rs) + beta[c] : 0.f;
    }
}

template <int FR, int CPT, int NTB>     // NTB threads per workgroup (512 or 1024)
__global__ __launch_bounds__(NTB) void cln_bwd_dx_reg_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                                float* __restrict__ dY, const float* __restrict__ mean_i,
                                                                const float* __restrict__ rstd_i, int M, int Ch, int K, int Kp,
                                                                const float* __restrict__ gamma, const float* __restrict__ alpha_p,
                                                                const float* __restrict__ add, const float* __restrict__ relu_ref,
                                                                float* __restrict__ dalpha_part) {
    constexpr int NG = NTB / FR;
    __shared__ float sh[2][NG][FR];
    __shared__ float red[NTB / 64];
    const int fr = threadIdx.x % FR, g = threadIdx.x / FR;
    const int kb = (Kp + FR - 1) / FR;
    const int m = blockIdx.x / kb, k = (blockIdx.x % kb) * FR + fr;
    const bool in = k < Kp, valid = k < K;
    const bool has_a = alpha_p != nullptr;
    const float al = has_a ? alpha_p[0] : 1.f;
    const size_t off = (size_t)m * Ch * Kp + (in ? k : 0);
    const float mu = in ? mean_i[(size_t)m * Kp + k] : 0.f, rs = in ? rstd_i[(size_t)m * Kp + k] : 0.f;
    float t[CPT], yv[CPT];           // gamma * dOut and the raw input of this thread's channels
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = g + NG * j;
        const bool ok = valid && c < Ch;
        yv[j] = ok ? Y[off + (size_t)c * Kp] : 0.f;
        t[j] = ok ? gamma[c] * dOut[off + (size_t)c * Kp] : 0.f;
        const float v = has_a ? prelu_f(yv[j], al) : yv[j];
        s1 += t[j];
        s2 += ok ? t[j] * ((v - mu) * rs) : 0.f;
    }
    sh[0][g][fr] = s1;
    sh[1][g][fr] = s2;
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
    for (int w = 0; w < NG; ++w) { m1 += sh[0][w][fr]; m2 += sh[1][w][fr]; }
    m1 /= (float)Ch;
    m2 /= (float)Ch;
    float dal = 0.f;
    if (in) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = g + NG * j;
            if (c < Ch) {
                const size_t o = off + (size_t)c * Kp;
                float r = 0.f;
                if (valid) {
                    const float v = has_a ? prelu_f(yv[j], al) : yv[j];
                    const float xh = (v - mu) * rs;
                    const float da = rs * (t[j] - m1 - xh * m2);
                    if (has_a && yv[j] < 0.f) dal += da * yv[j];
                    r = (has_a && yv[j] < 0.f) ? al * da : da;
                    if (add != nullptr) r += add[o];
                    if (relu_ref != nullptr && !(relu_ref[o] > 0.f)) r = 0.f;
                }
                dY[o] = r;
            }
        }
    }
    if (dalpha_part != nullptr) {
        dal = block_sum<float, NTB>(dal, red);
        if (threadIdx.x == 0) dalpha_part[blockIdx.x] = dal;
    }
}

// ---------------------------------------------------------------------------
// Round-2 channel-wise LayerNorm ("v4"): 16-byte accesses along frames and ONE backward pass.
// A 512-thread workgroup owns 32 frames x all channels: thread t holds the frame quad q = t % 8 (frames 4q..4q+3) of channel
// group g = t / 8, i.e. channels g, g + 64, ... (CPT per thread).  A wave instruction then moves 8 rows x 128 bytes with
// 64 float4 accesses instead of 4-byte ones.  Per-frame sums over channels: three xor-shuffles over the 8 groups of a wave,
// then 8 float4 partials through LDS.  The backward kernel also produces the parameter-gradient partials: dgamma / dbeta of
// a channel over this workgroup's 32 frames (its 4 frames per thread, then three xor-shuffles over the 8 quads), written to
// pc [2][blocks][Ch] and summed in fixed order by cln_bwd_finalize -- the separate cln_bwd_params pass (a third read of both
// tensors) is gone.
// ---------------------------------------------------------------------------
constexpr int C4_NT = 512, C4_FR = 32, C4_NG = C4_NT / (C4_FR / 4);      // 64 channel groups

template <int FR = C4_FR>
__device__ __forceinline__ float4 quad_group_sum(float4 v) {              // sum over the channel groups of a wave (lane bits above the FR / 4 frame quads)
#pragma unroll
    for (int o = FR / 4; o < 64; o <<= 1) {
        v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64);
        v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
    }
    return v;
}
// sum of `v` over all channel groups of the workgroup, result for this thread's frame quad in every thread
template <int NW = C4_NT / 64, int FR = C4_FR>
__device__ __forceinline__ float4 block_group_sum(float4 v, float4 (*sh)[FR / 4], int wave, int q, int lane) {
    v = quad_group_sum<FR>(v);
    __syncthreads();                                 // sh may still be read by a previous call
    if (lane < FR / 4) sh[wave][q] = v;
    __syncthreads();
    float4 r = sh[0][q];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        const float4 t = sh[w][q];
        r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
    }
    return r;
}

// (NTB, FR) = (512, 32) or (256, 16): the same channel groups and per-thread work; the small form fits more independent workgroups on a CU
template <int CPT, int NTB = C4_NT, int FR = C4_FR>
__global__ __launch_bounds__(NTB) void cln_fwd_v4_kernel(const float* __restrict__ Y, float* __restrict__ Out,
                                                           float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                           int M, int Ch, int K, int Kp, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ alpha_p,
                                                           unsigned* __restrict__ amax_out) {
    constexpr int NW = NTB / 64, NQ = FR / 4, C4_NG = NTB / NQ;
    static_assert(C4_NG == 64, "64 channel groups");
    __shared__ float4 sh[NW][NQ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = tid % NQ, g = tid / NQ;
    const int kb = Kp / FR;
    const int bx = FR == 32 ? (int)blockIdx.x : xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int m = bx / kb, k0 = (bx % kb) * FR + 4 * q;
    float amax = 0.f;
    const bool has_a = alpha_p != nullptr;
    const float al = has_a ? alpha_p[0] : 1.f;
    const size_t off = (size_t)m * Ch * Kp + k0;
    float4 v[CPT];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = g + C4_NG * j;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < Ch) t = ld4(Y + off + (size_t)c * Kp);
        if (has_a) { t.x = prelu_f(t.x, al); t.y = prelu_f(t.y, al); t.z = prelu_f(t.z, al); t.w = prelu_f(t.w, al); }
        v[j] = t;
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    float4 mu = block_group_sum<NW, FR>(s, sh, wave, q, lane);
    const float inv = 1.f / (float)Ch;
    mu.x *= inv; mu.y *= inv; mu.z *= inv; mu.w *= inv;
    float4 d2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < CPT; ++j)
        if (g + C4_NG * j < Ch) {
            d2.x += (v[j].x - mu.x) * (v[j].x - mu.x); d2.y += (v[j].y - mu.y) * (v[j].y - mu.y);
            d2.z += (v[j].z - mu.z) * (v[j].z - mu.z); d2.w += (v[j].w - mu.w) * (v[j].w - mu.w);
        }
    const float4 var = block_group_sum<NW, FR>(d2, sh, wave, q, lane);
    const float4 rs = make_float4(1.0f / sqrtf(var.x * inv + CTN_EPS), 1.0f / sqrtf(var.y * inv + CTN_EPS),
                                  1.0f / sqrtf(var.z * inv + CTN_EPS), 1.0f / sqrtf(var.w * inv + CTN_EPS));
    if (g == 0) {
        *reinterpret_cast<float4*>(mean_o + (size_t)m * Kp + k0) = mu;
        *reinterpret_cast<float4*>(rstd_o + (size_t)m * Kp + k0) = rs;
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = g + C4_NG * j;
        if (c < Ch) {
            const float ga = gamma[c], be = beta[c];
            float4 o;
            o.x = k0 + 0 < K ? ga * ((v[j].x - mu.x) * rs.x) + be : 0.f;
            o.y = k0 + 1 < K ? ga * ((v[j].y - mu.y) * rs.y) + be : 0.f;
            o.z = k0 + 2 < K ? ga * ((v[j].z - mu.z) * rs.z) + be : 0.f;
            o.w = k0 + 3 < K ? ga * ((v[j].w - mu.w) * rs.w) + be : 0.f;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            *reinterpret_cast<float4*>(Out + off + (size_t)c * Kp) = o;
        }
    }
    if (amax_out != nullptr)        // h3 arithmetic of the GEMM that reads Out: its maximum per utterance
        block_amax_atomic<NTB>(amax, reinterpret_cast<double*>(&sh[0][0]), amax_out + (size_t)m * CTN_AMAX_SLOTS, bx % kb);
}

// NTB threads = NTB/8 channel groups.  CPT = 8 at 512 threads needs 161 VGPRs (one workgroup = 2 waves per SIMD) and is
// still the fastest form for 512 channels: 1024 threads x CPT 4 (4 waves per SIMD) measured 84 us against 55 us, capping
// the registers at 128 (re-reading dOut in the second phase) 59 us.
// FR = frames per workgroup.  (512 threads, 32 frames) holds ONE workgroup per CU (161 VGPRs x 8 waves): it loads 128 KiB, reduces,
// then stores 64 KiB, and nothing on the CU overlaps those phases.  (256 threads, 16 frames) has the same channel groups, registers
// and instruction stream per thread, but three independent workgroups fit a CU; its 64-byte row pieces pair up into whole
// 128-byte lines with the neighbouring workgroup, which the XCD-contiguous block order keeps on the same L2.
// LEAN: the form the composite stacks launch -- every channel group full (Ch == NG * CPT), PReLU fused, no added gradient, no ReLU
// mask: the per-channel and per-element option tests become compile-time (1481 -> ~1000 VALU instructions per wave; the kernel spends
// about a third of its time issuing them).  Same arithmetic, same order: bitwise the general form.
template <int CPT, int NTB, int FR = C4_FR, bool LEAN = false>
__global__ __launch_bounds__(NTB) void cln_bwd_v4_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                           float* __restrict__ dY, const float* __restrict__ mean_i,
                                                           const float* __restrict__ rstd_i, int M, int Ch, int K, int Kp,
                                                           const float* __restrict__ gamma, const float* __restrict__ alpha_p,
                                                           const float* __restrict__ add, const float* __restrict__ relu_ref,
                                                           float* __restrict__ dalpha_part, float* __restrict__ pc,
                                                           unsigned* __restrict__ amax_out) {
    constexpr int NW = NTB / 64, NQ = FR / 4, NG = NTB / NQ;
    __shared__ float4 sh[NW][NQ];
    __shared__ float red[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = tid % NQ, g = tid / NQ;
    const int kb = Kp / FR, nblk = M * kb;
    const int bx = FR == C4_FR ? (int)blockIdx.x : xcd_remap((int)blockIdx.x, nblk);        // FR 16: the two halves of a 128-byte line on one XCD
    const int m = bx / kb, k0 = (bx % kb) * FR + 4 * q;
    const bool has_a = LEAN || alpha_p != nullptr;
    const float al = has_a ? alpha_p[0] : 1.f;
    const size_t off = (size_t)m * Ch * Kp + k0;
    const float4 mu = ld4(mean_i + (size_t)m * Kp + k0), rs = ld4(rstd_i + (size_t)m * Kp + k0);
    const float vm[4] = {k0 + 0 < K ? 1.f : 0.f, k0 + 1 < K ? 1.f : 0.f, k0 + 2 < K ? 1.f : 0.f, k0 + 3 < K ? 1.f : 0.f};
    float4 t[CPT], yv[CPT];           // gamma * dOut (0 for frames >= K) and the raw input of this thread's channels
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = g + NG * j;
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f), d = y;
        float ga = 0.f;
        if (LEAN || c < Ch) {
            y = ld4(Y + off + (size_t)c * Kp);
            d = ld4(dOut + off + (size_t)c * Kp);
            ga = gamma[c];
        }
        d.x *= vm[0]; d.y *= vm[1]; d.z *= vm[2]; d.w *= vm[3];
        const float4 v = has_a ? make_float4(prelu_f(y.x, al), prelu_f(y.y, al), prelu_f(y.z, al), prelu_f(y.w, al)) : y;
        const float4 xh = make_float4((v.x - mu.x) * rs.x, (v.y - mu.y) * rs.y, (v.z - mu.z) * rs.z, (v.w - mu.w) * rs.w);
        // parameter-gradient partials of channel c over this workgroup's frames: its 4 frames here, the 8 quads by shuffles
        float pg = (d.x * xh.x + d.y * xh.y) + (d.z * xh.z + d.w * xh.w), pb = (d.x + d.y) + (d.z + d.w);
#pragma unroll
        for (int o = 1; o < NQ; o <<= 1) { pg += __shfl_xor(pg, o, 64); pb += __shfl_xor(pb, o, 64); }
        if (q == 0 && (LEAN || c < Ch)) {
            pc[(size_t)bx * Ch + c] = pg;
            pc[((size_t)nblk + bx) * Ch + c] = pb;
        }
        yv[j] = y;
        t[j] = make_float4(ga * d.x, ga * d.y, ga * d.z, ga * d.w);
        s1.x += t[j].x; s1.y += t[j].y; s1.z += t[j].z; s1.w += t[j].w;
        s2.x += t[j].x * xh.x; s2.y += t[j].y * xh.y; s2.z += t[j].z * xh.z; s2.w += t[j].w * xh.w;
    }
    float4 m1 = block_group_sum<NW, FR>(s1, sh, wave, q, lane), m2 = block_group_sum<NW, FR>(s2, sh, wave, q, lane);
    const float inv = 1.f / (float)Ch;
    m1.x *= inv; m1.y *= inv; m1.z *= inv; m1.w *= inv;
    m2.x *= inv; m2.y *= inv; m2.z *= inv; m2.w *= inv;
    float dal = 0.f, amax = 0.f;
    const float mm[4] = {mu.x, mu.y, mu.z, mu.w}, rr[4] = {rs.x, rs.y, rs.z, rs.w};
    const float a1[4] = {m1.x, m1.y, m1.z, m1.w}, a2[4] = {m2.x, m2.y, m2.z, m2.w};
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = g + NG * j;
        if (LEAN || c < Ch) {
            const size_t o = off + (size_t)c * Kp;
            const float yy[4] = {yv[j].x, yv[j].y, yv[j].z, yv[j].w}, tt[4] = {t[j].x, t[j].y, t[j].z, t[j].w};
            float4 ad = make_float4(0.f, 0.f, 0.f, 0.f), rf = make_float4(1.f, 1.f, 1.f, 1.f);
            if constexpr (!LEAN) {
                if (add != nullptr) ad = ld4(add + o);
                if (relu_ref != nullptr) rf = ld4(relu_ref + o);
            }
            const float av[4] = {ad.x, ad.y, ad.z, ad.w}, rv[4] = {rf.x, rf.y, rf.z, rf.w};
            float r[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = has_a ? prelu_f(yy[e], al) : yy[e];
                const float xh = (v - mm[e]) * rr[e];
                const float da = rr[e] * (tt[e] - a1[e] - xh * a2[e]);
                float x = 0.f;
                if (vm[e] != 0.f) {
                    if (has_a && yy[e] < 0.f) dal += da * yy[e];
                    x = (has_a && yy[e] < 0.f) ? al * da : da;
                    if constexpr (!LEAN) {
                        x += av[e];
                        if (!(rv[e] > 0.f)) x = 0.f;
                    }
                }
                r[e] = x;
            }
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(r[0]), fabsf(r[1]))), fmaxf(fabsf(r[2]), fabsf(r[3])));
            *reinterpret_cast<float4*>(dY + o) = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
    if (dalpha_part != nullptr) {
        dal = block_sum<float, NTB>(dal, red);
        if (tid == 0) dalpha_part[bx] = dal;
    }
    if (amax_out != nullptr)
        block_amax_atomic<NTB>(amax, reinterpret_cast<double*>(&sh[0][0]), amax_out + (size_t)m * CTN_AMAX_SLOTS, bx % kb);
}

// per-(m,c) partial of dgamma = sum_k dOut*xh and dbeta = sum_k dOut ; pc[2][rows][Ch], rows >= M (fallback of the v4 kernel)
__global__ __launch_bounds__(NT) void cln_bwd_params_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                            const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                            int M, int Ch, int K, int Kp, const float* __restrict__ alpha_p,
                                                            float* __restrict__ pc, int rows) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hb = (Ch + ROWS - 1) / ROWS;
    const int m = blockIdx.x / hb;
    const int c = (blockIdx.x % hb) * ROWS + wave;
    if (c >= Ch) return;
    const bool has_a = alpha_p != nullptr;
    const float al = has_a ? alpha_p[0] : 1.f;
    const size_t row = ((size_t)m * Ch + c) * Kp;
    float dg = 0.f, db = 0.f;
    for (int k = lane * 4; k < Kp; k += 256) {
        const float4 d = ld4(dOut + row + k), y = ld4(Y + row + k);
        const float4 mu = ld4(mean_i + (size_t)m * Kp + k), rs = ld4(rstd_i + (size_t)m * Kp + k);
        const float dv[4] = {d.x, d.y, d.z, d.w}, yv[4] = {y.x, y.y, y.z, y.w};
        const float mv[4] = {mu.x, mu.y, mu.z, mu.w}, rv[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k + e < K) {
                const float v = has_a ? prelu_f(yv[e], al) : yv[e];
                dg += dv[e] * ((v - mv[e]) * rv[e]);
                db += dv[e];
            }
    }
    dg = wave_sum(dg);
    db = wave_sum(db);
    if (lane == 0) {
        pc[(size_t)m * Ch + c] = dg;
        pc[(size_t)rows * Ch + (size_t)m * Ch + c] = db;
    }
}

// out[f][i] = sum_mid in[f][mid][i]   (fixed order).  mode 0: thread per output; mode 1: workgroup per output.
__global__ __launch_bounds__(NT) void reduce_mid_kernel(const float* __restrict__ in, int F, int Mid, int Inner,
                                                        float* __restrict__ out, int wave_mode) {
    if (!wave_mode) {
        const long long o = (long long)blockIdx.x * NT + threadIdx.x;
        if (o >= (long long)F * Inner) return;
        const int f = (int)(o / Inner), i = (int)(o % Inner);
        float s = 0.f;
        for (int r = 0; r < Mid; ++r) s += in[((size_t)f * Mid + r) * Inner + i];
        out[o] = s;
    } else {                                        // one workgroup per output
        __shared__ float red[NT / 64];
        const long long o = blockIdx.x;
        const int f = (int)(o / Inner), i = (int)(o % Inner);
        float s = 0.f;
        for (int r = threadIdx.x; r < Mid; r += NT) s += in[((size_t)f * Mid + r) * Inner + i];
        s = block_sum<float, NT>(s, red);
        if (threadIdx.x == 0) out[o] = s;
    }
}

// Finish the per-(m,c) partials of dw_bwd<FUSED>:  pc [P+5, M, H] -> dD [H,P], dgamma2, dbeta2, dgamma1, dbeta1 [H],
// dalpha2 [1].  Blocks 0..nb-2 do the per-channel sums over m (thread per (f,h)); the last block sums dalpha2.
__global__ __launch_bounds__(NT) void dw_bwd_finalize_kernel(const float* __restrict__ pc, int P, int M, int H,
                                                             float* __restrict__ dD, float* __restrict__ dg2,
                                                             float* __restrict__ db2, float* __restrict__ dg1,
                                                             float* __restrict__ db1, float* __restrict__ da2,
                                                             const float* __restrict__ da1_part, int n_da1,
                                                             float* __restrict__ da1) {
    __shared__ float red[NT / 64];
    const size_t MH = (size_t)M * H;
    if (blockIdx.x == gridDim.x - 1) {                 // dalpha1 from gln_prelu_bwd's per-row partials (optional)
        if (da1_part == nullptr) return;
        float s = 0.f;
        for (int i = threadIdx.x; i < n_da1; i += NT) s += da1_part[i];
        s = block_sum<float, NT>(s, red);
        if (threadIdx.x == 0) da1[0] = s;
        return;
    }
    if (blockIdx.x == gridDim.x - 2) {
        float s = 0.f;
        const float* src = pc + (size_t)(P + 4) * MH;
        for (size_t i = threadIdx.x; i < MH; i += NT) s += src[i];
        s = block_sum<float, NT>(s, red);
        if (threadIdx.x == 0) da2[0] = s;
        return;
    }
    const int o = blockIdx.x * NT + threadIdx.x;
    if (o >= (P + 4) * H) return;
    const int f = o / H, h = o % H;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += pc[(size_t)f * MH + (size_t)m * H + h];
    if (f < P) dD[(size_t)h * P + f] = s;
    else if (f == P) dg2[h] = s;
    else if (f == P + 1) db2[h] = s;
    else if (f == P + 2) dg1[h] = s;
    else db1[h] = s;
}

// Finish ctn_cln_bwd's partials in one launch: pc [2][rows][Ch] -> dgamma[Ch], dbeta[Ch].  A workgroup owns 64 channels of one
// of the two outputs: lanes along channels (256-byte coalesced rows), the four waves take rows w, w+4, ... and their sums are
// added in wave order (fixed order: bitwise reproducible).  rows = workgroups of the backward kernel (800 at the paper shape).
// The last workgroup sums the nblk per-workgroup dalpha partials.
__global__ __launch_bounds__(NT) void cln_bwd_finalize_kernel(const float* __restrict__ pc, const float* __restrict__ dap,
                                                              int rows, int Ch, int nblk, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ dalpha) {
    __shared__ float red[NT / 64];
    __shared__ float part[NT / 64][64];
    if (blockIdx.x == gridDim.x - 1) {
        if (dap == nullptr) return;
        float s = 0.f;
        for (int i = threadIdx.x; i < nblk; i += NT) s += dap[i];
        s = block_sum<float, NT>(s, red);
        if (threadIdx.x == 0) dalpha[0] = s;
        return;
    }
    const int cb = (Ch + 63) / 64;
    const int f = blockIdx.x / cb, c = (blockIdx.x % cb) * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
    float s = 0.f;
    if (c < Ch) {
        const float* __restrict__ p = pc + (size_t)f * rows * Ch + c;
        int r = wave;
        for (; r + 28 < rows; r += 32) {                 // eight independent loads in flight per lane
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(r + 4 * j) * Ch];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; r < rows; r += 4) s += p[(size_t)r * Ch];
    }
    part[wave][threadIdx.x & 63] = s;
    __syncthreads();
    if (wave == 0 && c < Ch)
        (f == 0 ? dgamma : dbeta)[c] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
}

// Finish the per-(m,c) partials of dw_bwd<CLN>: pc [P+3, M, H] -> dD [H,P], dgamma2, dbeta2 [H] (sums over m in a fixed order), dalpha2 [1]
// (last workgroup).
__global__ __launch_bounds__(NT) void dw_bwd_cln_finalize_kernel(const float* __restrict__ pc, int P, int M, int H, float* __restrict__ dD,
                                                                 float* __restrict__ dg2, float* __restrict__ db2, float* __restrict__ da2) {
    __shared__ float red[NT / 64];
    const size_t MH = (size_t)M * H;
    if (blockIdx.x == gridDim.x - 1) {
        float s = 0.f;
        const float* src = pc + (size_t)(P + 2) * MH;
        for (size_t i = threadIdx.x; i < MH; i += NT) s += src[i];
        s = block_sum<float, NT>(s, red);
        if (threadIdx.x == 0) da2[0] = s;
        return;
    }
    const int o = blockIdx.x * NT + threadIdx.x;
    if (o >= (P + 2) * H) return;
    const int f = o / H, h = o % H;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += pc[(size_t)f * MH + (size_t)m * H + h];
    if (f < P) dD[(size_t)h * P + f] = s;
    else if (f == P) dg2[h] = s;
    else db2[h] = s;
}

// Per-frame statistics of a channel-wise LayerNorm from the column partials of ctn_pw_gemm_cln: (sum p, sum p^2) over the row tiles
// in fixed order (fp64) -> mean, rstd = 1 / sqrt(E[p^2] - mean^2 + eps) (fp64 until the last step; biased variance).
__global__ __launch_bounds__(NT) void cln_stats_frame_kernel(const double* __restrict__ part, int nparts, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int M, int Ch, int Kp) {
    const long long i = (long long)blockIdx.x * NT + threadIdx.x;
    if (i >= (long long)M * Kp) return;
    const int m = (int)(i / Kp), k = (int)(i % Kp);
    double s1 = 0.0, s2 = 0.0;
    for (int t = 0; t < nparts; ++t) {
        const double2 q = *reinterpret_cast<const double2*>(part + (((size_t)m * nparts + t) * Kp + k) * 2);
        s1 += q.x;
        s2 += q.y;
    }
    const double mu = s1 / (double)Ch;
    double var = s2 / (double)Ch - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[i] = (float)mu;
    rstd[i] = (float)(1.0 / sqrt(var + (double)CTN_EPS));
}

// Per-frame constants of a channel-wise LayerNorm's backward from the column partials of ctn_pw_dgrad_cln:
//   S1[k] = sum_t part[m][t][k][0], S2[k] = sum_t part[m][t][k][1]  (t = row tiles, fixed order, fp64)
//   fc[m][0..3][k] = (rstd, mean rstd, rstd S1 / Ch, rstd S2 / Ch)
__global__ __launch_bounds__(NT) void cln_bwd_frame_kernel(const double* __restrict__ part, int nparts, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ fc, int M, int Ch, int Kp) {
    const long long i = (long long)blockIdx.x * NT + threadIdx.x;
    if (i >= (long long)M * Kp) return;
    const int m = (int)(i / Kp), k = (int)(i % Kp);
    double s1 = 0.0, s2 = 0.0;
    for (int t = 0; t < nparts; ++t) {
        const double2 q = *reinterpret_cast<const double2*>(part + (((size_t)m * nparts + t) * Kp + k) * 2);
        s1 += q.x;
        s2 += q.y;
    }
    const float rs = rstd[i], mu = mean[i];
    float* const o = fc + (size_t)m * 4 * Kp + k;
    o[0] = rs;
    o[(size_t)Kp] = mu * rs;
    o[(size_t)2 * Kp] = rs * (float)(s1 / (double)Ch);
    o[(size_t)3 * Kp] = rs * (float)(s2 / (double)Ch);
}

// pc [P, M, H] (the un-fused ctn_dw_bwd's tap partials) -> dD [H, P], summed over m in a fixed order
__global__ __launch_bounds__(NT) void dw_bwd_taps_kernel(const float* __restrict__ pc, int P, int M, int H, float* __restrict__ dD) {
    const int o = blockIdx.x * NT + threadIdx.x;
    if (o >= P * H) return;
    const int j = o / H, h = o % H;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += pc[((size_t)j * M + m) * H + h];
    dD[(size_t)h * P + j] = s;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int ctn_absmax_rows(const float* x, int M, long long n, unsigned* amax, void* stream);      // ctn_gemm.hip
static int dw_fwd_launch(const DwFwdArgs& a, int pro, bool epi, bool small, void* stream);

extern "C" {

int ctn_dw_fwd(const float* Y, float* Z, const float* D, int M, int H, int K, int Kp, int P, int dilation, int causal,
               const double* pro_part, int pro_nparts, const float* pro_gamma, const float* pro_beta,
               const float* pro_alpha, float* pro_ms_out, const float* epi_alpha, double* epi_part, unsigned* amax_out,
               void* stream) {
    CTN_REQUIRE(Y && Z && D, "ctn_dw_fwd: null pointer");
    CTN_REQUIRE(!amax_out || epi_part, "ctn_dw_fwd: amax_out comes with the statistics epilogue");
    CTN_REQUIRE(M > 0 && H > 0 && K > 0 && Kp >= K && Kp % 4 == 0, "ctn_dw_fwd: bad sizes");
    CTN_REQUIRE(P >= 1 && P <= MAXP && dilation >= 1, "ctn_dw_fwd: kernel size %d unsupported (max %d)", P, MAXP);
    CTN_REQUIRE(aligned16(Y) && aligned16(Z), "ctn_dw_fwd: pointers must be 16-byte aligned");
    const int halo = (P - 1) * dilation;
    CTN_REQUIRE(causal || halo % 2 == 0, "ctn_dw_fwd: non-causal 'same' padding needs (P-1)*dilation even");
    const bool small = halo <= 192;
    const int seg = (((small ? FWD_BUF_S : FWD_BUF_L) - halo - 8) / 64) * 64;
    CTN_REQUIRE(seg >= 64, "ctn_dw_fwd: receptive field (P-1)*dilation=%d too large", halo);
    CTN_REQUIRE(!pro_part || (pro_gamma && pro_beta && pro_alpha && pro_nparts > 0), "ctn_dw_fwd: incomplete prologue arguments");
    CTN_REQUIRE(!epi_part || epi_alpha, "ctn_dw_fwd: stats epilogue needs alpha");
    DwFwdArgs a{};
    a.Y = Y; a.Z = Z; a.D = D; a.M = M; a.H = H; a.K = K; a.Kp = Kp; a.P = P; a.dil = dilation;
    a.padl = causal ? halo : halo / 2; a.seg = seg;
    a.pro_part = pro_part; a.pro_nparts = pro_nparts; a.pro_gamma = pro_gamma; a.pro_beta = pro_beta;
    a.pro_alpha = pro_alpha; a.pro_ms_out = pro_ms_out; a.epi_alpha = epi_alpha; a.epi_part = epi_part;
    a.amax_out = amax_out;
    return dw_fwd_launch(a, pro_part ? 1 : 0, epi_part != nullptr, halo <= 192, stream);
}

// channel-wise LayerNorm form (round 4): n = gamma ((prelu(Y, alpha) - mean[k]) rstd[k]) + beta with the per-frame statistics that
// ctn_cln_stats_frame made of the producing GEMM's column partials; no separate norm pass, the norm's output is never stored
int ctn_dw_fwd_cln(const float* Y, float* Z, const float* D, int M, int H, int K, int Kp, int P, int dilation, int causal,
                   const float* mean, const float* rstd, const float* gamma, const float* beta, const float* alpha, void* stream) {
    CTN_REQUIRE(Y && Z && D && mean && rstd && gamma && beta && alpha, "ctn_dw_fwd_cln: null pointer");
    CTN_REQUIRE(M > 0 && H > 0 && K > 0 && Kp >= K && Kp % 4 == 0, "ctn_dw_fwd_cln: bad sizes");
    CTN_REQUIRE(P >= 1 && P <= MAXP && dilation >= 1, "ctn_dw_fwd_cln: kernel size %d unsupported (max %d)", P, MAXP);
    CTN_REQUIRE(aligned16(Y) && aligned16(Z) && aligned16(mean) && aligned16(rstd), "ctn_dw_fwd_cln: pointers must be 16-byte aligned");
    const int halo = (P - 1) * dilation;
    CTN_REQUIRE(causal || halo % 2 == 0, "ctn_dw_fwd_cln: non-causal 'same' padding needs (P-1)*dilation even");
    const bool small = halo <= 192;
    const int seg = (((small ? FWD_BUF_S : FWD_BUF_L) - halo - 8) / 64) * 64;
    CTN_REQUIRE(seg >= 64, "ctn_dw_fwd_cln: receptive field (P-1)*dilation=%d too large", halo);
    DwFwdArgs a{};
    a.Y = Y; a.Z = Z; a.D = D; a.M = M; a.H = H; a.K = K; a.Kp = Kp; a.P = P; a.dil = dilation;
    a.padl = causal ? halo : halo / 2; a.seg = seg;
    a.pro_gamma = gamma; a.pro_beta = beta; a.pro_alpha = alpha; a.cln_mean = mean; a.cln_rstd = rstd;
    return dw_fwd_launch(a, 2, false, small, stream);
}

}  // extern "C"

static int dw_fwd_launch(const DwFwdArgs& a, int pro, bool epi, bool small, void* stream) {
    const int M = a.M, H = a.H, P = a.P;
    const dim3 grid((unsigned)(M * ctn_cdiv(H, ROWS))), block(NT);
    hipStream_t st = (hipStream_t)stream;
    const bool vec4 = (a.dil % 4 == 0) && (a.padl % 4 == 0);
#define CTN_DW_FWD_P(P_, E_, PT_)                                                                               \
    do {                                                                                                        \
        if (small && vec4) hipLaunchKernelGGL((dw_fwd_kernel<P_, E_, FWD_BUF_S, true, PT_>), grid, block, 0, st, a);  \
        else if (small) hipLaunchKernelGGL((dw_fwd_kernel<P_, E_, FWD_BUF_S, false, PT_>), grid, block, 0, st, a);    \
        else if (vec4) hipLaunchKernelGGL((dw_fwd_kernel<P_, E_, FWD_BUF_L, true, PT_>), grid, block, 0, st, a);      \
        else hipLaunchKernelGGL((dw_fwd_kernel<P_, E_, FWD_BUF_L, false, PT_>), grid, block, 0, st, a);               \
    } while (0)
    // kernel size 3 (every configuration of the paper) is compiled in; other sizes take the run-time-P variant
#define CTN_DW_FWD(P_, E_)                  \
    do {                                    \
        if (P == 3) CTN_DW_FWD_P(P_, E_, 3);  \
        else CTN_DW_FWD_P(P_, E_, 0);         \
    } while (0)
    if (pro == 2) CTN_DW_FWD(2, false);
    else if (pro && epi) CTN_DW_FWD(1, true);
    else if (pro) CTN_DW_FWD(1, false);
    else if (epi) CTN_DW_FWD(0, true);
    else CTN_DW_FWD(0, false);
#undef CTN_DW_FWD_P
#undef CTN_DW_FWD
    CTN_CHECK_LAUNCH("ctn_dw_fwd");
    return CTN_OK;
}

extern "C" {

static unsigned* g_dw_bwd_amax = nullptr;       // (hand-over from ctn_dw_bwd_gln2 to the shared launcher below: one host thread at a time)
int ctn_dw_bwd_rows(int P, int fused) { return fused == 1 ? P + 5 : (fused == 2 ? P + 3 : (fused == 3 ? P + 6 : P)); }

// see include/ctn_hip.h.  pc is [F, M, H] with F = ctn_dw_bwd_rows(P, fused)
int ctn_dw_bwd(const float* dN2, const float* Dz, const float* Y1, float* dN1, const float* D,
               int M, int H, int K, int Kp, int P, int dilation, int causal, int fused,
               const float* g1, const float* b1, const float* a1, const float* ms1,
               const float* g2, const float* a2, const float* ms2,
               const double* sums2_part, int sums2_nparts, float* pc, double* sums1_part, void* stream) {
    CTN_REQUIRE(dN2 && Y1 && dN1 && D && pc, "ctn_dw_bwd: null pointer");
    CTN_REQUIRE(M > 0 && H > 0 && K > 0 && Kp >= K && Kp % 4 == 0, "ctn_dw_bwd: bad sizes");
    CTN_REQUIRE(P >= 1 && P <= MAXP && dilation >= 1, "ctn_dw_bwd: kernel size %d unsupported (max %d)", P, MAXP);
    CTN_REQUIRE(aligned16(dN2) && aligned16(Y1) && aligned16(dN1) && (!fused || aligned16(Dz)), "ctn_dw_bwd: alignment");
    const int halo = (P - 1) * dilation;
    CTN_REQUIRE(causal || halo % 2 == 0, "ctn_dw_bwd: non-causal 'same' padding needs (P-1)*dilation even");
    // patch size by halo: the dilation-128 blocks of the paper stack (halo 256) ran at 107 us with the 1792-float patches
    // (2 workgroups per CU) against 52 us for the others; 1280 floats = 960-frame segments at 4 workgroups per CU
    const bool small = halo <= 128, medium = !small && halo <= 256;
    const int seg = (((small ? BWD_BUF_S : (medium ? BWD_BUF_M : BWD_BUF_L)) - halo - 8) / 64) * 64;
    CTN_REQUIRE(seg >= 64, "ctn_dw_bwd: receptive field (P-1)*dilation=%d too large", halo);
    CTN_REQUIRE(fused >= 0 && fused <= 3, "ctn_dw_bwd: fused must be 0, 1 (gLN), 2 (cLN: ctn_dw_bwd_cln) or 3 (gLN + first norm's backward: ctn_dw_bwd_gln2)");
    if (fused == 1 || fused == 3)
        CTN_REQUIRE(Dz && g1 && b1 && a1 && ms1 && g2 && a2 && ms2 && sums2_part && sums2_nparts > 0 && (fused == 3 || sums1_part),
                    "ctn_dw_bwd: fused mode needs every norm argument");
    const bool xcln = fused == 2 && g1 != nullptr;        // the cLN form with the first norm's output recomputed from Y1 = h1
    if (fused == 2) {       // (ms2 carries the per-frame constants fc [M][4][Kp]; ms1 / sums2_part the first norm's mean / rstd [M][Kp])
        CTN_REQUIRE(Dz && g2 && a2 && ms2 && aligned16(ms2), "ctn_dw_bwd_cln: null or unaligned argument");
        CTN_REQUIRE(!xcln || (b1 && a1 && ms1 && sums2_part && aligned16(ms1) && aligned16(sums2_part)), "ctn_dw_bwd_cln: incomplete first-norm arguments");
    }
    DwBwdArgs a{};
    a.dN2 = dN2; a.Dz = Dz; a.Y1 = Y1; a.dN1 = dN1; a.D = D;
    a.M = M; a.H = H; a.K = K; a.Kp = Kp; a.P = P; a.dil = dilation; a.padl = causal ? halo : halo / 2; a.seg = seg;
    a.g1 = g1; a.b1 = b1; a.a1 = a1; a.ms1 = ms1; a.g2 = g2; a.a2 = a2; a.ms2 = ms2;
    a.sums2_part = sums2_part; a.sums2_nparts = sums2_nparts; a.pc = pc; a.sums1_part = sums1_part;
    if (fused == 2) {
        a.fc2 = ms2; a.ms2 = nullptr;
        if (xcln) { a.mean1f = ms1; a.rstd1f = reinterpret_cast<const float*>(sums2_part); }
        a.ms1 = nullptr; a.sums2_part = nullptr; a.sums2_nparts = 0;
    }
    const dim3 grid((unsigned)(M * ctn_cdiv(H, ROWS))), block(NT);
    hipStream_t st = (hipStream_t)stream;
    // float4 compute path whenever the tap offsets keep 16-byte alignment (dilation and left pad multiples of 4): with the
    // kernel size compiled in it also wins for the small-patch variant (44.7 vs 48.5 us at dilation 4..32)
    const bool vec4 = (a.dil % 4 == 0) && (a.padl % 4 == 0);
#define CTN_DW_BWD_P(D_, X_, PT_)                                                                                   \
    do {                                                                                                        \
        if (small && vec4) hipLaunchKernelGGL((dw_bwd_kernel<D_, X_, BWD_BUF_S, true, PT_>), grid, block, 0, st, a);      \
        else if (small) hipLaunchKernelGGL((dw_bwd_kernel<D_, X_, BWD_BUF_S, false, PT_>), grid, block, 0, st, a);        \
        else if (medium && vec4) hipLaunchKernelGGL((dw_bwd_kernel<D_, X_, BWD_BUF_M, true, PT_>), grid, block, 0, st, a);  \
        else if (medium) hipLaunchKernelGGL((dw_bwd_kernel<D_, X_, BWD_BUF_M, false, PT_>), grid, block, 0, st, a);       \
        else if (vec4) hipLaunchKernelGGL((dw_bwd_kernel<D_, X_, BWD_BUF_L, true, PT_>), grid, block, 0, st, a);          \
        else hipLaunchKernelGGL((dw_bwd_kernel<D_, X_, BWD_BUF_L, false, PT_>), grid, block, 0, st, a);                   \
    } while (0)
#define CTN_DW_BWD(D_, X_)                  \
    do {                                    \
        if (P == 3) CTN_DW_BWD_P(D_, X_, 3);  \
        else CTN_DW_BWD_P(D_, X_, 0);         \
    } while (0)
    if (fused == 3) { a.amax_out = g_dw_bwd_amax; CTN_DW_BWD(1, 3); }
    else if (fused == 1) CTN_DW_BWD(1, 1);
    else if (xcln) CTN_DW_BWD(2, 2);
    else if (fused == 2) CTN_DW_BWD(2, 0);
    else CTN_DW_BWD(0, 0);
#undef CTN_DW_BWD_P
#undef CTN_DW_BWD
    CTN_CHECK_LAUNCH("ctn_dw_bwd");
    return CTN_OK;
}

int ctn_dw_bwd_finalize(const float* pc, int P, int M, int H, float* dD, float* dgamma2, float* dbeta2,
                        float* dgamma1, float* dbeta1, float* dalpha2, const float* dalpha1_part, int n_dalpha1,
                        float* dalpha1, void* stream) {
    CTN_REQUIRE(pc && dD && dgamma2 && dbeta2 && dgamma1 && dbeta1 && dalpha2, "ctn_dw_bwd_finalize: null pointer");
    CTN_REQUIRE(P >= 1 && P <= MAXP && M > 0 && H > 0, "ctn_dw_bwd_finalize: bad sizes");
    CTN_REQUIRE(!dalpha1_part || (dalpha1 && n_dalpha1 > 0), "ctn_dw_bwd_finalize: incomplete dalpha1 arguments");
    const unsigned nb = (unsigned)ctn_cdiv((P + 4) * H, NT) + 2;
    hipLaunchKernelGGL(dw_bwd_finalize_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, pc, P, M, H, dD, dgamma2,
                       dbeta2, dgamma1, dbeta1, dalpha2, dalpha1_part, n_dalpha1, dalpha1);
    CTN_CHECK_LAUNCH("ctn_dw_bwd_finalize");
    return CTN_OK;
}

// gLN form with the first norm's backward applied (round 4): see include/ctn_hip.h.  sums2_part [M, nparts, 8] from ctn_pw_dgrad_gln2;
// dY1 = gLN1' . PReLU1'(dN1) [M,H,Kp]; pc [P+6, M, H] (rows as fused = 1, plus P+5: dalpha1 partials); amax_out optional (h3).
int ctn_dw_bwd_gln2(const float* dN2, const float* Dz, const float* Y1, float* dY1, const float* D,
                    int M, int H, int K, int Kp, int P, int dilation, int causal,
                    const float* g1, const float* b1, const float* a1, const float* ms1,
                    const float* g2, const float* a2, const float* ms2,
                    const double* sums2_part, int sums2_nparts, float* pc, unsigned* amax_out, void* stream) {
    g_dw_bwd_amax = amax_out;
    const int rc = ctn_dw_bwd(dN2, Dz, Y1, dY1, D, M, H, K, Kp, P, dilation, causal, 3, g1, b1, a1, ms1, g2, a2, ms2, sums2_part, sums2_nparts,
                              pc, nullptr, stream);
    g_dw_bwd_amax = nullptr;
    return rc;
}

// cLN form (round 4): see include/ctn_hip.h
int ctn_dw_bwd_cln(const float* dN2, const float* Dz, const float* X1, float* dN1, const float* D,
                   int M, int H, int K, int Kp, int P, int dilation, int causal,
                   const float* g2, const float* a2, const float* fc,
                   const float* g1, const float* b1, const float* a1, const float* mean1, const float* rstd1,
                   float* pc, void* stream) {
    CTN_REQUIRE((g1 == nullptr) == (mean1 == nullptr) && (g1 == nullptr) == (rstd1 == nullptr), "ctn_dw_bwd_cln: first-norm arguments come together");
    return ctn_dw_bwd(dN2, Dz, X1, dN1, D, M, H, K, Kp, P, dilation, causal, 2, g1, b1, a1, mean1, g2, a2, fc,
                      reinterpret_cast<const double*>(rstd1), 0, pc, nullptr, stream);
}

int ctn_dw_bwd_cln_finalize(const float* pc, int P, int M, int H, float* dD, float* dgamma2, float* dbeta2, float* dalpha2,
                            void* stream) {
    CTN_REQUIRE(pc && dD && dgamma2 && dbeta2 && dalpha2, "ctn_dw_bwd_cln_finalize: null pointer");
    CTN_REQUIRE(P >= 1 && P <= MAXP && M > 0 && H > 0, "ctn_dw_bwd_cln_finalize: bad sizes");
    const unsigned nb = (unsigned)ctn_cdiv((P + 2) * H, NT) + 1;
    hipLaunchKernelGGL(dw_bwd_cln_finalize_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, pc, P, M, H, dD, dgamma2, dbeta2, dalpha2);
    CTN_CHECK_LAUNCH("ctn_dw_bwd_cln_finalize");
    return CTN_OK;
}

int ctn_cln_bwd_frame(const double* col_part, int nparts, const float* mean, const float* rstd, float* fc, int M, int Ch, int Kp,
                      void* stream) {
    CTN_REQUIRE(col_part && mean && rstd && fc && nparts > 0 && M > 0 && Ch > 0 && Kp > 0, "ctn_cln_bwd_frame: bad arguments");
    CTN_REQUIRE(aligned16(col_part), "ctn_cln_bwd_frame: col_part must be 16-byte aligned");
    hipLaunchKernelGGL(cln_bwd_frame_kernel, dim3((unsigned)ctn_cdivll((long long)M * Kp, NT)), dim3(NT), 0, (hipStream_t)stream,
                       col_part, nparts, mean, rstd, fc, M, Ch, Kp);
    CTN_CHECK_LAUNCH("ctn_cln_bwd_frame");
    return CTN_OK;
}

int ctn_cln_stats_frame(const double* col_part, int nparts, float* mean, float* rstd, int M, int Ch, int Kp, void* stream) {
    CTN_REQUIRE(col_part && mean && rstd && nparts > 0 && M > 0 && Ch > 0 && Kp > 0, "ctn_cln_stats_frame: bad arguments");
    CTN_REQUIRE(aligned16(col_part), "ctn_cln_stats_frame: col_part must be 16-byte aligned");
    hipLaunchKernelGGL(cln_stats_frame_kernel, dim3((unsigned)ctn_cdivll((long long)M * Kp, NT)), dim3(NT), 0, (hipStream_t)stream,
                       col_part, nparts, mean, rstd, M, Ch, Kp);
    CTN_CHECK_LAUNCH("ctn_cln_stats_frame");
    return CTN_OK;
}

int ctn_dw_bwd_taps(const float* pc, int P, int M, int H, float* dD, void* stream) {
    CTN_REQUIRE(pc && dD && P >= 1 && P <= MAXP && M > 0 && H > 0, "ctn_dw_bwd_taps: bad arguments");
    hipLaunchKernelGGL(dw_bwd_taps_kernel, dim3((unsigned)ctn_cdiv(P * H, NT)), dim3(NT), 0, (hipStream_t)stream, pc, P, M, H, dD);
    CTN_CHECK_LAUNCH("ctn_dw_bwd_taps");
    return CTN_OK;
}

int ctn_gln_prelu_bwd(const float* dN, const float* Y, float* dY, int M, int H, int K, int Kp,
                      const float* gamma, const float* alpha, const float* ms, const double* sums_part, int nparts,
                      float* dalpha_part, unsigned* amax_out, void* stream) {
    CTN_REQUIRE(dN && Y && dY && gamma && alpha && ms && sums_part && dalpha_part && nparts > 0, "ctn_gln_prelu_bwd: null pointer");
    CTN_REQUIRE(M > 0 && H > 0 && K > 0 && Kp >= K && Kp % 4 == 0, "ctn_gln_prelu_bwd: bad sizes");
    CTN_REQUIRE(aligned16(dN) && aligned16(Y) && aligned16(dY), "ctn_gln_prelu_bwd: alignment");
    hipLaunchKernelGGL(gln_prelu_bwd_kernel, dim3((unsigned)(M * ctn_cdiv(H, ROWS))), dim3(NT), 0, (hipStream_t)stream,
                       dN, Y, dY, M, H, K, Kp, gamma, alpha, ms, sums_part, nparts, dalpha_part, amax_out);
    CTN_CHECK_LAUNCH("ctn_gln_prelu_bwd");
    return CTN_OK;
}

int ctn_gln_bwd_sums(const float* dN, const float* Y, int M, int H, int K, int Kp, const float* gamma, const float* alpha,
                     const float* ms, double* sums_part, float* pc, void* stream) {
    CTN_REQUIRE(dN && Y && gamma && alpha && ms && sums_part && pc, "ctn_gln_bwd_sums: null pointer");
    CTN_REQUIRE(M > 0 && H > 0 && K > 0 && Kp >= K && Kp % 4 == 0, "ctn_gln_bwd_sums: bad sizes");
    CTN_REQUIRE(aligned16(dN) && aligned16(Y), "ctn_gln_bwd_sums: alignment");
    hipLaunchKernelGGL(gln_bwd_sums_kernel, dim3((unsigned)(M * ctn_cdiv(H, ROWS))), dim3(NT), 0, (hipStream_t)stream,
                       dN, Y, M, H, K, Kp, gamma, alpha, ms, sums_part, pc);
    CTN_CHECK_LAUNCH("ctn_gln_bwd_sums");
    return CTN_OK;
}

// frames per workgroup of the cln_*_v4 kernels: 16 (256 threads, three and more workgroups per CU; default since round 4) or 32
// (512 threads); ctn_tune("cln_fr", 16 | 32).  The backward partial buffers are sized and summed by this count for every kernel
// of the family.
int g_ctn_cln_fr = 16;
int g_ctn_cln_fuse = -1;         // ctn_tune("cln_fuse", 0 | 1 | 2): 1 = composite cLN stacks run the second norm's backward inside the input-gradient
                                 // GEMM's epilogue (per-frame sums) and the depthwise backward's dd image instead of as a pass of its own
int ctn_cln_fuse(void) {         // (-1: not read yet; CTN_CLN_FUSE=0|1|2 at first use, for fresh-process A/B runs; default 2)
    if (g_ctn_cln_fuse < 0) {
        const char* e = getenv("CTN_CLN_FUSE");
        g_ctn_cln_fuse = (e && *e >= '0' && *e <= '2' && !e[1]) ? *e - '0' : 2;
    }
    return g_ctn_cln_fuse;
}
 // ctn_tune("gln_fuse", 0 | 1): composite gLN stacks without the gLN-1' / PReLU-1' pass (ctn_pw_dgrad_gln2 + ctn_dw_bwd_gln2)
int g_ctn_gln_fuse = -1;
int ctn_gln_fuse(void) {         // (CTN_GLN_FUSE=0|1 at first use; default below)
    if (g_ctn_gln_fuse < 0) {
        const char* e = getenv("CTN_GLN_FUSE");
        g_ctn_gln_fuse = (e && (*e == '0' || *e == '1') && !e[1]) ? *e - '0' : 0;      // measured slower in the step (profiles/README.md r04_m): off
    }
    return g_ctn_gln_fuse;
}
int g_ctn_cln_lean = 1;          // ctn_tune("cln_lean", 0 | 1): the specialised backward kernel for the stacks' form

static bool cln_v4_ok(int Ch, int Kp, const void* a, const void* b, const void* c) {
    return Ch <= 8 * C4_NG && Kp % C4_FR == 0 && aligned16(a) && aligned16(b) && aligned16(c);
}

int ctn_cln_fwd(const float* Y, float* Out, float* mean, float* rstd, int M, int Ch, int K, int Kp,
                const float* gamma, const float* beta, const float* alpha, unsigned* amax_out, void* stream) {
    CTN_REQUIRE(Y && Out && mean && rstd && gamma && beta, "ctn_cln_fwd: null pointer");
    CTN_REQUIRE(M > 0 && Ch > 0 && K > 0 && Kp >= K, "ctn_cln_fwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (cln_v4_ok(Ch, Kp, Y, Out, mean) && aligned16(rstd)) {     // 16-byte accesses along frames (round 2)
        const bool small = g_ctn_cln_fr == 16;                     // (Kp % 32 == 0 was checked: 16 divides it)
        const dim3 grid((unsigned)(M * (Kp / (small ? 16 : C4_FR))));
#define CTN_CLN_FWD4(CPT_) do { if (small) hipLaunchKernelGGL((cln_fwd_v4_kernel<CPT_, 256, 16>), grid, dim3(256), 0, st, Y, Out, mean, rstd, M, Ch, K, Kp, gamma, beta, alpha, amax_out); \
                                else hipLaunchKernelGGL((cln_fwd_v4_kernel<CPT_>), grid, dim3(C4_NT), 0, st, Y, Out, mean, rstd, M, Ch, K, Kp, gamma, beta, alpha, amax_out); } while (0)
        const int cpt = ctn_cdiv(Ch, C4_NG);
        if (cpt <= 1) CTN_CLN_FWD4(1);
        else if (cpt <= 2) CTN_CLN_FWD4(2);
        else if (cpt <= 4) CTN_CLN_FWD4(4);
        else CTN_CLN_FWD4(8);
#undef CTN_CLN_FWD4
        CTN_CHECK_LAUNCH("ctn_cln_fwd");
        return CTN_OK;
    }
    const dim3 grid_r((unsigned)(M * ctn_cdiv(Kp, CLN_FR)));
#define CTN_CLN_FWD(CPT_) hipLaunchKernelGGL((cln_fwd_reg_kernel<CLN_FR, CPT_>), grid_r, dim3(CLN_NT), 0, st, Y, Out, mean, rstd, M, Ch, K, Kp, gamma, beta, alpha)
    const int cpt = ctn_cdiv(Ch, CLN_NT / CLN_FR);
    if (cpt <= 2) CTN_CLN_FWD(2);
    else if (cpt <= 4) CTN_CLN_FWD(4);
    else if (cpt <= 8) CTN_CLN_FWD(8);
    else if (cpt <= 16) CTN_CLN_FWD(16);
    else if (cpt <= 32) CTN_CLN_FWD(32);
    else hipLaunchKernelGGL(cln_fwd_kernel, dim3((unsigned)(M * ctn_cdiv(Kp, 64))), dim3(NT), 0, st, Y, Out, mean, rstd, M, Ch, K, Kp,
                            gamma, beta, alpha);
#undef CTN_CLN_FWD
    CTN_CHECK_LAUNCH("ctn_cln_fwd");
    if (amax_out != nullptr) return ctn_absmax_rows(Out, M, (long long)Ch * Kp, amax_out, stream);     // fallback kernels: a pass of its own
    return CTN_OK;
}

int ctn_cln_bwd_blocks(int M, int Kp) { return M * ctn_cdiv(Kp, g_ctn_cln_fr); }   // rows of the parameter-gradient partials
size_t ctn_cln_bwd_pc_floats(int M, int Ch, int Kp) { return (size_t)2 * ctn_cln_bwd_blocks(M, Kp) * Ch; }

// see include/ctn_hip.h.  pc is [2][ctn_cln_bwd_blocks(M, Kp)][Ch]
int ctn_cln_bwd(const float* dOut, const float* Y, float* dY, const float* mean, const float* rstd,
                int M, int Ch, int K, int Kp, const float* gamma, const float* alpha,
                const float* add, const float* relu_ref, float* dalpha_part, float* pc, unsigned* amax_out, void* stream) {
    CTN_REQUIRE(dOut && Y && dY && mean && rstd && gamma && pc, "ctn_cln_bwd: null pointer");
    CTN_REQUIRE(M > 0 && Ch > 0 && K > 0 && Kp >= K && Kp % 4 == 0, "ctn_cln_bwd: bad sizes");
    CTN_REQUIRE(!alpha || dalpha_part, "ctn_cln_bwd: dalpha_part required with alpha");
    CTN_REQUIRE(aligned16(dOut) && aligned16(Y) && aligned16(mean) && aligned16(rstd), "ctn_cln_bwd: alignment");
    hipStream_t st = (hipStream_t)stream;
    float* const dap = alpha ? dalpha_part : nullptr;
    const int rows = ctn_cln_bwd_blocks(M, Kp);
    if (cln_v4_ok(Ch, Kp, dOut, Y, dY) && Kp % g_ctn_cln_fr == 0 && (!add || aligned16(add)) && (!relu_ref || aligned16(relu_ref))) {
        // one pass: input gradient AND the parameter-gradient partials (dY may alias dOut: each thread reads its elements
        // of dOut before it writes them)
        const dim3 grid((unsigned)rows);
#define CTN_CLN_BWD4(CPT_) do { if (g_ctn_cln_fr == 16) hipLaunchKernelGGL((cln_bwd_v4_kernel<CPT_, 256, 16>), grid, dim3(256), 0, st, dOut, Y, dY, mean, rstd, M, Ch, K, Kp, gamma, alpha, add, relu_ref, dap, pc, amax_out); \
                                else hipLaunchKernelGGL((cln_bwd_v4_kernel<CPT_, C4_NT, C4_FR>), grid, dim3(C4_NT), 0, st, dOut, Y, dY, mean, rstd, M, Ch, K, Kp, gamma, alpha, add, relu_ref, dap, pc, amax_out); } while (0)
        // the stacks' form (every channel group full, PReLU fused, no added gradient, no ReLU mask) has a kernel of its own
        const bool lean = g_ctn_cln_lean && Ch == 8 * C4_NG && g_ctn_cln_fr == 16 && alpha && !add && !relu_ref;
        if (lean) hipLaunchKernelGGL((cln_bwd_v4_kernel<8, 256, 16, true>), grid, dim3(256), 0, st, dOut, Y, dY, mean, rstd, M, Ch, K, Kp, gamma, alpha, add, relu_ref, dap, pc, amax_out);
        else if (Ch <= C4_NG) CTN_CLN_BWD4(1);
        else if (Ch <= 2 * C4_NG) CTN_CLN_BWD4(2);
        else if (Ch <= 4 * C4_NG) CTN_CLN_BWD4(4);
        else CTN_CLN_BWD4(8);
#undef CTN_CLN_BWD4
        CTN_CHECK_LAUNCH("ctn_cln_bwd");
        return CTN_OK;
    }
    // fallback (very wide layers / unaligned frames): parameter partials first (dY may alias dOut), into the first M rows of pc;
    // these kernels work in 32-frame blocks whatever g_ctn_cln_fr is: the partial rows they do not write stay zero
    hipMemsetAsync(pc, 0, sizeof(float) * ctn_cln_bwd_pc_floats(M, Ch, Kp), st);
    if (dap) hipMemsetAsync(dap, 0, sizeof(float) * (size_t)rows, st);
    hipLaunchKernelGGL(cln_bwd_params_kernel, dim3((unsigned)(M * ctn_cdiv(Ch, ROWS))), dim3(NT), 0, st,
                       dOut, Y, mean, rstd, M, Ch, K, Kp, alpha, pc, rows);
    CTN_CHECK_LAUNCH("ctn_cln_bwd/params");
    const dim3 grid_r((unsigned)(M * ctn_cdiv(Kp, CLN_FR)));
#define CTN_CLN_BWD(CPT_, NTB_) hipLaunchKernelGGL((cln_bwd_dx_reg_kernel<CLN_FR, CPT_, NTB_>), grid_r, dim3(NTB_), 0, st, dOut, Y, dY, mean, rstd, M, Ch, K, Kp, gamma, alpha, add, relu_ref, dap)
    if (Ch <= 32) CTN_CLN_BWD(2, 512);
    else if (Ch <= 64) CTN_CLN_BWD(4, 512);
    else if (Ch <= 128) CTN_CLN_BWD(8, 512);
    else if (Ch <= 256) CTN_CLN_BWD(16, 512);
    else if (Ch <= 512) CTN_CLN_BWD(16, 1024);
    else {      // generic kernel; it fills only the first M*ceil(Kp/64) partials of the buffer
        hipLaunchKernelGGL(cln_bwd_dx_kernel, dim3((unsigned)(M * ctn_cdiv(Kp, 64))), dim3(NT), 0, st, dOut, Y, dY, mean, rstd, M,
                           Ch, K, Kp, gamma, alpha, add, relu_ref, dap);
    }
#undef CTN_CLN_BWD
    CTN_CHECK_LAUNCH("ctn_cln_bwd/dx");
    if (amax_out != nullptr) return ctn_absmax_rows(dY, M, (long long)Ch * Kp, amax_out, stream);      // fallback kernels: a pass of its own
    return CTN_OK;
}

int ctn_cln_bwd_finalize(const float* pc, const float* dalpha_part, int M, int Ch, int Kp, float* dgamma, float* dbeta,
                         float* dalpha, void* stream) {
    CTN_REQUIRE(pc && dgamma && dbeta && M > 0 && Ch > 0 && Kp > 0, "ctn_cln_bwd_finalize: bad arguments");
    CTN_REQUIRE(!dalpha_part || dalpha, "ctn_cln_bwd_finalize: dalpha required with dalpha_part");
    const int rows = ctn_cln_bwd_blocks(M, Kp);
    const unsigned nb = (unsigned)(2 * ctn_cdiv(Ch, 64)) + 1;
    hipLaunchKernelGGL(cln_bwd_finalize_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, pc, dalpha_part, rows, Ch, rows,
                       dgamma, dbeta, dalpha);
    CTN_CHECK_LAUNCH("ctn_cln_bwd_finalize");
    return CTN_OK;
}

int ctn_reduce_mid(const float* in, float* out, int F, int Mid, int Inner, void* stream) {
    CTN_REQUIRE(in && out && F > 0 && Mid > 0 && Inner > 0, "ctn_reduce_mid: bad arguments");
    const long long n = (long long)F * Inner;
    const int wave_mode = (Mid >= 256 && n <= 4096) ? 1 : 0;
    const unsigned blocks = wave_mode ? (unsigned)n : (unsigned)ctn_cdivll(n, NT);
    hipLaunchKernelGGL(reduce_mid_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, in, F, Mid, Inner, out, wave_mode);
    CTN_CHECK_LAUNCH("ctn_reduce_mid");
    return CTN_OK;
}

}  // extern "C"
