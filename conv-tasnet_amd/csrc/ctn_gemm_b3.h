// Split arithmetics of the 1x1-convolution GEMMs: fp32 operands multiplied on the bf16 / f16 matrix cores as pieces.
//
//   "h3" (library default for the composite stacks; template id AR = 5, A/B builds 4):
//     a s = a0 + a1 + r,  a0 = fp16_rne(a s), a1 = fp16_rne(a s - a0) (exact difference): two 11-bit significands and the sign of
//     a1 hold 22-23 bits: |a1| <= 2^-11 |a s|, |r| <= 2^-23 |a s| (half an ulp of a1; rms over random data 0.6 * 2^-24);
//     a.b ~= (a1.b0 + a0.b1 + a0.b0) / (s_a s_b): THREE v_mfma_f32_32x32x16_f16, fp32 accumulation; dropped per product:
//     a1.b1 + r_a.b + a.r_b, worst case (4 + 2 + 2) 2^-24 = 8 * 2^-24 |a.b| (4.8e-7: larger than one fp32 rounding), 1.2 * 2^-24 rms.
//     PER PRODUCT h3 is therefore no better than fp32; what makes its GEMM error smaller than the fp32 MFMA's is the
//     ACCUMULATION: the f16 MFMA adds a 16-deep step exactly and rounds the accumulator once per step, where the fp32 MFMA
//     (and b6, whose six products go through the same accumulator) round it after every 2-deep (16-deep x 6) step -- rms error
//     1.3 : 2.3 : 2.8e-8 of sum |a||b| ~ sqrt(number of accumulator roundings).  The dropped products are signed and cancel over
//     a contraction; when they do not (every operand just below a rounding midpoint, all products positive) the result is off
//     by 2^-22 = 2.4e-7 .. 4.8e-7 of sum |a||b| -- still under the 6e-7 gate of the tests
//     (tests/test_gpu_h3.py::test_h3_adversarial_coherent_low_pieces).
//     s = an exact power of two per operand from a bound on its magnitude -- the weight's own maximum, and for activations the
//     per-utterance maximum TRACKED BY THE PRODUCING KERNEL (CTN_AMAX_SLOTS words per utterance, atomic max of bit patterns: exact
//     and order-free).  The low piece is stored times 2^11 and its two cross products have an accumulator of their own, so it is a
//     normal fp16 number for every element down to 2^-27 of the bound (the bounds above hold for those elements; smaller ones lose
//     RELATIVE precision gracefully: absolute error <= 2^-50 of the bound).  Measured against fp64 (benchmarks/h3_check.py): max
//     error 1.5e-7 of sum |a||b| (rms 1.3e-8) on every form and on operands 1e-20 .. 1e6 / utterances 2^80 apart / heavy tails,
//     against 3.9e-7 (2.3e-8) for b6 and 3.8e-7 (2.8e-8) for the fp32 MFMA.
//   "b6" (NP = 3 bf16 pieces; every GEMM outside the composite stacks, and the stacks under CTN_GEMM_ARITH=b6):
//     a = a0 + a1 + a2 EXACTLY,  a0 = bf16_rne(a), a1 = bf16_rne(a - a0), a2 = a - a0 - a1   (both differences are exact in fp32;
//     |a1| <= 2^-8 |a|, |a2| <= 2^-16 |a|, and a2 has at most 8 significant bits, so its conversion is exact as well)
//     a.b ~= a2.b0 + a0.b2 + a1.b1 + a1.b0 + a0.b1 + a0.b0       six bf16 MFMAs per 16-deep step, fp32 accumulation;
//     dropped: a1.b2 + a2.b1 + a2.b2, |.| <= (2 * 2^-24 + 2^-32) |a.b| -- the size of ONE fp32 rounding of the product, which the
//     fp32 FMA chain of the reference commits at every step as well.
//   (Round 2's two-piece bf16 "b3" -- ~16-bit products, not reference precision -- is gone: h3 costs the same three MFMAs.)
//
// Why not the fp32 MFMA: it runs at 1/16 of the bf16 / f16 rate, and the fp32-MFMA training step sits AT the 1400 W package power
// cap where its time is its energy.
// CTN_GEMM_ARITH=h3|b6|fp32 / ctn_tune("arith", 3|2|0) select the arithmetic; fp32 = the fp32-MFMA kernels of ctn_gemm.hip.
//
// Same contracts as the fp32 kernels (PwArgs / WgArgs, prologues, epilogues, tile order, fixed-order reductions).  Activation
// operands are split on the fly while they are staged global -> registers -> LDS (after the fused prologue), so no pre-split
// activation copies exist in HBM; weights are either split the same way (raw ctn_pw_gemm on fp32 weights, b6) or pre-split once
// per stack call into MFMA-fragment order (ctn_split_b3_batch / ctn_split_h3_batch, the product path of the composite stacks).
// Operands whose contraction index is strided in memory (activations [channel][frame], weights given as [contraction][row]) are
// stored channel-major in LDS and read with ds_read_b64_tr_b16, the hardware transposing read; operands with a contiguous
// contraction (stored [row][contraction] weights, both operands of the weight gradient) are stored row-major and read with
// ds_read_b128.
// Included by ctn_gemm.hip (the argument structs live in that translation unit's anonymous namespace).
#pragma once
#include "ctn_gemm_common.h"
#include <type_traits>

namespace {

constexpr int XK = 32;               // contraction steps per k-tile (two MFMA steps of depth 16)
constexpr int XPA = XK + 8;          // row pitch of a row-major piece plane in bf16 (80 B: conflict-free ds_read_b128 over 16 rows)

// Arithmetic ids of the kernels below (template parameter AR): 3 = b6 (three bf16 pieces), 4 / 5 = h3 (two fp16 pieces).
// 5 (the one built into the library, CTN_H3_AR) stores the low piece multiplied by 2^11 and sums the two cross products in an
// accumulator of their own (W2): the low piece then stays a NORMAL fp16 number for every element down to 2^-27 of its operand's
// bound instead of 2^-16 (id 4: both pieces at one scale, one accumulator -- kept for A/B builds, -DCTN_H3_AR=4).
template <int AR> struct Ar { static constexpr int NP = AR; static constexpr bool F16 = false, W2 = false; };
template <> struct Ar<4> { static constexpr int NP = 2; static constexpr bool F16 = true, W2 = false; };
template <> struct Ar<5> { static constexpr int NP = 2; static constexpr bool F16 = true, W2 = true; };
#ifndef CTN_H3_AR
#define CTN_H3_AR 5
#endif
constexpr int H3AR = CTN_H3_AR;
constexpr float H3_LOW = 2048.f;          // 2^11: scale of the low piece under W2

// four consecutive fp32 -> NP bf16x4 pieces, most significant first (round-to-nearest-even each time; the differences are exact).
// Written on packed pairs: one v_cvt_pk_bf16_f32 per pair and piece, the bf16 -> fp32 widening as a shift / mask of the packed
// register -- 5.5 VALU instructions per element for three pieces (the element-wise form compiled to 7.5).
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, bf16x2v));
}
__device__ __forceinline__ float pk_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float pk_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ unsigned pk_f16(float a, float b) {          // round-to-nearest-even (v_cvt_pk_f16_f32)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, f16x2v));
}
// `s` (h3 only): the operand's power-of-two scale.  The 16-bit containers are declared bf16x4 / bf16x8 for every arithmetic
// (LDS traffic and fragment moves do not look at the format); only the split and the MFMA know what the bits mean.
// PRESCALED: v already carries the scale (folded into the operand prologue's constants: a power of two commutes with rounding).
template <int AR, bool PRESCALED = false>
__device__ __forceinline__ void split_x4(const float4& v, bf16x4 (&q)[Ar<AR>::NP], float s) {
    static_assert(AR >= 3 && AR <= 5, "b6 or h3");
    if constexpr (Ar<AR>::F16) {
        // a s = a0 + a1 + r:  a0 = f16_rne(a s), a1 = f16_rne(a s - a0) (the difference is exact in fp32), |r| <= 2^-23 |a s| as
        // long as a1 is a normal fp16 number (half an ulp of a1, |a1| <= 2^-11 |a s|), |r| <= 2^-25 in absolute terms (half an
        // fp16 subnormal step) below that.  W2: the stored low piece is 2^11 a1 (|a s - a0| <= 2^-11 |a s|, so it stays below the
        // scaled bound): normal for |a s| >= 2^-13
        const float x0 = PRESCALED ? v.x : v.x * s, x1 = PRESCALED ? v.y : v.y * s, x2 = PRESCALED ? v.z : v.z * s, x3 = PRESCALED ? v.w : v.w * s;
        const unsigned h0 = pk_f16(x0, x1), h1 = pk_f16(x2, x3);
        const f32x2v w0 = __builtin_convertvector(__builtin_bit_cast(f16x2v, h0), f32x2v);
        const f32x2v w1 = __builtin_convertvector(__builtin_bit_cast(f16x2v, h1), f32x2v);
        float r0 = x0 - w0.x, r1 = x1 - w0.y, r2 = x2 - w1.x, r3 = x3 - w1.y;
        if constexpr (Ar<AR>::W2) { r0 *= H3_LOW; r1 *= H3_LOW; r2 *= H3_LOW; r3 *= H3_LOW; }
        const unsigned m0 = pk_f16(r0, r1), m1 = pk_f16(r2, r3);
        q[0] = __builtin_bit_cast(bf16x4, u32x2v{h0, h1});
        q[1] = __builtin_bit_cast(bf16x4, u32x2v{m0, m1});
    } else {
        (void)s;
        const unsigned h0 = pk_bf16(v.x, v.y), h1 = pk_bf16(v.z, v.w);
        const float r0 = v.x - pk_lo(h0), r1 = v.y - pk_hi(h0), r2 = v.z - pk_lo(h1), r3 = v.w - pk_hi(h1);
        const unsigned m0 = pk_bf16(r0, r1), m1 = pk_bf16(r2, r3);
        q[0] = __builtin_bit_cast(bf16x4, u32x2v{h0, h1});
        q[1] = __builtin_bit_cast(bf16x4, u32x2v{m0, m1});
        if constexpr (AR == 3) {
            const unsigned l0 = pk_bf16(r0 - pk_lo(m0), r1 - pk_hi(m0)), l1 = pk_bf16(r2 - pk_lo(m1), r3 - pk_hi(m1));
            q[2] = __builtin_bit_cast(bf16x4, u32x2v{l0, l1});
        }
    }
}

// ---- h3 scales.  An operand with |x| <= bound is multiplied by 2^e, e = h3_exp(bound), so that 2^14 <= bound 2^e < 2^15 (fp16
// holds 65504: a factor 2 of headroom for the rounding of a computed bound).
__device__ __forceinline__ int h3_exp(float bound) {          // bound >= 0 (NaN / inf: the scaled operand overflows to inf -> NaN)
    const int ex = (int)((__float_as_uint(bound) >> 23) & 0xffu);       // bound < 2^(ex - 126)
    const int e = 141 - ex;
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
__device__ __forceinline__ float h3_pow2(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }   // |e| <= 126
// bound of gamma ((prelu(x) - mean) rstd) + beta over an utterance whose stored values satisfy |x| <= amax
__device__ __forceinline__ float h3_pro_bound(float amax, float alpha, float mean, float rstd, const float* gbmax) {
    const float pa = fmaxf(1.f, fabsf(alpha)) * amax;
    return gbmax[0] * fabsf(rstd) * (pa + fabsf(mean)) + gbmax[1];
}
// acc = (acc + 2^-11 acc2) 2^-(ea + eb): v_ldexp_f32 (exact; overflows / underflows only where the result itself does)
template <bool W2, int N, int N2>
__device__ __forceinline__ void h3_unscale(f32x16 (&acc)[N], const f32x16 (&acc2)[N2], int ea, int eb) {
    const int t = -(ea + eb);
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = acc[i][e];
            if constexpr (W2) v = fmaf(acc2[i][e], 1.f / H3_LOW, v);
            acc[i][e] = __builtin_amdgcn_ldexpf(v, t);
        }
}

// The piece products of one 16-deep step in issue order, smallest terms first: (piece of A, piece of B).
template <int NP> struct Prods;
template <> struct Prods<2> {
    static constexpr int N = 3;
    static constexpr int A[N] = {1, 0, 0}, B[N] = {0, 1, 0};
};
template <> struct Prods<3> {
    static constexpr int N = 6;
    static constexpr int A[N] = {2, 0, 1, 1, 0, 0}, B[N] = {0, 2, 1, 0, 1, 0};
};
// acc += sum over the piece products of a (NP fragments) and b (NP fragments); W2: the cross products go to acc2
template <int AR>
__device__ __forceinline__ void mfma_pieces(f32x16& acc, f32x16& acc2, const bf16x8 (&fa)[Ar<AR>::NP], const bf16x8 (&fb)[Ar<AR>::NP]) {
    constexpr int NP = Ar<AR>::NP;
    if constexpr (Ar<AR>::W2) {
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[1]), __builtin_bit_cast(f16x8, fb[0]), acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[0]), __builtin_bit_cast(f16x8, fb[0]), acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[0]), __builtin_bit_cast(f16x8, fb[1]), acc2, 0, 0, 0);
        return;
    }
#pragma unroll
    for (int t = 0; t < Prods<NP>::N; ++t) {
        if constexpr (Ar<AR>::F16)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[Prods<NP>::A[t]]),
                                                         __builtin_bit_cast(f16x8, fb[Prods<NP>::B[t]]), acc, 0, 0, 0);
        else
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[Prods<NP>::A[t]], fb[Prods<NP>::B[t]], acc, 0, 0, 0);
    }
}

template <typename TL, int TRANS_W, int NP>
struct B3 {
    static constexpr int PB = TL::TN + 32;                                   // pitch of a channel-major plane row in bf16
    static constexpr int PA = TL::TM + 32;
    static constexpr int A_ELEMS = TRANS_W ? NP * XK * PA : NP * TL::TM * XPA;  // bf16 elements, all pieces
    static constexpr int B_ELEMS = NP * XK * PB;
    static constexpr int MAIN_BYTES = (A_ELEMS + B_ELEMS) * 2;
    static constexpr int STAGE_BYTES = TL::STAGE_FLOATS * 4;
    static constexpr int SMEM_BYTES = MAIN_BYTES > STAGE_BYTES ? MAIN_BYTES : STAGE_BYTES;
};

typedef bf16x4 __attribute__((address_space(3))) * lds_b4;

// 8 contraction steps (k = 8 * (lane / 32) .. + 7 of a 16-deep MFMA step) of column `lane % 32` of a 32-column tile,
// out of a channel-major plane (row = contraction index, pitch P): two transposing reads of 4 rows each.
// Lane map of ds_read_b64_tr_b16: the 16-lane group g = lane / 16 covers columns 16 (g & 1) .. + 15 and rows
// 8 (g / 2) .. + 7 of the 16 x 32 block; lane 4 q + p of a group addresses row q, columns 4 p .. 4 p + 3.
__device__ __forceinline__ bf16x8 frag_tr(const __bf16* plane_at_tile, int P, int lane) {
    const int tr_q = (lane >> 2) & 3, tr_f = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const __bf16* base = plane_at_tile + ((lane >> 5) * 8 + tr_q) * P + tr_f;
    const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(base));
    const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4)(base + 4 * P));
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int AR, typename TL, int TRANS_W, int PRO, int EPI>
__global__ __launch_bounds__(TL::NTH, (TL::TM * TL::TN <= 4096) ? 4 : ((TL::TM * TL::TN <= 8192 && AR == 2) ? 3 : 2))
void pw_gemm_b3_kernel(PwArgs a) {
    static_assert(AR == 3, "fp32 weights split on the fly: b6 only (h3 needs the weight's range)");
    constexpr int NP = Ar<AR>::NP;
    constexpr int TM = TL::TM, TN = TL::TN, MT = TL::MT, NTL = TL::NTL, WM = TL::WM, WN = TL::WN, NTH = TL::NTH;
    using L = B3<TL, TRANS_W, NP>;
    constexpr int PB = L::PB, PA = L::PA;
    constexpr int B_L = XK * TN / 4 / NTH, A_L = XK * TM / 4 / NTH;         // float4 loads per thread per k-tile
    constexpr int AT = XK / 4;                                              // threads per stored weight row (TRANS_W = 0)
    static_assert(A_L >= 1 && B_L >= 1, "tile too small for the workgroup");
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[L::SMEM_BYTES];
    __shared__ double red[NTH / 64];
    __bf16* const Ap = reinterpret_cast<__bf16*>(smem_raw);        // TRANS_W: [NP][XK][PA]   else [NP][TM][XPA]
    __bf16* const Bp = Ap + L::A_ELEMS;                            // [NP][XK][PB]

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
        finalize_stats<NTH>(a.pro_part + (size_t)m * a.pro_nparts * 2, a.pro_nparts, (double)a.Cn * (double)a.K, red, p_mean, p_rstd);
        p_alpha = a.pro_alpha[0];
        if (a.pro_ms_out != nullptr && rt == 0 && ct == 0 && tid == 0) {
            a.pro_ms_out[2 * m] = p_mean;
            a.pro_ms_out[2 * m + 1] = p_rstd;
        }
    }
    const int nk = (a.Cn + XK - 1) / XK;
    // Buffer loads: contraction rows past Cn of W^T / X / gamma / beta fall off the end of their buffer and read 0 (the
    // k-tile offset is part of the per-lane offset, so the range check sees it).  TRANS_W = 0 with Cn % 32 != 0 reads
    // the next weight row instead: finite values against all-zero activation rows.  Rows / columns that overhang R / Kp
    // are never stored (and skipped by the statistics).
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
    const int sA = (TRANS_W == 0 ? XK : XK * a.R) * 4, sB = XK * a.Kp * 4;      // byte steps per k-tile

    auto load_tile = [&](int kt, float4 (&ra)[A_L], float4 (&rb)[B_L], float2 (&rp)[B_L]) {
#pragma unroll
        for (int j = 0; j < A_L; ++j) ra[j] = buf_ld4(rsW, voA[j] + kt * sA, 0);
#pragma unroll
        for (int j = 0; j < B_L; ++j) {
            rb[j] = buf_ld4(rsX, voB[j] + kt * sB, 0);
            if constexpr (PRO == PRO_PRELU_NORM)
                rp[j] = make_float2(buf_ld1(rsG, voP[j] + kt * XK * 4, 0), buf_ld1(rsBt, voP[j] + kt * XK * 4, 0));
            else
                rp[j] = make_float2(0.f, 0.f);
        }
    };
    // split while writing to LDS (after the prologue): the global loads have no consumer before the MFMA phase they overlap
    auto store_tile = [&](const float4 (&ra)[A_L], const float4 (&rb)[B_L], const float2 (&rp)[B_L]) {
#pragma unroll
        for (int j = 0; j < A_L; ++j) {
            bf16x4 q[NP];
            split_x4<AR>(ra[j], q, 1.f);
            if constexpr (TRANS_W == 0) {
                const int r = tid / AT + (NTH / AT) * j, c = (tid % AT) * 4;
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(Ap + (p * TM + r) * XPA + c) = q[p];
            } else {
                const int c = tid / (TM / 4) + (4 * NTH / TM) * j, r = (tid % (TM / 4)) * 4;
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(Ap + (p * XK + c) * PA + r) = q[p];
            }
        }
#pragma unroll
        for (int j = 0; j < B_L; ++j) {
            const int i = tid / (TN / 4) + (4 * NTH / TN) * j, k = (tid % (TN / 4)) * 4;
            float4 v = rb[j];
            if constexpr (PRO == PRO_PRELU_NORM) v = pro_apply(v, c0 + k, a.K, rp[j].x, rp[j].y, p_alpha, p_mean, p_rstd);
            bf16x4 q[NP];
            split_x4<AR>(v, q, 1.f);
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(Bp + (p * XK + i) * PB + k) = q[p];
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
    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < XK / 16; ++ks) {
            bf16x8 af[MT][NP], bfr[NTL][NP];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if constexpr (TRANS_W == 0)
                        af[i][p] = *reinterpret_cast<const bf16x8*>(Ap + (p * TM + wm * WM + i * 32 + l31) * XPA + ks * 16 + lhi * 8);
                    else
                        af[i][p] = frag_tr(Ap + (p * XK + ks * 16) * PA + wm * WM + i * 32, PA, lane);
                }
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p) bfr[j][p] = frag_tr(Bp + (p * XK + ks * 16) * PB + wn * WN + j * 32, PB, lane);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) mfma_pieces<AR>(acc[i][j], acc[i][j], af[i], bfr[j]);      // small terms first
        }
    };

    // One LDS buffer, two barriers per k-tile; the global loads of the next k-tile wait in registers.  Other resident
    // workgroups of the CU fill the barrier gaps (2-4 per CU).
    float4 pa[A_L], pb[B_L];
    float2 pp[B_L];
    load_tile(0, pa, pb, pp);
    for (int kt = 0; kt < nk; ++kt) {
        store_tile(pa, pb, pp);
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1, pa, pb, pp);
        compute();
        __syncthreads();
    }
    gemm_epilogue<TL, EPI>(a, acc, reinterpret_cast<float*>(smem_raw), red, m, rt, ct);
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient: dW[r,c] = sum_{m,k} dOut[m,r,k] * f(X[m,c,k]).  Both operands have the contraction (frames) contiguous:
// row-major piece planes, ds_read_b128 fragments.  128 x 128 output tile per 512-thread workgroup: 8 waves as 2 x 4, each
// 64 rows x 32 columns -- twice the waves of a 2 x 2 arrangement for the same tile, so the split of the two activation
// tiles (both operands are activations: ~100 VALU instructions per thread and k-tile at 256 threads) is spread over 8
// waves and 2-3 resident workgroups give every SIMD 4-6 waves to hide the global-load latency with.  Two LDS stages, one
// barrier per 32-frame k-tile; the next k-tile's global loads wait in registers.  Split over (utterance, frame chunk) into
// fp32 slabs summed in fixed order by slab_reduce_kernel.
// ---------------------------------------------------------------------------------------------------------
constexpr int WNT = 512;
template <int AR, int PRO>
__global__ __launch_bounds__(WNT, 2) void pw_wgrad_b3_kernel(WgArgs a) {
    constexpr int NP = Ar<AR>::NP;
    constexpr int PLANE = BM * XPA, STAGE = NP * PLANE;       // bf16 elements: one piece plane, all pieces of one operand
    __shared__ __attribute__((aligned(16))) __bf16 smem[4 * STAGE];        // (h3: 80 KiB, b6: 120 KiB)
    __bf16* const Ap = smem;                                               // [stage][piece][BM][XPA]
    __bf16* const Bp = smem + 2 * STAGE;
    static_assert(sizeof(smem) >= 32 * 128 * sizeof(float4), "the chained slab reduction stages 32 x 128 float4 in this buffer");
    __shared__ double red[WNT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
#ifdef CTN_EXP_B3_WGNULL
    if (a.R > 0) return;
#endif
    // XCD-aware order: the output tiles of one (utterance, chunk) split read the same activation rows -- give them
    // consecutive logical ids, i.e. the same XCD and its L2 (dealt round-robin they land on 8 different L2s and every
    // operand byte comes from HBM once per tile: 208 MB instead of 79 MB per launch at the paper shapes)
    int bid = xcd_remap(blockIdx.x, gridDim.x);
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
    // h3: per-utterance power-of-two scales of the two operands from their tracked maxima (a slab covers one utterance)
    int eg = 0, ex = 0;
    float sg = 1.f, sx = 1.f;
    if constexpr (Ar<AR>::F16) {
        eg = h3_exp(amax_read(a.g_amax + (size_t)m * CTN_AMAX_SLOTS));
        const float xm = amax_read(a.x_amax + (size_t)m * CTN_AMAX_SLOTS);
        ex = h3_exp(PRO == PRO_PRELU_NORM ? h3_pro_bound(xm, p_alpha, p_mean, p_rstd, a.pro_gbmax) : xm);
        sg = h3_pow2(eg);
        sx = h3_pow2(ex);
    }
    // staging map: 8 threads per row (32 frames = 8 float4), 64 rows per pass, 2 passes for 128 rows.  Rows past R / Cn
    // fall off the end of the utterance's matrix and read 0; frames past the chunk end are pushed out of range.
    const __amdgpu_buffer_rsrc_t rsG = make_rsrc(Gm, (unsigned)a.R * (unsigned)a.Kp * 4u);
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(Xm, (unsigned)a.Cn * (unsigned)a.Kp * 4u);
    constexpr int PF = 3;           // k-tiles of global loads in flight (register ring)
    float4 ra[PF][2], rb[PF][2];
    float2 rg[2];
    if constexpr (PRO == PRO_PRELU_NORM) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + (tid >> 3) + 64 * j;
            rg[j] = c < a.Cn ? make_float2(a.pro_gamma[c] * sx, a.pro_beta[c] * sx) : make_float2(0.f, 0.f);     // (h3: the scale rides in the constants)
        }
    }
    const int nk = (ke - kb + XK - 1) / XK;
    const int kq = (tid & 7) * 4;
    auto load_regs = [&](int kt, float4 (&qa)[2], float4 (&qb)[2]) {
        const int k = kb + kt * XK + kq;
        const unsigned oob = k < ke ? 0u : 0x80000000u;
#ifdef CTN_EXP_B3_WGNOLOAD
        if (kt > 2) return;
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (tid >> 3) + 64 * j;
            qa[j] = buf_ld4(rsG, (int)((unsigned)(((r0 + row) * a.Kp + k) * 4) + oob), 0);
            qb[j] = buf_ld4(rsX, (int)((unsigned)(((c0 + row) * a.Kp + k) * 4) + oob), 0);
        }
    };
    auto write_one = [&](__bf16* P, int row, const float4& v, float sc, auto prescaled) {
        bf16x4 q[NP];
#ifdef CTN_EXP_B3_WGNOSPLIT
        q[0] = __builtin_bit_cast(bf16x4, make_float2(v.x, v.y));
        q[NP - 1] = __builtin_bit_cast(bf16x4, make_float2(v.z, v.w));
#else
        split_x4<AR, decltype(prescaled)::value>(v, q, sc);
#endif
#ifndef CTN_EXP_B3_WGNOWRITE
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(P + p * PLANE + row * XPA + kq) = q[p];
#else
        if (q[0][0] == (__bf16)123.f) *reinterpret_cast<bf16x4*>(P + row * XPA + kq) = q[1];
#endif
    };
    auto write_lds = [&](int kt, int stage, const float4 (&qa)[2], const float4 (&qb)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (tid >> 3) + 64 * j;
            write_one(Ap + stage * STAGE, row, qa[j], sg, std::false_type{});
            float4 x = qb[j];
            // (frames past the chunk end are frames >= Kp >= K: the prologue zeroes them like every frame >= K)
            if constexpr (PRO == PRO_PRELU_NORM) x = pro_apply(x, kb + kt * XK + kq, a.K, rg[j].x, rg[j].y, p_alpha, p_mean, p_rstd);
            write_one(Bp + stage * STAGE, row, x, sx, std::integral_constant<bool, PRO == PRO_PRELU_NORM>{});
        }
    };

    f32x16 acc[2];                  // rows wm * 64 + 32 i, columns wn * 32
    f32x16 acc2[Ar<AR>::W2 ? 2 : 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[i][e] = 0.f; acc2[Ar<AR>::W2 ? i : 0][e] = 0.f; }

    const int l31 = lane & 31, lhi = lane >> 5;
    auto compute = [&](int stage) {
        const __bf16* const As = Ap + stage * STAGE;
        const __bf16* const Bs = Bp + stage * STAGE;
#ifdef CTN_EXP_B3_WGNOCOMPUTE
        return;
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2][NP], bfr[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    af[i][p] = *reinterpret_cast<const bf16x8*>(As + p * PLANE + (wm * 64 + i * 32 + l31) * XPA + ks * 16 + lhi * 8);
                bfr[p] = *reinterpret_cast<const bf16x8*>(Bs + p * PLANE + (wn * 32 + l31) * XPA + ks * 16 + lhi * 8);
            }
#ifdef CTN_EXP_B3_WGNOMFMA
#pragma unroll
            for (int p = 0; p < NP; ++p) asm volatile("" :: "v"(af[0][p]), "v"(af[1][p]), "v"(bfr[p]));
#else
#pragma unroll
            for (int i = 0; i < 2; ++i) mfma_pieces<AR>(acc[i], acc2[Ar<AR>::W2 ? i : 0], af[i], bfr);
#endif
        }
    };
    // Register ring of PF k-tiles: at the top of iteration kt, LDS stage kt % 2 holds tile kt, ring slots (kt + 1 .. kt + PF - 1)
    // % PF hold tiles kt + 1 .. in flight or landed.  One workgroup per CU (256 workgroups, the measured optimum in the step)
    // has no other workgroup to hide the load latency behind: distance 1 left ~1 us of every 1.4 us k-tile waiting.
    // Unrolled by 6: ring slot and stage indices are compile-time constants.
    // (Round 3, b6: running the two halves of each barrier interval -- MFMAs of tile kt, split of tile kt + 1 -- in opposite
    // order on waves 4-7, so that each wave's VALU half sits beside its SIMD partner's MFMA half, measured SLOWER: 48.7 vs
    // 44.9 us alone, 13.71 vs 13.43 ms per step; profiles/README.md.)
    // Chained launches: this grid also sums the slabs of the previous weight gradient of its stream (complete: stream order).
    // Workgroup b owns float4s [b * per, (b + 1) * per) of that gradient; per batch of 32 slabs four thread groups load eight
    // slabs each (loads in flight beside this launch's first three k-tiles), exchange them through LDS, and threads 0-127 add
    // them in slab order -- the sum sequence of slab_reduce_kernel, bit for bit.
    const bool chained = a.prev_slab != nullptr;
    const long long pn4 = a.prev_n >> 2;
    const long long pper = chained ? (pn4 + gridDim.x - 1) / gridDim.x : 0;
    const long long plo = (long long)blockIdx.x * pper, phi = plo + pper < pn4 ? plo + pper : pn4;
    const int pf = tid & 127, pg = tid >> 7;
    float4 pv[8];
    auto prev_load = [&](long long base, int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int sl = k0 + 8 * pg + j;
            pv[j] = (base + pf < phi && sl < a.prev_nsplit) ? *reinterpret_cast<const float4*>(a.prev_slab + ((size_t)sl * pn4 + base + pf) * 4)
                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto prev_sum = [&](float4& s) {
        float4* const L = reinterpret_cast<float4*>(smem);
#pragma unroll
        for (int j = 0; j < 8; ++j) L[(8 * pg + j) * 128 + pf] = pv[j];
        __syncthreads();
        if (pg == 0) {
#pragma unroll 8
            for (int sl = 0; sl < 32; ++sl) { const float4 v = L[sl * 128 + pf]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        }
        __syncthreads();
    };
    if (chained && plo < phi) prev_load(plo, 0);
    if (nk > 0) {
        load_regs(0, ra[0], rb[0]);
        if (nk > 1) load_regs(1, ra[1], rb[1]);
        if (nk > 2) load_regs(2, ra[2], rb[2]);
    }
    if (chained) {
        for (long long base = plo; base < phi; base += 128) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k0 = 0; k0 < a.prev_nsplit; k0 += 32) {
                if (base != plo || k0 != 0) prev_load(base, k0);
                prev_sum(s);
            }
            if (pg == 0 && base + pf < phi) *reinterpret_cast<float4*>(a.prev_out + (base + pf) * 4) = s;
        }
    }
    if (nk > 0) {
        write_lds(0, 0, ra[0], rb[0]);
        __syncthreads();
    }
    for (int kt0 = 0; kt0 < nk; kt0 += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int kt = kt0 + u;
            if (kt < nk) {
                // One scheduling region per k-tile, no branch inside: loads past the chunk end read 0 (out-of-range offset), the
                // split after the last k-tile lands in the stage nobody reads again.  The group barriers hand the scheduler
                // the order of a software pipeline -- first-step fragments and the global loads up front, then every MFMA
                // followed by its share of the NEXT tile's split (VALU) and, in the first half, one second-step fragment read:
                // the split runs in the shadow of the matrix pipe instead of after it (alone 33.8 -> 31.6 us dW1, 37.1 -> 35.2
                // dW2; step -1.1 %; profiles/README.md r04_e).
                load_regs(kt + PF, ra[u % PF], rb[u % PF]);           // slot of tile kt: already in LDS
                compute(u % 2);
                write_lds(kt + 1, (u + 1) % 2, ra[(u + 1) % PF], rb[(u + 1) % PF]);
                constexpr int VG = PRO == PRO_PRELU_NORM ? 10 : 5;    // VALU instructions per MFMA gap (64 / 120 per k-tile and wave)
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VG, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, VG, 0);
                }
                __syncthreads();
            }
        }
    }
    if constexpr (Ar<AR>::F16) h3_unscale<Ar<AR>::W2>(acc, acc2, eg, ex);
    float* __restrict__ S = a.slab + (size_t)sp * a.R * a.Cn;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = r0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
            const int c = c0 + wn * 32 + l31;
            if (r < a.R && c < a.Cn) S[(size_t)r * a.Cn + c] = acc[mt][e];
        }
}

// ---------------------------------------------------------------------------------------------------------
// Forward / input-gradient GEMM on PRE-SPLIT weights ("planes"): the product kernel of the composite stacks.
//
// The weights change once per step, the activations once per launch: ctn_split_b3_batch splits every 1x1 weight once
// into bf16 piece fragments stored in MFMA operand order, and this kernel reads them straight into registers --
//   block (rt, kt, p) = 1 KiB = the A operand of v_mfma_f32_32x32x16_bf16 for rows 32 rt .. + 31, contraction steps
//   16 kt .. + 15, piece p (0 = hi, 1 = lo): lane l holds row 32 rt + l % 32, steps 16 kt + 8 (l / 32) .. + 7 (16 bytes);
//   blocks ordered [rt][kt][p], so the four blocks of one 32-deep k-tile are 4 KiB contiguous (one per-lane address, four
//   immediate offsets) and a wave instruction reads 1 KiB contiguous.
// No LDS traffic, no conversion work and no sharing for the weight operand: the four waves of a workgroup own disjoint
// row ranges (4 x 1 wave grid, each wave WM rows x all TN columns).  Only the activation tile goes through LDS: loaded
// once per workgroup, PReLU+gLN prologue, split into two pieces, stored channel-major, read by every wave with the
// transposing ds_read_b64_tr_b16.  Two LDS stages -> one barrier per k-tile; weight fragments and the next activation
// tile are prefetched one k-tile ahead in registers (two k-tiles ahead measured slower: 35 vs 30 us, the extra register
// sets cost a resident workgroup).  Per wave and k-tile (128 x 64 tile): 4 + 2 global 16-byte loads, ~30 VALU (split of
// 8 values), 4 ds_write_b64, 16 ds_read_b64_tr_b16, 12 MFMAs.
// The MFMA order per accumulator is that of pw_gemm_b3_kernel (lo.hi, hi.lo, hi.hi per 16-deep step): same values.
// ---------------------------------------------------------------------------------------------------------
template <typename TL, int NP>
struct B3P {
    static constexpr int PB = TL::TN + 32;
    static constexpr int STAGE_ELEMS = NP * XK * PB;                           // all pieces of one k-tile
    static constexpr int MAIN_BYTES = 2 * STAGE_ELEMS * 2;                     // two stages
    static constexpr int EPI_BYTES = TL::STAGE_FLOATS * 4;
    static constexpr int SMEM_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
};

__device__ __forceinline__ bf16x8 buf_ld_frag(__amdgpu_buffer_rsrc_t r, int voff, int imm) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff + imm, 0, 0));
}

#ifdef CTN_EXP_B3_TIMELINE       // experiment builds only: per-workgroup phase stamps (benchmarks/gemm_timeline.py)
__device__ unsigned long long ctn_dbg_tl[8192 * 12];
#define CTN_TL_STAMP(i) do { if (tid == 0 && blockIdx.x < 8192) { ctn_dbg_tl[blockIdx.x * 12 + 2 * (i)] = __builtin_amdgcn_s_memtime(); \
                                  ctn_dbg_tl[blockIdx.x * 12 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define CTN_TL_STAMP(i) do { } while (0)
#endif

template <int AR, typename TL, int PRO, int EPI>
__global__ __launch_bounds__(TL::NTH, (TL::MT * TL::NTL <= 2) ? (Ar<AR>::NP == 2 ? 4 : 3) : 2)
void pw_gemm_b3p_kernel(PwArgs a) {
    constexpr int NP = Ar<AR>::NP;
    constexpr int TM = TL::TM, TN = TL::TN, MT = TL::MT, NTL = TL::NTL, WM = TL::WM, NTH = TL::NTH;
    static_assert(TL::WGN == 1, "each wave owns its rows: 4 x 1 wave grid");
    using L = B3P<TL, NP>;
    constexpr int PB = L::PB;
    constexpr int B_L = XK * TN / 4 / NTH;                                     // float4 loads per thread per k-tile
    static_assert(B_L >= 1, "tile too narrow for the workgroup");
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[L::SMEM_BYTES];
    __shared__ double red[NTH / 64];
    __bf16* const Bp = reinterpret_cast<__bf16*>(smem_raw);                    // [stage][piece][XK][PB]

    const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
    CTN_TL_STAMP(0);
#ifdef CTN_EXP_B3_TIMELINE
    if (tid == 0 && blockIdx.x < 8192) {
        ctn_dbg_tl[blockIdx.x * 12 + 8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
        ctn_dbg_tl[blockIdx.x * 12 + 9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
    }
#endif
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int rt = bid % a.tiles_r; bid /= a.tiles_r;
    const int ct = bid % a.tiles_c;
    const int m = bid / a.tiles_c;
    const int r0 = rt * TM, c0 = ct * TN;
    const float* __restrict__ Xm = a.X + (size_t)m * a.Cn * a.Kp;

    float p_mean = 0.f, p_rstd = 1.f, p_alpha = 0.f;
    if constexpr (PRO == PRO_PRELU_NORM) {
        finalize_stats<NTH>(a.pro_part + (size_t)m * a.pro_nparts * 2, a.pro_nparts, (double)a.Cn * (double)a.K, red, p_mean, p_rstd);
        p_alpha = a.pro_alpha[0];
        if (a.pro_ms_out != nullptr && rt == 0 && ct == 0 && tid == 0) {
            a.pro_ms_out[2 * m] = p_mean;
            a.pro_ms_out[2 * m + 1] = p_rstd;
        }
    }
#ifdef CTN_EXP_B3_NK1
    const int nk = 1, Cnp = (a.Cn + XK - 1) / XK * XK;
#else
    const int nk = (a.Cn + XK - 1) / XK, Cnp = nk * XK;
#endif
    const int Rp = (a.R + 31) / 32 * 32;
    // h3: max |W| sits behind the weight's pieces (ctn_split_h3_batch), the activation's scale comes from its tracked maximum
    int ew = 0, ex = 0;
    float sx = 1.f;
    if constexpr (Ar<AR>::F16) {
        ew = h3_exp(__uint_as_float(*reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(a.W) + (size_t)Rp * Cnp * (2 * NP))));
        const float xm = amax_read(a.x_amax + (size_t)m * CTN_AMAX_SLOTS);
        ex = h3_exp(PRO == PRO_PRELU_NORM ? h3_pro_bound(xm, p_alpha, p_mean, p_rstd, a.pro_gbmax) : xm);
        sx = h3_pow2(ex);
    }
    // weight fragments: rows past Rp fall off the end of the planes and read 0 (rows R .. Rp-1 are stored as zeros)
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(a.W, (unsigned)Rp * (unsigned)Cnp * (2u * NP));
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(Xm, (unsigned)a.Cn * (unsigned)a.Kp * 4u);
    __amdgpu_buffer_rsrc_t rsG = rsX, rsBt = rsX;
    if constexpr (PRO == PRO_PRELU_NORM) {
        rsG = make_rsrc(a.pro_gamma, (unsigned)a.Cn * 4u);
        rsBt = make_rsrc(a.pro_beta, (unsigned)a.Cn * 4u);
    }
    int voA[MT], voB[B_L], voP[B_L];
#pragma unroll
    for (int i = 0; i < MT; ++i) voA[i] = ((r0 + wm * WM) / 32 + i) * (Cnp / 16) * (NP * 1024) + lane * 16;      // NP KiB per (rt, 16-deep step)
#pragma unroll
    for (int j = 0; j < B_L; ++j) {
        const int i = tid / (TN / 4) + (4 * NTH / TN) * j;
        voB[j] = (i * a.Kp + c0 + (tid % (TN / 4)) * 4) * 4;
        voP[j] = i * 4;
    }
    const int sB = XK * a.Kp * 4;

    auto load_a = [&](int kt, bf16x8 (&fa)[MT][2][NP]) {         // [row tile][k step][piece]
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
#ifdef CTN_EXP_B3_NOA
                for (int p = 0; p < NP; ++p) { float4 z = make_float4((float)lane, 1.f, 2.f, (float)kt); fa[i][ks][p] = __builtin_bit_cast(bf16x8, z); }
#else
                for (int p = 0; p < NP; ++p) fa[i][ks][p] = buf_ld_frag(rsW, voA[i] + kt * (2 * NP * 1024), (ks * NP + p) * 1024);
#endif
    };
    auto load_b = [&](int kt, float4 (&rb)[B_L], float2 (&rp)[B_L]) {
#pragma unroll
        for (int j = 0; j < B_L; ++j) {
            rb[j] = buf_ld4(rsX, voB[j] + kt * sB, 0);
            if constexpr (PRO == PRO_PRELU_NORM)
                rp[j] = make_float2(buf_ld1(rsG, voP[j] + kt * XK * 4, 0), buf_ld1(rsBt, voP[j] + kt * XK * 4, 0));
            else
                rp[j] = make_float2(0.f, 0.f);
        }
    };
    auto store_b = [&](int stage, const float4 (&rb)[B_L], const float2 (&rp)[B_L]) {
        __bf16* const S = Bp + stage * L::STAGE_ELEMS;
#pragma unroll
        for (int j = 0; j < B_L; ++j) {
            const int i = tid / (TN / 4) + (4 * NTH / TN) * j, k = (tid % (TN / 4)) * 4;
            float4 v = rb[j];
            if constexpr (PRO == PRO_PRELU_NORM) {
                if constexpr (Ar<AR>::F16) v = pro_apply(v, c0 + k, a.K, rp[j].x * sx, rp[j].y * sx, p_alpha, p_mean, p_rstd);
                else v = pro_apply(v, c0 + k, a.K, rp[j].x, rp[j].y, p_alpha, p_mean, p_rstd);
            }
            bf16x4 q[NP];
#ifdef CTN_EXP_B3_NOSPLIT
            q[0] = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
            for (int p = 1; p < NP; ++p) q[p] = q[0];
#else
            split_x4<AR, PRO == PRO_PRELU_NORM>(v, q, sx);
#endif
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(S + (p * XK + i) * PB + k) = q[p];
        }
    };

    f32x16 acc[MT][NTL];
    f32x16 acc2[Ar<AR>::W2 ? MT : 1][Ar<AR>::W2 ? NTL : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; acc2[Ar<AR>::W2 ? i : 0][Ar<AR>::W2 ? j : 0][e] = 0.f; }

    auto compute = [&](int stage, const bf16x8 (&fa)[MT][2][NP]) {
        const __bf16* const S = Bp + stage * L::STAGE_ELEMS;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bfr[NTL][NP];
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
#ifdef CTN_EXP_B3_NOLDSRD
                for (int p = 0; p < NP; ++p) { float4 z = make_float4((float)lane, 1.f, (float)stage, (float)ks); bfr[j][p] = __builtin_bit_cast(bf16x8, z); }
#else
                for (int p = 0; p < NP; ++p) bfr[j][p] = frag_tr(S + (p * XK + ks * 16) * PB + j * 32, PB, lane);
#endif
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
#ifdef CTN_EXP_B3_NOMFMA
#pragma unroll
                    for (int p = 0; p < NP; ++p) {          // keep every fragment live: one VALU op per fragment register instead of the MFMAs
                        const float4 x = __builtin_bit_cast(float4, fa[i][ks][p]), y = __builtin_bit_cast(float4, bfr[j][p]);
                        acc[i][j][p] += x.x * y.x; acc[i][j][p + 4] += x.y * y.y; acc[i][j][p + 8] += x.z * y.z; acc[i][j][p + 12] += x.w * y.w;
                    }
#else
                    mfma_pieces<AR>(acc[i][j], acc2[Ar<AR>::W2 ? i : 0][Ar<AR>::W2 ? j : 0], fa[i][ks], bfr[j]);
#endif
                }
        }
    };

    // k-tile kt: weight fragments in one register set, activation tile in LDS stage kt & 1; while it is multiplied, tile
    // kt + 1 (already in registers) is split into the other stage and the loads of tile kt + 2 are issued.
    float4 rb[B_L];
    float2 rp[B_L];
    // Two accumulator sets (W2) on the 128 x 64 tile: ONE weight-fragment register set, reloaded right behind the MFMAs that read
    // it, keeps the kernel at 113-127 VGPRs = four workgroups per CU (two sets: 138 VGPRs, three workgroups; in-step 762 vs 747
    // utt/s on one box).  The other waves of the SIMD cover the reload.
    if constexpr (Ar<AR>::W2 && MT * NTL <= 2) {
        bf16x8 fa[MT][2][NP];
        load_a(0, fa);
        load_b(0, rb, rp);
        store_b(0, rb, rp);
        if (nk > 1) load_b(1, rb, rp);
        __syncthreads();
        CTN_TL_STAMP(1);
        for (int kt = 0; kt < nk; ++kt) {
            compute(kt & 1, fa);
            if (kt + 1 < nk) { load_a(kt + 1, fa); store_b((kt + 1) & 1, rb, rp); }
            __syncthreads();
            if (kt + 2 < nk) load_b(kt + 2, rb, rp);
        }
    } else {
    bf16x8 fa0[MT][2][NP], fa1[MT][2][NP];
    load_a(0, fa0);
    load_b(0, rb, rp);
    store_b(0, rb, rp);
    if (nk > 1) { load_a(1, fa1); load_b(1, rb, rp); }
    __syncthreads();
    CTN_TL_STAMP(1);
    for (int kt = 0; kt < nk; kt += 2) {
        compute(0, fa0);
        if (kt + 1 < nk) store_b(1, rb, rp);
        __syncthreads();
        if (kt + 2 < nk) { load_a(kt + 2, fa0); load_b(kt + 2, rb, rp); }
        if (kt + 1 < nk) {
            compute(1, fa1);
            if (kt + 2 < nk) store_b(0, rb, rp);
            __syncthreads();
            if (kt + 3 < nk) { load_a(kt + 3, fa1); load_b(kt + 3, rb, rp); }
        }
    }
    }
    CTN_TL_STAMP(2);
#ifdef CTN_EXP_B3_NOEPI
    if (a.K < 0) a.Out[tid] = acc[0][0][0] + acc[MT - 1][NTL - 1][15];      // never taken: keeps the accumulators live
#else
    if constexpr (Ar<AR>::F16) {
#pragma unroll
        for (int i = 0; i < MT; ++i) h3_unscale<Ar<AR>::W2>(acc[i], acc2[Ar<AR>::W2 ? i : 0], ew, ex);
    }
    gemm_epilogue<TL, EPI>(a, acc, reinterpret_cast<float*>(smem_raw), red, m, rt, ct);
#endif
    CTN_TL_STAMP(3);
}

// W fp32 -> fragment-ordered bf16 pieces (layout above).  value(r, k) = W[r * sr + k * sk]: (sr, sk) = (Cn, 1) for a stored
// [R, Cn] matrix, (1, R) for a stored [Cn, R] matrix used transposed.  One thread per (rt, kt, lane); zero fill to Rp x Cnp.
constexpr int SPLIT_MAX = 64;
struct SplitArgs {
    const float* src[SPLIT_MAX];
    __bf16* dst[SPLIT_MAX];
    int R, Cn, sr, sk, nkt;        // nkt = Cnp / 16
};
template <int AR>
__global__ __launch_bounds__(256) void split_b3_kernel(SplitArgs a) {
    constexpr int NP = Ar<AR>::NP;
    const float* __restrict__ W = a.src[blockIdx.z];
    __bf16* __restrict__ D = a.dst[blockIdx.z];
    const int lane = threadIdx.x & 63;
    const int kt = blockIdx.x * 4 + (threadIdx.x >> 6), rt = blockIdx.y;
    if (kt >= a.nkt) return;
    float sc = 1.f;
    if constexpr (Ar<AR>::F16) {        // h3: max |W| was put behind the pieces by absmax_batch_kernel; the GEMMs read it there too
        const size_t pieces = (size_t)gridDim.y * 32 * (size_t)a.nkt * 16 * (2 * NP);
        sc = h3_pow2(h3_exp(__uint_as_float(*reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(D) + pieces))));
    }
    const int r = rt * 32 + (lane & 31), k0 = kt * 16 + (lane >> 5) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (r < a.R && k0 + e < a.Cn) ? W[(size_t)r * a.sr + (size_t)(k0 + e) * a.sk] : 0.f;
    bf16x4 q0[NP], q1[NP];
    split_x4<AR>(make_float4(v[0], v[1], v[2], v[3]), q0, sc);
    split_x4<AR>(make_float4(v[4], v[5], v[6], v[7]), q1, sc);
    __bf16* const blk = D + ((size_t)(rt * a.nkt + kt) * NP) * 512 + lane * 8;     // 512 bf16 = 1 KiB per block
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x8*>(blk + p * 512) = __builtin_shufflevector(q0[p], q1[p], 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- range tracking for the h3 arithmetic -----------------------------------------------------------------------------
// dst[z][0] = bit pattern of max |src[z][0 .. n)| : one workgroup per (small) array -- weights, gamma / beta vectors.
struct AbsmaxArgs {
    const float* src[SPLIT_MAX];
    unsigned* dst[SPLIT_MAX];
    int n;
};
__global__ __launch_bounds__(1024) void absmax_batch_kernel(AbsmaxArgs a) {
    __shared__ unsigned sc[16];
    const float* __restrict__ S = a.src[blockIdx.x];
    unsigned b = 0u;
    const int n4 = a.n & ~3;
    auto upd = [&](const float4& v) {
        const unsigned t0 = __float_as_uint(fabsf(v.x)), t1 = __float_as_uint(fabsf(v.y));
        const unsigned t2 = __float_as_uint(fabsf(v.z)), t3 = __float_as_uint(fabsf(v.w));
        const unsigned u = (t0 > t1 ? t0 : t1), w = (t2 > t3 ? t2 : t3);
        b = b > u ? b : u;
        b = b > w ? b : w;
    };
    int i = threadIdx.x * 4;
    for (; i + 3 * 4096 < n4; i += 4 * 4096) {       // four 16-byte loads in flight per thread
        const float4 v0 = *reinterpret_cast<const float4*>(S + i), v1 = *reinterpret_cast<const float4*>(S + i + 4096);
        const float4 v2 = *reinterpret_cast<const float4*>(S + i + 2 * 4096), v3 = *reinterpret_cast<const float4*>(S + i + 3 * 4096);
        upd(v0); upd(v1); upd(v2); upd(v3);
    }
    for (; i < n4; i += 4096) upd(*reinterpret_cast<const float4*>(S + i));
    for (int j = n4 + threadIdx.x; j < a.n; j += 1024) { const unsigned t = __float_as_uint(fabsf(S[j])); b = b > t ? b : t; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)b, o, 64); b = b > t ? b : t; }
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned r = sc[0];
        for (int w = 1; w < 16; ++w) r = r > sc[w] ? r : sc[w];
        a.dst[blockIdx.x][0] = r;
    }
}
// slots of amax[m] <- max |x[m][0 .. n)| : activations entering a stack (the caller zeroes amax); n % 4 == 0, 16-byte rows
__global__ __launch_bounds__(256) void absmax_rows_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ amax) {
    __shared__ double red[4];
    const int m = blockIdx.y;
    const float* __restrict__ X = x + (size_t)m * n;
    float v = 0.f;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
        const float4 q = *reinterpret_cast<const float4*>(X + i);
        v = fmaxf(fmaxf(v, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
        // (fmaxf drops NaNs; a NaN input still poisons the GEMM result through the operand itself)
    }
    block_amax_atomic<256>(v, red, amax + (size_t)m * CTN_AMAX_SLOTS, blockIdx.x);
}

template <int AR, typename TL>
void launch_b3p_tile(const PwArgs& a, bool pro, bool residual, int stats, bool relu, int gln_bwd, hipStream_t st) {      // gln_bwd: 1 = EPI_GLN_BWD, 2 = EPI_CLN_BWD, 3 = EPI_GLN_BWD2; stats: 1 = EPI_PRELU_STATS, 2 = EPI_CLN_STATS
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * a.M)), block(TL::NTH);
    if (gln_bwd == 3) hipLaunchKernelGGL((pw_gemm_b3p_kernel<AR, TL, PRO_NONE, EPI_GLN_BWD2>), grid, block, 0, st, a);
    else if (gln_bwd == 2) hipLaunchKernelGGL((pw_gemm_b3p_kernel<AR, TL, PRO_NONE, EPI_CLN_BWD>), grid, block, 0, st, a);
    else if (gln_bwd) hipLaunchKernelGGL((pw_gemm_b3p_kernel<AR, TL, PRO_NONE, EPI_GLN_BWD>), grid, block, 0, st, a);
    else if (pro && residual) hipLaunchKernelGGL((pw_gemm_b3p_kernel<AR, TL, PRO_PRELU_NORM, EPI_RESIDUAL>), grid, block, 0, st, a);
    else if (pro) hipLaunchKernelGGL((pw_gemm_b3p_kernel<AR, TL, PRO_PRELU_NORM, EPI_NONE>), grid, block, 0, st, a);
    else if (stats == 2) hipLaunchKernelGGL((pw_gemm_b3p_kernel<AR, TL, PRO_NONE, EPI_CLN_STATS>), grid, block, 0, st, a);
    else if (stats) hipLaunchKernelGGL((pw_gemm_b3p_kernel<AR, TL, PRO_NONE, EPI_PRELU_STATS>), grid, block, 0, st, a);
    else if (residual) hipLaunchKernelGGL((pw_gemm_b3p_kernel<AR, TL, PRO_NONE, EPI_RESIDUAL>), grid, block, 0, st, a);
    else if (relu) hipLaunchKernelGGL((pw_gemm_b3p_kernel<AR, TL, PRO_NONE, EPI_RELU>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pw_gemm_b3p_kernel<AR, TL, PRO_NONE, EPI_NONE>), grid, block, 0, st, a);
}

template <int AR, typename TL>
void launch_b3_tile(const PwArgs& a, int trans_w, bool pro, bool residual, int stats, bool relu, int gln_bwd, hipStream_t st) {
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * a.M)), block(TL::NTH);
    if (gln_bwd == 3) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 1, PRO_NONE, EPI_GLN_BWD2>), grid, block, 0, st, a);
    else if (gln_bwd == 2) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 1, PRO_NONE, EPI_CLN_BWD>), grid, block, 0, st, a);
    else if (gln_bwd) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 1, PRO_NONE, EPI_GLN_BWD>), grid, block, 0, st, a);
    else if (trans_w) {
        if (pro && residual) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 1, PRO_PRELU_NORM, EPI_RESIDUAL>), grid, block, 0, st, a);
        else if (pro) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 1, PRO_PRELU_NORM, EPI_NONE>), grid, block, 0, st, a);
        else if (stats == 2) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 1, PRO_NONE, EPI_CLN_STATS>), grid, block, 0, st, a);
        else if (stats) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 1, PRO_NONE, EPI_PRELU_STATS>), grid, block, 0, st, a);
        else if (residual) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 1, PRO_NONE, EPI_RESIDUAL>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 1, PRO_NONE, EPI_NONE>), grid, block, 0, st, a);
    } else if (pro) {
        if (residual) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 0, PRO_PRELU_NORM, EPI_RESIDUAL>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 0, PRO_PRELU_NORM, EPI_NONE>), grid, block, 0, st, a);
    } else if (stats == 2) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 0, PRO_NONE, EPI_CLN_STATS>), grid, block, 0, st, a);
    else if (stats) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 0, PRO_NONE, EPI_PRELU_STATS>), grid, block, 0, st, a);
    else if (residual) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 0, PRO_NONE, EPI_RESIDUAL>), grid, block, 0, st, a);
    else if (relu) hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 0, PRO_NONE, EPI_RELU>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pw_gemm_b3_kernel<AR, TL, 0, PRO_NONE, EPI_NONE>), grid, block, 0, st, a);
}

}  // namespace

// ---- host side, used by the entry points of ctn_gemm.hip (ar = arithmetic id of the kernels: 3 = b6, 4 = h3) ----------------------
static int g_ctn_b3_tile = 1;               // 0: 128x128, 1: 128x64, 2: 256x64, 3: 256x64 on 8 waves   (ctn_tune("b3_tile", id)); in-step (h3) . / 10.70 / 11.19 / 10.85 ms
static int g_ctn_b3_wgrad_blocks = 256;     // target workgroups per weight-gradient launch   (ctn_tune("b3_wgrad_blocks", n))

static int g_ctn_b3_tile_k3 = 3;            // tile of the K3 form (operand prologue + residual on pre-split weights, <= 256 rows): 256x64 =
                                            // one workgroup per column tile, so the PReLU+gLN prologue and the split of the activation tile run
                                            // once instead of once per 128-row tile (b6, alone: 45 vs 54 us); id 3 = the same tile on 8 waves of
                                            // 32 rows (two accumulator sets fit 128 VGPRs: 4 waves per SIMD instead of 2; h3 in-step 10.85 vs
                                            // 10.91 ms); ctn_tune("b3_tile_k3", id)
static void ctn_b3_tile_dims(int* tm, int* tn) {    // the tile of every form that writes statistics partials (their count is part of the ABI)
    static const int d[4][2] = {{128, 128}, {128, 64}, {256, 64}, {256, 64}};
    *tm = d[g_ctn_b3_tile][0];
    *tn = d[g_ctn_b3_tile][1];
}
// tile id of one launch: forms without a statistics epilogue may pick their own
static int ctn_b3_pick_tile(const PwArgs& a, int trans_w, bool pro, bool residual, int stats, int gln_bwd) {
    if (trans_w == 2 && residual && !stats && !gln_bwd && a.R <= 256 && a.R > 128) return g_ctn_b3_tile_k3;      // K3 and B5
    return g_ctn_b3_tile;
}

template <int AR>
static void ctn_b3_launch_fwd_np(int tile, PwArgs& a, int trans_w, bool pro, bool residual, int stats, bool relu, int gln_bwd, hipStream_t st) {
    if (trans_w == 2) {             // a.W = fragment-ordered pieces (ctn_split_b3_batch)
        switch (tile) {
            case 1: launch_b3p_tile<AR, Tile<128, 64, 4, 1>>(a, pro, residual, stats, relu, gln_bwd, st); break;
            case 2: launch_b3p_tile<AR, Tile<256, 64, 4, 1>>(a, pro, residual, stats, relu, gln_bwd, st); break;
            case 3: launch_b3p_tile<AR, Tile<256, 64, 8, 1>>(a, pro, residual, stats, relu, gln_bwd, st); break;      // 8 waves of 32 rows
            default: launch_b3p_tile<AR, Tile<128, 128, 4, 1>>(a, pro, residual, stats, relu, gln_bwd, st); break;
        }
        return;
    }
    if constexpr (AR < 4) {         // fp32 weights split on the fly: the bf16 arithmetics
        switch (tile) {
            case 1: launch_b3_tile<AR, T128x64>(a, trans_w, pro, residual, stats, relu, gln_bwd, st); break;
            case 2: case 3: launch_b3_tile<AR, Tile<256, 64, 2, 2>>(a, trans_w, pro, residual, stats, relu, gln_bwd, st); break;
            default: launch_b3_tile<AR, T128x128>(a, trans_w, pro, residual, stats, relu, gln_bwd, st); break;
        }
    }
}

static void ctn_b3_launch_fwd(int ar, PwArgs& a, int trans_w, bool pro, bool residual, int stats, bool relu, int gln_bwd, hipStream_t st) {
    static const int d[4][2] = {{128, 128}, {128, 64}, {256, 64}, {256, 64}};
    const int tile = ctn_b3_pick_tile(a, trans_w, pro, residual, stats, gln_bwd);
    a.tiles_r = ctn_cdiv(a.R, d[tile][0]);
    a.tiles_c = ctn_cdiv(a.Kp, d[tile][1]);
    if (ar == 4) ctn_b3_launch_fwd_np<H3AR>(tile, a, 2, pro, residual, stats, relu, gln_bwd, st);       // h3 (pre-split weights only)
    else ctn_b3_launch_fwd_np<3>(tile, a, trans_w, pro, residual, stats, relu, gln_bwd, st);
}

// bytes of one pre-split weight operand; h3 appends 16 bytes: the bit pattern of max |W| (the scale is derived from it)
static size_t ctn_b3_pieces_bytes(int ar, int R, int Cn) {
    return (size_t)((R + 31) / 32 * 32) * (size_t)((Cn + XK - 1) / XK * XK) * 2 * (size_t)(ar == 4 ? 2 : ar);
}
static size_t ctn_b3_planes_bytes(int ar, int R, int Cn) { return ctn_b3_pieces_bytes(ar, R, Cn) + (ar == 4 ? 16 : 0); }

static void ctn_b3_launch_absmax(const void* const* src, void* const* dst, size_t dst_offset, int n, int len, hipStream_t st) {
    for (int o = 0; o < n; o += SPLIT_MAX) {
        AbsmaxArgs aa{};
        const int cnt = n - o < SPLIT_MAX ? n - o : SPLIT_MAX;
        for (int i = 0; i < cnt; ++i) {
            aa.src[i] = (const float*)src[o + i];
            aa.dst[i] = (unsigned*)((char*)dst[o + i] + dst_offset);
        }
        aa.n = len;
        hipLaunchKernelGGL(absmax_batch_kernel, dim3(cnt), dim3(1024), 0, st, aa);
    }
}

static void ctn_b3_launch_split(int ar, const void* const* src, void* const* dst, int n, int R, int Cn, int k_major, hipStream_t st) {
    if (ar == 4) ctn_b3_launch_absmax(src, dst, ctn_b3_pieces_bytes(4, R, Cn), n, R * Cn, st);
    for (int o = 0; o < n; o += SPLIT_MAX) {
        SplitArgs sa{};
        const int cnt = n - o < SPLIT_MAX ? n - o : SPLIT_MAX;
        for (int i = 0; i < cnt; ++i) {
            sa.src[i] = (const float*)src[o + i];
            sa.dst[i] = (__bf16*)dst[o + i];
        }
        sa.R = R; sa.Cn = Cn;
        sa.sr = k_major ? 1 : Cn; sa.sk = k_major ? R : 1;
        sa.nkt = (Cn + XK - 1) / XK * 2;
        const dim3 grid(ctn_cdiv(sa.nkt, 4), (R + 31) / 32, cnt);
        if (ar == 4) hipLaunchKernelGGL(split_b3_kernel<H3AR>, grid, dim3(256), 0, st, sa);
        else hipLaunchKernelGGL(split_b3_kernel<3>, grid, dim3(256), 0, st, sa);
    }
}

static void ctn_b3_wgrad_plan(int M, int R, int Cn, int Kp, int* chunk, int* chunks_per_m) {
    const int tiles = ctn_cdiv(R, BM) * ctn_cdiv(Cn, BN);
    int cpm = ctn_cdiv(g_ctn_b3_wgrad_blocks, tiles * M);
    const int max_cpm = ctn_cdiv(Kp, 256);         // keep >= 256 frames of contraction per slab
    if (cpm > max_cpm) cpm = max_cpm;
    if (cpm < 1) cpm = 1;
    const int c = ctn_cdiv(ctn_cdiv(Kp, cpm), XK) * XK;
    *chunk = c;
    *chunks_per_m = ctn_cdiv(Kp, c);
}

template <int AR>
static void ctn_b3_launch_wgrad_np(const WgArgs& a, bool pro, dim3 grid, hipStream_t st) {
    const dim3 block(WNT);
    if (pro) hipLaunchKernelGGL((pw_wgrad_b3_kernel<AR, PRO_PRELU_NORM>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pw_wgrad_b3_kernel<AR, PRO_NONE>), grid, block, 0, st, a);
}

// a.chunk / a.chunks_per_m / a.slab already set by the caller from ctn_b3_wgrad_plan; returns the number of slabs
static int ctn_b3_launch_wgrad(int ar, WgArgs& a, bool pro, hipStream_t st) {
    a.tiles_r = ctn_cdiv(a.R, BM);
    a.tiles_c = ctn_cdiv(a.Cn, BN);
    const int nsplit = a.M * a.chunks_per_m;
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * nsplit));
    if (ar == 4) ctn_b3_launch_wgrad_np<H3AR>(a, pro, grid, st);
    else ctn_b3_launch_wgrad_np<3>(a, pro, grid, st);
    return nsplit;
}
