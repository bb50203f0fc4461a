// Shared device/host helpers for the Conv-TasNet gfx950 kernels.
// gfx950 only: wave = 64 lanes, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CTN_EPS 1e-8f   // EPS of src/conv_tasnet.py:10 and src/pit_criterion.py:9

// ---- status / error reporting across the C ABI (never throws) -------------
enum {
    CTN_OK = 0,
    CTN_ERR_ARG = -1,      // bad shape / null pointer / unsupported size
    CTN_ERR_LAUNCH = -2,   // hipGetLastError() after a launch
    CTN_ERR_WORKSPACE = -3 // caller-provided workspace too small
};

void ctn_set_error(const char* fmt, ...);

#define CTN_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) {                                          \
            ctn_set_error(__VA_ARGS__);                         \
            return CTN_ERR_ARG;                                 \
        }                                                       \
    } while (0)

#define CTN_CHECK_LAUNCH(name)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            ctn_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return CTN_ERR_LAUNCH;                                          \
        }                                                                   \
    } while (0)

static inline int ctn_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long ctn_cdivll(long long a, long long b) { return (a + b - 1) / b; }

// ---- chained weight gradients (library-internal: the composite stacks of ctn_block.hip call these, ctn_gemm.hip defines them) ----
// A chained launch leaves its split-K slabs unsummed and records them in *chain; the next chained launch ON THE SAME STREAM sums
// them inside its own kernel (same addition order as slab_reduce_kernel: bitwise the un-chained result) and records its own.
// Every launch of a chain needs a slab buffer different from the pending one (alternate between two);
// ctn_wgrad_chain_flush() sums what is still pending with a slab_reduce launch.  chain == nullptr: the plain entry point.
struct CtnWgradChain { const float* slab = nullptr; float* out = nullptr; long long n = 0; int nsplit = 0; };
int ctn_pw_wgrad_chained(const float* dOut, const float* X, float* dW, int M, int R, int Cn, int K, int Kp,
                         const float* pro_gamma, const float* pro_beta, const float* pro_alpha, const float* pro_ms,
                         void* workspace, size_t workspace_bytes, void* stream, CtnWgradChain* chain);
int ctn_pw_wgrad_h3_chained(const float* dOut, const float* X, float* dW, int M, int R, int Cn, int K, int Kp,
                            const float* pro_gamma, const float* pro_beta, const float* pro_alpha, const float* pro_ms,
                            const unsigned* g_amax, const unsigned* x_amax, const float* pro_gbmax,
                            void* workspace, size_t workspace_bytes, void* stream, CtnWgradChain* chain);
int ctn_wgrad_chain_flush(CtnWgradChain* chain, void* stream);

#ifdef __HIPCC__
// ---- wave64 / block reductions ---------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a whole block of NT threads (NT multiple of 64, <= 1024).
// Result valid in every thread.  `scratch` holds >= NT/64 elements of T.
// The order of additions is fixed -> bitwise reproducible.
template <typename T, int NT>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
    constexpr int NW = NT / 64;
    v = wave_sum(v);
    __syncthreads();  // scratch may still be read by a previous call
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = scratch[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r += scratch[w];
    return r;
}

// Range tracking for the h3 GEMM arithmetic.  The maximum of |x| over one utterance's tensor lives in CTN_AMAX_SLOTS unsigned
// words (bit patterns of non-negative floats: unsigned order = float order); a producer workgroup merges its block maximum into
// ONE of them with an atomic max (slot = its index within the utterance mod 64 -- 128..1024 atomics per launch on ONE address
// cost the producers 5-9 us, profiles/README.md), a consumer takes the maximum over the 64.  max is exact and order-free, so
// results stay bitwise reproducible.  `scratch` holds >= NT/64 doubles.
#ifndef CTN_AMAX_SLOTS
#define CTN_AMAX_SLOTS 64
#endif
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float amax_read(const unsigned* __restrict__ slots) {     // every lane returns the maximum
    unsigned b = slots[threadIdx.x & 63];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)b, o, 64); b = b > t ? b : t; }
    return __uint_as_float(b);
}
template <int NT>
__device__ __forceinline__ void block_amax_atomic(float v, double* scratch, unsigned* slots, int slot) {
    constexpr int NW = NT / 64;
    unsigned b = __float_as_uint(fabsf(v));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)b, o, 64); b = b > t ? b : t; }
    unsigned* const sc = reinterpret_cast<unsigned*>(scratch);
    __syncthreads();  // scratch may still be read by a previous reduction
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned r = sc[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) r = r > sc[w] ? r : sc[w];
        if (r != 0u) atomicMax(slots + (slot & (CTN_AMAX_SLOTS - 1)), r);
    }
}

__device__ __forceinline__ float prelu_f(float v, float a) { return v >= 0.f ? v : a * v; }

// Workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an L2).  Give each XCD a contiguous range of LOGICAL ids, so
// that neighbours in the logical order -- the row tiles of a GEMM that re-read the same activation columns, the two 64-byte halves
// of a 128-byte line -- are served by one private L2 instead of the fabric.  Bijective for any grid size; affects speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// Finalise (mean, rstd) of one utterance from [nparts][2] double partial (sum, sumsq).
// Every thread of the block returns the same values.  count = Ch*K valid elements.
template <int NT>
__device__ __forceinline__ void finalize_stats(const double* __restrict__ part, int nparts, double count,
                                               double* scratch, float& mean, float& rstd) {
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nparts; i += NT) {
        s += part[2 * i];
        q += part[2 * i + 1];
    }
    s = block_sum<double, NT>(s, scratch);
    q = block_sum<double, NT>(q, scratch);
    const double mu = s / count;
    double var = q / count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + (double)CTN_EPS));
}
#endif  // __HIPCC__
