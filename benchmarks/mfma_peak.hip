// Bare fp32-MFMA issue-rate probe: what does v_mfma_f32_32x32x2_f32 sustain on THIS box, on random vs zero operands,
// at 1, 2 and 4 waves per SIMD?  (MI355X_MICROARCH.md, DVFS give-back: MFMA-dense loops hold 1.5-1.7 GHz.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, float* __restrict__ out, int iters,
                                             unsigned long long* clk) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-9f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void probe16(const float* __restrict__ in, float* __restrict__ out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        a += 1e-9f;
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 4; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// bf16 32x32x16: NACC independent accumulators, each receiving CHAIN back-to-back dependent MFMAs per iteration
template <int NACC, int CHAIN>
__global__ __launch_bounds__(256) void probe_bf16(const float* __restrict__ in, float* __restrict__ out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[(threadIdx.x + e) & 511]; b[e] = (__bf16)in[(threadIdx.x + 17 * e) & 511]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, 256 * 2048 * 4 * 4); hipMalloc(&clk, 16);
    float h[512];
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 512; ++i) h[i] = mode ? 0.f : (float)rand() / RAND_MAX - 0.5f;
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        for (int bpc = 1; bpc <= 4; bpc *= 2) {
            const int blocks = 256 * bpc, iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
            double flop = (double)blocks * 4 /*waves*/ * iters * 4 /*acc*/ * 4096.0;
            printf("%s operands, %d wave(s)/SIMD: %.1f TFLOP/s, %.2f ms, in-kernel clock %.2f GHz\n", mode ? "zero  " : "random",
                   bpc, flop / ms / 1e9, ms, (double)hc[0] / (double)hc[1] * 0.1);
        }
    }
    // accumulator-chain experiment (random operands): one dependent chain vs several independent ones
    for (int i = 0; i < 512; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        const int blocks = 256 * bpc, iters = 20000;
        float m1 = time_ms([&] { hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters * 4, clk); });
        float m2 = time_ms([&] { hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, in, out, iters * 2, clk); });
        float m4 = time_ms([&] { hipLaunchKernelGGL(probe16<4>, dim3(blocks), dim3(256), 0, 0, in, out, iters * 2); });
        float m8 = time_ms([&] { hipLaunchKernelGGL(probe16<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters * 8); });
        const double f32 = (double)blocks * 4 * iters * 4 * 4096.0;          // same FLOPs in every variant
        printf("%d wave(s)/SIMD: 32x32x2 1 chain %.1f TF | 2 chains %.1f TF | 16x16x4 4 chains %.1f TF | 16x16x4 1 chain %.1f TF\n", bpc,
               f32 / m1 / 1e9, f32 / m2 / 1e9, f32 / m4 / 1e9, f32 / m8 / 1e9);
    }
    // bf16 32x32x16: dependent-chain experiment (random operands).  FLOPs per MFMA = 2*32*32*16 = 32768
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        const int blocks = 256 * bpc, iters = 20000;
        const double fl = (double)blocks * 4 * iters * 6 * 32768.0;     // 6 MFMAs per iteration in every variant
        float t1 = time_ms([&] { hipLaunchKernelGGL((probe_bf16<1, 6>), dim3(blocks), dim3(256), 0, 0, in, out, iters); });
        float t2 = time_ms([&] { hipLaunchKernelGGL((probe_bf16<2, 3>), dim3(blocks), dim3(256), 0, 0, in, out, iters); });
        float t6 = time_ms([&] { hipLaunchKernelGGL((probe_bf16<6, 1>), dim3(blocks), dim3(256), 0, 0, in, out, iters); });
        printf("bf16 32x32x16, %d wave(s)/SIMD: 1 acc x 6-chain %.0f TF | 2 acc x 3-chain %.0f TF | 6 independent acc %.0f TF\n", bpc,
               fl / t1 / 1e9, fl / t2 / 1e9, fl / t6 / 1e9);
    }
    return 0;
}
