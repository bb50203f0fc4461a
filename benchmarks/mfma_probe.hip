// fp32-MFMA cost model probe for gfx950 (MI355X).  One binary, four questions:
//   P1  peak: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 issue rate at 1..8 waves per SIMD and 1..4 independent
//       accumulators per wave, random operands, NO other instruction in the loop body (x16 unrolled);
//       in-kernel clock from s_memtime / s_memrealtime.  (MI355X_MICROARCH.md quotes 155 TF at one wave per SIMD.)
//   P2  one wave: k VALU fillers (v_fma_f32) issued between consecutive fp32 MFMAs -- how many hide?
//   P3  two waves per SIMD, roles split by wave id (>= 4): waves 0-3 run the MFMA loop, waves 4-7 one of
//       {nothing, v_fma_f32, v_add_f64, ds_read_b32, ds_write_b32, ds_write_b128, global_load_dwordx4}.
//       Times: MFMA alone, partner alone, both.  both ~ max => the pipes overlap; both ~ sum => they serialise.
//   P4  same as P3 with the bf16 32x32x16 MFMA (control: the matrix pipe the guide's numbers come from).
// Build: hipcc --offload-arch=gfx950 -O3 -o benchmarks/mfma_probe.bin benchmarks/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Clk { unsigned long long cyc, real; };

// ---------------------------------------------------------------- P1
template <int NACC>
__global__ __launch_bounds__(256) void peak32(const float* __restrict__ in, float* __restrict__ out, int iters, Clk* clk) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { clk->cyc = t1 - t0; clk->real = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void peak16(const float* __restrict__ in, float* __restrict__ out, int iters, Clk* clk) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    const float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc[i][e];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { clk->cyc = t1 - t0; clk->real = r1 - r0; }
}

// ---------------------------------------------------------------- P2: k fillers between MFMAs, one wave per SIMD
template <int FILL, int NACC>
__global__ __launch_bounds__(256) void fill32(const float* __restrict__ in, float* __restrict__ out, int iters, Clk* clk) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = in[(threadIdx.x + j) & 511];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < FILL; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[j % 8]) : "v"(a), "v"(b));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { clk->cyc = t1 - t0; clk->real = r1 - r0; }
}

// ---------------------------------------------------------------- P3 / P4: role split inside a 512-thread workgroup
enum { PART_NONE = 0, PART_FMA32, PART_ADD64, PART_DSREAD, PART_DSWRITE32, PART_DSWRITE128, PART_GLOAD, PART_MFMA };

template <int PARTNER, bool RUN_MFMA, bool BF16>
__global__ __launch_bounds__(512) void coexec(const float* __restrict__ in, float* __restrict__ out, const float4* __restrict__ big,
                                              int iters, int piters, Clk* clk) {
    __shared__ float lds[8192];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = in[i & 511];
    __syncthreads();
    float s = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (wave < 4) {
        if (RUN_MFMA) {
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            if (BF16) {
                bf16x8 a, b;
#pragma unroll
                for (int e = 0; e < 8; ++e) { a[e] = (__bf16)in[(threadIdx.x + e) & 511]; b[e] = (__bf16)in[(threadIdx.x + 17 * e) & 511]; }
                for (int it = 0; it < iters; ++it)
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
            } else {
                const float a = in[threadIdx.x & 255], b = in[(threadIdx.x & 255) + 256];
                for (int it = 0; it < iters; ++it)
#pragma unroll
                    for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][e];
        }
    } else {
        const float a = in[threadIdx.x & 255], b = in[(threadIdx.x & 255) + 256];
        if (PARTNER == PART_FMA32) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = in[(threadIdx.x + j) & 511];
            for (int it = 0; it < piters; ++it)
#pragma unroll
                for (int u = 0; u < 32; ++u) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[u & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < 8; ++j) s += f[j];
        } else if (PARTNER == PART_ADD64) {
            double f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (double)in[(threadIdx.x + j) & 511];
            const double da = (double)a;
            for (int it = 0; it < piters; ++it)
#pragma unroll
                for (int u = 0; u < 32; ++u) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[u & 7]) : "v"(da));
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)f[j];
        } else if (PARTNER == PART_DSREAD) {
            const float* p = lds + (threadIdx.x & 63);
            float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int it = 0; it < piters; ++it) {
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    float v;
                    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)p), "n"(u * 256));
                    asm volatile("s_waitcnt lgkmcnt(8)");
                    f[u & 7] = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int j = 0; j < 8; ++j) s += f[j];
        } else if (PARTNER == PART_DSWRITE32) {
            float* p = lds + (threadIdx.x & 63) + 64 * (wave - 4);
            for (int it = 0; it < piters; ++it)
#pragma unroll
                for (int u = 0; u < 32; ++u) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"((unsigned)(size_t)p), "v"(a), "n"(u * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (PARTNER == PART_DSWRITE128) {
            float* p = lds + 4 * (threadIdx.x & 63) + 256 * (wave - 4);
            f32x4 v = {a, b, a, b};
            for (int it = 0; it < piters; ++it)
#pragma unroll
                for (int u = 0; u < 7; ++u) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"((unsigned)(size_t)p), "v"(v), "n"(u * 4096));
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (PARTNER == PART_GLOAD) {
            // every wave streams its own 1 MiB window of a 64 MiB buffer (L2 / MALL resident after the first pass)
            const float4* p = big + ((size_t)(blockIdx.x * 4 + (wave - 4)) & 63) * 65536 + (threadIdx.x & 63);
            float f = 0.f;
            for (int it = 0; it < piters; ++it) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[((it * 8 + u) & 1023) * 64];
#pragma unroll
                for (int u = 0; u < 8; ++u) f += v[u].x;
            }
            s += f;
        } else if (PARTNER == PART_MFMA) {
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            for (int it = 0; it < piters; ++it)
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][e];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { clk->cyc = t1 - t0; clk->real = r1 - r0; }
}

// ---------------------------------------------------------------- host
static hipEvent_t g_e0, g_e1;
template <typename F>
static float time_ms(F f, int reps = 3) {
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(g_e0));
        f();
        CK(hipEventRecord(g_e1));
        CK(hipEventSynchronize(g_e1));
        float ms;
        CK(hipEventElapsedTime(&ms, g_e0, g_e1));
        if (ms < best) best = ms;
    }
    return best;
}

static float *d_in, *d_out;
static float4* d_big;
static Clk* d_clk;
static double clock_ghz() {
    Clk h;
    CK(hipMemcpy(&h, d_clk, sizeof(h), hipMemcpyDeviceToHost));
    return h.real ? (double)h.cyc / (double)h.real * 0.1 : 0.0;
}

template <int PARTNER, bool BF16>
static void run_coexec(const char* name, int iters, int piters) {
    const dim3 grid(256), block(512);
    const float tm = time_ms([&] { hipLaunchKernelGGL((coexec<PART_NONE, true, BF16>), grid, block, 0, 0, d_in, d_out, d_big, iters, piters, d_clk); });
    const double c0 = clock_ghz();
    const float tp = time_ms([&] { hipLaunchKernelGGL((coexec<PARTNER, false, BF16>), grid, block, 0, 0, d_in, d_out, d_big, iters, piters, d_clk); });
    const float tb = time_ms([&] { hipLaunchKernelGGL((coexec<PARTNER, true, BF16>), grid, block, 0, 0, d_in, d_out, d_big, iters, piters, d_clk); });
    const double c2 = clock_ghz();
    printf("  %-5s MFMA + %-14s: mfma alone %7.3f ms (%.2f GHz) | partner alone %7.3f ms | both %7.3f ms (%.2f GHz) | both/max %.2f  both/sum %.2f\n",
           BF16 ? "bf16" : "fp32", name, tm, c0, tp, tb, c2, tb / (tm > tp ? tm : tp), tb / (tm + tp));
}

int main(int argc, char** argv) {
    const int which = argc > 1 ? atoi(argv[1]) : 0;   // 0 = all
    CK(hipMalloc(&d_in, 512 * 4));
    CK(hipMalloc(&d_out, (size_t)256 * 8 * 512 * 4));
    CK(hipMalloc(&d_big, (size_t)64 << 20));
    CK(hipMemset(d_big, 0, (size_t)64 << 20));
    CK(hipMalloc(&d_clk, sizeof(Clk)));
    CK(hipEventCreate(&g_e0));
    CK(hipEventCreate(&g_e1));
    float h[512];
    srand(1);
    for (int i = 0; i < 512; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice));

    if (which == 0 || which == 1) {
        printf("P1 fp32 MFMA peak (random operands, no other instruction in the loop); spec 157.3 TF at 2.4 GHz = 64 FLOP/clk/SIMD\n");
        // warm the clocks: ~1 s of back-to-back launches
        for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(peak32<4>, dim3(256 * 2), dim3(256), 0, 0, d_in, d_out, 2000, d_clk);
        CK(hipDeviceSynchronize());
        for (int bpc = 1; bpc <= 8; bpc *= 2) {
            const int blocks = 256 * bpc, iters = 16000 / bpc;
            const double flop = (double)blocks * 4 * iters * 16 * 4096.0;
            float t1 = time_ms([&] { hipLaunchKernelGGL(peak32<1>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, d_clk); });
            double c1 = clock_ghz();
            float t2 = time_ms([&] { hipLaunchKernelGGL(peak32<2>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, d_clk); });
            double c2 = clock_ghz();
            float t4 = time_ms([&] { hipLaunchKernelGGL(peak32<4>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, d_clk); });
            double c4 = clock_ghz();
            printf("  32x32x2  %d wave(s)/SIMD: 1 acc %6.1f TF (%.2f GHz, %.1f cyc/MFMA) | 2 acc %6.1f TF (%.2f GHz, %.1f) | 4 acc %6.1f TF (%.2f GHz, %.1f)\n", bpc,
                   flop / t1 / 1e9, c1, t1 * 1e-3 * c1 * 1e9 / ((double)iters * 16 * bpc), flop / t2 / 1e9, c2,
                   t2 * 1e-3 * c2 * 1e9 / ((double)iters * 16 * bpc), flop / t4 / 1e9, c4, t4 * 1e-3 * c4 * 1e9 / ((double)iters * 16 * bpc));
        }
        for (int bpc = 1; bpc <= 8; bpc *= 2) {
            const int blocks = 256 * bpc, iters = 16000 / bpc;
            const double flop = (double)blocks * 4 * iters * 32 * 2048.0;
            float t1 = time_ms([&] { hipLaunchKernelGGL(peak16<1>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, d_clk); });
            double c1 = clock_ghz();
            float t4 = time_ms([&] { hipLaunchKernelGGL(peak16<4>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, d_clk); });
            double c4 = clock_ghz();
            float t8 = time_ms([&] { hipLaunchKernelGGL(peak16<8>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, d_clk); });
            double c8 = clock_ghz();
            printf("  16x16x4  %d wave(s)/SIMD: 1 acc %6.1f TF (%.2f GHz) | 4 acc %6.1f TF (%.2f GHz) | 8 acc %6.1f TF (%.2f GHz)\n", bpc,
                   flop / t1 / 1e9, c1, flop / t4 / 1e9, c4, flop / t8 / 1e9, c8);
        }
        fflush(stdout);
    }
    if (which == 0 || which == 2) {
        printf("P2 one wave per SIMD, k x v_fma_f32 between consecutive fp32 32x32x2 MFMAs (cycles per MFMA slot; 64 = free)\n");
        const int blocks = 256, iters = 20000;
#define FILLRUN(K, NACC)                                                                                                          \
    {                                                                                                                             \
        float t = time_ms([&] { hipLaunchKernelGGL((fill32<K, NACC>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, d_clk); }); \
        double c = clock_ghz();                                                                                                   \
        printf("  k=%2d acc=%d: %7.3f ms  %.2f GHz  %6.1f cyc/MFMA\n", K, NACC, t, c, t * 1e-3 * c * 1e9 / ((double)iters * 8));  \
    }
        FILLRUN(0, 4) FILLRUN(2, 4) FILLRUN(4, 4) FILLRUN(8, 4) FILLRUN(12, 4) FILLRUN(16, 4) FILLRUN(24, 4)
        FILLRUN(0, 1) FILLRUN(4, 1) FILLRUN(8, 1) FILLRUN(16, 1)
        fflush(stdout);
    }
    if (which == 0 || which == 3) {
        printf("P3 two waves per SIMD (512-thread WG): waves 0-3 fp32 MFMA (4 acc), waves 4-7 the partner stream\n");
        const int it = 4000;                       // 4000*16 MFMAs * 64 cyc = 4.1 M cycles
        run_coexec<PART_FMA32, false>("v_fma_f32", it, it * 16 * 64 / 4 / 32);            // 4 cyc each -> same span
        run_coexec<PART_FMA32, false>("v_fma_f32 x1/4", it, it * 16 * 64 / 4 / 32 / 4);
        run_coexec<PART_ADD64, false>("v_add_f64", it, it * 16 * 64 / 8 / 32);
        run_coexec<PART_DSREAD, false>("ds_read_b32", it, it * 16 * 64 / 8 / 32);
        run_coexec<PART_DSWRITE32, false>("ds_write_b32", it, it * 16 * 64 / 8 / 32);
        run_coexec<PART_DSWRITE128, false>("ds_write_b128", it, it * 16 * 64 / 16 / 7);
        run_coexec<PART_GLOAD, false>("global_load x4", it, it * 16 * 64 / 64 / 8);
        run_coexec<PART_MFMA, false>("fp32 MFMA", it, it);
        fflush(stdout);
    }
    if (which == 5) {
        // power mode (benchmarks/power_lab.sh mfma): each configuration runs back to back for ~3 s while the shell samples
        // rocm-smi; prints "case <name> <t0> <t1> <launches> <us per launch>" with wall-clock timestamps
        auto now = [] { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; };
        auto run = [&](const char* name, auto launch, double flop_per_launch) {
            launch(); CK(hipDeviceSynchronize());
            const double t0 = now();
            long n = 0;
            while (now() - t0 < 3.0) { for (int i = 0; i < 20; ++i) launch(); CK(hipDeviceSynchronize()); n += 20; }
            const double t1 = now();
            printf("case %s %.3f %.3f %ld %.2f  # %.1f TFLOP/s\n", name, t0, t1, n, (t1 - t0) / n * 1e6, flop_per_launch * n / (t1 - t0) / 1e12);
            fflush(stdout);
            struct timespec sl = {0, 400000000}; nanosleep(&sl, nullptr);
        };
        const int it = 4000;
        const double f32 = 256.0 * 4 * 4 * it * 16 * 4096.0;        // 4 blocks per CU
        run("mfma32x32x2_4w_regs", [&] { hipLaunchKernelGGL(peak32<4>, dim3(1024), dim3(256), 0, 0, d_in, d_out, it, d_clk); }, f32);
        run("mfma32x32x2_1w_regs", [&] { hipLaunchKernelGGL(peak32<4>, dim3(256), dim3(256), 0, 0, d_in, d_out, it * 4, d_clk); }, f32);
        run("mfma16x16x4_4w_regs", [&] { hipLaunchKernelGGL(peak16<4>, dim3(1024), dim3(256), 0, 0, d_in, d_out, it, d_clk); }, 256.0 * 4 * 4 * it * 32 * 2048.0);
        const double fco = 256.0 * 4 * it * 16 * 4096.0;
        run("mfma+ds_read_b32", [&] { hipLaunchKernelGGL((coexec<PART_DSREAD, true, false>), dim3(256), dim3(512), 0, 0, d_in, d_out, d_big, it, it * 16 * 64 / 8 / 32, d_clk); }, fco);
        run("mfma+ds_write_b128", [&] { hipLaunchKernelGGL((coexec<PART_DSWRITE128, true, false>), dim3(256), dim3(512), 0, 0, d_in, d_out, d_big, it, it * 16 * 64 / 16 / 7, d_clk); }, fco);
        run("mfma+global_load", [&] { hipLaunchKernelGGL((coexec<PART_GLOAD, true, false>), dim3(256), dim3(512), 0, 0, d_in, d_out, d_big, it, it * 16 * 64 / 64 / 8, d_clk); }, fco);
        run("mfma_alone_2w_wg", [&] { hipLaunchKernelGGL((coexec<PART_NONE, true, false>), dim3(256), dim3(512), 0, 0, d_in, d_out, d_big, it, 0, d_clk); }, fco);
        run("ds_read_b32_alone", [&] { hipLaunchKernelGGL((coexec<PART_DSREAD, false, false>), dim3(256), dim3(512), 0, 0, d_in, d_out, d_big, it, it * 16 * 64 / 8 / 32, d_clk); }, 0);
        run("global_load_alone", [&] { hipLaunchKernelGGL((coexec<PART_GLOAD, false, false>), dim3(256), dim3(512), 0, 0, d_in, d_out, d_big, it, it * 16 * 64 / 64 / 8, d_clk); }, 0);
        run("v_fma_alone", [&] { hipLaunchKernelGGL((coexec<PART_FMA32, false, false>), dim3(256), dim3(512), 0, 0, d_in, d_out, d_big, it, it * 16 * 64 / 4 / 32, d_clk); }, 0);
        run("bf16_mfma_2w_wg", [&] { hipLaunchKernelGGL((coexec<PART_NONE, true, true>), dim3(256), dim3(512), 0, 0, d_in, d_out, d_big, it * 2, 0, d_clk); }, 256.0 * 4 * it * 2 * 16 * 32768.0);
    }
    if (which == 0 || which == 4) {
        printf("P4 control: the same with v_mfma_f32_32x32x16_bf16 in waves 0-3\n");
        const int it = 8000;                       // 8000*16 MFMAs * 32 cyc = 4.1 M cycles
        run_coexec<PART_FMA32, true>("v_fma_f32", it, it * 16 * 32 / 4 / 32);
        run_coexec<PART_DSREAD, true>("ds_read_b32", it, it * 16 * 32 / 8 / 32);
        run_coexec<PART_DSWRITE32, true>("ds_write_b32", it, it * 16 * 32 / 8 / 32);
        fflush(stdout);
    }
    return 0;
}
