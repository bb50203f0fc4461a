/* A torch-free, Python-free host of the C ABI: the reference's Encoder (src/conv_tasnet.py:106-121,
 * relu(conv1d(mixture, U, stride=L/2))) driven from plain C through include/ctn_hip.h, checked against a scalar loop.
 * Built and run by tests/test_c_abi_host.py:  cc host_demo.c -lctn_hip -lamdhip64.
 * Exit code 0 and "OK" on success. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ctn_hip.h"

#define HIP_OK(x)                                                          \
    do {                                                                   \
        hipError_t e_ = (x);                                               \
        if (e_ != hipSuccess) {                                            \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));        \
            return 2;                                                      \
        }                                                                  \
    } while (0)
#define CTN_OK_(x)                                                         \
    do {                                                                   \
        int r_ = (x);                                                      \
        if (r_ != 0) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #x, r_, ctn_last_error());   \
            return 3;                                                      \
        }                                                                  \
    } while (0)

static float frand(unsigned* s) {
    *s = *s * 1664525u + 1013904223u;
    return (float)((*s >> 8) & 0xffff) / 32768.0f - 1.0f;
}

int main(void) {
    const int M = 2, T = 4000, L = 20, S = L / 2, N = 64;
    const int K = (T - L) / S + 1;
    const int Kp = ctn_padded_frames(K);
    unsigned seed = 12345u;
    float* mix = (float*)malloc(sizeof(float) * M * T);
    float* U = (float*)malloc(sizeof(float) * N * L);
    float* got = (float*)malloc(sizeof(float) * (size_t)M * N * Kp);
    for (int i = 0; i < M * T; ++i) mix[i] = frand(&seed);
    for (int i = 0; i < N * L; ++i) U[i] = 0.3f * frand(&seed);

    float *d_mix, *d_U, *d_col, *d_w;
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    HIP_OK(hipMalloc((void**)&d_mix, sizeof(float) * M * T));
    HIP_OK(hipMalloc((void**)&d_U, sizeof(float) * N * L));
    HIP_OK(hipMalloc((void**)&d_col, sizeof(float) * (size_t)M * L * Kp));
    HIP_OK(hipMalloc((void**)&d_w, sizeof(float) * (size_t)M * N * Kp));
    HIP_OK(hipMemcpy(d_mix, mix, sizeof(float) * M * T, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_U, U, sizeof(float) * N * L, hipMemcpyHostToDevice));

    printf("libctn_hip version %d, K = %d, Kp = %d\n", ctn_version(), K, Kp);
    CTN_OK_(ctn_im2col(d_mix, d_col, M, T, L, L, K, Kp, st));
    CTN_OK_(ctn_pw_gemm(d_U, d_col, d_w, M, N, L, K, Kp, 0, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 1, st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(got, d_w, sizeof(float) * (size_t)M * N * Kp, hipMemcpyDeviceToHost));

    double worst = 0.0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < Kp; ++k) {
                double ref = 0.0;
                if (k < K) {
                    for (int l = 0; l < L; ++l) ref += (double)U[n * L + l] * (double)mix[m * T + k * S + l];
                    if (ref < 0.0) ref = 0.0;
                }
                const double d = fabs(ref - (double)got[((size_t)m * N + n) * Kp + k]);
                if (d > worst) worst = d;
            }
    printf("encoder max abs err vs scalar loop: %.3e (pad frames must be exactly zero)\n", worst);

    /* error behaviour: bad sizes return a negative code and a message, nothing is launched */
    const int rc = ctn_pw_gemm(d_U, d_col, d_w, M, N, L, K, Kp - 1, 0, NULL, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 1, st);
    printf("bad Kp -> %d (%s)\n", rc, ctn_last_error());

    hipFree(d_mix); hipFree(d_U); hipFree(d_col); hipFree(d_w);
    hipStreamDestroy(st);
    free(mix); free(U); free(got);
    if (!(worst < 2e-5) || rc >= 0) { printf("FAIL\n"); return 1; }
    printf("OK\n");
    return 0;
}
