// Flat-buffer optimiser step (src/solver.py:194-196 with Adam from src/train.py:92-95), gfx950.
//
// The whole model lives in one flat fp32 buffer (and so do its gradient and the two Adam
// moments), so the per-step tail of the training loop is three streaming kernels instead of
// 294 x (norm, clip-scale, 4 Adam ops):  sum-of-squares partials -> clip coefficient + Adam.
// The same flat gradient is the single RCCL all-reduce payload in data-parallel runs.
#include "ctn_common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXPART = 1024;

__global__ __launch_bounds__(NT) void sumsq_kernel(const float* __restrict__ g, long long n, double* __restrict__ part) {
    __shared__ double red[NT / 64];
    double s = 0.0;
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        const float4 v = *reinterpret_cast<const float4*>(g + 4 * i);
        s += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0)
        for (long long i = n4 * 4 + threadIdx.x; i < n; i += NT) s += (double)g[i] * g[i];
    s = block_sum<double, NT>(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// grad_scale multiplies g before everything (1/world after a sum all-reduce).
// total_norm_out[0] = ||grad_scale * g||_2 ; coef = min(1, max_norm / (norm + 1e-6)) as clip_grad_norm_.
__global__ __launch_bounds__(NT) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, long long n,
                                                       const double* __restrict__ part, int nparts, float grad_scale,
                                                       float max_norm, float lr, float b1, float b2, float eps,
                                                       float bc1, float bc2_sqrt, float* __restrict__ total_norm_out) {
    __shared__ double red[NT / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += NT) s += part[i];
    s = block_sum<double, NT>(s, red);
    const float total = (float)sqrt(s) * fabsf(grad_scale);
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (total + 1e-6f), 1.f);
    if (blockIdx.x == 0 && threadIdx.x == 0 && total_norm_out != nullptr) total_norm_out[0] = total;
    const float gs = grad_scale * coef;
    const float step = lr / bc1;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const float gi = g[i] * gs;
        const float mi = m[i] * b1 + gi * (1.f - b1);
        const float vi = v[i] * b2 + (gi * gi) * (1.f - b2);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step * (mi / denom);
    }
}

}  // namespace

extern "C" {

int ctn_optim_parts(void) { return MAXPART; }

// workspace: ctn_optim_parts() doubles
int ctn_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                       float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps, int step,
                       float* total_norm_out, double* workspace, void* stream) {
    CTN_REQUIRE(params && grads && exp_avg && exp_avg_sq && workspace, "ctn_clip_adam_step: null pointer");
    CTN_REQUIRE(n > 0 && step >= 1, "ctn_clip_adam_step: bad sizes");
    CTN_REQUIRE((reinterpret_cast<uintptr_t>(grads) & 15) == 0, "ctn_clip_adam_step: grads must be 16-byte aligned");
    long long nb = ctn_cdivll(n / 4 + 1, NT);
    if (nb > MAXPART) nb = MAXPART;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nb), dim3(NT), 0, st, grads, n, workspace);
    CTN_CHECK_LAUNCH("ctn_clip_adam_step/sumsq");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    long long nb2 = ctn_cdivll(n, NT);
    if (nb2 > 2048) nb2 = 2048;
    hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)nb2), dim3(NT), 0, st, params, grads, exp_avg, exp_avg_sq, n,
                       (const double*)workspace, (int)nb, grad_scale, max_norm, lr, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2), total_norm_out);
    CTN_CHECK_LAUNCH("ctn_clip_adam_step/adam");
    return CTN_OK;
}

}  // extern "C"
