// BatchNorm1d over (utterances, frames) per channel, optionally behind a single-slope PReLU -- the
// norm_type="BN" branch of chose_norm (src/conv_tasnet.py:305-309), used at :225 and :260.
//
// HBM-bound: every kernel streams [M, Ch, Kp] rows with one wave per (m, c) row and float4 accesses.  Statistics
// are fp64 per-row partials summed in a fixed order (m = 0..M-1) by a one-thread-per-channel finalize kernel, so
// results are bitwise reproducible; frames >= K never enter a sum and are written as zeros (the zero-pad
// invariant the depthwise and GEMM kernels rely on).
#include "ctn_common.h"

namespace {

constexpr int NT = 256;
constexpr int ROWS = 4;   // one wave per (m, c) row

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// part[c][m] = (sum, sum of squares) of prelu(Y[m,c,0:K])
__global__ __launch_bounds__(NT) void bn_row_moments_kernel(const float* __restrict__ Y, const float* __restrict__ alpha_p,
                                                            int M, int Ch, int K, int Kp, double* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = (Ch + ROWS - 1) / ROWS;
    const int m = blockIdx.x / cb, c = (blockIdx.x % cb) * ROWS + wave;
    if (c >= Ch) return;
    const float al = alpha_p ? alpha_p[0] : 1.f;
    const float* __restrict__ y = Y + ((size_t)m * Ch + c) * Kp;
    double s = 0.0, q = 0.0;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 v4 = ld4(y + k);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
        float ls = 0.f, lq = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k + e < K) {
                const float p = prelu_f(v[e], al);
                ls += p;
                lq += p * p;
            }
        s += (double)ls;
        q += (double)lq;
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) {
        part[((size_t)c * M + m) * 2] = s;
        part[((size_t)c * M + m) * 2 + 1] = q;
    }
}

// training: batch mean / biased variance from the row partials, running statistics updated as nn.BatchNorm1d does
// (momentum blend, unbiased variance); eval: the running statistics themselves.  mr[c] = (mean, 1/sqrt(var + eps)).
__global__ __launch_bounds__(NT) void bn_finalize_kernel(const double* __restrict__ part, int M, int Ch, double n, float eps,
                                                         float momentum, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, float* __restrict__ mr) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= Ch) return;
    double mu, var;
    if (part != nullptr) {
        double s = 0.0, q = 0.0;
        for (int m = 0; m < M; ++m) {
            s += part[((size_t)c * M + m) * 2];
            q += part[((size_t)c * M + m) * 2 + 1];
        }
        mu = s / n;
        var = q / n - mu * mu;
        if (var < 0.0) var = 0.0;
        if (running_mean != nullptr) {
            const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mu);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
    } else {
        mu = (double)running_mean[c];
        var = (double)running_var[c];
    }
    mr[2 * c] = (float)mu;
    mr[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(NT) void bn_apply_kernel(const float* __restrict__ Y, float* __restrict__ Out,
                                                      const float* __restrict__ alpha_p, const float* __restrict__ mr,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int M, int Ch, int K, int Kp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = (Ch + ROWS - 1) / ROWS;
    const int m = blockIdx.x / cb, c = (blockIdx.x % cb) * ROWS + wave;
    if (c >= Ch) return;
    const float al = alpha_p ? alpha_p[0] : 1.f;
    const float mean = mr[2 * c], sc = mr[2 * c + 1] * gamma[c], sh = beta[c];
    const size_t row = ((size_t)m * Ch + c) * Kp;
    for (int k = lane * 4; k < Kp; k += 256) {
        const float4 v4 = ld4(Y + row + k);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (k + e) < K ? (prelu_f(v[e], al) - mean) * sc + sh : 0.f;
        *reinterpret_cast<float4*>(Out + row + k) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// part[c][m] = (sum dOut, sum dOut * xhat),  xhat = (prelu(y) - mean) * rstd
__global__ __launch_bounds__(NT) void bn_bwd_row_sums_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                             const float* __restrict__ alpha_p, const float* __restrict__ mr,
                                                             int M, int Ch, int K, int Kp, double* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = (Ch + ROWS - 1) / ROWS;
    const int m = blockIdx.x / cb, c = (blockIdx.x % cb) * ROWS + wave;
    if (c >= Ch) return;
    const float al = alpha_p ? alpha_p[0] : 1.f;
    const float mean = mr[2 * c], rstd = mr[2 * c + 1];
    const size_t row = ((size_t)m * Ch + c) * Kp;
    double s1 = 0.0, s2 = 0.0;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 d4 = ld4(dOut + row + k);
        const float4 y4 = ld4(Y + row + k);
        const float d[4] = {d4.x, d4.y, d4.z, d4.w};
        const float y[4] = {y4.x, y4.y, y4.z, y4.w};
        float l1 = 0.f, l2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (k + e < K) {
                l1 += d[e];
                l2 += d[e] * ((prelu_f(y[e], al) - mean) * rstd);
            }
        s1 += (double)l1;
        s2 += (double)l2;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        part[((size_t)c * M + m) * 2] = s1;
        part[((size_t)c * M + m) * 2 + 1] = s2;
    }
}

// dgamma[c] = S2, dbeta[c] = S1, coef[c] = (S1/n, S2/n) in training mode (batch statistics depend on the input) or
// (0, 0) in eval mode (running statistics are constants).
__global__ __launch_bounds__(NT) void bn_bwd_finalize_kernel(const double* __restrict__ part, int M, int Ch, double n,
                                                             int training, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ coef) {
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= Ch) return;
    double s1 = 0.0, s2 = 0.0;
    for (int m = 0; m < M; ++m) {
        s1 += part[((size_t)c * M + m) * 2];
        s2 += part[((size_t)c * M + m) * 2 + 1];
    }
    dgamma[c] = (float)s2;
    dbeta[c] = (float)s1;
    coef[2 * c] = training ? (float)(s1 / n) : 0.f;
    coef[2 * c + 1] = training ? (float)(s2 / n) : 0.f;
}

// dY = gamma * rstd * (dOut - c1 - xhat * c2) * prelu'(y);  dalpha_part[m*Ch + c] = sum over the row of dP * y [y < 0]
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const float* __restrict__ dOut, const float* __restrict__ Y,
                                                          float* __restrict__ dY, const float* __restrict__ alpha_p,
                                                          const float* __restrict__ mr, const float* __restrict__ gamma,
                                                          const float* __restrict__ coef, int M, int Ch, int K, int Kp,
                                                          float* __restrict__ dalpha_part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cb = (Ch + ROWS - 1) / ROWS;
    const int m = blockIdx.x / cb, c = (blockIdx.x % cb) * ROWS + wave;
    if (c >= Ch) return;
    const bool has_a = alpha_p != nullptr;
    const float al = has_a ? alpha_p[0] : 1.f;
    const float mean = mr[2 * c], rstd = mr[2 * c + 1], gs = gamma[c] * rstd, c1 = coef[2 * c], c2 = coef[2 * c + 1];
    const size_t row = ((size_t)m * Ch + c) * Kp;
    float dal = 0.f;
    for (int k = lane * 4; k < Kp; k += 256) {
        const float4 d4 = ld4(dOut + row + k);
        const float4 y4 = ld4(Y + row + k);
        const float d[4] = {d4.x, d4.y, d4.z, d4.w};
        const float y[4] = {y4.x, y4.y, y4.z, y4.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (prelu_f(y[e], al) - mean) * rstd;
            const float dp = gs * (d[e] - c1 - xh * c2);
            const bool valid = (k + e) < K;
            if (has_a && valid && y[e] < 0.f) dal += dp * y[e];
            o[e] = valid ? ((!has_a || y[e] >= 0.f) ? dp : al * dp) : 0.f;
        }
        *reinterpret_cast<float4*>(dY + row + k) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (has_a) {
        dal = wave_sum(dal);
        if (lane == 0) dalpha_part[(size_t)m * Ch + c] = dal;
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int ctn_bn_fwd(const float* Y, float* Out, const float* alpha, const float* gamma, const float* beta,
               float* running_mean, float* running_var, int training, float eps, float momentum,
               int M, int Ch, int K, int Kp, double* part, float* mr, void* stream) {
    CTN_REQUIRE(Y && Out && gamma && beta && mr, "ctn_bn_fwd: null pointer");
    CTN_REQUIRE(M > 0 && Ch > 0 && K > 0 && Kp >= K && Kp % 4 == 0, "ctn_bn_fwd: bad sizes M=%d Ch=%d K=%d Kp=%d", M, Ch, K, Kp);
    CTN_REQUIRE(training ? part != nullptr : (running_mean && running_var),
                "ctn_bn_fwd: training needs the partials workspace, eval needs running statistics");
    CTN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "ctn_bn_fwd: running_mean / running_var go together");
    CTN_REQUIRE(aligned16(Y) && aligned16(Out), "ctn_bn_fwd: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const dim3 rows((unsigned)(M * ctn_cdiv(Ch, ROWS))), chans((unsigned)ctn_cdiv(Ch, NT));
    if (training) {
        hipLaunchKernelGGL(bn_row_moments_kernel, rows, dim3(NT), 0, st, Y, alpha, M, Ch, K, Kp, part);
        CTN_CHECK_LAUNCH("ctn_bn_fwd/moments");
    }
    hipLaunchKernelGGL(bn_finalize_kernel, chans, dim3(NT), 0, st, training ? part : (const double*)nullptr, M, Ch,
                       (double)M * (double)K, eps, momentum, running_mean, running_var, mr);
    CTN_CHECK_LAUNCH("ctn_bn_fwd/finalize");
    hipLaunchKernelGGL(bn_apply_kernel, rows, dim3(NT), 0, st, Y, Out, alpha, mr, gamma, beta, M, Ch, K, Kp);
    CTN_CHECK_LAUNCH("ctn_bn_fwd/apply");
    return CTN_OK;
}

int ctn_bn_bwd(const float* dOut, const float* Y, float* dY, const float* alpha, const float* gamma, const float* mr,
               int training, int M, int Ch, int K, int Kp, double* part, float* coef, float* dgamma, float* dbeta,
               float* dalpha_part, void* stream) {
    CTN_REQUIRE(dOut && Y && dY && gamma && mr && part && coef && dgamma && dbeta, "ctn_bn_bwd: null pointer");
    CTN_REQUIRE(!alpha || dalpha_part, "ctn_bn_bwd: dalpha_part required with alpha");
    CTN_REQUIRE(M > 0 && Ch > 0 && K > 0 && Kp >= K && Kp % 4 == 0, "ctn_bn_bwd: bad sizes M=%d Ch=%d K=%d Kp=%d", M, Ch, K, Kp);
    CTN_REQUIRE(aligned16(dOut) && aligned16(Y) && aligned16(dY), "ctn_bn_bwd: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const dim3 rows((unsigned)(M * ctn_cdiv(Ch, ROWS))), chans((unsigned)ctn_cdiv(Ch, NT));
    hipLaunchKernelGGL(bn_bwd_row_sums_kernel, rows, dim3(NT), 0, st, dOut, Y, alpha, mr, M, Ch, K, Kp, part);
    CTN_CHECK_LAUNCH("ctn_bn_bwd/sums");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, chans, dim3(NT), 0, st, part, M, Ch, (double)M * (double)K, training,
                       dgamma, dbeta, coef);
    CTN_CHECK_LAUNCH("ctn_bn_bwd/finalize");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, rows, dim3(NT), 0, st, dOut, Y, dY, alpha, mr, gamma, coef, M, Ch, K, Kp,
                       dalpha_part);
    CTN_CHECK_LAUNCH("ctn_bn_bwd/apply");
    return CTN_OK;
}

}  // extern "C"
