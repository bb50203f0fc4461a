// Encoder / decoder glue kernels around the MFMA GEMMs, gfx950.  All HBM-bound, frames on lanes.
//
//  encoder  (src/conv_tasnet.py:106-121):  w = relu(U . frames(x))      = im2col + ctn_pw_gemm(act=relu)
//  decoder  (src/conv_tasnet.py:140-145):  est = OLA(V . (w * mask))    = mask_apply + ctn_pw_gemm + ola
//  overlap_and_add (src/utils.py:9-47) is a gather here (each output sample sums the <= ceil(L/S)
//  frame taps that cover it): deterministic, no atomics, unlike index_add_.
#include "ctn_common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXC = 8;   // max speakers for the softmax mask

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Xcol[m][l][k] = x[m][k*S + l]  (k < K, l < L), zero elsewhere.  Xcol is [M, Lp, Kp].
__global__ __launch_bounds__(NT) void im2col_kernel(const float* __restrict__ x, float* __restrict__ xc,
                                                    int M, int T, int L, int Lp, int S, int K, int Kp) {
    const long long n = (long long)M * Lp * Kp;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const int k = (int)(i % Kp);
        const int l = (int)((i / Kp) % Lp);
        const int m = (int)(i / ((long long)Kp * Lp));
        float v = 0.f;
        if (k < K && l < L) v = x[(size_t)m * T + (size_t)k * S + l];
        xc[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Encoder, src/conv_tasnet.py:106-121:  w[m, n, k] = relu( sum_l U[n, l] * mix[m, k*S + l] ),  S = L/2  (Conv1d(1, N, L, stride
// L/2, bias=False) + ReLU), zero for frames k >= K.  HBM-bound on its output (M*N*K*4 bytes; the input is 1/N of that):
// the L-sample sliding windows of a block of 256 frames are staged in LDS once (256*S + S samples, coalesced) together
// with the basis rows of this workgroup's 64 channels; lane = frame, so every output store of a wave is 256 contiguous
// bytes along the frame axis; a basis row is read as broadcast float4s.  No im2col buffer on the forward path.
// ---------------------------------------------------------------------------------------------------------
constexpr int ENC_FR = 256, ENC_CH = 64;
template <int L>
__global__ __launch_bounds__(ENC_FR) void encoder_fwd_kernel(const float* __restrict__ mix, const float* __restrict__ U,
                                                            float* __restrict__ w, int T, int N, int K, int Kp) {
    constexpr int S = L / 2;
    __shared__ __attribute__((aligned(16))) float win[ENC_FR * S + L];
    __shared__ __attribute__((aligned(16))) float ub[ENC_CH * L];
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * ENC_FR, n0 = blockIdx.y * ENC_CH, m = blockIdx.z;
    const float* __restrict__ x = mix + (size_t)m * T;
    for (int i = tid; i < ENC_FR * S + L; i += ENC_FR) {
        const int t = k0 * S + i;
        win[i] = t < T ? x[t] : 0.f;
    }
    for (int i = tid; i < ENC_CH * L; i += ENC_FR) {
        const int n = n0 + i / L;
        ub[i] = n < N ? U[(size_t)n * L + i % L] : 0.f;
    }
    __syncthreads();
    const int k = k0 + tid;
    float xv[L];
#pragma unroll
    for (int l = 0; l < L; ++l) xv[l] = win[tid * S + l];
    if (k >= Kp) return;
    const bool valid = k < K;
    float* __restrict__ out = w + ((size_t)m * N + n0) * Kp + k;
    for (int c = 0; c < ENC_CH && n0 + c < N; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int l4 = 0; l4 < L / 4; ++l4) {
            const float4 u = *reinterpret_cast<const float4*>(ub + c * L + 4 * l4);       // same address in every lane: broadcast
            acc = fmaf(u.x, xv[4 * l4 + 0], acc);
            acc = fmaf(u.y, xv[4 * l4 + 1], acc);
            acc = fmaf(u.z, xv[4 * l4 + 2], acc);
            acc = fmaf(u.w, xv[4 * l4 + 3], acc);
        }
        out[(size_t)c * Kp] = valid ? fmaxf(acc, 0.f) : 0.f;
    }
}

// sw[m,c,n,k] = w[m,n,k] * act(score[m,c,n,k]);  act = relu (mode 0), softmax over c (mode 1) or identity (mode 2: the
// stand-alone Decoder.forward, whose est_mask argument already is a mask, src/conv_tasnet.py:140)
__global__ __launch_bounds__(NT) void mask_apply_kernel(const float* __restrict__ score, const float* __restrict__ w,
                                                        float* __restrict__ sw, int M, int C, long long NK, int mode) {
    const long long n4 = (long long)M * NK / 4;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        const long long e = i * 4;
        const int m = (int)(e / NK);
        const long long r = e % NK;
        const float4 wv = ld4(w + e);
        const float ww[4] = {wv.x, wv.y, wv.z, wv.w};
        const size_t base = (size_t)m * C * NK + r;
        if (mode == 0) {
            for (int c = 0; c < C; ++c) {
                const float4 s = ld4(score + base + (size_t)c * NK);
                *reinterpret_cast<float4*>(sw + base + (size_t)c * NK) =
                    make_float4(ww[0] * fmaxf(s.x, 0.f), ww[1] * fmaxf(s.y, 0.f), ww[2] * fmaxf(s.z, 0.f), ww[3] * fmaxf(s.w, 0.f));
            }
        } else if (mode == 2) {
            for (int c = 0; c < C; ++c) {
                const float4 s = ld4(score + base + (size_t)c * NK);
                *reinterpret_cast<float4*>(sw + base + (size_t)c * NK) = make_float4(ww[0] * s.x, ww[1] * s.y, ww[2] * s.z, ww[3] * s.w);
            }
        } else {
            float sc[MAXC][4];
            float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, den[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) {
                    const float4 s = ld4(score + base + (size_t)c * NK);
                    sc[c][0] = s.x; sc[c][1] = s.y; sc[c][2] = s.z; sc[c][3] = s.w;
#pragma unroll
                    for (int q = 0; q < 4; ++q) mx[q] = fmaxf(mx[q], sc[c][q]);
                }
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { sc[c][q] = expf(sc[c][q] - mx[q]); den[q] += sc[c][q]; }
                }
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C)
                    *reinterpret_cast<float4*>(sw + base + (size_t)c * NK) =
                        make_float4(ww[0] * (sc[c][0] / den[0]), ww[1] * (sc[c][1] / den[1]),
                                    ww[2] * (sc[c][2] / den[2]), ww[3] * (sc[c][3] / den[3]));
        }
    }
}

// backward of mask_apply: dscore (may alias dsw) and dw[m,n,k] = sum_c dsw*mask
__global__ __launch_bounds__(NT) void mask_apply_bwd_kernel(const float* __restrict__ dsw, const float* __restrict__ score,
                                                            const float* __restrict__ w, float* __restrict__ dscore,
                                                            float* __restrict__ dw, int M, int C, long long NK, int mode) {
    const long long n4 = (long long)M * NK / 4;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
        const long long e = i * 4;
        const int m = (int)(e / NK);
        const long long r = e % NK;
        const float4 wv = ld4(w + e);
        const float ww[4] = {wv.x, wv.y, wv.z, wv.w};
        const size_t base = (size_t)m * C * NK + r;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (mode == 0) {
            for (int c = 0; c < C; ++c) {
                const float4 s = ld4(score + base + (size_t)c * NK);
                const float4 g = ld4(dsw + base + (size_t)c * NK);
                const float sv[4] = {s.x, s.y, s.z, s.w}, gv[4] = {g.x, g.y, g.z, g.w};
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[q] += gv[q] * fmaxf(sv[q], 0.f);
                    o[q] = sv[q] > 0.f ? gv[q] * ww[q] : 0.f;
                }
                *reinterpret_cast<float4*>(dscore + base + (size_t)c * NK) = make_float4(o[0], o[1], o[2], o[3]);
            }
        } else if (mode == 2) {
            for (int c = 0; c < C; ++c) {
                const float4 s = ld4(score + base + (size_t)c * NK);
                const float4 g = ld4(dsw + base + (size_t)c * NK);
                acc[0] += g.x * s.x; acc[1] += g.y * s.y; acc[2] += g.z * s.z; acc[3] += g.w * s.w;
                *reinterpret_cast<float4*>(dscore + base + (size_t)c * NK) = make_float4(g.x * ww[0], g.y * ww[1], g.z * ww[2], g.w * ww[3]);
            }
        } else {
            float pr[MAXC][4], gm[MAXC][4];
            float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, den[4] = {0.f, 0.f, 0.f, 0.f}, dotv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) {
                    const float4 s = ld4(score + base + (size_t)c * NK);
                    const float4 g = ld4(dsw + base + (size_t)c * NK);
                    pr[c][0] = s.x; pr[c][1] = s.y; pr[c][2] = s.z; pr[c][3] = s.w;
                    gm[c][0] = g.x; gm[c][1] = g.y; gm[c][2] = g.z; gm[c][3] = g.w;
#pragma unroll
                    for (int q = 0; q < 4; ++q) mx[q] = fmaxf(mx[q], pr[c][q]);
                }
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { pr[c][q] = expf(pr[c][q] - mx[q]); den[q] += pr[c][q]; }
                }
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        pr[c][q] /= den[q];
                        acc[q] += gm[c][q] * pr[c][q];          // dw
                        gm[c][q] *= ww[q];                      // dmask
                        dotv[q] += gm[c][q] * pr[c][q];
                    }
                }
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C)
                    *reinterpret_cast<float4*>(dscore + base + (size_t)c * NK) =
                        make_float4(pr[c][0] * (gm[c][0] - dotv[0]), pr[c][1] * (gm[c][1] - dotv[1]),
                                    pr[c][2] * (gm[c][2] - dotv[2]), pr[c][3] * (gm[c][3] - dotv[3]));
        }
        *reinterpret_cast<float4*>(dw + e) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// est[b][t] = sum_{k,l: k*S+l = t} fr[b][l][k],  b over M*C rows; est is [B, T], zero for t >= (K-1)S+L
__global__ __launch_bounds__(NT) void ola_kernel(const float* __restrict__ fr, float* __restrict__ est,
                                                 int Bn, int T, int L, int Lp, int S, int K, int Kp) {
    const long long n = (long long)Bn * T;
    const int q = (L + S - 1) / S;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const int t = (int)(i % T);
        const int b = (int)(i / T);
        const float* __restrict__ f = fr + (size_t)b * Lp * Kp;
        const int kh = t / S;
        float v = 0.f;
        for (int j = q - 1; j >= 0; --j) {        // ascending frame index, like the reference's accumulation order
            const int k = kh - j, l = t - k * S;
            if (k >= 0 && k < K && l < L) v += f[(size_t)l * Kp + k];
        }
        est[i] = v;
    }
}

// dfr[b][l][k] = (k < K && l < L) ? dest[b][k*S + l] : 0
__global__ __launch_bounds__(NT) void unfold_kernel(const float* __restrict__ dest, float* __restrict__ dfr,
                                                    int Bn, int T, int L, int Lp, int S, int K, int Kp) {
    const long long n = (long long)Bn * Lp * Kp;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const int k = (int)(i % Kp);
        const int l = (int)((i / Kp) % Lp);
        const int b = (int)(i / ((long long)Kp * Lp));
        float v = 0.f;
        if (k < K && l < L) v = dest[(size_t)b * T + (size_t)k * S + l];
        dfr[i] = v;
    }
}

// General overlap-and-add, src/utils.py:9-47: out[b][j*step + l] += sig[b][j][l] for any frame_step (the reference
// splits frames into gcd(frame_length, frame_step) sub-frames and index_add_s them).  Gather form: every output sample
// sums, in ascending frame order, the <= ceil(frame_length / step) frames that cover it.  sig is [Bn, F, L] row-major
// (the reference's [..., frames, frame_length]), out [Bn, (F-1)*step + L].
__global__ __launch_bounds__(NT) void ola_general_kernel(const float* __restrict__ sig, float* __restrict__ out,
                                                         int Bn, int F, int L, int step, int T) {
    const long long n = (long long)Bn * T;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const int t = (int)(i % T);
        const int b = (int)(i / T);
        const float* __restrict__ s = sig + (size_t)b * F * L;
        int j0 = (t - L + step) / step;                  // first frame with j*step + L > t
        if (t - L + step < 0) j0 = 0;
        int j1 = t / step;
        if (j1 > F - 1) j1 = F - 1;
        float v = 0.f;
        for (int j = j0; j <= j1; ++j) v += s[(size_t)j * L + (t - j * step)];
        out[i] = v;
    }
}

// backward: dsig[b][j][l] = dout[b][j*step + l]
__global__ __launch_bounds__(NT) void ola_general_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dsig,
                                                             int Bn, int F, int L, int step, int T) {
    const long long n = (long long)Bn * F * L;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
        const int l = (int)(i % L);
        const int j = (int)((i / L) % F);
        const int b = (int)(i / ((long long)L * F));
        dsig[i] = dout[(size_t)b * T + (size_t)j * step + l];
    }
}

unsigned grid_for(long long n) {
    long long b = ctn_cdivll(n, NT);
    if (b > 256 * 16) b = 256 * 16;   // grid-stride beyond 16 workgroups per CU
    if (b < 1) b = 1;
    return (unsigned)b;
}
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int ctn_im2col(const float* mix, float* xcol, int M, int T, int L, int Lp, int K, int Kp, void* stream) {
    CTN_REQUIRE(mix && xcol, "ctn_im2col: null pointer");
    CTN_REQUIRE(M > 0 && L >= 2 && Lp >= L && T >= L && K == (T - L) / (L / 2) + 1 && Kp >= K,
                "ctn_im2col: inconsistent sizes (T=%d L=%d K=%d Kp=%d)", T, L, K, Kp);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((long long)M * Lp * Kp)), dim3(NT), 0, (hipStream_t)stream,
                       mix, xcol, M, T, L, Lp, L / 2, K, Kp);
    CTN_CHECK_LAUNCH("ctn_im2col");
    return CTN_OK;
}

int ctn_encoder_supported(int L) { return L == 16 || L == 20 || L == 32 || L == 40; }

int ctn_encoder_fwd(const float* mix, const float* U, float* w, int M, int T, int N, int L, int K, int Kp, void* stream) {
    CTN_REQUIRE(mix && U && w, "ctn_encoder_fwd: null pointer");
    CTN_REQUIRE(ctn_encoder_supported(L), "ctn_encoder_fwd: filter length %d not compiled in (16, 20, 32, 40): use ctn_im2col + ctn_pw_gemm", L);
    CTN_REQUIRE(M > 0 && N > 0 && T >= L && K == (T - L) / (L / 2) + 1 && Kp >= K, "ctn_encoder_fwd: inconsistent sizes (T=%d L=%d K=%d Kp=%d)", T, L, K, Kp);
    const dim3 grid((unsigned)ctn_cdiv(Kp, ENC_FR), (unsigned)ctn_cdiv(N, ENC_CH), (unsigned)M), block(ENC_FR);
    hipStream_t st = (hipStream_t)stream;
    switch (L) {
        case 16: hipLaunchKernelGGL(encoder_fwd_kernel<16>, grid, block, 0, st, mix, U, w, T, N, K, Kp); break;
        case 20: hipLaunchKernelGGL(encoder_fwd_kernel<20>, grid, block, 0, st, mix, U, w, T, N, K, Kp); break;
        case 32: hipLaunchKernelGGL(encoder_fwd_kernel<32>, grid, block, 0, st, mix, U, w, T, N, K, Kp); break;
        default: hipLaunchKernelGGL(encoder_fwd_kernel<40>, grid, block, 0, st, mix, U, w, T, N, K, Kp); break;
    }
    CTN_CHECK_LAUNCH("ctn_encoder_fwd");
    return CTN_OK;
}

int ctn_mask_apply(const float* score, const float* w, float* sw, int M, int C, int N, int Kp, int softmax, void* stream) {
    CTN_REQUIRE(score && w && sw, "ctn_mask_apply: null pointer");
    CTN_REQUIRE(M > 0 && C > 0 && N > 0 && Kp > 0 && Kp % 4 == 0, "ctn_mask_apply: bad sizes");
    CTN_REQUIRE(softmax >= 0 && softmax <= 2, "ctn_mask_apply: mode must be 0 (relu), 1 (softmax) or 2 (identity)");
    CTN_REQUIRE(softmax != 1 || C <= MAXC, "ctn_mask_apply: softmax mask supports at most %d speakers", MAXC);
    CTN_REQUIRE(aligned16(score) && aligned16(w) && aligned16(sw), "ctn_mask_apply: alignment");
    const long long NK = (long long)N * Kp;
    hipLaunchKernelGGL(mask_apply_kernel, dim3(grid_for((long long)M * NK / 4)), dim3(NT), 0, (hipStream_t)stream,
                       score, w, sw, M, C, NK, softmax);
    CTN_CHECK_LAUNCH("ctn_mask_apply");
    return CTN_OK;
}

int ctn_mask_apply_bwd(const float* dsw, const float* score, const float* w, float* dscore, float* dw,
                       int M, int C, int N, int Kp, int softmax, void* stream) {
    CTN_REQUIRE(dsw && score && w && dscore && dw, "ctn_mask_apply_bwd: null pointer");
    CTN_REQUIRE(M > 0 && C > 0 && N > 0 && Kp > 0 && Kp % 4 == 0, "ctn_mask_apply_bwd: bad sizes");
    CTN_REQUIRE(softmax >= 0 && softmax <= 2, "ctn_mask_apply_bwd: mode must be 0 (relu), 1 (softmax) or 2 (identity)");
    CTN_REQUIRE(softmax != 1 || C <= MAXC, "ctn_mask_apply_bwd: softmax mask supports at most %d speakers", MAXC);
    CTN_REQUIRE(aligned16(dsw) && aligned16(score) && aligned16(w) && aligned16(dscore) && aligned16(dw), "ctn_mask_apply_bwd: alignment");
    const long long NK = (long long)N * Kp;
    hipLaunchKernelGGL(mask_apply_bwd_kernel, dim3(grid_for((long long)M * NK / 4)), dim3(NT), 0, (hipStream_t)stream,
                       dsw, score, w, dscore, dw, M, C, NK, softmax);
    CTN_CHECK_LAUNCH("ctn_mask_apply_bwd");
    return CTN_OK;
}

int ctn_ola(const float* frames, float* est, int Bn, int T, int L, int Lp, int K, int Kp, void* stream) {
    CTN_REQUIRE(frames && est, "ctn_ola: null pointer");
    CTN_REQUIRE(Bn > 0 && L >= 2 && Lp >= L && K > 0 && Kp >= K && T >= (K - 1) * (L / 2) + L, "ctn_ola: inconsistent sizes");
    hipLaunchKernelGGL(ola_kernel, dim3(grid_for((long long)Bn * T)), dim3(NT), 0, (hipStream_t)stream,
                       frames, est, Bn, T, L, Lp, L / 2, K, Kp);
    CTN_CHECK_LAUNCH("ctn_ola");
    return CTN_OK;
}

int ctn_unfold(const float* dest, float* dframes, int Bn, int T, int L, int Lp, int K, int Kp, void* stream) {
    CTN_REQUIRE(dest && dframes, "ctn_unfold: null pointer");
    CTN_REQUIRE(Bn > 0 && L >= 2 && Lp >= L && K > 0 && Kp >= K && T >= (K - 1) * (L / 2) + L, "ctn_unfold: inconsistent sizes");
    hipLaunchKernelGGL(unfold_kernel, dim3(grid_for((long long)Bn * Lp * Kp)), dim3(NT), 0, (hipStream_t)stream,
                       dest, dframes, Bn, T, L, Lp, L / 2, K, Kp);
    CTN_CHECK_LAUNCH("ctn_unfold");
    return CTN_OK;
}

// general overlap_and_add(signal [Bn, frames, frame_length], frame_step) -> out [Bn, (frames-1)*frame_step + frame_length]
int ctn_overlap_add(const float* signal, float* out, int Bn, int frames, int frame_length, int frame_step, void* stream) {
    CTN_REQUIRE(signal && out, "ctn_overlap_add: null pointer");
    CTN_REQUIRE(Bn > 0 && frames > 0 && frame_length > 0 && frame_step > 0, "ctn_overlap_add: bad sizes");
    const int T = (frames - 1) * frame_step + frame_length;
    hipLaunchKernelGGL(ola_general_kernel, dim3(grid_for((long long)Bn * T)), dim3(NT), 0, (hipStream_t)stream, signal, out,
                       Bn, frames, frame_length, frame_step, T);
    CTN_CHECK_LAUNCH("ctn_overlap_add");
    return CTN_OK;
}

int ctn_overlap_add_bwd(const float* dout, float* dsignal, int Bn, int frames, int frame_length, int frame_step, void* stream) {
    CTN_REQUIRE(dout && dsignal, "ctn_overlap_add_bwd: null pointer");
    CTN_REQUIRE(Bn > 0 && frames > 0 && frame_length > 0 && frame_step > 0, "ctn_overlap_add_bwd: bad sizes");
    const int T = (frames - 1) * frame_step + frame_length;
    hipLaunchKernelGGL(ola_general_bwd_kernel, dim3(grid_for((long long)Bn * frames * frame_length)), dim3(NT), 0,
                       (hipStream_t)stream, dout, dsignal, Bn, frames, frame_length, frame_step, T);
    CTN_CHECK_LAUNCH("ctn_overlap_add_bwd");
    return CTN_OK;
}

}  // extern "C"
