// Pointwise (1x1) convolution GEMMs on fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces every nn.Conv1d(..., 1) of the reference (src/conv_tasnet.py:174,191,223,262)
// in forward, input-gradient and weight-gradient form.  Exact fp32 products and
// accumulation (bf16 inputs break the 1e-3 dB budget).
//
// Data layout: activations [M, Ch, Kp] fp32, frames fastest, Kp = K rounded up
// (multiple of 4); columns k in [K, Kp) hold exact zeros in every activation
// and gradient tensor (invariant kept by every kernel's store path).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each
// wave 64x64 = 2x2 MFMA tiles of 32x32, 64 accumulator VGPRs), contraction
// step 16 staged through LDS with register double-buffering.  One MFMA
// (64 cycles) consumes one A and one B dword per lane, so LDS bandwidth is
// never the bound; the fused prologue/epilogue work rides in the VALU shadow.
#include "ctn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int NT = 256;
constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDA = 132, LDB = 132;   // row stride (floats) of the LDS images; 16-B aligned rows

enum { PRO_NONE = 0, PRO_PRELU_NORM = 1 };
enum { EPI_NONE = 0, EPI_RESIDUAL = 1, EPI_PRELU_STATS = 2, EPI_GLN_BWD = 3, EPI_RELU = 4 };

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
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int TRANS_W, int PRO, int EPI>
__global__ __launch_bounds__(NT) void pw_gemm_kernel(PwArgs a) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
    __shared__ double red[NT / 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bid = blockIdx.x;
    const int rt = bid % a.tiles_r; bid /= a.tiles_r;
    const int ct = bid % a.tiles_c;
    const int m = bid / a.tiles_c;
    const int r0 = rt * BM, c0 = ct * BN;
    const float* __restrict__ Xm = a.X + (size_t)m * a.Cn * a.Kp;
    float* __restrict__ Om = a.Out + (size_t)m * a.R * a.Kp;

    float p_mean = 0.f, p_rstd = 1.f, p_alpha = 0.f;
    if constexpr (PRO == PRO_PRELU_NORM) {
        finalize_stats<NT>(a.pro_part + (size_t)m * a.pro_nparts * 2, a.pro_nparts,
                           (double)a.Cn * (double)a.K, red, p_mean, p_rstd);
        p_alpha = a.pro_alpha[0];
        if (a.pro_ms_out != nullptr && rt == 0 && ct == 0 && tid == 0) {
            a.pro_ms_out[2 * m] = p_mean;
            a.pro_ms_out[2 * m + 1] = p_rstd;
        }
    }

    // ---- global -> register staging maps -----------------------------------
    // A (weights): TRANS_W=0 reads W[r][c..c+3] (thread: c4 = tid&3, r = tid>>2 (+64));
    //              TRANS_W=1 reads W[c][r..r+3] (thread: r4 = tid&31, c = tid>>5 (+8)).
    // B (activations): X[i][k..k+3] (thread: k4 = tid&31, i = tid>>5 (+8)).
    float4 ra[2], rb[2];
    const int nk = (a.Cn + BK - 1) / BK;

    auto load_tile = [&](int kt) {
        const int kc = kt * BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (TRANS_W == 0) {
                const int r = r0 + (tid >> 2) + 64 * j, c = kc + (tid & 3) * 4;
                if (r < a.R && c < a.Cn) v = ld4(a.W + (size_t)r * a.Cn + c);
            } else {
                const int c = kc + (tid >> 5) + 8 * j, r = r0 + (tid & 31) * 4;
                if (c < a.Cn && r < a.R) v = ld4(a.W + (size_t)c * a.R + r);
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = kc + (tid >> 5) + 8 * j, k = c0 + (tid & 31) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < a.Cn && k < a.Kp) {
                v = ld4(Xm + (size_t)i * a.Kp + k);
                if constexpr (PRO == PRO_PRELU_NORM) {
                    const float g = a.pro_gamma[i], b = a.pro_beta[i];
                    v.x = (k + 0 < a.K) ? g * ((prelu_f(v.x, p_alpha) - p_mean) * p_rstd) + b : 0.f;
                    v.y = (k + 1 < a.K) ? g * ((prelu_f(v.y, p_alpha) - p_mean) * p_rstd) + b : 0.f;
                    v.z = (k + 2 < a.K) ? g * ((prelu_f(v.z, p_alpha) - p_mean) * p_rstd) + b : 0.f;
                    v.w = (k + 3 < a.K) ? g * ((prelu_f(v.w, p_alpha) - p_mean) * p_rstd) + b : 0.f;
                }
            }
            rb[j] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (TRANS_W == 0) {
                const int r = (tid >> 2) + 64 * j, c = (tid & 3) * 4;
                As[buf][c + 0][r] = ra[j].x;
                As[buf][c + 1][r] = ra[j].y;
                As[buf][c + 2][r] = ra[j].z;
                As[buf][c + 3][r] = ra[j].w;
            } else {
                const int c = (tid >> 5) + 8 * j, r = (tid & 31) * 4;
                *reinterpret_cast<float4*>(&As[buf][c][r]) = ra[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = (tid >> 5) + 8 * j, k = (tid & 31) * 4;
            *reinterpret_cast<float4*>(&Bs[buf][i][k]) = rb[j];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int kk = 2 * s + lhi;
            const float a0 = As[buf][kk][wm * 64 + l31];
            const float a1 = As[buf][kk][wm * 64 + 32 + l31];
            const float b0 = Bs[buf][kk][wn * 64 + l31];
            const float b1 = Bs[buf][kk][wn * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (e&3)+8*(e>>2)+4*(lane>>5) ----
    float e_alpha = 0.f, b_mean = 0.f, b_rstd = 1.f;
    if constexpr (EPI == EPI_PRELU_STATS) e_alpha = a.epi_alpha[0];
    if constexpr (EPI == EPI_GLN_BWD) {
        e_alpha = a.bwd_alpha[0];
        b_mean = a.bwd_ms[2 * m];
        b_rstd = a.bwd_ms[2 * m + 1];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = r0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
            float g = 0.f;
            if constexpr (EPI == EPI_GLN_BWD) g = (r < a.R) ? a.bwd_gamma[r] : 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int k = c0 + wn * 64 + nt * 32 + l31;
                float v = acc[mt][nt][e];
                const bool ok = (r < a.R) && (k < a.Kp);
                const size_t off = (size_t)r * a.Kp + k;
                if constexpr (EPI == EPI_RESIDUAL) {
                    if (ok) v += a.residual[(size_t)m * a.R * a.Kp + off];
                }
                if constexpr (EPI == EPI_RELU) v = fmaxf(v, 0.f);
                if constexpr (EPI == EPI_PRELU_STATS) {
                    const float p = prelu_f(v, e_alpha);
                    s1 += p;
                    s2 += p * p;
                }
                if constexpr (EPI == EPI_GLN_BWD) {
                    if (ok) {
                        const float y = a.bwd_y[(size_t)m * a.R * a.Kp + off];
                        const float xh = (prelu_f(y, e_alpha) - b_mean) * b_rstd;
                        const float t = g * v;
                        s1 += t;
                        s2 += t * xh;
                    }
                }
                if (ok) Om[off] = v;
            }
        }
    }
    if constexpr (EPI == EPI_PRELU_STATS || EPI == EPI_GLN_BWD) {
        const double d1 = block_sum<double, NT>((double)s1, red);
        const double d2 = block_sum<double, NT>((double)s2, red);
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
};

template <int PRO>
__global__ __launch_bounds__(NT) void pw_wgrad_kernel(WgArgs a) {
    __shared__ float As[2][BM][LDW];
    __shared__ float Bs[2][BN][LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bid = blockIdx.x;
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

    float4 ra[2], rb[2];
    const int nk = (ke - kb + WK - 1) / WK;
    auto load_tile = [&](int kt) {
        const int k = kb + kt * WK + (tid & 3) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (tid >> 2) + 64 * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + row < a.R && k < ke) v = ld4(Gm + (size_t)(r0 + row) * a.Kp + k);
            ra[j] = v;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c = c0 + row;
            if (c < a.Cn && k < ke) {
                x = ld4(Xm + (size_t)c * a.Kp + k);
                if constexpr (PRO == PRO_PRELU_NORM) {
                    const float g = a.pro_gamma[c], b = a.pro_beta[c];
                    x.x = (k + 0 < a.K) ? g * ((prelu_f(x.x, p_alpha) - p_mean) * p_rstd) + b : 0.f;
                    x.y = (k + 1 < a.K) ? g * ((prelu_f(x.y, p_alpha) - p_mean) * p_rstd) + b : 0.f;
                    x.z = (k + 2 < a.K) ? g * ((prelu_f(x.z, p_alpha) - p_mean) * p_rstd) + b : 0.f;
                    x.w = (k + 3 < a.K) ? g * ((prelu_f(x.w, p_alpha) - p_mean) * p_rstd) + b : 0.f;
                }
            }
            rb[j] = x;
        }
    };
    auto store_tile = [&](int buf) {
        const int kq = (tid & 3) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (tid >> 2) + 64 * j;
            As[buf][row][kq + 0] = ra[j].x; As[buf][row][kq + 1] = ra[j].y;
            As[buf][row][kq + 2] = ra[j].z; As[buf][row][kq + 3] = ra[j].w;
            Bs[buf][row][kq + 0] = rb[j].x; Bs[buf][row][kq + 1] = rb[j].y;
            Bs[buf][row][kq + 2] = rb[j].z; Bs[buf][row][kq + 3] = rb[j].w;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lhi = lane >> 5;
    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int s = 0; s < WK / 2; ++s) {
            const int kk = 2 * s + lhi;
            const float a0 = As[buf][wm * 64 + l31][kk];
            const float a1 = As[buf][wm * 64 + 32 + l31][kk];
            const float b0 = Bs[buf][wn * 64 + l31][kk];
            const float b1 = Bs[buf][wn * 64 + 32 + l31][kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }
    float* __restrict__ S = a.slab + (size_t)sp * a.R * a.Cn;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = r0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int c = c0 + wn * 64 + nt * 32 + l31;
                if (r < a.R && c < a.Cn) S[(size_t)r * a.Cn + c] = acc[mt][nt][e];
            }
        }
}

// out[i] = sum_s slab[s][i], fixed order.
__global__ __launch_bounds__(NT) void slab_reduce_kernel(const float* __restrict__ slab, int nsplit, long long n,
                                                         float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += slab[(size_t)k * n + i];
    out[i] = s;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_common(const char* fn, const float* W, const float* X, const float* Out, int M, int R, int Cn, int K, int Kp) {
    CTN_REQUIRE(W && X && Out, "%s: null pointer", fn);
    CTN_REQUIRE(M > 0 && R > 0 && Cn > 0 && K > 0 && Kp >= K, "%s: bad sizes M=%d R=%d Cn=%d K=%d Kp=%d", fn, M, R, Cn, K, Kp);
    CTN_REQUIRE(Kp % 4 == 0 && R % 4 == 0 && Cn % 4 == 0, "%s: Kp, rows and contraction must be multiples of 4 (Kp=%d R=%d Cn=%d)", fn, Kp, R, Cn);
    CTN_REQUIRE(aligned16(W) && aligned16(X) && aligned16(Out), "%s: pointers must be 16-byte aligned", fn);
    return CTN_OK;
}

}  // namespace

extern "C" {

int ctn_pw_stats_parts(int R, int Kp) { return ctn_cdiv(R, BM) * ctn_cdiv(Kp, BN); }

// Out[m] = op(W) . f(X[m]) (+ residual) ; see include/ctn_hip.h
int ctn_pw_gemm(const float* W, const float* X, float* Out, int M, int R, int Cn, int K, int Kp, int trans_w,
                const double* pro_part, int pro_nparts, const float* pro_gamma, const float* pro_beta,
                const float* pro_alpha, float* pro_ms_out,
                const float* residual, const float* epi_alpha, double* epi_part, int relu_out, void* stream) {
    int rc = check_common("ctn_pw_gemm", W, X, Out, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(!relu_out || !(residual || epi_part || pro_part || trans_w), "ctn_pw_gemm: relu_out only on the plain forward GEMM");
    CTN_REQUIRE(!(residual && epi_part), "ctn_pw_gemm: residual and stats epilogues are exclusive");
    CTN_REQUIRE(!pro_part || (pro_gamma && pro_beta && pro_alpha && pro_nparts > 0), "ctn_pw_gemm: incomplete prologue arguments");
    CTN_REQUIRE(!epi_part || epi_alpha, "ctn_pw_gemm: stats epilogue needs alpha");
    CTN_REQUIRE(!(trans_w && (pro_part || epi_part)), "ctn_pw_gemm: fused prologue/stats only with trans_w=0");
    PwArgs a{};
    a.W = W; a.X = X; a.Out = Out; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.tiles_r = ctn_cdiv(R, BM); a.tiles_c = ctn_cdiv(Kp, BN);
    a.pro_part = pro_part; a.pro_nparts = pro_nparts; a.pro_gamma = pro_gamma; a.pro_beta = pro_beta;
    a.pro_alpha = pro_alpha; a.pro_ms_out = pro_ms_out;
    a.residual = residual; a.epi_alpha = epi_alpha; a.epi_part = epi_part;
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * M)), block(NT);
    hipStream_t st = (hipStream_t)stream;
    if (trans_w) {
        if (residual) hipLaunchKernelGGL((pw_gemm_kernel<1, PRO_NONE, EPI_RESIDUAL>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_gemm_kernel<1, PRO_NONE, EPI_NONE>), grid, block, 0, st, a);
    } else if (pro_part) {
        if (residual) hipLaunchKernelGGL((pw_gemm_kernel<0, PRO_PRELU_NORM, EPI_RESIDUAL>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((pw_gemm_kernel<0, PRO_PRELU_NORM, EPI_NONE>), grid, block, 0, st, a);
    } else if (epi_part) {
        hipLaunchKernelGGL((pw_gemm_kernel<0, PRO_NONE, EPI_PRELU_STATS>), grid, block, 0, st, a);
    } else if (residual) {
        hipLaunchKernelGGL((pw_gemm_kernel<0, PRO_NONE, EPI_RESIDUAL>), grid, block, 0, st, a);
    } else if (relu_out) {
        hipLaunchKernelGGL((pw_gemm_kernel<0, PRO_NONE, EPI_RELU>), grid, block, 0, st, a);
    } else {
        hipLaunchKernelGGL((pw_gemm_kernel<0, PRO_NONE, EPI_NONE>), grid, block, 0, st, a);
    }
    CTN_CHECK_LAUNCH("ctn_pw_gemm");
    return CTN_OK;
}

// dN[m] = W^T . dOut[m]  (W stored [Cn=O_fwd, R=I_fwd]) plus the two per-utterance sums that
// gLN backward needs:  S1 = sum gamma*dN, S2 = sum gamma*dN*xhat, xhat = (prelu(y)-mean)*rstd.
int ctn_pw_dgrad_gln(const float* W, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                     const float* y, const float* gamma, const float* alpha, const float* ms, double* sums_part,
                     void* stream) {
    int rc = check_common("ctn_pw_dgrad_gln", W, dOut, dN, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(y && gamma && alpha && ms && sums_part, "ctn_pw_dgrad_gln: null pointer");
    PwArgs a{};
    a.W = W; a.X = dOut; a.Out = dN; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.tiles_r = ctn_cdiv(R, BM); a.tiles_c = ctn_cdiv(Kp, BN);
    a.bwd_y = y; a.bwd_gamma = gamma; a.bwd_alpha = alpha; a.bwd_ms = ms; a.bwd_part = sums_part;
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * M)), block(NT);
    hipLaunchKernelGGL((pw_gemm_kernel<1, PRO_NONE, EPI_GLN_BWD>), grid, block, 0, (hipStream_t)stream, a);
    CTN_CHECK_LAUNCH("ctn_pw_dgrad_gln");
    return CTN_OK;
}

static void wgrad_plan(int M, int R, int Cn, int Kp, int* chunk, int* chunks_per_m) {
    const int tiles = ctn_cdiv(R, BM) * ctn_cdiv(Cn, BN);
    int cpm = ctn_cdiv(512, tiles * M);            // aim for ~2 workgroups per CU
    const int max_cpm = ctn_cdiv(Kp, 256);         // but keep >= 256 frames of contraction per slab
    if (cpm > max_cpm) cpm = max_cpm;
    if (cpm < 1) cpm = 1;
    int c = ctn_cdiv(ctn_cdiv(Kp, cpm), WK) * WK;
    *chunk = c;
    *chunks_per_m = ctn_cdiv(Kp, c);
}

size_t ctn_pw_wgrad_workspace(int M, int R, int Cn, int Kp) {
    int chunk, cpm;
    wgrad_plan(M, R, Cn, Kp, &chunk, &cpm);
    return (size_t)M * cpm * R * Cn * sizeof(float);
}

// dW[R,Cn] = sum_{m,k} dOut[m,r,k] * f(X[m,c,k])
int ctn_pw_wgrad(const float* dOut, const float* X, float* dW, int M, int R, int Cn, int K, int Kp,
                 const float* pro_gamma, const float* pro_beta, const float* pro_alpha, const float* pro_ms,
                 void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common("ctn_pw_wgrad", dW, X, (const float*)dOut, M, R, Cn, K, Kp);
    if (rc) return rc;
    CTN_REQUIRE(!pro_ms || (pro_gamma && pro_beta && pro_alpha), "ctn_pw_wgrad: incomplete prologue arguments");
    WgArgs a{};
    a.dOut = dOut; a.X = X; a.slab = (float*)workspace; a.M = M; a.R = R; a.Cn = Cn; a.K = K; a.Kp = Kp;
    a.tiles_r = ctn_cdiv(R, BM); a.tiles_c = ctn_cdiv(Cn, BN);
    wgrad_plan(M, R, Cn, Kp, &a.chunk, &a.chunks_per_m);
    const int nsplit = M * a.chunks_per_m;
    if (workspace == nullptr || workspace_bytes < (size_t)nsplit * R * Cn * sizeof(float)) {
        ctn_set_error("ctn_pw_wgrad: workspace too small (%zu < %zu)", workspace_bytes, (size_t)nsplit * R * Cn * sizeof(float));
        return CTN_ERR_WORKSPACE;
    }
    a.pro_gamma = pro_gamma; a.pro_beta = pro_beta; a.pro_alpha = pro_alpha; a.pro_ms = pro_ms;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((long long)a.tiles_r * a.tiles_c * nsplit)), block(NT);
    if (pro_ms) hipLaunchKernelGGL((pw_wgrad_kernel<PRO_PRELU_NORM>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pw_wgrad_kernel<PRO_NONE>), grid, block, 0, st, a);
    CTN_CHECK_LAUNCH("ctn_pw_wgrad");
    const long long n = (long long)R * Cn;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ctn_cdivll(n, NT)), block, 0, st, a.slab, nsplit, n, dW);
    CTN_CHECK_LAUNCH("ctn_pw_wgrad/reduce");
    return CTN_OK;
}

}  // extern "C"
