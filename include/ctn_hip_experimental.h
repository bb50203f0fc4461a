/* EXPERIMENTAL entry points of libctn_hip.so -- present only in a library built with CTN_BUILD_X6=1
 * (conv-tasnet_amd/csrc/experimental/ctn_gemm_x6.hip) and bound by Python only under CTN_EXPERIMENTAL=1.
 * Not part of the product surface: round 1 measured that the split-bf16 GEMMs do not beat the fp32-MFMA kernels at
 * the paper shapes (profiles/README.md).  Same conventions as include/ctn_hip.h. */
#ifndef CTN_HIP_EXPERIMENTAL_H
#define CTN_HIP_EXPERIMENTAL_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- split-bf16 ("x6") forms of the same GEMMs ------------------------------------------------
 * fp32-accurate products on the bf16 matrix cores: every fp32 operand is split exactly into three bf16 pieces and the
 * six piece-products of weight >= 2^-16 are accumulated in fp32 (dropped terms <= 2^-24 |a.b|).  Same contracts as the
 * fp32-MFMA entry points above; the weights are passed as pre-split planes made by ctn_split_bf16. */
int ctn_split_cols(int Cn);                 /* contraction length padded to the kernels' k-tile (32) */
/* planes: [3][R][ctn_split_cols(Cn)] bf16 with (R, Cn) = transpose ? (cols, rows) : (rows, cols); W is [rows, cols].
 * transpose = 1 prepares the input-gradient (W^T) form. */
int ctn_split_bf16(const float* W, void* planes, int rows, int cols, int transpose, void* stream);
int ctn_pw_gemm_x6(const void* Wp, const float* X, float* Out, int M, int R, int Cn, int K, int Kp,
                   const double* pro_part, int pro_nparts, const float* pro_gamma, const float* pro_beta,
                   const float* pro_alpha, float* pro_ms_out,
                   const float* residual, const float* epi_alpha, double* epi_part, int relu_out, void* stream);
int ctn_pw_dgrad_gln_x6(const void* Wp, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                        const float* y, const float* gamma, const float* alpha, const float* ms, double* sums_part,
                        void* stream);
int ctn_pw_wgrad_x6(const float* dOut, const float* X, float* dW, int M, int R, int Cn, int K, int Kp,
                    const float* pro_gamma, const float* pro_beta, const float* pro_alpha, const float* pro_ms,
                    void* workspace, size_t workspace_bytes, void* stream);
size_t ctn_pw_wgrad_x6_workspace(int M, int R, int Cn, int Kp);
/* "p6": both operands already split in HBM.  ctn_split_act is the stand-alone form of what the producers' epilogues
 * emit (planes [3][n] bf16).  ctn_pw_gemm_p6: Out[m] = Wp(m) . Xp[m] (+ row_bias[m,r] for k < K) (+ residual), stored
 * as fp32 (Out) and / or as bf16 planes (out_planes [3][M,R,Kp]); Wp [3][R][Cnp], per utterance ([M][3][R][Cnp]) when
 * w_per_m != 0; one of the residual / PReLU-statistics / ReLU / gLN-backward epilogues as in the fp32 entry points. */
int ctn_split_act(const float* X, void* planes, long long n, void* stream);
int ctn_pw_gemm_p6(const void* Wp, int w_per_m, const void* Xp, float* Out, void* out_planes, int M, int R, int Cn,
                   int K, int Kp, const float* row_bias, const float* residual, const float* epi_alpha, double* epi_part,
                   int relu_out, const float* bwd_y, const float* bwd_gamma, const float* bwd_alpha, const float* bwd_ms,
                   double* bwd_part, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTN_HIP_EXPERIMENTAL_H */
