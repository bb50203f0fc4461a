/* libctn_hip.so -- C ABI of the MI355X (gfx950) Conv-TasNet hot path.
 *
 * The reference (OfekCohen1/Conv-TasNet) has no FFI of its own: its hot path is torch.nn
 * modules.  This header is the boundary a maintainer binds instead (ctypes
 * stub in INTEGRATION.md); each entry point names the reference code it replaces, paths
 * relative to the reference root.
 *
 * Conventions
 *  - plain pointers + sizes only; every pointer is DEVICE memory unless marked "host".
 *  - the caller owns and allocates every buffer, workspaces included; the library never
 *    allocates, frees or retains a pointer, and never synchronises the device.
 *  - every launch goes to `stream` (a hipStream_t passed as void*; NULL = default stream).
 *  - return 0 on success, <0 on error (CTN_ERR_*); ctn_last_error() gives the message
 *    for the calling thread.  Nothing throws across the boundary.
 *  - activations are fp32 [M, Ch, Kp], frames fastest; Kp = ctn_padded_frames(K); columns
 *    k in [K, Kp) hold exact zeros in every activation / gradient tensor (every kernel keeps
 *    that invariant).  Channel counts must be multiples of 4; tensors 16-byte aligned.
 *  - all reductions have a fixed order: same inputs -> bitwise same outputs.
 */
#ifndef CTN_HIP_H
#define CTN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTN_OK 0
#define CTN_ERR_ARG (-1)
#define CTN_ERR_LAUNCH (-2)
#define CTN_ERR_WORKSPACE (-3)
#define CTN_AMAX_SLOTS 64      /* words per utterance of a tracked-maximum array (h3 arithmetic, below) */

int ctn_version(void);
const char* ctn_last_error(void);
int ctn_padded_frames(int K);               /* K rounded up to a multiple of 64 */
/* Orders two HIP streams of the current device: work enqueued on `from` so far completes before work enqueued on `to`
 * after the call.  Device-scope event (no timing, no system-scope fence): the cheap form of
 * torch.cuda.Stream.wait_stream for the weight-gradient stream of the backward pass. */
int ctn_stream_order(void* from, void* to);

/* ---- 1x1 convolutions = GEMMs on the matrix cores, fp32-faithful ------------------------
 * replaces nn.Conv1d(*, *, 1, bias=False): src/conv_tasnet.py:174 (bottleneck), :191 (mask),
 * :223 and :262 (TemporalBlock), and the Linear of the decoder (:128,:143); autograd's
 * convolution_backward for the same layers. */

/* Out[m] = op(W) . f(X[m])  (+ residual[m]),   X:[M,Cn,Kp]  Out:[M,R,Kp]
 *   trans_w = 0: W is [R,Cn] (forward);  trans_w = 1: W is [Cn,R] (input gradient, or forward on a transposed copy:
 *     the fast form of the fp32-MFMA arithmetic -- ctn_transpose_batch);  trans_w = 2: W is a block of pre-split bf16 pieces
 *     from ctn_split_b3_batch (b6 arithmetic, R >= 64: the fast form there).
 *   pro_part != NULL: f(x)[i,k] = gamma[i]*((prelu(x,alpha)-mean_m)*rstd_m)+beta[i]
 *     for k < K, 0 otherwise; (mean_m, rstd_m) are finalised from the [M, pro_nparts, 2] fp64
 *     (sum, sum of squares) partials of prelu(x) -- global LayerNorm, src/conv_tasnet.py:358-360 --
 *     and written to pro_ms_out [M,2] when that is non-NULL.
 *   residual != NULL: TemporalBlock's "out + residual" (src/conv_tasnet.py:243).
 *   epi_part != NULL: also emits the partials of prelu(Out, epi_alpha) for the NEXT gLN,
 *     layout [M, ctn_pw_stats_parts(M,R,Kp), 2] fp64.
 *   relu_out != 0: Out = relu(.)  (encoder, src/conv_tasnet.py:120). */
int ctn_pw_gemm(const float* W, const float* X, float* Out, int M, int R, int Cn, int K, int Kp, int trans_w,
                const double* pro_part, int pro_nparts, const float* pro_gamma, const float* pro_beta,
                const float* pro_alpha, float* pro_ms_out,
                const float* residual, const float* epi_alpha, double* epi_part, int relu_out, void* stream);
int ctn_pw_stats_parts(int M, int R, int Kp);

/* dst[i] = src[i]^T for n equally shaped [rows, cols] fp32 matrices (HOST arrays of device pointers): the forward pass
 * keeps a [I, O] copy of every 1x1 weight [O, I] so that ctn_pw_gemm(trans_w = 1) -- 16-byte row writes into LDS, no
 * transposing scatter -- serves forward and input gradient alike (src/conv_tasnet.py:223,262). */
int ctn_transpose_batch(const void* const* src, void* const* dst, int n, int rows, int cols, void* stream);

/* dN[m] = W^T . dOut[m]   (W:[Cn,R] as stored by the forward layer, dOut:[M,Cn,Kp], dN:[M,R,Kp])
 * and, fused, the two sums gLN backward needs per utterance, as partials
 * sums_part [M, ctn_pw_stats_parts(M,R,Kp), 2] fp64:  S1 = sum gamma*dN,  S2 = sum gamma*dN*xhat,
 * xhat = (prelu(y,alpha)-ms[m][0])*ms[m][1],  y:[M,R,Kp] the pre-activation input of that norm. */
int ctn_pw_dgrad_gln(const float* W, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                     const float* y, const float* gamma, const float* alpha, const float* ms, double* sums_part,
                     void* stream);

/* dW[R,Cn] = sum_{m,k} dOut[m,r,k] * f(X[m,c,k]);  f as above when pro_ms ([M,2]) != NULL.
 * workspace: ctn_pw_wgrad_workspace() bytes of split-K slabs, summed in a fixed order. */
int ctn_pw_wgrad(const float* dOut, const float* X, float* dW, int M, int R, int Cn, int K, int Kp,
                 const float* pro_gamma, const float* pro_beta, const float* pro_alpha, const float* pro_ms,
                 void* workspace, size_t workspace_bytes, void* stream);
size_t ctn_pw_wgrad_workspace(int M, int R, int Cn, int Kp);
/* Library switches (process-global; the library is driven by ONE host thread at a time: set them from that thread, between
 * steps -- workspace sizes and statistics layouts depend on them).  Keys: "arith" (below); "b3_tile" / "b3_tile_k3" 0|1|2|3 =
 * 128x128 / 128x64 / 256x64 / 256x64 on 8 waves: tile of the split forward / input-gradient kernels (the prologue + residual form has its
 * own); "b3_wgrad_blocks", "wgrad_blocks": target workgroups per weight-gradient launch (split-bf16 / fp32 MFMA);
 * "pw_tile" -1|0..3: tile of the fp32-MFMA forward kernel (also CTN_PW_TILE); "b3_ws" 0|1 (default 0): run the h3 forward /
 * input-gradient GEMMs on the wave-specialised persistent kernel (csrc/ctn_gemm_ws.h; same values, measured slower: kept as a tested
 * experiment), "b3_ws_blocks": its workgroup count; "cln_fr" 16|32: frames per workgroup of the channel-wise LayerNorm backward
 * kernel (16: three 256-thread workgroups per CU; changes ctn_cln_bwd_blocks()); "cln_lean" 0|1 (default 1): ctn_cln_bwd at 512 channels with PReLU and without `add` /
 * `relu_ref` runs a kernel specialised for that form (same bits, 11 % faster alone); "wgrad_chain" 0|1 (default 0): inside the
 * composite stacks the split-K slabs of a weight gradient are summed by the NEXT weight-gradient launch of the stream instead of a
 * slab_reduce launch of their own (same addition order, same bits; measured equal in the step); "cln_fuse" 0|1|2 (default 2): 1 = the
 * composite cLN stacks run the second norm's backward inside the input-gradient GEMM's epilogue and the depthwise backward
 * (ctn_pw_dgrad_cln / ctn_cln_bwd_frame / ctn_dw_bwd_cln) instead of as a ctn_cln_bwd pass; 2 = also the first norm's forward
 * inside the first 1x1 conv's epilogue and the depthwise kernel's prologue (ctn_pw_gemm_cln / ctn_cln_stats_frame / ctn_dw_fwd_cln:
 * n1s is then neither written nor read; set it between steps, forward and backward under the same value; CTN_CLN_FUSE=0|1|2 at first
 * use); ctn_cln_fuse() reads it;
 * "bwd_events" 0|1|2 (default 0 = gLN stacks 1, cLN stacks 2; CTN_BWD_EVENTS=1|2 at first use): forks of the weight-gradient stream
 * per block of the composite backward passes -- 2: dW2 behind B1 and dW1 behind the norm backward; 1: one fork behind B5 (dW1 and the
 * sums of that block, then dW2 of the next block, which needs only that B5's output); same gradients bit for bit; "gln_fuse" 0|1
 * (default 0; CTN_GLN_FUSE at first use; ctn_gln_fuse() reads it): 1 = the composite gLN stacks run without the gLN-1' / PReLU-1' pass
 * (ctn_pw_dgrad_gln2 + ctn_dw_bwd_gln2 instead of ctn_pw_dgrad_gln + ctn_dw_bwd + ctn_gln_prelu_bwd): three tensor passes of 20 less, same
 * gradients to fp32 rounding -- and 1.7 % SLOWER in the step (profiles/README.md r04_m): a tested option.  Defaults are the
 * measured best. */
int ctn_tune(const char* key, int value);
int ctn_cln_fuse(void);
int ctn_gln_fuse(void);
/* Arithmetic of the 1x1-convolution GEMMs (ctn_pw_gemm, ctn_pw_dgrad_gln, ctn_pw_wgrad and the composites over them):
 *   3 = "h3" (default): the GEMMs of the composite stacks (ctn_tcn_*) run on the ctn_*_h3 entry points below -- two fp16 pieces
 *       per fp32 operand under a tracked power-of-two scale, three f16 MFMAs, fp32 accumulation; every other GEMM as b6;
 *   2 = "b6": every fp32 operand is split EXACTLY into three bf16 pieces (a = a0 + a1 + a2, round-to-nearest-even)
 *       and a.b is formed from the six piece-products of weight >= 2^-17 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation;
 *       the dropped terms are <= 2^-23 |a.b| -- one fp32 rounding of the product.  Measured against fp64 the error of every
 *       GEMM form is that of the fp32 MFMA (3.4e-7 vs 4.0e-7 of sum |a||b|, profiles/r03_a_b6_check.txt);
 *   0 = fp32 MFMA (v_mfma_f32_32x32x2_f32), bit-exact fp32 FMA chains;
 *   (1 was round 2's two-piece bf16 "b3", ~16-bit products: removed -- h3 costs the same three MFMAs at reference precision.)
 * Selected by CTN_GEMM_ARITH=h3|b6|fp32 at first use or ctn_tune("arith", 3|2|0) between steps; layers with fewer than 64
 * output rows (and weight gradients with a side below 32) always use the fp32-MFMA kernels. */
int ctn_gemm_arith(void);
/* b6 arithmetic (also the plain entry points under h3): the weight operand pre-split once per step.  dst[i] receives the bf16 pieces of the GEMM operand
 * A [R, Cn] (rows = output channels of THAT GEMM, Cn = its contraction) in MFMA fragment order, zero-filled to multiples
 * of 32: ctn_split_b3_bytes(R, Cn) bytes each (three pieces), 16-byte aligned.  k_major = 0: src[i] is
 * stored [R, Cn] (forward layers); k_major = 1: src[i] is stored [Cn, R] and used transposed (input gradients of the same
 * layers).  HOST arrays of device pointers, any n.  ctn_pw_gemm(trans_w = 2) and ctn_pw_dgrad_gln_planes take such a block as
 * W: no conversion work and no LDS traffic for the weights inside the GEMM; results are bitwise those of the fp32-weight forms. */
size_t ctn_split_b3_bytes(int R, int Cn);
int ctn_split_b3_batch(const void* const* src, void* const* dst, int n, int R, int Cn, int k_major, void* stream);
int ctn_pw_dgrad_gln_planes(const void* Wp, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                            const float* y, const float* gamma, const float* alpha, const float* ms, double* sums_part,
                            void* stream);

/* ---- "h3" arithmetic: fp32-faithful products from TWO fp16 pieces per operand (three f16 MFMAs instead of b6's six) ------
 * replaces the same fp32 nn.Conv1d(*, *, 1) layers (src/conv_tasnet.py:223,262) inside the composite stacks.
 *   a s = a0 + a1 + r,  a0 = fp16_rne(a s), a1 = fp16_rne(a s - a0):  two 11-bit significands + the sign of a1 hold 22-23 bits,
 *   |a1| <= 2^-11 |a s|, |r| <= 2^-23 |a s|;   a.b ~= (a1.b0 + a0.b1 + a0.b0) / (s_a s_b)  on v_mfma_f32_32x32x16_f16, fp32
 *   accumulation; dropped per product: a1.b1 + r_a.b + a.r_b, <= 8 * 2^-24 |a.b| in the worst case (4.8e-7), 1.2 * 2^-24 rms (b6
 *   drops 2 * 2^-24).  A single product is thus NOT better than an fp32 one; the GEMM is, because the f16 MFMA rounds its fp32
 *   accumulator once per 16-deep step where the fp32 MFMA rounds it every 2-deep step (measured rms error of every form 1.3e-8
 *   of sum |a||b| against 2.8e-8, max 1.5e-7 against 3.8e-7).  Coherent worst case (all operands positive, each just below a
 *   rounding midpoint): 2.4e-7 .. 4.8e-7 of sum |a||b|, tested (tests/test_gpu_h3.py).
 * fp16 has 5 exponent bits, so every operand is brought into range by an exact power-of-two scale s derived from a bound on
 * its magnitude: the weight's own maximum (computed by ctn_split_h3_batch), and for activations the per-utterance maximum
 * max |X[m]| TRACKED BY THE KERNEL THAT PRODUCED X -- `amax` arrays: [M][CTN_AMAX_SLOTS] unsigned, bit patterns of non-negative
 * floats; a producer workgroup merges its maximum into one of an utterance's slots with an atomic max (exact, order-free: results
 * stay bitwise reproducible; 64 slots because 1000 atomics on one address cost a producer 5-9 us), a consumer takes the maximum
 * over the slots; the caller zeroes the array before the producer runs.  Producers: ctn_absmax_rows (any tensor), ctn_pw_gemm_h3 (residual epilogue, out_amax), ctn_dw_fwd and
 * ctn_gln_prelu_bwd (amax_out).  With an operand prologue the scale comes from the bound
 * max|gamma| * rstd * (max(1,|alpha|) * amax + |mean|) + max|beta|  (pro_gbmax = {max|gamma|, max|beta|}: ctn_absmax_batch).
 * Scaled values stay below 2^15 (fp16 holds 65504).  The bounds above hold for every element down to 2^-27 of its operand's bound
 * (the low piece is stored times 2^11 and accumulated separately, so it stays a normal fp16 number); smaller elements lose RELATIVE
 * precision gracefully (absolute error <= 2^-50 of the bound).  The scale is per UTTERANCE (per weight matrix): an element far below
 * its utterance's maximum is covered by that window, not by a scale of its own.  Same tiles, statistics
 * layouts (ctn_pw_stats_parts of the split arithmetics), epilogues and fixed-order reductions as the plain entry points.
 * These entry points do not depend on ctn_tune("arith"); R >= 64 (weight gradient: both sides >= 32). */
size_t ctn_split_h3_bytes(int R, int Cn);
int ctn_split_h3_batch(const void* const* src, void* const* dst, int n, int R, int Cn, int k_major, void* stream);
/* dst[i][0] = bit pattern of max |src[i][0 .. len)|  (HOST arrays of device pointers; dst[i]: 4 bytes) */
int ctn_absmax_batch(const void* const* src, void* const* dst, int n, int len, void* stream);
/* slots of amax[m] <- bits of max |x[m][0 .. n)|,  x: [M, n] fp32, n % 4 == 0, amax: [M][CTN_AMAX_SLOTS] */
int ctn_absmax_rows(const float* x, int M, long long n, unsigned* amax, void* stream);
/* ctn_pw_gemm(trans_w = 2) on h3 pieces.  x_amax: maximum of X as stored; pro_gbmax with the prologue; out_amax
 * (optional, with residual): receives the maximum of Out. */
int ctn_pw_gemm_h3(const void* Wp, const float* X, float* Out, int M, int R, int Cn, int K, int Kp,
                   const double* pro_part, int pro_nparts, const float* pro_gamma, const float* pro_beta,
                   const float* pro_alpha, float* pro_ms_out, const float* residual, const float* epi_alpha, double* epi_part,
                   const unsigned* x_amax, const float* pro_gbmax, unsigned* out_amax, void* stream);
int ctn_pw_dgrad_gln_h3(const void* Wp, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                        const float* y, const float* gamma, const float* alpha, const float* ms, double* sums_part,
                        const unsigned* g_amax, void* stream);
int ctn_pw_wgrad_h3(const float* dOut, const float* X, float* dW, int M, int R, int Cn, int K, int Kp,
                    const float* pro_gamma, const float* pro_beta, const float* pro_alpha, const float* pro_ms,
                    const unsigned* g_amax, const unsigned* x_amax, const float* pro_gbmax,
                    void* workspace, size_t workspace_bytes, void* stream);
size_t ctn_pw_wgrad_h3_workspace(int M, int R, int Cn, int Kp);

/* ---- depthwise dilated conv (+ fused PReLU / gLN) ---------------------------------------
 * replaces DepthwiseSeparableConv.net[0] (+Chomp1d), src/conv_tasnet.py:253-256,281-295, with the
 * PReLU (:224,:259) and GlobalLayerNorm (:225,:260,:338-361) on either side fused in. */

/* Z[m,h,k] = sum_j D[h,j] * n[m,h,k + j*dilation - pad_left],  zeros outside [0,K);
 *   pad_left = (P-1)*dilation/2 (non-causal "same") or (P-1)*dilation (causal, = pad + Chomp1d).
 *   pro_part != NULL: n = gLN(prelu(Y)) as in ctn_pw_gemm; else n = Y.
 *   epi_part != NULL: partials of prelu(Z, epi_alpha), layout [M, H, 2] fp64;
 *   amax_out != NULL (with epi_part): [M][CTN_AMAX_SLOTS], receives max |Z[m]| -- see the h3 section. */
int ctn_dw_fwd(const float* Y, float* Z, const float* D, int M, int H, int K, int Kp, int P, int dilation, int causal,
               const double* pro_part, int pro_nparts, const float* pro_gamma, const float* pro_beta,
               const float* pro_alpha, float* pro_ms_out, const float* epi_alpha, double* epi_part, unsigned* amax_out,
               void* stream);

/* Backward of the above.  fused = 1 walks  gLN2 <- PReLU2 <- depthwise <- (gLN1 output)  in one pass:
 *   in : dN2 (grad of gLN2's output), Dz (= Z of the forward), Y1 (= Y of the forward),
 *        (g1,b1,a1,ms1) / (g2,a2,ms2) of the two norms, sums2_part from ctn_pw_dgrad_gln
 *   out: dN1 (grad of gLN1's output), sums1_part [M,H,2] fp64 (S1,S2 for gLN1's backward) and
 *        pc [F, M, H] per-(utterance, channel) partials, F = ctn_dw_bwd_rows(P, fused):
 *        rows 0..P-1 dD taps; fused adds P: dgamma2, P+1: dbeta2, P+2: dgamma1, P+3: dbeta1, P+4: dalpha2.
 * fused = 0: dN2 = dZ, Y1 = the forward input; outputs dN1 = dY and pc rows 0..P-1.
 * (fused = 2 is the channel-wise LayerNorm form behind ctn_dw_bwd_cln below: call that.) */
int ctn_dw_bwd(const float* dN2, const float* Dz, const float* Y1, float* dN1, const float* D,
               int M, int H, int K, int Kp, int P, int dilation, int causal, int fused,
               const float* g1, const float* b1, const float* a1, const float* ms1,
               const float* g2, const float* a2, const float* ms2,
               const double* sums2_part, int sums2_nparts, float* pc, double* sums1_part, void* stream);
int ctn_dw_bwd_rows(int P, int fused);
/* Fixed-order finish of the fused backward's partials: pc [P+5,M,H] -> dD [H,P] (the depthwise weight's layout),
 * dgamma2, dbeta2, dgamma1, dbeta1 [H] and dalpha2 [1]; when dalpha1_part (the [n_dalpha1] per-row partials of
 * ctn_gln_prelu_bwd) is non-NULL the same launch also sums dalpha1 [1].  Destinations may be views into a flat
 * gradient buffer. */
int ctn_dw_bwd_finalize(const float* pc, int P, int M, int H, float* dD, float* dgamma2, float* dbeta2,
                        float* dgamma1, float* dbeta1, float* dalpha2, const float* dalpha1_part, int n_dalpha1,
                        float* dalpha1, void* stream);

/* gLN block WITHOUT the gLN-1' / PReLU-1' pass (round 4; GlobalLayerNorm backward of src/conv_tasnet.py:338-361 inside a TemporalBlock,
 * :223-225).  The first norm's backward needs S1' = sum gamma1 dN1 and S2' = sum gamma1 dN1 xhat1 over the utterance, dN1 being the
 * depthwise conv's input gradient.  The conv's adjoint moves both sums onto its OUTPUT gradient dd:
 *     S1' = sum_{c,k} dd[c,k] gamma1[c] V[c,k],   S2' = sum_{c,k} dd[c,k] (d[c,k] - beta1[c] V[c,k])
 * (d = the forward depthwise output = dw(gamma1 xhat1 + beta1), V[c,k] = sum of the taps of frame k that stay inside [0, K)), and
 * dd = prelu'(d) rstd2 (gamma2 dN2 - c1 - xhat2 c2) is affine in the second norm's (c1, c2) -- so six more per-utterance sums of
 * quantities that the second 1x1 conv's input-gradient GEMM already holds in its epilogue give S1', S2' BEFORE the depthwise
 * backward runs:   ctn_pw_dgrad_gln2 = ctn_pw_dgrad_gln with sums_part [M, ctn_pw_stats_parts(M,R,Kp), 8]
 *     (S1, S2, sum u t g1V, sum u g1V, sum u xh2 g1V, sum u t e, sum u e, sum u xh2 e;  u = prelu'(d), t = gamma2 dN, g1V = gamma1 V,
 *      e = d - beta1 V; W / w_form as ctn_pw_dgrad_cln; gamma1, beta1 [R], D [R,P], P, dilation, causal: the depthwise conv),
 * and ctn_dw_bwd_gln2 = ctn_dw_bwd(fused = 1) that applies the first norm's backward to its result on the fly: it writes
 *     dY1 = rstd1 (gamma1 dN1 - S1'/n - xhat1 S2'/n) prelu'(Y1)   instead of dN1,
 * the dalpha1 partials as row P+5 of pc [P+6, M, H] (ctn_dw_bwd_rows(P, 3); finish with ctn_dw_bwd_finalize, dalpha1_part = that
 * row) and, with amax_out != NULL, max |dY1[m]| (h3 section).  ctn_gln_prelu_bwd (three tensor passes) is not needed then. */
int ctn_pw_dgrad_gln2(const void* W, int w_form, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                      const float* y, const float* gamma, const float* alpha, const float* ms,
                      const float* gamma1, const float* beta1, const float* D, int P, int dilation, int causal,
                      double* sums_part, const unsigned* g_amax, void* stream);
int ctn_dw_bwd_gln2(const float* dN2, const float* Dz, const float* Y1, float* dY1, const float* D,
                    int M, int H, int K, int Kp, int P, int dilation, int causal,
                    const float* g1, const float* b1, const float* a1, const float* ms1,
                    const float* g2, const float* a2, const float* ms2,
                    const double* sums2_part, int sums2_nparts, float* pc, unsigned* amax_out, void* stream);

/* Fixed-order finish of the UN-fused ctn_dw_bwd's tap partials: pc [P, M, H] -> dD [H, P] (the depthwise weight's layout). */
int ctn_dw_bwd_taps(const float* pc, int P, int M, int H, float* dD, void* stream);

/* cLN form of the backward (round 4; ChannelwiseLayerNorm, src/conv_tasnet.py:313-335, on the causal config's blocks :257-266):
 * walks  cLN2 <- PReLU2 <- depthwise  in one pass, i.e. the second norm's whole backward rides in the depthwise kernel:
 *   in : dN2 (grad of cLN2's output, from ctn_pw_dgrad_cln), Dz (= Z of the forward = cLN2's input), X1 (= the forward's input =
 *        cLN1's output), gamma2 / alpha2, fc [M][4][Kp] per-frame constants from ctn_cln_bwd_frame;
 *        g1 != NULL: the first norm's output was never stored -- X1 is its INPUT (the first 1x1 conv's output) and the kernel recomputes
 *        cLN1(prelu(X1, a1)) from (g1, b1, a1) and the per-frame statistics mean1, rstd1 [M,Kp]; else pass NULL for all five
 *   out: dN1 (grad of X1) and pc [P+3, M, H]: rows 0..P-1 dD taps, P: dgamma2, P+1: dbeta2, P+2: dalpha2 partials
 *        (ctn_dw_bwd_rows(P, 2) rows), finished in fixed order by ctn_dw_bwd_cln_finalize: dD [H,P], dgamma2 / dbeta2 [H], dalpha2 [1]. */
int ctn_dw_bwd_cln(const float* dN2, const float* Dz, const float* X1, float* dN1, const float* D,
                   int M, int H, int K, int Kp, int P, int dilation, int causal,
                   const float* g2, const float* a2, const float* fc,
                   const float* g1, const float* b1, const float* a1, const float* mean1, const float* rstd1,
                   float* pc, void* stream);
int ctn_dw_bwd_cln_finalize(const float* pc, int P, int M, int H, float* dD, float* dgamma2, float* dbeta2, float* dalpha2,
                            void* stream);

/* dY = rstd*(gamma*dN - S1/n - xhat*S2/n) * prelu'(Y);  dalpha_part [M*H] = per-row sum over Y<0 of (..)*Y.
 * Backward of  gLN(prelu(Y)), src/conv_tasnet.py:224-225.  dY may alias dN.
 * amax_out != NULL: [M][CTN_AMAX_SLOTS], receives max |dY[m]| -- see the h3 section. */
int ctn_gln_prelu_bwd(const float* dN, const float* Y, float* dY, int M, int H, int K, int Kp,
                      const float* gamma, const float* alpha, const float* ms, const double* sums_part, int nparts,
                      float* dalpha_part, unsigned* amax_out, void* stream);

/* First pass of the STAND-ALONE gLN(prelu(Y)) backward (GlobalLayerNorm used as a module of its own,
 * src/conv_tasnet.py:338-361; inside a TemporalBlock these sums come out of the GEMM / depthwise epilogues):
 * sums_part [M, H, 2] fp64 per-row (S1 = sum gamma*dN, S2 = sum gamma*dN*xhat) for ctn_gln_prelu_bwd, and
 * pc [2, M, H]: row 0 the dgamma partials (sum dN*xhat), row 1 the dbeta partials (sum dN) -> ctn_reduce_mid. */
int ctn_gln_bwd_sums(const float* dN, const float* Y, int M, int H, int K, int Kp, const float* gamma, const float* alpha,
                     const float* ms, double* sums_part, float* pc, void* stream);

/* ---- composite: the whole stack of gLN TemporalBlocks in one call ---------------------------------
 * replaces `temporal_conv_net = nn.Sequential(*repeats)`, src/conv_tasnet.py:176-186, i.e. nblocks = X*R
 * TemporalBlock.forward calls (:233-243) and their autograd backward.  The host side of a block (3 launches forward,
 * 8 backward) is issued from C++ in exactly the order of the per-kernel entry points above, so results are bitwise
 * those of calling them one by one.
 *   params  : HOST array [nblocks][9] of device pointers, per block in the order
 *             w1 [H,B], alpha1 [1], gamma1 [H], beta1 [H], D [H,P], alpha2 [1], gamma2 [H], beta2 [H], w2 [B,H]
 *             (src/conv_tasnet.py:223-225 and :253-262).   grads: the same layout, gradient destinations.
 *   dilation: HOST int [nblocks] (2^x, :170).
 *   x0 [M,B,Kp]: input of block 0.  Forward writes, backward reads (all device, caller-owned):
 *     xs  [nblocks][M,B,Kp]  block outputs (xs[nblocks-1] is the stack's output),
 *     h1s [nblocks][M,H,Kp]  first 1x1 outputs,  ds [nblocks][M,H,Kp] depthwise outputs,
 *     ms  [nblocks][2][M][2] (mean, rstd) of the two gLNs,
 *     amax [nblocks][2][M][CTN_AMAX_SLOTS] unsigned (h3 arithmetic; may be NULL otherwise): the tracked maxima of every block's input and of
 *          its depthwise output (see the h3 section); written by the forward pass (which zeroes it first), read by backward.
 *     save = 0 (inference): xs needs 2 slots, h1s / ds / ms one slot each (amax all nblocks); the output is xs[(nblocks-1) & 1].
 *   forward with side_stream != NULL and M >= 2: the batch runs as two half-batch chains on the two streams (utterances are
 *     independent; one half's HBM-bound phases overlap the other half's MFMA phases); same values bit for bit; on return `stream`
 *     is ordered after both.
 *   backward: dout [M,B,Kp] gradient of the stack's output; dxs [nblocks][M,B,Kp] receives the gradient of every
 *     block's input (dxs[0] = gradient w.r.t. x0); dn1s [nblocks][M,H,Kp] scratch (one slot per block, read by the
 *     weight-gradient stream).  side_stream != NULL: weight-gradient GEMMs and the fixed-order parameter-gradient sums go
 *     there (forked / joined with ctn_stream_order); on return `stream` is ordered after all of them -- unless flags bit 0
 *     is set: then the second stream is left un-joined, so that the caller can put more work behind this call's gradients
 *     (a data-parallel trainer: the all-reduce of this bucket of blocks) while `stream` already runs the next call; such a
 *     call needs a workspace of its own, and the caller joins the streams itself (ctn_stream_order) before the gradients
 *     are read on `stream`.  A stack may be run as several calls over consecutive block ranges (last blocks first).
 *   workspace: ctn_tcn_gln_{fwd,bwd}_workspace() bytes, 256-byte aligned. */
int ctn_tcn_gln_fwd(const void* const* params, const int* dilation, int nblocks, const float* x0,
                    float* xs, float* h1s, float* ds, float* ms, unsigned* amax, int save,
                    int M, int B, int H, int K, int Kp, int P, int causal,
                    void* workspace, size_t workspace_bytes, void* stream, void* side_stream);
size_t ctn_tcn_gln_fwd_workspace(int M, int B, int H, int Kp, int nblocks);
int ctn_tcn_gln_bwd(const void* const* params, void* const* grads, const int* dilation, int nblocks,
                    const float* x0, const float* xs, const float* h1s, const float* ds, const float* ms, const unsigned* amax,
                    const float* dout, float* dxs, float* dn1s,
                    int M, int B, int H, int K, int Kp, int P, int causal,
                    void* workspace, size_t workspace_bytes, void* stream, void* side_stream, int flags);
size_t ctn_tcn_gln_bwd_workspace(int M, int B, int H, int Kp, int P, int nblocks);

/* Measurement hook for the composite stacks (bench.py's roofline leg): ctn_probe_enable(1) makes every launch group issued
 * by ctn_tcn_*_fwd / _bwd on this thread's process record a HIP-event pair on the stream it is launched to;
 * ctn_probe_read waits for them, fills fam[i] (0..15: K1 K2 K3 B1 B2 B3 B4 B5 B6 finalize weight-prep cln_fwd cln_bwd taps last-slab-sums cln-frame-constants)
 * and us[i] (microseconds) in issue order for up to cap groups, returns the number recorded and ends the recording.
 * Off by default: no events, no overhead. */
int ctn_probe_enable(int on);
int ctn_probe_read(int* fam, float* us, int cap);

/* The same for a stack of cLN TemporalBlocks (norm_type = 'cLN': the causal BASELINE config), un-fused norms: per block
 * forward  1x1 -> cLN(PReLU) -> depthwise -> cLN(PReLU) -> 1x1 + residual,  backward the adjoint chain with the two weight
 * gradients on side_stream.  Saved by forward (slot per block): xs [nblocks][M,B,Kp]; h1s, n1s, ds, n2s [nblocks][M,H,Kp]
 * (1x1 output, first norm output, depthwise output, second norm output); st [nblocks][4][M,Kp] = mean1, rstd1, mean2, rstd2.
 * amax [nblocks][2][M][CTN_AMAX_SLOTS] (h3 arithmetic; may be NULL otherwise): tracked maxima of every block's input and of its
 * second norm's output, written by forward, read by backward.
 * save = 0: two x slots, one slot of everything else (amax all nblocks).  backward scratch: dxs [nblocks][M,B,Kp], dh1s [nblocks][M,H,Kp]. */
int ctn_tcn_cln_fwd(const void* const* params, const int* dilation, int nblocks, const float* x0,
                    float* xs, float* h1s, float* n1s, float* ds, float* n2s, float* st, unsigned* amax, int save,
                    int M, int B, int H, int K, int Kp, int P, int causal,
                    void* workspace, size_t workspace_bytes, void* stream, void* side_stream);
size_t ctn_tcn_cln_fwd_workspace(int M, int B, int H, int Kp, int nblocks);
int ctn_tcn_cln_bwd(const void* const* params, void* const* grads, const int* dilation, int nblocks,
                    const float* x0, const float* xs, const float* h1s, const float* n1s, const float* ds, const float* n2s,
                    const float* st, const unsigned* amax, const float* dout, float* dxs, float* dh1s,
                    int M, int B, int H, int K, int Kp, int P, int causal,
                    void* workspace, size_t workspace_bytes, void* stream, void* side_stream, int flags);
size_t ctn_tcn_cln_bwd_workspace(int M, int B, int H, int Kp, int P, int nblocks);

/* ---- channel-wise LayerNorm, src/conv_tasnet.py:313-335 (per frame, biased variance) -----
 * Out = gamma*((a-mean_k)*rstd_k)+beta with a = prelu(Y,alpha) if alpha != NULL else Y.
 * mean, rstd: [M,Kp] saved for backward.  amax_out != NULL: [M][CTN_AMAX_SLOTS], receives max |Out[m]| (h3 section). */
int ctn_cln_fwd(const float* Y, float* Out, float* mean, float* rstd, int M, int Ch, int K, int Kp,
                const float* gamma, const float* beta, const float* alpha, unsigned* amax_out, void* stream);
/* dY = [cLN/PReLU backward of dOut  (+ add)] masked by (relu_ref > 0) when relu_ref != NULL -- and, in the same pass, the
 * parameter-gradient partials: pc [2][ctn_cln_bwd_blocks(M,Kp)][Ch] (ctn_cln_bwd_pc_floats() floats: dgamma, dbeta of every
 * workgroup's 16 frames -- 32 with ctn_tune("cln_fr", 32) --) and dalpha_part [ctn_cln_bwd_blocks(M,Kp)]; ctn_cln_bwd_finalize
 * sums them in fixed order.  Size both with the functions below AFTER any ctn_tune("cln_fr", ...).
 * amax_out != NULL: [M][CTN_AMAX_SLOTS], receives max |dY[m]| (h3 section). */
int ctn_cln_bwd(const float* dOut, const float* Y, float* dY, const float* mean, const float* rstd,
                int M, int Ch, int K, int Kp, const float* gamma, const float* alpha,
                const float* add, const float* relu_ref, float* dalpha_part, float* pc, unsigned* amax_out, void* stream);
int ctn_cln_bwd_blocks(int M, int Kp);
size_t ctn_cln_bwd_pc_floats(int M, int Ch, int Kp);
/* cLN backward WITHOUT a pass of its own (round 4), for a norm that sits between a 1x1 conv and the depthwise conv
 * (src/conv_tasnet.py:257-266): the input-gradient GEMM of the 1x1 conv produces, besides dN = W^T . dOut, the two per-frame sums
 * over channels that the norm's backward needs,
 *     S1[k] = sum_c gamma_c dN[c,k],   S2[k] = sum_c gamma_c dN[c,k] xhat[c,k],   xhat = (prelu(y, alpha) - mean[k]) rstd[k],
 * as column partials of its row tiles: col_part [M][ctn_pw_col_parts(M,R,Kp,w_form)][Kp][2] fp64 (fixed order inside a tile);
 * ctn_cln_bwd_frame sums them over the row tiles and writes fc [M][4][Kp] = (rstd, mean rstd, rstd S1/Ch, rstd S2/Ch)[k], from which
 * ctn_dw_bwd_cln forms  dy = rstd (gamma dN - S1/Ch - xhat S2/Ch) prelu'(y)  on the fly.
 *   W / w_form: 1 = the stored fp32 [Cn, R] matrix used transposed (arithmetic by ctn_tune("arith")), 2 = b6 pieces
 *   (ctn_split_b3_batch, k_major = 1), 3 = h3 pieces (ctn_split_h3_batch, k_major = 1; g_amax = tracked maximum of dOut, else NULL).
 *   y: the norm's input [M,R,Kp]; mean, rstd: [M,Kp] saved by ctn_cln_fwd. */
int ctn_pw_dgrad_cln(const void* W, int w_form, const float* dOut, float* dN, int M, int R, int Cn, int K, int Kp,
                     const float* y, const float* gamma, const float* alpha, const float* mean, const float* rstd,
                     double* col_part, const unsigned* g_amax, void* stream);
int ctn_pw_col_parts(int M, int R, int Kp, int w_form);
/* cLN forward WITHOUT a pass of its own (round 4), for the norm between the first 1x1 conv and the depthwise conv
 * (src/conv_tasnet.py:223-225 with norm_type = 'cLN'): ctn_pw_gemm_cln is the 1x1 conv Out = op(W) . X that also leaves, per frame,
 * (sum_c p, sum_c p^2), p = prelu(Out, alpha), as column partials of its row tiles (col_part as above; w_form 0 = fp32 [R, Cn],
 * 1 = fp32 [Cn, R] used transposed, 2 = b6 pieces, 3 = h3 pieces with x_amax = tracked maximum of X); ctn_cln_stats_frame sums them
 * over the row tiles into mean, rstd [M,Kp] (fp64, biased variance, eps as ctn_cln_fwd); ctn_dw_fwd_cln is ctn_dw_fwd with
 * n = gamma ((prelu(Y, alpha) - mean[k]) rstd[k]) + beta applied while the row is staged: the norm's output is never stored. */
int ctn_pw_gemm_cln(const void* W, int w_form, const float* X, float* Out, int M, int R, int Cn, int K, int Kp,
                    const float* alpha, double* col_part, const unsigned* x_amax, void* stream);
int ctn_cln_stats_frame(const double* col_part, int nparts, float* mean, float* rstd, int M, int Ch, int Kp, void* stream);
int ctn_dw_fwd_cln(const float* Y, float* Z, const float* D, int M, int H, int K, int Kp, int P, int dilation, int causal,
                   const float* mean, const float* rstd, const float* gamma, const float* beta, const float* alpha, void* stream);
int ctn_cln_bwd_frame(const double* col_part, int nparts, const float* mean, const float* rstd, float* fc, int M, int Ch, int Kp,
                      void* stream);
/* One launch that finishes the partials above in fixed order: dgamma[Ch], dbeta[Ch] from pc, and dalpha[1] from
 * dalpha_part [ctn_cln_bwd_blocks(M,Kp)] when that is non-NULL. */
int ctn_cln_bwd_finalize(const float* pc, const float* dalpha_part, int M, int Ch, int Kp, float* dgamma, float* dbeta,
                         float* dalpha, void* stream);

/* ---- BatchNorm1d over (utterances, frames) per channel, optionally behind PReLU ------------------
 * replaces nn.BatchNorm1d as returned by chose_norm's else-branch, src/conv_tasnet.py:305-309, at its two uses
 * (:225 after PReLU :224, :260 after PReLU :259).  gamma/beta are nn.BatchNorm1d's weight/bias [Ch].
 *   training != 0: batch statistics over the M*K valid frames (biased variance for the normalisation); when
 *     running_mean/running_var are non-NULL they are updated in place with `momentum` (unbiased variance), as
 *     torch does.  part: [Ch*M*2] fp64 workspace.
 *   training == 0: normalises with running_mean / running_var.
 *   alpha != NULL: the input is prelu(Y, alpha) (fused), else Y itself.
 *   mr [Ch,2] out: the (mean, 1/sqrt(var+eps)) actually used -- saved for ctn_bn_bwd.  Frames >= K are written 0. */
int ctn_bn_fwd(const float* Y, float* Out, const float* alpha, const float* gamma, const float* beta,
               float* running_mean, float* running_var, int training, float eps, float momentum,
               int M, int Ch, int K, int Kp, double* part, float* mr, void* stream);
/* dY (may alias dOut) = gradient w.r.t. Y (through the PReLU when alpha != NULL); dgamma/dbeta [Ch];
 * dalpha_part [M*Ch] per-row partials (sum them in order for dalpha);  part [Ch*M*2] fp64 and coef [Ch,2] workspaces. */
int ctn_bn_bwd(const float* dOut, const float* Y, float* dY, const float* alpha, const float* gamma, const float* mr,
               int training, int M, int Ch, int K, int Kp, double* part, float* coef, float* dgamma, float* dbeta,
               float* dalpha_part, void* stream);

/* out[f][i] = sum_r in[f][r][i]  -- fixed-order finish of the per-(m,c) partials above */
int ctn_reduce_mid(const float* in, float* out, int F, int Mid, int Inner, void* stream);

/* ---- encoder / decoder glue -----------------------------------------------------------
 * encoder  src/conv_tasnet.py:106-121 : ctn_encoder_fwd (forward); ctn_im2col + ctn_pw_wgrad (basis gradient)
 * decoder  src/conv_tasnet.py:140-145 : ctn_mask_apply + ctn_pw_gemm + ctn_ola (src/utils.py:9-47)
 * mask non-linearity src/conv_tasnet.py:208-214 (relu | softmax over speakers). */
int ctn_im2col(const float* mix, float* xcol, int M, int T, int L, int Lp, int K, int Kp, void* stream);
/* The encoder in one kernel (forward): w[m,n,k] = relu(sum_l U[n,l] * mix[m, k*L/2 + l]), zero for k >= K; the L-sample
 * sliding windows of 256 frames and the basis rows are staged in LDS, no im2col buffer.  U: [N, L] (the Conv1d weight
 * [N,1,L]); w: [M,N,Kp].  Filter lengths compiled in: ctn_encoder_supported(L) (16, 20, 32, 40); other lengths and the
 * weight gradient use ctn_im2col + the GEMM entry points. */
int ctn_encoder_supported(int L);
int ctn_encoder_fwd(const float* mix, const float* U, float* w, int M, int T, int N, int L, int K, int Kp, void* stream);
/* softmax: 0 = relu, 1 = softmax over speakers, 2 = identity (sw = w * score: the stand-alone Decoder.forward, :140) */
int ctn_mask_apply(const float* score, const float* w, float* sw, int M, int C, int N, int Kp, int softmax, void* stream);
int ctn_mask_apply_bwd(const float* dsw, const float* score, const float* w, float* dscore, float* dw,
                       int M, int C, int N, int Kp, int softmax, void* stream);
/* est[b, t] = sum_{k*S+l = t} frames[b, l, k]; zeros for t >= (K-1)S+L (the F.pad of :59). frames:[Bn,Lp,Kp] */
int ctn_ola(const float* frames, float* est, int Bn, int T, int L, int Lp, int K, int Kp, void* stream);
int ctn_unfold(const float* dest, float* dframes, int Bn, int T, int L, int Lp, int K, int Kp, void* stream);
/* overlap_and_add(signal, frame_step) for ANY frame_step, src/utils.py:9-47 (the reference splits frames into
 * gcd(frame_length, frame_step) sub-frames and index_add_s them; here a deterministic gather in ascending frame order):
 * signal [Bn, frames, frame_length] row-major -> out [Bn, (frames-1)*frame_step + frame_length]; _bwd is its adjoint. */
int ctn_overlap_add(const float* signal, float* out, int Bn, int frames, int frame_length, int frame_step, void* stream);
int ctn_overlap_add_bwd(const float* dout, float* dsignal, int Bn, int frames, int frame_length, int frame_step, void* stream);

/* ---- PIT SI-SNR loss, src/pit_criterion.py:12-77 -----------------------------------------
 * source, estimate: [B,C,T]; lengths: [B] int64; perms: [nperm,C] int32 in itertools order.
 * estimate is masked IN PLACE for t >= len (reference :38).  Outputs: max_snr [B], best_idx [B] int64,
 * loss [1] = -mean(max_snr), snr_out [B,C,C] (optional), and the backward tables coef [B,C,4], jsel [B,C]. */
int ctn_sisnr_pit_fwd(const float* source, float* estimate, const long long* lengths, const int* perms, int nperm,
                      int B, int C, int T, float* max_snr, long long* best_idx, float* loss, float* snr_out,
                      float* coef, int* jsel, void* workspace, size_t workspace_bytes, void* stream);
size_t ctn_sisnr_workspace(int B, int C, int T);
int ctn_sisnr_chunks(int T);
/* d_estimate = dloss/d estimate for upstream grads g_loss [1] and/or g_max [B] (either may be NULL) */
int ctn_sisnr_pit_bwd(const float* source, const float* estimate, const long long* lengths, const float* coef,
                      const int* jsel, const float* g_loss, const float* g_max, int B, int C, int T, float* d_estimate,
                      void* stream);

/* ---- optimiser tail, src/solver.py:194-196 (clip_grad_norm_ + Adam.step) on flat buffers ---
 * g' = grad_scale*g; total = ||g'||2 -> total_norm_out; g' *= min(1, max_norm/(total+1e-6)) if max_norm > 0;
 * torch.optim.Adam update (no weight decay / amsgrad) with bias corrections for `step` (1-based).
 * workspace: ctn_optim_parts() doubles. */
int ctn_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                       float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps, int step,
                       float* total_norm_out, double* workspace, void* stream);
int ctn_optim_parts(void);

#ifdef __cplusplus
}
#endif
#endif /* CTN_HIP_H */
