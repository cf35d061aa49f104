/* skf.h - C ABI of libskf.so, the MI355X (gfx950) implementation of the
 * sketch-transformer-tf2 training hot path of leosampaio/sketchformer.
 *
 * The reference has no native layer: every entry point below replaces arithmetic
 * that the reference delegates to TensorFlow 2.1 / Keras from the cited Python call
 * site (paths relative to the reference repository).  A reference-side binding
 * (ctypes) is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers unless the
 *    parameter name ends in _host.  Activations are row-major (B, L, features).
 *  - the caller owns every buffer; kernels never allocate.  Scratch is passed as
 *    (workspace, workspace_bytes); sizes come from the *_workspace_bytes queries.
 *  - launches are asynchronous on `stream` (a hipStream_t cast to void*), re-entrant,
 *    and keep no global state besides a thread-local error string (the opt-in launch
 *    profiler below is a measurement facility of the calling process, off by default).
 *    The library reads no environment variable: tuning / ablation knobs exist only in
 *    -DSKF_MEASURE=1 builds (csrc/skf_common.h: skf_knob), never in the shipped .so.
 *  - return 0 on success, negative on error (never throws across the ABI);
 *    skf_last_error() describes the last failure on the calling thread.
 */
#ifndef SKF_H_
#define SKF_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* skf_stream_t; /* hipStream_t */

#define SKF_OK 0
#define SKF_EINVAL (-1)       /* bad argument */
#define SKF_EUNSUPPORTED (-2) /* valid in the reference, not implemented here */
#define SKF_EHIP (-3)         /* HIP runtime / launch failure */

const char* skf_last_error(void);
int skf_version(void);
/* name of the device the library will launch on + number of visible devices (host query) */
int skf_device_info(char* name_host, size_t name_len, int* n_devices_host);

/* Per-launch timing with HIP events on the launch stream (bench.py's roofline leg).  enable(1) clears the
 * records and starts recording every kernel launch made through this library on the calling process;
 * report() synchronises the recorded events and writes one JSON array
 * [{"tag","count","ms","flops","bytes"}...] (algorithmic flops / bytes summed over the launches of a tag). */
int skf_profiler_enable(int on);
int skf_profiler_report(char* buf_host, size_t len);

/* ------------------------------------------------------------------ Dense
 * tf.keras.layers.Dense forward / dgrad / wgrad (+bias grad).
 * builders/layers/transformer.py:154-158,196-197; models/sketchformer.py:85-104.
 *   C[M,N] (+)= opA(A)[M,K] . opB(B)[K,N] (+bias) (act) (relu mask)
 *   a_kcontig: 1 = A stored [M][K] (lda = row stride), 0 = A stored [K][M]
 *   b_kcontig: 0 = B stored [K][N] (ldb = row stride), 1 = B stored [N][K]
 *   act: 0 none, 1 relu, 2 tanh;  relu_src (optional, ld_relu): C = 0 where relu_src <= 0
 *   splits > 1 (or bias_grad != NULL): split-K through `workspace`; bias_grad[n] = sum_k B[k][n]
 *   (the bias gradient of a wgrad call, B stored [K][N]); no fused epilogue on that path. */
/* `precision` - the arithmetic of a Dense matmul (an argument of every entry that multiplies, and a field of SkfConfig):
 * SKF_PREC_F32 (0)    = v_mfma_f32_16x16x4_f32 on the fp32 operands (bit-for-bit a k-ordered fmaf chain);
 * SKF_PREC_BF16X6 (6) = every fp32 operand split exactly into three bf16 pieces, the six largest piece products summed
 *     in fp32 on the bf16 matrix cores (dropped terms <= 2^-23 |a||b| per product: the size of one fp32 rounding;
 *     against float64 its error is below the fp32-MFMA kernel's: the piece products are exact and the small ones are
 *     summed before the large ones);
 * SKF_PREC_BF16X3 (3) = two pieces / three products (dropped terms <= 2^-15 |a||b|): opt-in fast mode, never a default.
 * Operands and results are fp32 tensors in every mode.  Non-finite operands: an output element is non-finite in the
 * split modes exactly where it is non-finite in mode 0 (an inf operand gives NaN there instead of +-inf: the remainder
 * of inf is inf - inf); finite outputs are unaffected.  fp32 subnormal operands / pieces below the bf16 normal range are
 * flushed by the bf16 matrix cores: absolute error <= 2^-126 per product (tests/test_gpu_ops.py). */
#define SKF_PREC_F32 0
#define SKF_PREC_BF16X3 3
#define SKF_PREC_BF16X6 6
/* May be OR'ed into the `precision` argument of skf_attention_bwd / skf_attention_bwd_rows: take the two-pass backward
 * (skf_attention_bwd2.hip: a dQ pass over query tiles, a dK/dV pass over key tiles, bf16 matrix cores) wherever it is built
 * (head sizes 16 and 32, Lq, Lk <= 512, split modes) even where the dispatch would pick the one-pass kernel (head size 16). */
#define SKF_ATTN_TWO_PASS 0x100
size_t skf_gemm_workspace_bytes(int M, int N, int K, int splits, int with_bias_grad);
int skf_gemm_default_splits(int M, int N, int K);
int skf_gemm_f32(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                 float* C, int ldc, const float* bias, int act, const float* relu_src, int ld_relu, int accumulate,
                 int splits, float* bias_grad, int bias_grad_accumulate, void* workspace, size_t workspace_bytes,
                 int precision, skf_stream_t stream);

/* wgrad split into its two phases so that a caller can run MANY wgrads and reduce all their slabs with one launch:
 * skf_gemm_wgrad_partial = the partial-tile kernel only (A = X [K][M], B = dY [K][N]; slab layout
 * [splits][M][N] followed by [splits][N] column sums), skf_splitk_reduce_batch = one launch over a DEVICE array of
 * descriptors whose block_begin fields are the running sum of skf_splitk_reduce_blocks(M, N). */
typedef struct SkfReduceDesc {
  const float* slab; float* C; float* bias_grad; /* bias_grad may be NULL */
  int32_t splits, M, N, ldc, block_begin, pad;
} SkfReduceDesc;
int skf_gemm_wgrad_partial(int M, int N, int K, const float* A, int lda, const float* B, int ldb, int splits,
                           int with_bias_grad, float* slab, size_t slab_bytes, int* splits_used_host, int precision,
                           skf_stream_t stream);
/* ---- rows that are known to be zero (decoder-side backward of a padded batch, token mode) ----
 * The masked cross-entropy (builders/losses.py; models/sketchformer.py:313-349) gives padded target positions a zero
 * gradient and the decoder is causal, so every decoder-side gradient row at or behind a sample's last unmasked position
 * is exactly zero.  skf_target_live_len: live_len[b] = 1 + last t < Ld with tar[b][t + 1] != 0 (tar = the (B, tar_ld)
 * int64 target tokens, decoder row t is trained on tar[b][t + 1]).  skf_row_blocks_build: the blocks of `granule`
 * consecutive rows of the flattened (B * rows_per_sample) rows, as {n_live, n_blocks, live ids ascending, dead ids
 * ascending, one 0/1 live flag per block in block order} (skf_row_blocks_bytes bytes).  skf_gemm_f32_rows (dgrad form, A [M][K], granule 16: rows of A in dead blocks
 * are zero -> their C rows are stored as zeros without being computed, or left alone when accumulating) and
 * skf_gemm_wgrad_partial_rows (granule 32: contraction rows in dead blocks are zero in B = dY and are not visited) are
 * EXACT: what is skipped is x * 0.  Kernels other than the split-arithmetic weight-stationary ones ignore the list. */
int skf_target_live_len(const long long* tar, int tar_ld, int B, int Ld, int* live_len, skf_stream_t stream);
size_t skf_row_blocks_bytes(int rows, int granule);
int skf_row_blocks_build(const int* live_len, int B, int rows_per_sample, int granule, int* blocks, skf_stream_t stream);
int skf_gemm_f32_rows(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                      float* C, int ldc, const float* bias, int act, const float* relu_src, int ld_relu, int accumulate,
                      int splits, float* bias_grad, int bias_grad_accumulate, void* workspace, size_t workspace_bytes,
                      int precision, const int* row_blocks, int row_block_rows, skf_stream_t stream);
/* ReLU sign bits (ffn: builders/layers/transformer.py:196-197 Dense(dff, relu) -> Dense(d); its tape gradient multiplies
 * d(hidden) by relu'(hidden)).  A forward launch with act = relu can leave ONE BIT per output element ("> 0") in
 * relu_bits_out, in the layout of its own tiles, and the input-gradient launch of the same (M, N, K) reads relu_bits_in
 * instead of the hidden tensor (relu_src): identical results, 1/32 of the bytes.  Exists where the split-arithmetic
 * weight-stationary kernel takes the launch: skf_gemm_relu_bits_bytes returns the buffer size (64-byte aligned buffer), or 0
 * when the shape has no such path - then pass NULL / use relu_src; bits on an unsupported launch are an error. */
size_t skf_gemm_relu_bits_bytes(int M, int N, int K, int precision);
int skf_gemm_f32_bits(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                      float* C, int ldc, const float* bias, int act, const float* relu_src, int ld_relu, int accumulate,
                      int splits, float* bias_grad, int bias_grad_accumulate, void* workspace, size_t workspace_bytes,
                      int precision, const int* row_blocks, int row_block_rows, void* relu_bits_out, const void* relu_bits_in,
                      skf_stream_t stream);
/* Dense + residual + dropout + LayerNorm in ONE launch: the reference's `layernorm1(x + dropout1(attn_output))`
 * (builders/layers/transformer.py:216-224 EncoderLayer.call, :258-272 DecoderLayer.call) where attn_output is the MultiHeadAttention
 * output projection `self.dense(concat_attention)` (:186):
 *   z = x + dropout(A[M,K] . W[K,N] + bias, rate, site);  out = (z - mean) * rstd * gamma + beta;  stats[row] = (mean, rstd)
 * with the arithmetic, the dropout mask and the epsilon (1e-6) of skf_gemm_f32 followed by skf_layernorm_residual_fwd (z is
 * bit-identical to that pair's; mean / rstd are summed in a different order).  z, out and x are row-major with pitch N.  Exists where
 * one workgroup of the split-arithmetic weight-stationary kernel owns whole output rows: skf_gemm_ln_residual_supported
 * (K = N = 128, precision != SKF_PREC_F32); anything else is an error - use the two calls. */
int skf_gemm_ln_residual_supported(int M, int N, int K, int precision);
int skf_gemm_ln_residual_f32(int M, int N, int K, const float* A, int lda, const float* W, int ldw, const float* bias,
                             const float* x, const float* gamma, const float* beta, float* z, float* out, float* stats,
                             float rate, unsigned site, const void* step_state, int precision, skf_stream_t stream);
/* The feed-forward block of a layer in ONE launch per direction: the reference's `point_wise_feed_forward_network`
 * (builders/layers/transformer.py:194-198: Dense(dff, relu) -> Dense(d_model)) together with the `layernorm(out + dropout(ffn(out)))`
 * that wraps it in EncoderLayer.call (:221-224) / DecoderLayer.call (:270-272), and the tape gradient of the two Dense layers:
 *   forward:   h = relu(x[M,d] . W1[d,dff] + b1);  y = h . W2[dff,d] + b2;  z = x + dropout(y, rate, site);
 *              out = (z - mean) * rstd * gamma + beta;  stats[row] = (mean, rstd);  relu_bits_out = sign bits of h
 *   backward:  dh = (dy[M,d] . W2^T) o relu'(h)  (from the sign bits);  dx (+)= dh . W1^T
 * with the split arithmetic of skf_gemm_f32 (SKF_PREC_BF16X6 / BF16X3), the dropout mask of skf_layernorm_residual_fwd and its
 * LayerNorm arithmetic; h / dh are written in full (the weight gradients read them), never read back.  The weights are read from
 * PRE-SPLIT images in the kernel's MFMA operand order: skf_ffn_weight_images builds n of them in one launch (image k from
 * W1[k] [d][dff] pitch ld1[k], W2[k] [dff][d] pitch ld2[k]; transpose[k] = 0: the forward's image, 1: the backward's;
 * skf_ffn_image_bytes each, 16-byte aligned) - rebuild them whenever the weights change (once per optimizer step).  Exists for
 * d = 128, dff = 512 in the split modes (skf_ffn_fused_supported; anything else is an error - use the separate calls).
 * row_blocks (backward, optional): the 16-row block list of skf_row_blocks_build - dead blocks have dy == 0: their dh rows are
 * stored as zeros, their dx rows left alone (accumulate) or zeroed. */
int skf_ffn_fused_supported(int M, int d, int dff, int precision);
size_t skf_ffn_image_bytes(int d, int dff, int precision);
size_t skf_ffn_relu_bits_bytes(int M, int d, int dff, int precision);
int skf_ffn_weight_images(int n, const float* const* W1, const int* ld1, const float* const* W2, const int* ld2,
                          const int* transpose, void* const* images, int d, int dff, int precision, skf_stream_t stream);
int skf_ffn_fused_fwd_f32(int M, int d, int dff, const float* x, const void* image, const float* b1, const float* b2,
                          float* h, void* relu_bits_out, const float* gamma, const float* beta, float* z, float* out,
                          float* stats, float rate, unsigned site, const void* step_state, int precision, skf_stream_t stream);
int skf_ffn_fused_bwd_f32(int M, int d, int dff, const float* dy, const void* image_t, const void* relu_bits_in,
                          float* dh, float* dx, int accumulate, const int* row_blocks, int row_block_rows,
                          int precision, skf_stream_t stream);
/* The forward launch going one step further: proj_out[M, proj_n] = out . Wp + proj_bias for the Dense that consumes the block's
 * LayerNorm output - the NEXT layer's fused q|k|v projection (builders/layers/transformer.py:154-158, 216-217; proj_n = 384) or a
 * query projection (128) - with proj_image = skf_dense_weight_images(Wp [d][proj_n], transpose = 0).  Same arithmetic as
 * skf_gemm_f32 on `out`; everything else as skf_ffn_fused_fwd_f32. */
int skf_ffn_fused_fwd_proj_f32(int M, int d, int dff, const float* x, const void* image, const float* b1, const float* b2,
                               float* h, void* relu_bits_out, const float* gamma, const float* beta, float* z, float* out,
                               float* stats, float rate, unsigned site, const void* step_state, const void* proj_image,
                               const float* proj_bias, int proj_n, float* proj_out, int precision, skf_stream_t stream);
/* The forward launch in its general form: the feed-forward block with an optional LEADING stage and an optional CHAINED projection.
 * Leading stage (pre_image != NULL): the launch starts at the attention output a (= x [M, d]):
 *   pre_z = pre_residual + dropout(a . Wo + pre_bias, rate, pre_site);  pre_out = LayerNorm(pre_z; pre_gamma, pre_beta);  pre_stats = (mean, rstd)
 * - the MultiHeadAttention output projection with its residual LayerNorm (builders/layers/transformer.py:186, 216-224 / 262-268), i.e.
 * skf_gemm_ln_residual_f32 - and pre_out is the block's input and residual (pre_image = skf_dense_weight_images(Wo [d][d], transpose 0)).
 * Chained projection (proj_image != NULL): as skf_ffn_fused_fwd_proj_f32.  struct_size = sizeof(SkfFfnBlockFwd).
 * image == NULL (with pre_image and proj_image): NO feed-forward block - the leading stage followed by the projection of ITS LayerNorm
 * output, proj_out = pre_out . Wp + proj_bias: the decoder's self-attention tail with the cross-attention query projection behind it
 * (builders/layers/transformer.py:258-262); b1, b2, h, relu_bits_out, gamma, beta, z, out, stats, site are not used. */
typedef struct SkfFfnBlockFwd {
  uint32_t struct_size; int32_t M, d, dff, precision;
  const float* x; const void* image; const float* b1; const float* b2; float* h; void* relu_bits_out;
  const float* gamma; const float* beta; float* z; float* out; float* stats;
  float rate; uint32_t site; const void* step_state;
  const void* pre_image; const float* pre_bias; const float* pre_residual; const float* pre_gamma; const float* pre_beta;
  float* pre_z; float* pre_out; float* pre_stats; uint32_t pre_site; int32_t proj_n;
  const void* proj_image; const float* proj_bias; float* proj_out;
} SkfFfnBlockFwd;
int skf_ffn_block_fwd_f32(const SkfFfnBlockFwd* block, skf_stream_t stream);
/* The backward launch starting one step earlier, at the gradient `dout` of the LayerNorm that closes the block
 * (out = LayerNorm(z), z = x + dropout(ffn(x))): dz = LayerNorm'(dout) from (z, stats, gamma) with the arithmetic of
 * skf_layernorm_residual_bwd, dy = dropout'(dz) (written: the second Dense's weight gradient reads it), dh as above, and
 * dx = dz + dh . W1^T (the residual path added in registers; dx is written, not accumulated).  dgamma / dbeta leave as
 * skf_ffn_fused_ln_partials(M) partial row pairs [n][2][d] in ln_partials - sum them with skf_splitk_reduce(ln_partials, n, 1, 2 * d,
 * ...) or ride them in a skf_splitk_reduce_batch descriptor like the LayerNorm launch's partials.  Dead row blocks (row_blocks):
 * dout == 0 there; their dy, dh and dx rows are stored as zeros. */
int skf_ffn_fused_ln_partials(int M);
int skf_ffn_fused_bwd_ln_f32(int M, int d, int dff, const float* dout, const float* z, const float* stats, const float* gamma,
                             float rate, unsigned site, const void* step_state, const void* image_t, const void* relu_bits_in,
                             float* dy, float* dh, float* dx, float* ln_partials, size_t ln_partials_bytes,
                             const int* row_blocks, int row_block_rows, int precision, skf_stream_t stream);
/* Pre-split operand images in general: image j of B_j [K][N] (K % 32 == 0, N % 16 == 0; transpose[j] = 0: B[k][n] = src[k * ld + n],
 * 1: B[k][n] = src[n * ld + k]), skf_dense_image_bytes(K, N, precision) bytes each, all in one launch per 40 images. */
size_t skf_dense_image_bytes(int K, int N, int precision);
int skf_dense_weight_images(int n, const float* const* src, const int* ld, const int* transpose, const int* K, const int* N,
                            void* const* images, int precision, skf_stream_t stream);
/* The backward of `out = LayerNorm(x + dropout(Dense(a)))` - the MultiHeadAttention output projection with the residual LayerNorm
 * behind it (builders/layers/transformer.py:186, 216-224, 258-268) - from the gradient of `out` to the gradient of `a` in ONE launch:
 *   dz = LayerNorm'(dout) (written: the residual path's gradient), dy = dropout'(dz) (written: the Dense's weight gradient reads it),
 *   da = dy[M,d] . W^T   (image_t = the transposed image of W [d][d]: skf_dense_weight_images(.., transpose = 1, K = d, N = d))
 * with the arithmetic of skf_layernorm_residual_bwd followed by skf_gemm_f32; dgamma / dbeta as skf_layernorm_bwd_dgrad_partials(M)
 * partial row pairs [n][2][d] (see skf_ffn_fused_bwd_ln_f32).  Exists for d = 128 in the split modes.  row_blocks: 16-row blocks,
 * dead blocks have dout == 0 and get zero rows in dz, dy and da. */
int skf_layernorm_bwd_dgrad_supported(int M, int d, int precision);
int skf_layernorm_bwd_dgrad_partials(int M);
int skf_layernorm_bwd_dgrad_f32(int M, int d, const float* dout, const float* z, const float* stats, const float* gamma,
                                float rate, unsigned site, const void* step_state, const void* image_t, float* dz, float* dy,
                                float* da, float* ln_partials, size_t ln_partials_bytes, const int* row_blocks,
                                int row_block_rows, int precision, skf_stream_t stream);
/* the same with a LEADING product: the gradient of `out` is dout + lead_a[M,d] . Wl^T (lead_image_t = the transposed image of Wl [d][d]) -
 * the input gradient of a Dense that reads `out` (the decoder's cross-attention query projection, builders/layers/transformer.py:258-262),
 * formed in the launch instead of being accumulated into dout by a GEMM launch of its own; lead_a == NULL: skf_layernorm_bwd_dgrad_f32 */
int skf_layernorm_bwd_dgrad_lead_f32(int M, int d, const float* dout, const float* lead_a, const void* lead_image_t, const float* z,
                                     const float* stats, const float* gamma, float rate, unsigned site, const void* step_state,
                                     const void* image_t, float* dz, float* dy, float* da, float* ln_partials, size_t ln_partials_bytes,
                                     const int* row_blocks, int row_block_rows, int precision, skf_stream_t stream);
int skf_gemm_wgrad_partial_rows(int M, int N, int K, const float* A, int lda, const float* B, int ldb, int splits,
                                int with_bias_grad, float* slab, size_t slab_bytes, int* splits_used_host, int precision,
                                const int* row_blocks, int row_block_rows, skf_stream_t stream);
/* the partial-tile kernels of up to 8 weight gradients in one launch (a transformer layer's, issued together); a HOST array,
 * splits_used is written per problem; problems the split-arithmetic kernel cannot take fall back to one launch each */
typedef struct SkfWgradProblem {
  const float* A; const float* B; float* slab; size_t slab_bytes; const int* row_blocks;   /* row_blocks may be NULL */
  int32_t M, N, K, lda, ldb, splits, with_bias_grad, row_block_rows, splits_used, pad;
} SkfWgradProblem;
int skf_gemm_wgrad_partial_group(SkfWgradProblem* probs, int n, int precision, skf_stream_t stream);
/* one slab ([splits][M][N] then [splits][N] when bias_grad != NULL) -> C (+)= sum, bias_grad (+)= column sums */
int skf_splitk_reduce(const float* slab, int splits, int M, int N, float* C, int ldc, int accumulate, float* bias_grad,
                      int bias_grad_accumulate, skf_stream_t stream);
int skf_splitk_reduce_blocks(int M, int N);
int skf_splitk_reduce_batch(const SkfReduceDesc* descs_dev, int ndesc, int total_blocks, skf_stream_t stream);

/* ------------------------------------------------------------------ attention
 * builders/utils.py:71-105 scaled_dot_product_attention + the head split / merge of
 * builders/layers/transformer.py:160-186 + the masks of builders/utils.py:35-68.
 * Q (B,Lq,ldq) K,V (B,Lk,ld*) O (B,Lq,ldo); head h = columns [h*dh,(h+1)*dh).
 * key_mask: (B, key_mask_ld) bytes, 1 = padded key (create_padding_mask), may be NULL.
 * precision != SKF_PREC_F32: the score products of the dh = 16 kernels run on the bf16 matrix cores with split operands.
 * causal: add the look-ahead mask (needs Lq == Lk).  stats: (B,H,Lq,2) row max of the base-2 logits
 * (q.k * log2(e)/sqrt(dh)) and 1/row-sum - opaque to the caller, kept for the backward.  dh in {16,32,64}. */
int skf_attention_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                      const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk, int dh,
                      float* O, int ldo, float* stats, int precision, skf_stream_t stream);
int skf_attention_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                      const float* dO, int lddo, const float* stats, const unsigned char* key_mask, int key_mask_ld,
                      int causal, int B, int H, int Lq, int Lk, int dh, float* dQ, int lddq, float* dK, int lddk,
                      float* dV, int lddv, int precision, skf_stream_t stream);
/* the same with per-sample live query counts (skf_target_live_len): query rows >= q_live_len[b] have dO == 0 exactly; the
 * one-pass kernel neither stages nor visits their tiles and stores zeros into their dQ rows (the two-pass kernel finds
 * all-zero dO tiles by itself) */
int skf_attention_bwd_rows(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                           const float* dO, int lddo, const float* stats, const unsigned char* key_mask, int key_mask_ld,
                           int causal, int B, int H, int Lq, int Lk, int dh, float* dQ, int lddq, float* dK, int lddk,
                           float* dV, int lddv, int precision, const int* q_live_len, skf_stream_t stream);
/* Padded batches: a (sample, head) workgroup costs what its sample's length makes it cost, and consecutive workgroups of an XCD are
 * handed to its four shader engines in turn and never leave them - so the engine that draws the long samples finishes last.
 * skf_sample_order: order[0..B) = the samples sorted by the number of UNMASKED positions in up to two padding-mask matrices (either may
 * be NULL), most first, stable (B <= 4096).  The *_ordered forms take that list (or NULL = the plain numbering) and deal the sorted samples over the
 * 8 XCDs, heaviest first, the heads of a sample over the engines of its XCD; used when B is a multiple of 8, ignored otherwise and by the
 * plain fp32 kernels of other head sizes.  The numbering never changes a result bit.  sample_order must be a PERMUTATION of [0, B) in device
 * memory (what skf_sample_order writes); the kernels do not verify it - entries are clamped into [0, B), so a malformed list cannot touch
 * memory outside the tensors, but it computes some samples twice and leaves the outputs of others unwritten.  Measured (B 128, H 8, L 200, 58 % padding): the three
 * backward calls of a layer 168 -> 143 us, the forward calls 98 -> 90 us (profiles/r05o_attn_order.txt). */
int skf_sample_order(const unsigned char* mask_a, int lda, int La, const unsigned char* mask_b, int ldb, int Lb, int B, int* order,
                     skf_stream_t stream);
int skf_attention_fwd_ordered(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                              const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk, int dh,
                              float* O, int ldo, float* stats, int precision, const int* sample_order, skf_stream_t stream);
int skf_attention_bwd_ordered(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                              const float* dO, int lddo, const float* stats, const unsigned char* key_mask, int key_mask_ld,
                              int causal, int B, int H, int Lq, int Lk, int dh, float* dQ, int lddq, float* dK, int lddk,
                              float* dV, int lddv, int precision, const int* q_live_len, const int* sample_order, skf_stream_t stream);
/* The second return value of scaled_dot_product_attention (builders/utils.py:105: `return output, attention_weights`):
 * W (B,H,Lq,Lk) = softmax(q.k/sqrt(dh) + mask * -1e9) over the keys, same mask arguments as skf_attention_fwd.  Only the builders
 * front-end calls it, on request (the train step of the reference drops the weights: models/sketchformer.py:140-145); plain fp32
 * kernel, any dh <= 128, Lq, Lk <= 1024. */
int skf_attention_weights(const float* Q, int ldq, const float* K, int ldk, const unsigned char* key_mask, int key_mask_ld,
                          int causal, int B, int H, int Lq, int Lk, int dh, float* W, skf_stream_t stream);
/* builders/utils.py:71-105 scaled_dot_product_attention with ANY float mask broadcastable to (B,H,Lq,Lk) (`:96-97`:
 * `scaled_attention_logits += (mask * -1e9)` - the mask is ADDED, a fractional value lowers a logit, it does not remove the key).
 * mask may be NULL; mask_stride_b / _h / _q are ELEMENT strides of the sample / head / query axes (0 = broadcast), the key axis has
 * unit stride.  O (B,Lq,H*dh) with row stride ldo; W = the (B,H,Lq,Lk) weights or NULL.  Plain fp32 kernel of skf_generic.hip (one
 * wave per query row: correct, not fast) - the padding and padding + look-ahead masks the model builds go to skf_attention_fwd as
 * bytes; the front-end (builders.utils.scaled_dot_product_attention) routes every other mask here instead of refusing it. */
int skf_attention_fwd_float_mask(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* mask,
                                 long mask_stride_b, long mask_stride_h, long mask_stride_q, int B, int H, int Lq, int Lk, int dh,
                                 float* O, int ldo, float* W, skf_stream_t stream);
/* Row means behind LossManager.add_mean_loss / add_mae_loss / add_mse_loss (builders/losses.py:68-75; tf.keras.losses.MAE / MSE
 * reduce the LAST axis): out[r] = mean_c a[r][c] (mode 0), mean_c |a - b| (mode 1), mean_c (a - b)^2 (mode 2); a, b (rows, cols)
 * contiguous. */
int skf_row_mean(const float* a, const float* b, long rows, int cols, int mode, float* out, skf_stream_t stream);

/* ------------------------------------------------------------------ embedding stage
 * Encoder.call / Decoder.call head, builders/layers/transformer.py:288-296, 325-334:
 * out = Dropout(Embedding(tok) * sqrt(d) + pos[:L]).  tokens (B, tok_ld) int64 (first L used). */
int skf_embed_fwd(const long long* tokens, int tok_ld, int B, int L, const float* table, int vocab, int d,
                  const float* pos, float* out, float rate, unsigned site, const void* step_state, skf_stream_t stream);
/* dtable (vocab,d) must be zeroed by the caller; receives the dense embedding gradient. */
int skf_embed_bwd(const long long* tokens, int tok_ld, int B, int L, const float* dx, int vocab, int d, float* dtable,
                  float rate, unsigned site, const void* step_state, skf_stream_t stream);
/* The same gradient without pre-zeroed table and (almost) without atomics, in two launches: skf_embed_sort depends on the
 * tokens only (counting sort of the B*L positions by id into `ws`, one workgroup; rows of ids that need more than one
 * 256-position chunk are zeroed in `zero_table` when it is not NULL), skf_embed_bwd_sorted then writes EVERY row of
 * dtable (zeros for unused ids). */
size_t skf_embed_sort_workspace_bytes(int B, int L, int vocab);
int skf_embed_sort(const long long* tokens, int tok_ld, int B, int L, int vocab, float* zero_table, int d, void* ws,
                   size_t ws_bytes, skf_stream_t stream);
int skf_embed_bwd_sorted(const void* ws, int B, int L, const float* dx, int vocab, int d, float* dtable, float rate,
                         unsigned site, const void* step_state, skf_stream_t stream);
/* key padding mask bytes (create_padding_mask, builders/utils.py:35-43): out[b][t] = tokens[b][t]==0 */
int skf_padding_mask(const long long* tokens, int tok_ld, int B, int L, unsigned char* out, skf_stream_t stream);

/* ------------------------------------------------------------------ residual + LayerNorm
 * out = LayerNormalization(eps=1e-6)(x + Dropout(y)); builders/layers/transformer.py:217-222,247-260.
 * y is overwritten with z = x + Dropout(y); stats (rows,2) = mean, rstd. */
int skf_layernorm_residual_fwd(const float* x, float* y_inout_z, const float* gamma, const float* beta, float* out,
                               float* stats, int rows, int d, float rate, unsigned site, const void* step_state,
                               skf_stream_t stream);
size_t skf_layernorm_bwd_workspace_bytes(int rows, int d);
/* dz = d(x + drop(y)); dy = dz * dropout mask (only written when rate > 0; with rate == 0 dy == dz). */
/* dgamma == dbeta == NULL: only the per-workgroup partials are written, workspace = [g][2][d] floats with
 * g = skf_layernorm_bwd_workspace_bytes(rows, d) / (8*d) (row 2k = dgamma part, as one [g][2d] matrix); the caller
 * column-sums them later (the train step batches all of them into its split-K reduction launch). */
int skf_layernorm_residual_bwd(const float* dout, const float* z, const float* stats, const float* gamma, float* dz,
                               float* dy, float* dgamma, float* dbeta, int rows, int d, float rate, unsigned site,
                               const void* step_state, void* workspace, size_t workspace_bytes, skf_stream_t stream);
/* rows = B * rows_per_sample; row t of sample b with t >= live_len[b] has dout == 0 exactly (skf_target_live_len): it is not
 * read, zeros are stored (d <= 256; wider rows take the dense kernel) */
int skf_layernorm_residual_bwd_rows(const float* dout, const float* z, const float* stats, const float* gamma, float* dz,
                                    float* dy, float* dgamma, float* dbeta, int rows, int d, float rate, unsigned site,
                                    const void* step_state, void* workspace, size_t workspace_bytes, const int* live_len,
                                    int rows_per_sample, skf_stream_t stream);
int skf_colsum(const float* in, int nrows, int ld, int ncols, float* out, int accumulate, skf_stream_t stream);

/* ------------------------------------------------------------------ loss heads + metrics
 * Sparse softmax CE from logits, fused accuracy + in-place gradient.
 * builders/losses.py:26-41 (mask_pad=1: loss*(target!=0), mean over ALL rows -> scale = weight/rows),
 * builders/losses.py:21-24 (mask_pad=0), builders/keras_metrics.py:25 (first-index argmax == target).
 * target element for row r: target[(r / tgt_cols) * tgt_ld + (r % tgt_cols) + tgt_off].
 * probs_out (rows,ncls) optional (classify_layer's softmax output, models/sketchformer.py:99). */
int skf_softmax_ce(float* logits, int ld, int rows, int ncls, const long long* target, int tgt_ld, int tgt_cols,
                   int tgt_off, int mask_pad, float scale, float* row_loss, float* row_hit, float* probs_out,
                   int write_grad, skf_stream_t stream);
/* metrics (32 floats): [0..4] this step recon_loss, recon_acc, class_loss, class_acc, total_loss;
 * [8..12] running totals, [16..20] running counts (Keras Mean / SparseCategoricalAccuracy). */
int skf_metrics_update(const float* recon_loss, const float* recon_hit, int recon_rows, float recon_weight,
                       const float* class_loss, const float* class_hit, int class_rows, float class_weight,
                       const float* recon_scalar /* continuous mode: the already reduced recon loss, else NULL */,
                       float* metrics, skf_stream_t stream);

/* ------------------------------------------------------------------ continuous stroke-5 mode (use_continuous_data=True)
 * x (B, x_ld_rows, 5) float32 stroke-5 rows; the first L rows of every sample are used.
 * builders/utils.py:35-43 (pad bit), builders/layers/transformer.py:276,288-296 (embedding = Dense(5->d)),
 * builders/losses.py:43-66 (location MSE + GLOBAL mean pen-state CE, masked, mean over all positions). */
int skf_padding_mask_continuous(const float* x, int x_ld_rows, int B, int L, unsigned char* out, skf_stream_t stream);
int skf_embed_continuous_fwd(const float* x, int x_ld_rows, int B, int L, const float* W, const float* bias, int d,
                             const float* pos, float* out, float rate, unsigned site, const void* step_state,
                             skf_stream_t stream);
size_t skf_embed_continuous_bwd_workspace_bytes(int rows, int d);
int skf_embed_continuous_bwd(const float* x, int x_ld_rows, int B, int L, const float* dx, int d, float* dW, float* dbias,
                             float rate, unsigned site, const void* step_state, void* workspace, size_t workspace_bytes,
                             skf_stream_t stream);
/* pred (rows,5) is overwritten with the gradient when write_grad; target row r =
 * target[(r / tgt_cols) * tgt_ld_rows + r % tgt_cols + tgt_off]; scalars (4 floats): sum(loc*mask), sum(ce),
 * sum(mask), weighted loss. */
int skf_continuous_loss(float* pred_inout_grad, const float* target, int tgt_ld_rows, int tgt_cols, int tgt_off, int rows,
                        float weight, float* row_loc, float* row_ce, float* row_mask, float* scalars, int write_grad,
                        skf_stream_t stream);

/* ------------------------------------------------------------------ bottleneck + expander
 * SelfAttnV1.call after u = tanh(xW+b): builders/layers/transformer.py:70-73. */
int skf_pool_fwd(const float* u, const float* Vw, const float* x, int B, int L, int U, int d, float* a_out, float* emb,
                 skf_stream_t stream);
/* u is overwritten with d(pre-tanh); dx receives a[t]*demb[c]; workspace >= B*U floats */
int skf_pool_bwd(float* u_inout_dpre, const float* Vw, const float* x, const float* a, const float* demb, int B, int L,
                 int U, int d, float* dx, float* dV, void* workspace, size_t workspace_bytes, skf_stream_t stream);
/* DenseExpander.call: builders/layers/transformer.py:370-376. workspace >= 2*B*L floats */
int skf_expander_fwd(const float* emb, const float* w, const float* bias, int B, int L, int d, float* pre,
                     skf_stream_t stream);
int skf_expander_bwd(const float* dpre, const float* emb, const float* w, int B, int L, int d, float* demb,
                     int demb_accumulate, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                     skf_stream_t stream);

/* ------------------------------------------------------------------ optimizer
 * builders/schedulers.py:13-46 + tf.keras.optimizers.Adam (models/sketchformer.py:112-124,348).
 * step_state is skf_step_state_bytes() of device memory {int64 iterations; float lr, alpha; u32 drop_key}.
 * schedule 0: WarmupDecay p0=d_model, p1=warmup^-1.5; 1: StepDecay p0=init_lr p1=rate p2=steps p3=min_ratio; 2: const p0 */
size_t skf_step_state_bytes(void);
int skf_step_prologue(void* step_state, int schedule, float p0, float p1, float p2, float p3, float beta1, float beta2,
                      unsigned seed, skf_stream_t stream);
int skf_step_epilogue(void* step_state, skf_stream_t stream);
int skf_adam_step(float* w, const float* g, float* m, float* v, size_t n, const void* step_state, float grad_scale,
                  float beta1, float beta2, float eps, skf_stream_t stream);
/* tf.keras.optimizers.SGD(lr_schedule, momentum), nesterov=False: velocity = momentum*velocity - lr*g*grad_scale; w += velocity */
int skf_sgd_momentum_step(float* w, const float* g, float* velocity, size_t n, const void* step_state, float grad_scale,
                          float momentum, skf_stream_t stream);
/* y = inverted dropout of x (n floats, y may alias x) with the mask of (step key, site, element index);
 * applied to an upstream gradient it is the backward.  rate 0 = copy. */
int skf_dropout(const float* x, float* y, size_t n, float rate, unsigned site, const void* step_state, skf_stream_t stream);
/* host helper for parity tests: the keep-mask the kernels derive for (drop_key, site) */
int skf_dropout_keep_mask(unsigned drop_key, unsigned site, float rate, size_t n, unsigned char* out_host);

/* ---- single-query attention + token selection of the KV-cached greedy decode ----
 * skf_attention_decode: scaled_dot_product_attention (builders/utils.py:71-105) for ONE query row per (sample, head):
 *   Q (B, H*dh) row stride ldq; K/V rows of sample b start at K + b*kv_batch_stride, row stride ld_kv, Lk rows used;
 *   key_mask (B, key_mask_ld) bytes, 1 = masked (additive -1e9); keys >= key_limit[b] (or key_limit_all if > 0) masked.
 *   Graph-replayable form: step_dev (device int) non-NULL and K_new/V_new (B, ld_new) given -> the number of keys is
 *   *step_dev + 1 (Lk = cache capacity), the newest key/value row is read from K_new/V_new and appended to the cache at
 *   row *step_dev; limit_from_step: keys >= *step_dev + 1 are masked where key_limit is NULL or negative.
 * skf_decode_init: column 0 of the output = SOS (tokens) or (0,0,1,0,0) (continuous); clears the flags (and *step_dev).
 * skf_decode_embed: decoder input of position *step_dev: table[token] (or Dense(5->d) of the stroke-5 row) * sqrt(d) + pos.
 * skf_decode_select_tokens / _continuous: models/sketchformer.py:285-301 - append argmax (first index on ties) or
 *   (x, y, softmax(pen)); maintain the target padding mask, the sticky EOS flags and done_step (-1 until the stop test
 *   holds).  With step_dev/dyn non-NULL the step, n_valid (dyn[0]) and eos (dyn[1]) are read from device memory and
 *   *step_dev is incremented, so that one captured launch sequence serves every step. */
int skf_attention_decode(const float* Q, int ldq, const float* K, const float* V, int ld_kv, long long kv_batch_stride,
                         const unsigned char* key_mask, int key_mask_ld, const int* key_limit, int key_limit_all, int B,
                         int H, int Lk, int dh, float* O, int ldo, const int* step_dev, const float* K_new,
                         const float* V_new, int ld_new, int limit_from_step, skf_stream_t stream);
int skf_decode_init(long long* tokens, int tok_ld, float* cont, int cont_ld_rows, unsigned char* selfmask, int mask_ld,
                    int* eos_seen, int* done_step, int B, long long sos, int* step_dev, skf_stream_t stream);
int skf_decode_embed(const long long* tokens, const float* cont, int ld, int B, const float* table, int vocab,
                     const float* W, const float* bias, int d, const float* pos, const int* step_dev, float* out,
                     skf_stream_t stream);
int skf_decode_select_tokens(const float* logits, int ld, int B, int V, int n_valid, int step, long long eos,
                             long long* tokens, int tok_ld, unsigned char* selfmask, int mask_ld, int* eos_seen,
                             int* done_step, int* step_dev, const long long* dyn, skf_stream_t stream);
int skf_decode_select_continuous(const float* pred, int ld, int B, int n_valid, int step, float* out, int out_ld_rows,
                                 unsigned char* selfmask, int mask_ld, int* done_step, int* step_dev,
                                 const long long* dyn, skf_stream_t stream);

/* ------------------------------------------------------------------ bf16 path (BASELINE cfg 5)
 * bf16 storage + v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 master weights / optimizer state (SkfConfig.act_dtype
 * = SKF_ACT_BF16).  `void*` tensors below are bf16 (2 bytes per element, pitches in elements); parameters, statistics,
 * losses and parameter gradients stay fp32.  Same reference lines as the fp32 entries they mirror.
 *
 * Dense (builders/layers/transformer.py:154-158,196-197): C[M][N] (+)= A[M][K] . B_nk[N][K]^T (+bias) (act: 0 none, 1 relu,
 * 2 tanh) (relu mask); both operands contraction-contiguous: the forward passes the [out][in] image of the weight, the input
 * gradient the [in][out] image (skf_cast_weight_bf16 makes both from the fp32 master).  lda / ldb multiples of 8 and
 * >= K rounded up to 8 (pad columns must hold zeros: they take part in the contraction); ldc, N multiples of 4.
 * C_f32: optional fp32 copy of the result. */
int skf_gemm_bf16(int M, int N, int K, const void* A, int lda, const void* B_nk, int ldb, void* C, int ldc, const float* bias,
                  int act, const void* relu_src, int ld_relu, int accumulate, float* C_f32, int ldc_f32, skf_stream_t stream);
/* weight gradient dW[P][Q] = sum_r X[r][P] dY[r][Q] (+ column sums of dY = bias gradient): fp32 partial tiles
 * [splits][P][Q] (+ [splits][Q]) into `slab`, reduced by skf_splitk_reduce_batch like the fp32 path's. */
int skf_gemm_bf16_wgrad_splits(int P, int Q, int R);
size_t skf_gemm_bf16_wgrad_workspace_bytes(int P, int Q, int R, int splits);
int skf_gemm_bf16_wgrad_partial(int P, int Q, int R, const void* X, int ldx, const void* dY, int lddy, int splits,
                                int with_bias_grad, float* slab, size_t slab_bytes, int* splits_used_host, skf_stream_t stream);
/* the same + its reduction into the fp32 gradient dW[P][Q] (row stride ldw) and bias_grad[Q] (may be NULL); workspace >=
 * skf_gemm_bf16_wgrad_workspace_bytes(P, Q, R, skf_gemm_bf16_wgrad_splits(P, Q, R)) (fewer splits are used if it is smaller) */
int skf_gemm_bf16_wgrad(int P, int Q, int R, const void* X, int ldx, const void* dY, int lddy, float* dW, int ldw,
                        float* bias_grad, void* workspace, size_t workspace_bytes, skf_stream_t stream);
/* the same over live rows (see skf_row_blocks_build): the dgrad form takes a live-ROW list (granule 1) and runs its m
 * tiles over the compacted live rows - rows of A / C / relu_src are gathered through the list, which costs nothing with
 * per-lane load addresses (zero_dead: the dead rows of C are stored as zeros; never when accumulating); the weight
 * gradient contracts over the live 64-row blocks (256 x 256 kernel only; the 128 x 128 kernel visits every row). */
int skf_gemm_bf16_tile_rows(int M, int N, int K, int act);   /* 256 or 128: the m-tile height skf_gemm_bf16 uses for this problem */
int skf_gemm_bf16_rows(int M, int N, int K, const void* A, int lda, const void* B_nk, int ldb, void* C, int ldc,
                       const float* bias, int act, const void* relu_src, int ld_relu, int accumulate, float* C_f32,
                       int ldc_f32, const int* row_list, int zero_dead, skf_stream_t stream);
/* ReLU sign bits of the bf16 path (ffn: builders/layers/transformer.py:196-197): bit (n & 7) of byte [m][n >> 3] (row pitch
 * ld_bits bytes) = "C[m][n] > 0".  A relu forward launch writes them (relu_bits_out), the input-gradient launch of the layer
 * reads them (relu_bits_in) instead of the hidden tensor: same result, 1/16 of the bytes.  N must be a multiple of 8. */
size_t skf_gemm_bf16_relu_bits_bytes(int M, int N);
int skf_gemm_bf16_bits(int M, int N, int K, const void* A, int lda, const void* B_nk, int ldb, void* C, int ldc,
                       const float* bias, int act, const void* relu_src, int ld_relu, int accumulate, float* C_f32, int ldc_f32,
                       const int* row_list, int zero_dead, void* relu_bits_out, const void* relu_bits_in, int ld_bits,
                       skf_stream_t stream);
int skf_gemm_bf16_wgrad_partial_rows(int P, int Q, int R, const void* X, int ldx, const void* dY, int lddy, int splits,
                                     int with_bias_grad, float* slab, size_t slab_bytes, int* splits_used_host,
                                     const int* row_blocks_64, skf_stream_t stream);
int skf_gemm_bf16_wgrad_rows(int P, int Q, int R, const void* X, int ldx, const void* dY, int lddy, float* dW, int ldw,
                             float* bias_grad, void* workspace, size_t workspace_bytes, const int* row_blocks_64,
                             skf_stream_t stream);
/* scaled_dot_product_attention (builders/utils.py:71-105), streaming / online softmax, head size 64, any Lq / Lk;
 * mask semantics and `stats` as skf_attention_fwd.  The backward is two passes (dQ, then dK / dV) and needs
 * skf_attention_bf16_bwd_workspace_bytes of scratch (rowsum(dO o O)).  O_lo (optional, shape and pitch of O): the forward
 * stores the bf16 rounding residual of O there and the backward adds it back when it forms rowsum(dO o O) - that sum is
 * subtracted from dO.V^T, which it nearly cancels where the softmax gradient is small, and 8 significand bits of O leave an
 * error larger than the result in such rows. */
int skf_attention_bf16_fwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const unsigned char* key_mask,
                           int key_mask_ld, int causal, int B, int H, int Lq, int Lk, int dh, void* O, int ldo, void* O_lo,
                           float* stats, skf_stream_t stream);
size_t skf_attention_bf16_bwd_workspace_bytes(int B, int H, int Lq);
int skf_attention_bf16_bwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O, int ldo,
                           const void* O_lo, const void* dO, int lddo, const float* stats, const unsigned char* key_mask, int key_mask_ld,
                           int causal, int B, int H, int Lq, int Lk, int dh, void* dQ, int lddq, void* dK, int lddk, void* dV,
                           int lddv, void* workspace, size_t workspace_bytes, skf_stream_t stream);
/* the same with per-sample live query counts (skf_target_live_len): query rows >= q_live_len[b] have dO == 0 exactly, so the
 * dQ pass stores zeros for their blocks and the dK / dV pass stops at the last live block */
int skf_attention_bf16_bwd_rows(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O, int ldo,
                                const void* O_lo, const void* dO, int lddo, const float* stats, const unsigned char* key_mask,
                                int key_mask_ld, int causal, int B, int H, int Lq, int Lk, int dh, void* dQ, int lddq, void* dK,
                                int lddk, void* dV, int lddv, void* workspace, size_t workspace_bytes, const int* q_live_len,
                                skf_stream_t stream);
/* the same with the sorted sample list of skf_sample_order (or NULL): the (sample, head) pairs of the list are dealt over the 8 XCDs and,
 * inside one, over its shader engines (B a multiple of 8, ignored otherwise); a numbering, never a result bit */
int skf_attention_bf16_fwd_ordered(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const unsigned char* key_mask,
                                   int key_mask_ld, int causal, int B, int H, int Lq, int Lk, int dh, void* O, int ldo, void* O_lo,
                                   float* stats, const int* sample_order, skf_stream_t stream);
int skf_attention_bf16_bwd_ordered(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O, int ldo,
                                   const void* O_lo, const void* dO, int lddo, const float* stats, const unsigned char* key_mask,
                                   int key_mask_ld, int causal, int B, int H, int Lq, int Lk, int dh, void* dQ, int lddq, void* dK,
                                   int lddk, void* dV, int lddv, void* workspace, size_t workspace_bytes, const int* q_live_len,
                                   const int* sample_order, skf_stream_t stream);
/* row kernels: the fp32 entries of the same name with bf16 activations (d in {128, 256, 512, 1024}) */
int skf_embed_fwd_bf16(const long long* tokens, int tok_ld, int B, int L, const float* table, int vocab, int d, const float* pos,
                       void* out, float rate, unsigned site, const void* step_state, skf_stream_t stream);
int skf_embed_bwd_sorted_bf16(const void* ws, int B, int L, const void* dx, int vocab, int d, float* dtable, float rate,
                              unsigned site, const void* step_state, skf_stream_t stream);
int skf_layernorm_residual_fwd_bf16(const void* x, void* y_inout_z, const float* gamma, const float* beta, void* out,
                                    float* stats, int rows, int d, float rate, unsigned site, const void* step_state,
                                    skf_stream_t stream);
size_t skf_layernorm_bwd_bf16_workspace_bytes(int rows, int d);
int skf_layernorm_residual_bwd_bf16(const void* dout, const void* z, const float* stats, const float* gamma, void* dz, void* dy,
                                    float* dgamma, float* dbeta, int rows, int d, float rate, unsigned site,
                                    const void* step_state, void* workspace, size_t workspace_bytes, skf_stream_t stream);
/* rows = B * rows_per_sample; row t of sample b with t >= live_len[b] has dout == 0 exactly: zeros are stored for it */
int skf_layernorm_residual_bwd_bf16_rows(const void* dout, const void* z, const float* stats, const float* gamma, void* dz,
                                         void* dy, float* dgamma, float* dbeta, int rows, int d, float rate, unsigned site,
                                         const void* step_state, void* workspace, size_t workspace_bytes, const int* live_len,
                                         int rows_per_sample, skf_stream_t stream);
/* logits (rows, ld) bf16, even ncls <= ld <= 2048; the gradient is written in place, columns [ncls, ld) as zeros */
int skf_softmax_ce_bf16(void* logits, int ld, int rows, int ncls, const long long* target, int tgt_ld, int tgt_cols, int tgt_off,
                        int mask_pad, float scale, float* row_loss, float* row_hit, int write_grad, skf_stream_t stream);
int skf_pool_fwd_bf16(const void* u, const float* Vw, const void* x, int B, int L, int U, int d, float* a_out, float* emb,
                      skf_stream_t stream);
int skf_pool_bwd_bf16(void* u_inout_dpre, const float* Vw, const void* x, const float* a, const float* demb, int B, int L, int U,
                      int d, void* dx, float* dV, void* workspace, size_t workspace_bytes, skf_stream_t stream);
int skf_expander_fwd_bf16(const float* emb, const float* w, const float* bias, int B, int L, int d, void* pre, skf_stream_t stream);
int skf_expander_bwd_bf16(const void* dpre, const float* emb, const float* w, int B, int L, int d, float* demb,
                          int demb_accumulate, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                          skf_stream_t stream);
/* fp32 master weight [R][C] (row stride ld_src) -> bf16 images dst [R][ld_dst] and / or dst_t [C][ld_t] (NULL = skip);
 * pad columns are written as zeros.  Plain casts for activations at the fp32 <-> bf16 seams. */
int skf_cast_weight_bf16(const float* src, int R, int C, int ld_src, void* dst, int ld_dst, void* dst_t, int ld_t,
                         skf_stream_t stream);
/* many images in one launch: a DEVICE array of descriptors whose block_begin fields are the running sum of
 * skf_cast_weight_bf16_blocks(R, C, ld_dst, ld_t, &blocks_x) (dst / dst_t may be NULL individually) */
typedef struct SkfCastDesc {
  const float* src; void* dst; void* dst_t;
  int32_t R, C, ld_src, ld_dst, ld_t, block_begin, blocks_x, pad;
} SkfCastDesc;
int skf_cast_weight_bf16_blocks(int R, int C, int ld_dst, int ld_t, int* blocks_x);
int skf_cast_weight_bf16_batch(const SkfCastDesc* descs_dev, int n, int total_blocks, skf_stream_t stream);
int skf_cast_f32_to_bf16(const float* src, void* dst, size_t n, skf_stream_t stream);
int skf_cast_bf16_to_f32(const void* src, float* dst, size_t n, skf_stream_t stream);

/* ------------------------------------------------------------------ the train step
 * Transformer.build_model / call / model_trainer, models/sketchformer.py:63-147, 313-349. */
typedef struct SkfConfig {
  /* sizeof(SkfConfig) as the CALLER compiled / declared it.  skf_config_validate (and with it every entry that takes a
   * config) refuses any other value: a binding that is a field short or long is an error, never a read of heap garbage.
   * skf_config_size() returns the library's figure. */
  uint32_t struct_size;
  int32_t batch, seq_len, d_model, num_heads, dff, num_layers;
  int32_t vocab_size, n_classes, lowerdim, attn_version;
  int32_t continuous, blind_decoder_mask, max_pos;
  float dropout_rate, recon_weight, class_weight;
  int32_t schedule;
  float sched_p0, sched_p1, sched_p2, sched_p3;
  float beta1, beta2, eps;
  uint32_t seed;
  int32_t use_graph; /* 0: eager launches, weight gradients on a side stream; 1: the step captured into hipGraphs on ONE stream and replayed;
                      * 2: the two-stream step captured (the side stream is forked / joined inside the capture; the first step runs eagerly) -
                      * only with SKF_MODEL_TWO_STREAM_GRAPH set through skf_model_set_flags, otherwise mode 2 issues the same launches
                      * eagerly (bit-equal results): hipGraphLaunch of a multi-branch graph can crash inside the HIP runtime, see the flag.
                      * The default is 0 and is also the fastest */
  int32_t optimizer; /* 0 = Keras Adam (beta1, beta2, eps above), 1 = Keras SGD with momentum (models/sketchformer.py:120-126) */
  float momentum;
  int32_t class_buffer_layers; /* Dense(lowerdim, relu) + Dropout(class_dropout) layers before classify (models/sketchformer.py:44-45,101-104) */
  float class_dropout;
  /* models/sketchformer.py:42,46,76-108: the class head needs lowerdim > 0 and do_classification; the decoder / output
   * layer / expander need do_reconstruction; lowerdim == 0 = no bottleneck, the decoder attends to the encoder output */
  int32_t do_classification, do_reconstruction;
  int32_t gemm_precision; /* SKF_PREC_*: arithmetic of every Dense / attention matmul of the step (0 = fp32 MFMA) */
  /* SKF_ACT_F32 (0): fp32 tensors everywhere (the reference's arithmetic, cfg 1-4).  SKF_ACT_BF16 (1): bf16 activations
   * and weight images, bf16 MFMA with fp32 accumulation, fp32 master weights / Adam (BASELINE cfg 5); token mode with the
   * default structure (attn_version 1, bottleneck + classifier + decoder, no class buffers), head size 64 */
  int32_t act_dtype;
} SkfConfig;
#define SKF_ACT_F32 0
#define SKF_ACT_BF16 1

typedef struct SkfParamEntry {
  char name[96];
  int64_t offset;      /* in floats, into the flat parameter / gradient / moment buffers */
  int32_t rows, cols;  /* logical 2-D shape (1-D variables: rows = 1) */
  int32_t row_stride;  /* floats between rows (fused wq|wk|wv blocks are strided views) */
} SkfParamEntry;

typedef struct SkfModel SkfModel;

size_t skf_config_size(void);                                   /* sizeof(SkfConfig) of this build of the library */
int skf_config_validate(const SkfConfig* cfg);
size_t skf_model_param_floats(const SkfConfig* cfg);            /* length of the flat buffers */
int skf_model_param_entries(const SkfConfig* cfg, SkfParamEntry* out_host, int max_entries); /* returns count */
size_t skf_model_workspace_bytes(const SkfConfig* cfg);

int skf_model_create(const SkfConfig* cfg, SkfModel** out);
void skf_model_destroy(SkfModel* m);
/* Per-model switches (state of THIS model object, not of the library).  SKF_MODEL_DECODE_LAYERWISE: skf_model_greedy_decode
 * runs the layer-by-layer path (one launch per operator, the form the oracle tests pinned first) instead of the one launch
 * per position of skf_decode_fused.hip - same tokens, kept as the cross-check of the fused kernel. */
#define SKF_MODEL_DECODE_LAYERWISE 1u
/* SKF_MODEL_FFN_LAUNCHES: the feed-forward blocks of the step run as separate Dense / LayerNorm launches instead of the one launch
 * per direction of skf_ffn_fused_fwd_f32 / _bwd_f32 (same arithmetic, the sums in a different order): cross-check and A/B. */
#define SKF_MODEL_FFN_LAUNCHES 2u
/* SKF_MODEL_TWO_STREAM_GRAPH: opt-in for SkfConfig.use_graph = 2.  Root cause of the crash recorded in profiles/r05y_two_stream_graph_crash.txt
 * (round 6, from the disassembly of the runtime that ships with torch 2.10+rocm7.0): hip::Graph::UpdateStreams walks the graph exec's internal
 * stream vector WITHOUT a bound while it skips entries that compare equal to the launch stream (the comparison is a virtual call on a
 * member object of the two streams - which object, the binary does not say) - one such entry and it dereferences whatever lies behind
 * the vector.  Which streams compare equal is decided inside the runtime, so a process
 * whose runtime state makes an internal stream of the exec compare equal to the launch stream faults at a replay.  Nothing this library
 * creates or destroys is read on that path.  tools/micro/graph_parallel_stream_alias.hip tries to provoke it with plain HIP calls
 * (stream counts, build / replay / destroy cycles, leaked execs, stream 0): none of its 72 cases faults (profiles/r06b_graph_alias.txt) -
 * the trigger needs the history of the process that showed it (pytest over two test files) and was not isolated.  Set the flag only in
 * a process of its own (the test and bench.py use a child process). */
#define SKF_MODEL_TWO_STREAM_GRAPH 4u
int skf_model_set_flags(SkfModel* m, uint32_t flags);
/* params/grads/adam_m/adam_v: skf_model_param_floats floats each; pos: (max_pos, d_model) table
 * (builders/utils.py:17-32, computed by the host in float64 like the reference);
 * metrics: 32 floats; step_state: skf_step_state_bytes bytes.  All device memory owned by the caller. */
int skf_model_bind(SkfModel* m, float* params, float* grads, float* adam_m, float* adam_v, const float* pos,
                   void* workspace, size_t workspace_bytes, float* metrics, void* step_state);
/* forward only (Transformer.call); training != 0 enables dropout.  inp/tar: (B,L) int64 tokens, or (B,L,5) float32
 * stroke-5 rows when cfg.continuous; tar has row stride tar_ld (in sequence positions), its first L-1 positions feed the decoder. */
int skf_model_forward(SkfModel* m, const void* inp, const void* tar, int tar_ld, int training,
                      skf_stream_t stream);
/* Every call that takes inp / tar / labels copies them into the model's workspace first (one launch on `stream`) and records the model's
 * "inputs staged" event behind that copy.  A caller that passes DEVICE tensors which it will overwrite for the next batch makes the
 * stream that overwrites them wait here - the step itself is not waited for, and nothing has to be cloned in front of the call. */
int skf_model_wait_inputs_staged(SkfModel* m, skf_stream_t stream);
/* model_trainer minus apply_gradients: forward, losses, metrics, backward into `grads`. labels (B,1) int64. */
int skf_model_forward_backward(SkfModel* m, const void* inp, const void* tar, int tar_ld,
                               const long long* labels, skf_stream_t stream);
/* optimizer.apply_gradients with grads pre-multiplied by grad_scale (1/world_size under data parallelism) */
int skf_model_apply_gradients(SkfModel* m, float grad_scale, skf_stream_t stream);
/* ---- data-parallel hooks: the flat gradient buffer becomes final in pieces ("buckets", in production order:
 * [decoder embedding .. output layer] after the decoder backward, then [encoder .. expander]).  A caller that
 * all-reduces gradients makes its communication stream wait for a bucket, reduces that slice of `grads`, and applies the
 * optimizer per slice - the all-reduce of bucket 0 overlaps the encoder backward, that of bucket 1 the optimizer sweep of
 * bucket 0.  skf_model_grad_buckets returns the number of buckets (1 when the step is replayed from hipGraphs);
 * offsets/counts in floats.  skf_model_apply_gradients_range == skf_model_apply_gradients on one slice
 * (last != 0: also ends the step, iterations += 1). */
int skf_model_grad_buckets(SkfModel* m, int max_buckets, size_t* offsets_host, size_t* counts_host);
int skf_model_wait_grad_bucket(SkfModel* m, int bucket, skf_stream_t stream);
int skf_model_apply_gradients_range(SkfModel* m, size_t offset, size_t count, float grad_scale, int last,
                                    skf_stream_t stream);
/* ---- inference API of the plugin (models/sketchformer.py:162-168, 201-311) ----
 * skf_model_encode: encode_from_seq / predict_class - encoder + bottleneck + classifier with dropout off; results in
 *   the buffers "embedding" (B,d), "class_probs" (B,C), "enc_output" (B*L,d).
 * skf_model_greedy_decode: predict_from_embedding - greedy reconstruction with a per-layer K/V cache.
 *   embedding: (B,d) device floats, or NULL to use the model's own "embedding" buffer (after skf_model_encode);
 *   expected_len_host: per-sample key limit of the cross attention, used only when blind_decoder_mask == 0 (NULL = i+1);
 *   only the first n_valid samples take part in the stop test; sos/eos: tokenizer ids (ignored in continuous mode);
 *   out: device buffer (B, max_steps+1) int64 tokens [column 0 = SOS] or (B, max_steps+1, 5) float stroke-5 rows
 *   [row 0 = (0,0,1,0,0)]; *out_len_host = number of valid columns (the reference's output length).
 *   Blocking: synchronises the stream every 8 tokens to test the stop condition. */
int skf_model_encode(SkfModel* m, const void* inp, skf_stream_t stream);
int skf_model_greedy_decode(SkfModel* m, const float* embedding, const int* expected_len_host, int n_valid,
                            long long sos, long long eos, int max_steps, void* out, int* out_len_host,
                            skf_stream_t stream);
/* look up an internal activation by name ("logits", "class_probs", "embedding", "enc_output", ...); skf_model_buffer serves
 * fp32 buffers, skf_model_buffer_info any buffer with its row pitch (elements) and element type (bf16 models keep their
 * activations in bf16) */
int skf_model_buffer_info(SkfModel* m, const char* name, void** ptr, int* rows, int* cols, int* ld, int* is_bf16);
int skf_model_buffer(SkfModel* m, const char* name, float** ptr, int* rows, int* cols);

#ifdef __cplusplus
}
#endif
#endif /* SKF_H_ */
