// The position-wise feed-forward block of builders/layers/transformer.py:194-198 (`point_wise_feed_forward_network`:
// Dense(dff, relu) -> Dense(d_model)) as ONE launch per direction (skf_ffn_fused.hip); shared with skf_model.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct FfnFusedParams {
  const float* A; int lda; int M;      // [M][128] input rows: x (forward) or dy (backward)
  const char* img1; const char* img2;  // pre-split weight images (skf_ffn_weight_images): B1 [128][512], B2 [512][128]
  const float* bias1; const float* bias2;   // forward only
  float* H;                            // [M][512] hidden tensor: h = relu(x.W1 + b1) (forward) / dh (backward), pitch 512
  unsigned long long* bits_out;        // forward: sign bits of h: word [tile][block][wave][r], bit 16g + i <-> h[16 tile + i][128 block + 16 wave + 4g + r] > 0
  const unsigned long long* bits_in;   // backward: the same words
  float* C;                            // forward: z = res + dropout(y) [M][128]; backward: dx (+)= dh.W1^T [M][128]
  int accumulate;                      // backward: C += result (else C = result)
  const float* res;                    // forward: residual rows [M][128]
  const float* gamma; const float* beta; float* out; float* stats;   // forward: LayerNorm(z)
  float rate; unsigned site; const void* state;                     // forward: dropout of y (SkfStepState*)
  const int* row_blocks;               // 16-row block list (skf_row_blocks_build) or null
  // forward starting at the attention output (pre_img != null): A = a [M][128]; x1 = LayerNorm(pre_res + dropout(a . Bp + pre_bias)) is the block's input
  const char* pre_img; const float* pre_bias; const float* pre_res; const float* pre_gamma; const float* pre_beta; unsigned pre_site;
  float* pre_z; float* pre_out; float* pre_stats;
  // forward with a chained projection (img3 != null): out2[M][n2] = out . B3 + bias3, B3 [128][n2] as a pre-split image, n2 in {128, 256, 384}
  const char* img3; const float* bias3; float* out2; int n2;
  // backward with the LayerNorm-backward prologue (ln_dout != null): A is not read; gamma / rate / site / state are the LayerNorm's
  const float* ln_dout; const float* ln_z; const float* ln_stats;   // gradient of the LayerNorm output, z = x + dropout(y), (mean, rstd)
  float* ln_dy;                        // dy = dropout'(LayerNorm'(dout)) [M][128] (the second Dense's weight gradient reads it)
  float* ln_part;                      // [gridDim.x][2][128] partial column sums (dgamma, dbeta)
};

// pieces = 3 (six products) or 2 (three products); direction 0 forward, 1 backward
int skf_ffn_fused_launch(const FfnFusedParams& p, int pieces, int direction, hipStream_t st);
