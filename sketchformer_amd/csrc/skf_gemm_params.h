// Shared between the generic tiled GEMM (skf_gemm.hip) and the weight-stationary one (skf_gemm_ws.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string>

struct GemmParams {
  const float* A; const float* B; float* C;
  int M, N, K;
  int lda, ldb, ldc;
  const float* bias;
  int act;                 // 0 none, 1 relu, 2 tanh
  const float* relu_src;   // optional: C *= (relu_src > 0)
  int ld_relu;
  int accumulate;          // C += result
  int a_vec, b_vec;        // 16-byte vector loads legal
  // split-K
  int k_chunk;             // k range per blockIdx.z (multiple of BK)
  float* slab;             // [splits][M][N] raw partial tiles (split-K only)
  float* colsum_slab;      // [splits][N] partial column sums of B (bias grad), or null
  int tiles_m, tiles_n;
  long long* dbg;          // diagnostics only: per-phase s_memtime stamps of a few workgroups (-DSKF_MEASURE=1 builds only, env SKF_GEMM_DBG; always null in the shipped library)
  int xcd_remap;           // ws kernel: XCD-contiguous logical ids (env SKF_WS_XCD, A/B knob)
  int precision;           // SKF_PREC_*: 0 fp32 MFMA, 6 / 3 = split fp32 operands on the bf16 matrix cores
  // Row-block list (skf_row_blocks_build): {n_live, n_blocks, live block ids ..., dead block ids ...} over blocks of
  // `row_block_rows` consecutive rows of the M (dgrad: output / A rows) or K (wgrad: contraction rows) dimension whose A
  // (and, for the wgrad, dY) rows are known to be all zero when dead.  The weight-stationary bf16x kernels visit only the
  // live blocks (dead output rows are stored as zeros, or left alone when accumulating); the generic tiled kernel skips the
  // tiles without a live 16-row block (dgrad form); every other kernel ignores it.
  const int* row_blocks;
  int row_block_rows;
  int ablate;              // generic kernel, diagnostics only (env SKF_GEMM_ABLATE): 1 no MFMA, 2 no C store, 3 no global reload
  // ReLU sign bits (split-arithmetic weight-stationary kernels only; skf_gemm_relu_bits_bytes): a forward launch with
  // act = relu leaves one bit per output element ("> 0") in the layout of its own tiles - word [tile][column wave][r * NB + nb]
  // = ballot over the wave's lanes - and the input-gradient launch of the SAME (M, N, K) multiplies by them instead of
  // re-reading the hidden tensor (52 MB per launch at the cfg-2 size -> 1.6 MB)
  unsigned long long* relu_bits_out;
  const unsigned long long* relu_bits_in;
  // Last slice of a contraction that is not a multiple of the kernel's K (the logits input gradient: K = vocabulary = 1004 =
  // 512 + 492): the launch runs with K = 512, weight elements k >= k_valid count as zeros and the A rows are read up to
  // a_cut bytes short of the matrix end (the columns behind k_valid of a row are whatever follows it - the next row -
  // multiplied by those zeros; behind the last row they are out of the descriptor's range and read as 0).  0 = plain launch.
  int k_valid, a_cut;
  // Residual + dropout + LayerNorm epilogue (split-arithmetic weight-stationary kernel, K = N = 128: one workgroup owns whole
  // output rows; skf_gemm_ln_residual_f32): C = z = ln_x + dropout(A.B + bias), ln_out = LayerNorm(z) * gamma + beta,
  // ln_stats[row] = (mean, rstd).  ln_out == null: plain launch.
  const float* ln_x; const float* ln_gamma; const float* ln_beta;
  float* ln_out; float* ln_stats;
  float ln_rate; unsigned ln_site; const void* ln_state;   // dropout of the product (SkfStepState*, read on the device)
};

// weight-stationary fast path; sets *handled when it launched the problem
int skf_gemm_ws_dispatch(const GemmParams& p, int a_kcontig, int b_kcontig, hipStream_t st, int* handled);
// the same problem on the bf16 matrix cores with split fp32 operands (pieces = 3: six products, 2: three products)
int skf_gemm_wsx_launch(const GemmParams& p, int b_kcontig, int pieces, hipStream_t st);
size_t skf_gemm_wsx_relu_bits_bytes(int M, int N, int K);
// wgrad (X^T.dY) fast path writing the split-K slab; sets *handled when it launched the problem
int skf_gemm_wgrad_dispatch(const GemmParams& p, int a_kcontig, int b_kcontig, int splits, hipStream_t st, int* handled);
// several wgrad fast-path problems (k_chunk, slab, colsum_slab, row_blocks set as for skf_gemm_wgrad_dispatch) in one launch
int skf_gemm_wgrad_group_dispatch(const GemmParams* ps, const int* splits, int n, hipStream_t st, int* handled);
// small-problem path (M*N*K <= 2^25): any layout, all epilogues, optional bias gradient (column sums of B) in the same launch
int skf_gemm_small_dispatch(const GemmParams& p, int a_kcontig, int b_kcontig, float* bias_grad, int bias_grad_accumulate,
                            hipStream_t st, int* handled);
