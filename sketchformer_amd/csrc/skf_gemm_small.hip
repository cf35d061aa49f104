// Small-problem fp32 GEMM (M*N*K <= 2^25): the classifier / class-buffer / bottleneck-projection matmuls of the
// train step (batch-sized M) and the one-row-per-sample matmuls of the greedy decode.  The big kernels tile for
// 25k-row operands; on these shapes they run 2..6 workgroups with scalar loads (20..45 us).  Here: 32x32 output
// tiles (one 16x16 MFMA block per wave), BK = 32, any operand layout through element strides, all epilogues of
// skf_gemm_f32 (bias, relu/tanh, relu-grad mask, accumulate) plus the bias gradient (column sums of B) in the
// same launch.  C = opA(A)[M,K] . opB(B)[K,N].
#include "skf_common.h"
#include "skf_gemm_params.h"

namespace {

struct SmallParams {
  const float* A; const float* B; float* C;
  int M, N, K;
  long long sam, sak, sbk, sbn;   // element strides: A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
  int ldc;
  const float* bias; int act;
  const float* relu_src; int ld_relu;
  int accumulate;
  float* bias_grad; int bias_grad_accumulate;   // column sums of B over k (needs sbn == 1 semantics only for speed)
  int a_kfast, b_kfast;                         // which index runs fastest over the threads of a tile load
};

constexpr int SBK = 32;

__global__ __launch_bounds__(256) void gemm_small_kernel(SmallParams p) {
  __shared__ float As[32][SBK + 1];   // [m][k]
  __shared__ float Bs[SBK][32 + 1];   // [k][n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int wm = (wave >> 1) * 16, wn = (wave & 1) * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;                   // thread n (< 32) of the first row of tiles: column sum of B
  const bool do_colsum = p.bias_grad != nullptr && blockIdx.y == 0;
  float av[4], bv[4];
  auto load_slab = [&](int k0) {      // global -> registers (guards give exact zeros outside the matrices)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int e = tid + v * 256;
      const int am = p.a_kfast ? e >> 5 : e & 31, ak = p.a_kfast ? e & 31 : e >> 5;
      const int bk = p.b_kfast ? e & 31 : e >> 5, bn = p.b_kfast ? e >> 5 : e & 31;
      av[v] = (m0 + am < p.M && k0 + ak < p.K) ? p.A[(long long)(m0 + am) * p.sam + (long long)(k0 + ak) * p.sak] : 0.f;
      bv[v] = (k0 + bk < p.K && n0 + bn < p.N) ? p.B[(long long)(k0 + bk) * p.sbk + (long long)(n0 + bn) * p.sbn] : 0.f;
    }
  };
  load_slab(0);
  for (int k0 = 0; k0 < p.K; k0 += SBK) {
    __syncthreads();                  // previous slab fully consumed
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int e = tid + v * 256;
      const int am = p.a_kfast ? e >> 5 : e & 31, ak = p.a_kfast ? e & 31 : e >> 5;
      const int bk = p.b_kfast ? e & 31 : e >> 5, bn = p.b_kfast ? e >> 5 : e & 31;
      As[am][ak] = av[v];
      Bs[bk][bn] = bv[v];
    }
    __syncthreads();
    if (k0 + SBK < p.K) load_slab(k0 + SBK);   // next slab in flight during the MFMAs
#pragma unroll
    for (int s = 0; s < SBK / 4; ++s)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(As[wm + i][4 * s + g], Bs[4 * s + g][wn + i], acc, 0, 0, 0);
    if (do_colsum && tid < 32) {
#pragma unroll
      for (int k = 0; k < SBK; ++k) csum += Bs[k][tid];
    }
  }
  // lane (i,g) holds C[m0+wm+4g+r][n0+wn+i]
  const int n = n0 + wn + i;
  if (n < p.N) {
    const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm + 4 * g + r;
      if (m < p.M) {
        float v = acc[r] + bias;
        if (p.act == 1) v = fmaxf(v, 0.f);
        else if (p.act == 2) v = tanhf(v);
        if (p.relu_src) v = p.relu_src[(size_t)m * p.ld_relu + n] > 0.f ? v : 0.f;
        float* dst = p.C + (size_t)m * p.ldc + n;
        *dst = p.accumulate ? *dst + v : v;
      }
    }
  }
  if (do_colsum && tid < 32 && n0 + tid < p.N) {
    float* dst = p.bias_grad + n0 + tid;
    *dst = p.bias_grad_accumulate ? *dst + csum : csum;
  }
}

}  // namespace

// Returns SKF_OK and sets *handled = 1 when the small-problem path applies.
int skf_gemm_small_dispatch(const GemmParams& p, int a_kcontig, int b_kcontig, float* bias_grad, int bias_grad_accumulate,
                            hipStream_t st, int* handled) {
  *handled = 0;
  const char* off = skf_knob("SKF_GEMM_NO_SMALL");
  if (off && off[0] == '1') return SKF_OK;
  if ((double)p.M * p.N * p.K > 33554432.0) return SKF_OK;
  // a long contraction over few outputs (weight gradient of a narrow layer over the B*L rows, e.g. Dense(d -> 5) of the
  // continuous mode: 256 x 5 x 25472) would run on a handful of workgroups here (measured 150 us at cfg 3): leave it to
  // the split-K path
  if (p.K > 4096 && (double)p.M * p.N <= 65536.0) return SKF_OK;
  *handled = 1;
  SmallParams q{};
  q.A = p.A; q.B = p.B; q.C = p.C; q.M = p.M; q.N = p.N; q.K = p.K; q.ldc = p.ldc;
  q.sam = a_kcontig ? p.lda : 1; q.sak = a_kcontig ? 1 : p.lda;     // A stored [M][K] or [K][M]
  q.sbk = b_kcontig ? 1 : p.ldb; q.sbn = b_kcontig ? p.ldb : 1;     // B stored [N][K] or [K][N]
  q.a_kfast = a_kcontig; q.b_kfast = b_kcontig;
  q.bias = p.bias; q.act = p.act; q.relu_src = p.relu_src; q.ld_relu = p.ld_relu; q.accumulate = p.accumulate;
  q.bias_grad = bias_grad; q.bias_grad_accumulate = bias_grad_accumulate;
  dim3 grid(skf_cdiv(p.N, 32), skf_cdiv(p.M, 32)), block(256);
  SkfProfScope ps(st, "gemm_small<32x32>", 2.0 * p.M * p.N * p.K, 4.0 * ((double)p.M * p.K + (double)p.K * p.N + (double)p.M * p.N));
  hipLaunchKernelGGL(gemm_small_kernel, grid, block, 0, st, q);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
