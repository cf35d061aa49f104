// Does v_mfma_f32_16x16x4_f32 overlap with VALU work on gfx950?  (f32-input MFMA runs at the f32 vector rate.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: MFMA only, 1: VALU only, 2: both interleaved in one wave, 3: even waves MFMA / odd waves VALU
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, y0 = 1.f, y1 = 2.f, y2 = 3.f, y3 = 4.f, y4 = 5.f, y5 = 6.f, y6 = 7.f, y7 = 8.f;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a3, 0, 0, 0);
    }
    if (do_v) {   // 32 independent-ish FMAs (8 chains x 4)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        y0 = fmaf(y0, 1.0001f, x); y1 = fmaf(y1, 1.0001f, x); y2 = fmaf(y2, 1.0001f, x); y3 = fmaf(y3, 1.0001f, x);
        y4 = fmaf(y4, 1.0001f, x); y5 = fmaf(y5, 1.0001f, x); y6 = fmaf(y6, 1.0001f, x); y7 = fmaf(y7, 1.0001f, x);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
}

template <int MODE>
float run(int threads, int iters) {
  float* d; hipMalloc(&d, 1024 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d); return ms * 1e3f;
}

int main() {
  const int it = 20000;
  printf("256 threads/WG (1 wave/SIMD), %d iters: per iter = 4 MFMA(16x16x4 f32) and/or 32 v_fma\n", it);
  printf("  MFMA only        %8.1f us\n", run<0>(256, it));
  printf("  VALU only        %8.1f us\n", run<1>(256, it));
  printf("  both, one wave   %8.1f us\n", run<2>(256, it));
  printf("512 threads/WG (2 waves/SIMD)\n");
  printf("  MFMA only        %8.1f us\n", run<0>(512, it));
  printf("  VALU only        %8.1f us\n", run<1>(512, it));
  printf("  both, every wave %8.1f us\n", run<2>(512, it));
  printf("  even MFMA/odd VALU %6.1f us\n", run<3>(512, it));
  return 0;
}
