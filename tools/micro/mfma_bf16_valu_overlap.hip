// Does v_mfma_f32_16x16x32_bf16 (16 cycles) overlap with VALU work on gfx950 - inside one wave, and between the two
// waves of a SIMD?   hipcc --offload-arch=gfx950 -O3 -w tools/micro/mfma_bf16_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>   // 0: MFMA only, 1: VALU only, 2: NV VALU ops after every MFMA (one wave), 3: waves 0-3 MFMA / waves 4-7 VALU,
                              // 4: blocks of 16 MFMAs then 16*NV VALU (one wave)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const u32x4 xb = {threadIdx.x, 1, 2, 3};
  const bf16x8 x = __builtin_bit_cast(bf16x8, xb);
  float s = threadIdx.x * 1e-3f, y[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  const bool do_m = MODE == 0 || MODE == 2 || MODE == 4 || (MODE == 3 && wave < 4);
  const bool do_v = MODE == 1 || MODE == 2 || MODE == 4 || (MODE == 3 && wave >= 4);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a0, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) y[(4 * u + v) & 7] = fmaf(y[(4 * u + v) & 7], 1.0001f, s);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a1, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) y[(4 * u + v + 2) & 7] = fmaf(y[(4 * u + v + 2) & 7], 1.0001f, s);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a2, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) y[(4 * u + v + 4) & 7] = fmaf(y[(4 * u + v + 4) & 7], 1.0001f, s);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a3, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) y[(4 * u + v + 6) & 7] = fmaf(y[(4 * u + v + 6) & 7], 1.0001f, s);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a3, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (do_v) {
#pragma unroll
        for (int u = 0; u < 16 * NV; ++u) y[u & 7] = fmaf(y[u & 7], 1.0001f, s);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + y[0] + y[1] + y[2] + y[3] + y[4] + y[5] + y[6] + y[7];
}

template <int MODE, int NV>
float run(int threads, int iters) {
  float* d; hipMalloc(&d, 1024 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d); return ms * 1e3f;
}


typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int CHAINS>   // 8 v_mfma_f32_32x32x16_bf16 per iteration on CHAINS accumulators, NV v_fma after each
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  f32x16 a0 = {0}, a1 = {0};
  const u32x4 xb = {threadIdx.x, 1, 2, 3};
  const bf16x8 x = __builtin_bit_cast(bf16x8, xb);
  float s = threadIdx.x * 1e-3f, y[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (CHAINS == 1 || (u & 1) == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a0, 0, 0, 0);
      else a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a1, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) y[(u + v) & 7] = fmaf(y[(u + v) & 7], 1.0001f, s);
    }
  }
  float r = 0; for (int j = 0; j < 16; ++j) r += a0[j] + a1[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r + y[0] + y[1] + y[2] + y[3] + y[4] + y[5] + y[6] + y[7];
}
template <int NV, int CHAINS>
float run32(int iters) {
  float* d; hipMalloc(&d, 1024 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k32<NV, CHAINS>), dim3(256), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k32<NV, CHAINS>), dim3(256), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d); return ms * 1e3f;
}

int main() {
  const int it = 20000;
  printf("per iteration: 16 v_mfma_f32_16x16x32_bf16 and/or 16*NV v_fma_f32; %d iterations, 256 workgroups\n", it);
  printf("1 wave/SIMD:  MFMA only %7.1f us | VALU only NV=2 %7.1f, NV=3 %7.1f us\n", run<0, 2>(256, it), run<1, 2>(256, it), run<1, 3>(256, it));
  printf("1 wave/SIMD:  interleaved (NV after each MFMA): NV=1 %7.1f  NV=2 %7.1f  NV=3 %7.1f us\n", run<2, 1>(256, it), run<2, 2>(256, it), run<2, 3>(256, it));
  printf("1 wave/SIMD:  blocks (16 MFMA then 16*NV VALU): NV=2 %7.1f  NV=3 %7.1f us\n", run<4, 2>(256, it), run<4, 3>(256, it));
  printf("2 waves/SIMD: MFMA only %7.1f us | VALU only NV=2 %7.1f us | both in blocks, every wave NV=2 %7.1f us\n", run<0, 2>(512, it), run<1, 2>(512, it), run<4, 2>(512, it));
  printf("2 waves/SIMD: one wave MFMA, the other VALU: NV=2 %7.1f  NV=3 %7.1f us\n", run<3, 2>(512, it), run<3, 3>(512, it));
  printf("v_mfma_f32_32x32x16_bf16 (32 cycles), 8 per iteration, 1 wave/SIMD, NV v_fma after each MFMA:\n");
  printf("  two accumulators: NV=0 %7.1f  NV=2 %7.1f  NV=4 %7.1f  NV=5 %7.1f  NV=6 %7.1f  NV=8 %7.1f us\n", run32<0, 2>(it), run32<2, 2>(it), run32<4, 2>(it), run32<5, 2>(it), run32<6, 2>(it), run32<8, 2>(it));
  printf("  one accumulator:  NV=0 %7.1f  NV=2 %7.1f  NV=4 %7.1f  NV=6 %7.1f us\n", run32<0, 1>(it), run32<2, 1>(it), run32<4, 1>(it), run32<6, 1>(it));
  return 0;
}
