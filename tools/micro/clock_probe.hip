// Shader clock under load: s_memtime ticks of one workgroup vs. the wall time of the kernel (hipEvent).
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_only(long long* ticks, float* sink, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ __launch_bounds__(256) void stream_copy(const float4* in, float4* out, size_t n, long long* ticks) {
  const long long t0 = clock64();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
  const long long t1 = clock64();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
  long long* ticks; float* sink; float4 *a, *b;
  const size_t n = (size_t)1 << 26;   // 1 GiB per buffer
  hipMalloc(&ticks, 4096 * 8); hipMalloc(&sink, 4096 * 256 * 4); hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  long long h[1024];
  for (int wgs : {256, 1024}) {
    const int iters = 20000;
    hipLaunchKernelGGL(mfma_only, dim3(wgs), dim3(256), 0, 0, ticks, sink, iters);
    hipEventRecord(e0); hipLaunchKernelGGL(mfma_only, dim3(wgs), dim3(256), 0, 0, ticks, sink, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, ticks, wgs * 8, hipMemcpyDeviceToHost);
    long long mx = 0; for (int i = 0; i < wgs; ++i) mx = h[i] > mx ? h[i] : mx;
    const double flops = 2.0 * 16 * 16 * 4 * 4.0 * iters * 4 * wgs;
    printf("mfma_only %4d WGs: %.3f ms, max WG ticks %lld -> %.3f ticks/ns (clock GHz if 1 tick = 1 cycle), %.1f TF, cycles per MFMA %.1f\n",
           wgs, ms, mx, mx / (ms * 1e6), flops / (ms * 1e-3) / 1e12, (double)mx / (4.0 * iters) / (wgs / 256));
  }
  hipLaunchKernelGGL(stream_copy, dim3(2048), dim3(256), 0, 0, a, b, n, ticks);
  hipEventRecord(e0); hipLaunchKernelGGL(stream_copy, dim3(2048), dim3(256), 0, 0, a, b, n, ticks); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, ticks, 1024 * 8, hipMemcpyDeviceToHost);
  long long mx = 0; for (int i = 0; i < 1024; ++i) mx = h[i] > mx ? h[i] : mx;
  printf("stream_copy: %.3f ms (%.2f TB/s read+write), max WG ticks %lld -> %.3f ticks/ns\n", ms, 2.0 * n * 16 / (ms * 1e-3) / 1e12, mx, mx / (ms * 1e6));
  return 0;
}
