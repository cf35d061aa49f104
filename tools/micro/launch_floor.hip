// What a 10..40 us kernel can cost at best on this box: per-launch time (back-to-back launches in one stream,
// hipEvent around 200 of them) of (a) an empty grid, (b) a row-streaming kernel moving the bytes of one Dense
// GEMM of the train step (read M x K fp32, write M x N fp32) with no arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void empty_kernel(float* p) { if (p == nullptr && threadIdx.x == 9999) p[0] = 0; }
// each workgroup walks 16-row tiles (stride = grid): reads 16 x K floats, writes 16 x N floats
__global__ __launch_bounds__(256) void stream_rows(const float4* __restrict__ A, float4* __restrict__ C, int M, int K4, int N4) {
  const int ntiles = M / 16;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    float4 acc = {0, 0, 0, 0};
    const float4* a = A + (size_t)t * 16 * K4;
    for (int e = threadIdx.x; e < 16 * K4; e += 256) { const float4 v = a[e]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    float4* c = C + (size_t)t * 16 * N4;
    for (int e = threadIdx.x; e < 16 * N4; e += 256) c[e] = acc;
  }
}
template <typename F> float per_launch_us(F launch, int n = 200) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) launch();
  hipEventRecord(e0); for (int i = 0; i < n; ++i) launch(); hipEventRecord(e1);
  hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}
int main() {
  const int M = 25600;
  float4 *A, *C; hipMalloc(&A, (size_t)M * 1024 * 4); hipMalloc(&C, (size_t)M * 1024 * 4);
  hipMemset(A, 0, (size_t)M * 1024 * 4);
  for (int wgs : {256, 512, 1600}) {
    printf("empty grid %4d x 256 threads:            %.2f us/launch\n", wgs, per_launch_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(256), 0, 0, (float*)A); }));
    printf("empty grid %4d x 256 threads, 26 KB LDS: %.2f us/launch\n", wgs, per_launch_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(256), 26112, 0, (float*)A); }));
  }
  const int shapes[][2] = {{128, 128}, {128, 384}, {128, 512}, {512, 128}, {128, 1004}};
  for (auto& s : shapes)
    for (int wgs : {512, 1600}) {
      const float us = per_launch_us([&] { hipLaunchKernelGGL(stream_rows, dim3(wgs), dim3(256), 0, 0, A, C, M, s[0] / 4, s[1] / 4); });
      const double bytes = 4.0 * M * (s[0] + s[1]);
      printf("stream_rows K=%3d N=%4d, %4d WGs: %.2f us/launch = %.2f TB/s (%.1f MB)\n", s[0], s[1], wgs, us, bytes / us / 1e6, bytes / 1e6);
    }
  return 0;
}
