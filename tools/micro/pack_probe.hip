// Does v_pack_b32_f16 with op_sel:[1,1,0] move the HIGH halves of its two sources bit-exactly (dst = src0.hi | src1.hi << 16), including
// bit patterns that are fp16 subnormals / NaNs?  It would replace v_perm_b32 (half rate) in the bf16 piece packing.  And its rate.
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/pack_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const unsigned* a, const unsigned* b, unsigned* o, int n) {
  const int i = threadIdx.x + blockIdx.x * blockDim.x;
  if (i >= n) return;
  unsigned r, x = a[i], y = b[i];
  asm volatile("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(r) : "v"(x), "v"(y));
  o[i] = r;
}
template <int OP>
__global__ __launch_bounds__(1024) void rate(unsigned* out, int iters) {
  unsigned u[8];
  for (int j = 0; j < 8; ++j) u[j] = threadIdx.x * 2654435761u + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (OP == 0) asm volatile("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(u[j]) : "v"(u[j]), "v"(u[(j + 1) & 7]));
        if (OP == 1) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[j]) : "v"(u[j]), "v"(u[(j + 1) & 7]), "s"(0x07060302u));
        if (OP == 2) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[j]) : "s"(0xffff0000u), "v"(u[j]));
        if (OP == 3) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(u[j]) : "v"(u[j]), "v"(u[(j + 1) & 7]));
        if (OP == 4) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(u[j]) : "v"(u[j]), "s"(0xffff0000u), "v"(u[(j + 1) & 7]));
        if (OP == 5) asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(u[j]) : "v"(u[j]));
      }
  }
  unsigned s = 0; for (int j = 0; j < 8; ++j) s += u[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> float run(int threads) {
  unsigned* d; hipMalloc(&d, 256 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL((rate<OP>), dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate<OP>), dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d);
  return ms * 1e6f / ((float)iters * 64) / (threads / 256);      // ns per instruction per SIMD
}
int main() {
  const int n = 1 << 16;
  unsigned *ha = new unsigned[n], *hb = new unsigned[n], *ho = new unsigned[n];
  for (int i = 0; i < n; ++i) { ha[i] = (unsigned)i << 16 | 0x1234u; hb[i] = (unsigned)((i * 40503u) & 0xffffu) << 16 | 0xabcdu; }
  unsigned *a, *b, *o; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&o, n * 4);
  hipMemcpy(a, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, a, b, o, n);
  hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) { const unsigned want = (ha[i] >> 16) | (hb[i] & 0xffff0000u); if (ho[i] != want) { if (bad < 5) printf("mismatch %08x %08x -> %08x want %08x\n", ha[i], hb[i], ho[i], want); ++bad; } }
  printf("v_pack_b32_f16 op_sel:[1,1,0] on all 65536 high-half patterns: %d mismatches\n", bad);
  printf("ns per instruction per SIMD at 4 waves/SIMD: v_pack_b32_f16 %.2f | v_perm_b32 %.2f | v_and_b32 %.2f | v_sub_f32 %.2f | v_and_or_b32 %.2f | v_lshrrev_b32 %.2f\n",
         run<0>(1024), run<1>(1024), run<2>(1024), run<3>(1024), run<4>(1024), run<5>(1024));
  return 0;
}
