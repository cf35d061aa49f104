// Issue cost of the VALU instructions the operand split and the dropout hash are made of (gfx950, one wave per SIMD, 8 independent chains):
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/valu_rates.hip -o /tmp/vr && /tmp/vr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned sel) {
  float x[8]; unsigned u[8]; f2 p[8];
  for (int j = 0; j < 8; ++j) { x[j] = threadIdx.x * 1e-3f + j; u[j] = threadIdx.x * 2654435761u + j; p[j] = (f2){x[j], x[j] + 1.f}; }
  const bf2 pc = __builtin_bit_cast(bf2, sel);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (OP == 0) x[j] = fmaf(x[j], 1.0001f, 0.5f);
        if (OP == 1) x[j] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, u[j]), pc, x[j], false);
        if (OP == 2) u[j] = __builtin_amdgcn_perm(u[j], u[(j + 1) & 7], 0x07060302u);
        if (OP == 3) p[j] = p[j] + (f2){1.5f, 2.5f};
        if (OP == 4) u[j] = (u[j] & 0xffff0000u) + 3u;      // and + add: 2 ops
        if (OP == 5) u[j] = u[j] * 0x7feb352du;
        if (OP == 6) p[j] = p[j] * (f2){1.0001f, 0.9999f} + (f2){1.5f, 2.5f};   // v_pk_fma_f32
        if (OP == 7) x[j] = __builtin_amdgcn_exp2f(x[j]);
        if (OP == 8) { unsigned r_; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r_) : "v"(x[j]), "v"(x[(j + 1) & 7])); u[j] = r_; }
        if (OP == 9) x[j] = x[j] - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[j]) & 0xffff0000u) + 1.0f;   // and + sub + add: 3 ops
      }
  }
  float s = 0; for (int j = 0; j < 8; ++j) s += x[j] + (float)u[j] + p[j][0] + p[j][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static int g_threads = 256;
template <int OP> float run(int iters) {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(g_threads), 0, 0, d, iters, 0xbf800000u);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(g_threads), 0, 0, d, iters, 0xbf800000u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d);
  return ms * 1e6f / ((float)iters * 64);      // ns per instruction-slot of the source (64 per iteration)
}
int main() {
  const int it = 20000;
  printf("ns per source op (one wave per SIMD, 64 ops per iteration in 8 independent chains; 2.1-2.4 GHz: 4 cycles ~ 1.8 ns)\n");
  printf("v_fma_f32 %.2f | v_dot2c_f32_bf16 %.2f | v_perm_b32 %.2f | v_pk_add_f32 %.2f | and+add (2 ops) %.2f | v_mul_lo_u32 %.2f | v_pk_fma_f32 %.2f\n",
         run<0>(it), run<1>(it), run<2>(it), run<3>(it), run<4>(it), run<5>(it), run<6>(it));
  printf("v_exp_f32 %.2f | v_cvt_pk_bf16_f32 %.2f | and+sub+add (3 ops) %.2f\n", run<7>(it), run<8>(it), run<9>(it));
  for (g_threads = 512; g_threads <= 1024; g_threads *= 2) {
    printf("%d waves per SIMD, ns per source op PER WAVE-SLOT (divide by the waves for the SIMD's rate):\n", g_threads / 256);
    printf("  v_fma_f32 %.2f | v_dot2c_f32_bf16 %.2f | v_perm_b32 %.2f | and+add (2 ops) %.2f | v_exp_f32 %.2f | v_cvt_pk_bf16_f32 %.2f | and+sub+add (3 ops) %.2f\n",
           run<0>(it), run<1>(it), run<2>(it), run<4>(it), run<7>(it), run<8>(it), run<9>(it));
  }
  return 0;
}
