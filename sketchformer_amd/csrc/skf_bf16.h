// bf16 storage helpers of the cfg-5 path (bf16 activations / weight images, fp32 arithmetic in registers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 skf_bf16;
typedef __bf16 skf_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 skf_bf16x4 __attribute__((ext_vector_type(4)));

// fp32 -> bf16 bits, round to nearest even (NaN stays NaN: the quiet bit is forced)
__host__ __device__ __forceinline__ uint32_t skf_f2bf(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__host__ __device__ __forceinline__ float skf_bf2f(uint32_t b) { return __builtin_bit_cast(float, b << 16); }

// device packing: v_cvt_pk_bf16_f32 (round to nearest even), one instruction per pair
typedef float skf_f32x2 __attribute__((ext_vector_type(2)));
typedef float skf_f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 skf_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t skf_pack2(float lo, float hi) {
  const skf_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, skf_bf16x2));
}
__device__ __forceinline__ uint2 skf_pack4(const float (&v)[4]) { return make_uint2(skf_pack2(v[0], v[1]), skf_pack2(v[2], v[3])); }
__device__ __forceinline__ uint4 skf_pack8(const float (&v)[8]) {
  return make_uint4(skf_pack2(v[0], v[1]), skf_pack2(v[2], v[3]), skf_pack2(v[4], v[5]), skf_pack2(v[6], v[7]));
}
__device__ __forceinline__ skf_bf16x8 skf_cvt8(float a, float b, float c, float d, float e, float f, float g, float h) {
  const skf_f32x8 v = {a, b, c, d, e, f, g, h};
  return __builtin_convertvector(v, skf_bf16x8);
}
__device__ __forceinline__ void skf_unpack2(uint32_t w, float& lo, float& hi) {
  lo = __builtin_bit_cast(float, w << 16); hi = __builtin_bit_cast(float, w & 0xffff0000u);
}
__device__ __forceinline__ void skf_unpack4(uint2 w, float (&v)[4]) { skf_unpack2(w.x, v[0], v[1]); skf_unpack2(w.y, v[2], v[3]); }
__device__ __forceinline__ void skf_unpack8(uint4 w, float (&v)[8]) {
  skf_unpack2(w.x, v[0], v[1]); skf_unpack2(w.y, v[2], v[3]); skf_unpack2(w.z, v[4], v[5]); skf_unpack2(w.w, v[6], v[7]);
}
