#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 4];
  const int l = threadIdx.x;
  // lane l stores 4 contiguous 16-bit values l*4+e at its own 8-byte slot
  for (int e = 0; e < 4; ++e) lds[l * 4 + e] = (uint16_t)(l * 4 + e);
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + l * 4));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (uint16_t)v[e];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  probe<<<1, 64>>>(d);
  uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (src lane %2d, elem %d)", h[l*4+e] / 4, h[l*4+e] % 4); printf("\n"); }
  return 0;
}
