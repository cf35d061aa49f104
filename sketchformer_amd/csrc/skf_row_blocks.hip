// Which rows of the decoder-side backward can be non-zero, as lists the GEMM kernels walk.
//
// Token mode (builders/losses.py masked cross-entropy; models/sketchformer.py:313-349): the reconstruction loss is masked
// where the target token is PAD, so d(logits) of such a position is exactly zero; the decoder is causal (look-ahead mask), so
// the gradient of EVERY decoder tensor at position t is exactly zero once all positions >= t are masked: its own row
// receives nothing from the loss, and as a self-attention key it is only seen by queries >= t, whose dO is zero.  QuickDraw-
// shaped batches are 58 % padding (83 % at seq_len 512).  live_len[b] = 1 + the last unmasked position of sample b; rows
// [b * Ld + live_len[b], (b + 1) * Ld) of every decoder-side gradient are zero, the dgrad GEMMs need not compute them and
// the weight gradients need not contract over them - exact, not an approximation (the products left out are x * 0).
// (Continuous mode has no such rows: the pen-state cross-entropy is a global mean over all positions, losses.py:43-66.)
#include "skf_common.h"
#include "../../include/skf.h"

namespace {

__global__ __launch_bounds__(256) void target_live_len_kernel(const long long* __restrict__ tar, int tar_ld, int B, int Ld,
                                                              int* __restrict__ live_len) {
  // one wave per sample: position t (decoder row) is unmasked iff tar[b][t + 1] != 0
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  int last = -1;
  for (int t = lane; t < Ld; t += 64)
    if (tar[(size_t)b * tar_ld + t + 1] != 0) last = t;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
  if (lane == 0) live_len[b] = last + 1;
}

// One workgroup.  Block k covers rows [k * g, (k + 1) * g) of the flattened (B * rps) rows; it is live when any of its
// rows r = b * rps + t has t < live_len[b].  Stable compaction: live ids ascending, then dead ids ascending; behind the ids
// one flag per block in block order (1 = live) for kernels that are launched per block rather than walking the list.
__global__ __launch_bounds__(1024) void row_blocks_kernel(const int* __restrict__ live_len, int B, int rps, int g,
                                                          int* __restrict__ blocks) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int rows = B * rps, nb = (rows + g - 1) / g;
  const int per = (nb + 1023) / 1024;
  auto is_live = [&](int k) -> bool {
    const int r0 = k * g, r1 = min(rows, r0 + g);
    for (int b = r0 / rps; b * rps < r1; ++b) {                 // the samples the block touches
      const int lo = max(r0, b * rps), hi = b * rps + live_len[b];
      if (lo < hi) return true;
    }
    return false;
  };
  int cnt = 0;
  for (int k = tid * per; k < min(nb, (tid + 1) * per); ++k) cnt += is_live(k) ? 1 : 0;
  part[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int a = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += a;
    __syncthreads();
  }
  const int nlive = part[1023];
  int lpos = part[tid] - cnt;                                   // live blocks before this thread's range
  for (int k = tid * per; k < min(nb, (tid + 1) * per); ++k) {
    const bool lv = is_live(k);
    blocks[2 + nb + k] = lv ? 1 : 0;
    if (lv) blocks[2 + lpos++] = k;
    else blocks[2 + nlive + (k - lpos)] = k;                    // dead blocks before k = k - (live blocks before k)
  }
  if (tid == 0) { blocks[0] = nlive; blocks[1] = nb; }
}

// granule 1 (row lists of 10^4..10^5 rows): the live rows of a sample are its first live_len[b] rows, so a row's slot follows
// from the running sum of the live lengths alone - every workgroup scans the B lengths and places its 1024 rows
__global__ __launch_bounds__(1024) void row_list_kernel(const int* __restrict__ live_len, int B, int rps, int* __restrict__ blocks) {
  extern __shared__ int pre[];                 // [B + 1] exclusive running sum of min(live_len, rps)
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (B + 1023) / 1024;
  int cnt = 0;
  for (int b = tid * per; b < min(B, (tid + 1) * per); ++b) cnt += min(max(live_len[b], 0), rps);
  part[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int a = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += a;
    __syncthreads();
  }
  int run = part[tid] - cnt;
  for (int b = tid * per; b < min(B, (tid + 1) * per); ++b) { pre[b] = run; run += min(max(live_len[b], 0), rps); }
  if (tid == 1023) pre[B] = part[1023];
  __syncthreads();
  const int rows = B * rps, nlive = pre[B];
  const int r = blockIdx.x * 1024 + tid;
  if (r < rows) {
    const int b = r / rps, t = r - b * rps, len = min(max(live_len[b], 0), rps);
    const bool lv = t < len;
    const int live_before = pre[b] + min(t, len);
    blocks[2 + (lv ? live_before : nlive + (r - live_before))] = r;
    blocks[2 + rows + r] = lv ? 1 : 0;
  }
  if (blockIdx.x == 0 && tid == 0) { blocks[0] = nlive; blocks[1] = rows; }
}

// Samples sorted by the number of unmasked positions of up to two mask matrices (stable, most first): one workgroup on the step's
// critical path, so every load is independent of the others (four mask bytes per request where the rows allow it, the sample's count in
// LDS by atomic add); then rank by counting.  (First version: a wave per sample with the byte loads of a row behind one another - 17 us.)
__device__ __forceinline__ void count_unmasked(const unsigned char* __restrict__ m, int ld, int L, int B, int* key, int tid) {
  if (!m) return;
  if (ld == L && L >= 4 && ((uintptr_t)m & 3) == 0) {  // contiguous rows: the matrix as words, a word may straddle two samples (L >= 4: never three)
    const int total = B * L, n = (total + 3) >> 2;
    for (int i0 = tid; i0 < n; i0 += 4 * 1024) {
      unsigned v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {           // four requests in flight per thread
        const int i = i0 + 1024 * u, t0 = 4 * i;
        v[u] = 0x01010101u;
        if (t0 + 3 < total) v[u] = *reinterpret_cast<const unsigned*>(m + t0);
        else if (t0 < total) v[u] = (unsigned)m[t0] | (t0 + 1 < total ? (unsigned)m[t0 + 1] << 8 : 0x100u) | (t0 + 2 < total ? (unsigned)m[t0 + 2] << 16 : 0x10000u) | 0x1000000u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t0 = 4 * (i0 + 1024 * u), b = t0 / L, split = (b + 1) * L - t0;       // bytes [0, split) of the word belong to sample b
        int z0 = 0, z1 = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int z = ((v[u] >> (8 * e)) & 0xffu) == 0u;
          if (e < split) z0 += z; else z1 += z;
        }
        if (z0) atomicAdd(&key[b], z0);
        if (z1) atomicAdd(&key[b + 1], z1);
      }
    }
  } else {
    const int n = B * L;
    for (int i = tid; i < n; i += 1024) {
      const int b = i / L, t = i - b * L;
      if (!m[(size_t)b * ld + t]) atomicAdd(&key[b], 1);
    }
  }
}
__global__ __launch_bounds__(1024) void sample_order_kernel(const unsigned char* __restrict__ ma, int lda, int La,
                                                            const unsigned char* __restrict__ mb, int ldb, int Lb, int B, int* __restrict__ order) {
  extern __shared__ int key[];                 // [B] counts, [B] ranks
  int* rank = key + B;
  const int tid = threadIdx.x;
  for (int b = tid; b < 2 * B; b += 1024) key[b] = 0;
  __syncthreads();
  count_unmasked(ma, lda, La, B, key, tid);
  count_unmasked(mb, ldb, Lb, B, key, tid);
  __syncthreads();
  // rank[b] = samples in front of b: `parts` threads per sample share the comparisons
  const int parts = B >= 1024 ? 1 : 1024 / B, per_pass = 1024 / parts;
  for (int b0 = 0; b0 < B; b0 += per_pass) {
    const int b = b0 + tid / parts, part = tid - (tid / parts) * parts;
    if (tid < per_pass * parts && b < B) {
      const int k = key[b];
      int cnt = 0;
      for (int j = part; j < B; j += parts) { const int kj = key[j]; cnt += (kj > k || (kj == k && j < b)) ? 1 : 0; }
      if (cnt) atomicAdd(&rank[b], cnt);
    }
  }
  __syncthreads();
  for (int b = tid; b < B; b += 1024) order[rank[b]] = b;
}

}  // namespace

extern "C" int skf_sample_order(const unsigned char* mask_a, int lda, int La, const unsigned char* mask_b, int ldb, int Lb, int B, int* order,
                                skf_stream_t stream) {
  SKF_CHECK_ARG(order && B > 0 && B <= 4096 && (mask_a || mask_b), "bad argument (B <= 4096: the counts and ranks of a batch sit in 32 KB of LDS)");
  SKF_CHECK_ARG((!mask_a || (La > 0 && lda >= La)) && (!mask_b || (Lb > 0 && ldb >= Lb)), "bad mask shape");
  hipLaunchKernelGGL(sample_order_kernel, dim3(1), dim3(1024), (size_t)2 * B * sizeof(int), (hipStream_t)stream, mask_a, lda, La, mask_b, ldb, Lb, B, order);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_target_live_len(const long long* tar, int tar_ld, int B, int Ld, int* live_len, skf_stream_t stream) {
  SKF_CHECK_ARG(tar && live_len && B > 0 && Ld > 0 && tar_ld > Ld, "bad argument");
  hipLaunchKernelGGL(target_live_len_kernel, dim3(skf_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, tar, tar_ld, B, Ld, live_len);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" size_t skf_row_blocks_bytes(int rows, int granule) {
  return granule > 0 ? (size_t)(2 + 2 * skf_cdiv(rows, granule)) * sizeof(int) : 0;
}

extern "C" int skf_row_blocks_build(const int* live_len, int B, int rows_per_sample, int granule, int* blocks, skf_stream_t stream) {
  SKF_CHECK_ARG(live_len && blocks && B > 0 && rows_per_sample > 0 && granule > 0, "bad argument");
  if (granule == 1 && B <= 8192)
    hipLaunchKernelGGL(row_list_kernel, dim3(skf_cdiv(B * rows_per_sample, 1024)), dim3(1024), (size_t)(B + 1) * sizeof(int), (hipStream_t)stream,
                       live_len, B, rows_per_sample, blocks);
  else
  hipLaunchKernelGGL(row_blocks_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, live_len, B, rows_per_sample, granule, blocks);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
