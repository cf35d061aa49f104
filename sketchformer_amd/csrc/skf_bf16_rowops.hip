// Row kernels of the bf16 path (BASELINE cfg 5): the HBM-bound stages between the matmuls, reading and writing bf16
// activations (half the traffic of the fp32 kernels in skf_rowops.hip), arithmetic in fp32 registers, parameters and
// parameter gradients in fp32 (master weights).  Same formulas, dropout hash and site / element indexing as the fp32
// path, so skf_dropout_keep_mask reproduces the masks on the host for both.
//
//   embed_fwd        builders/layers/transformer.py:288-296, 325-334   Embedding * sqrt(d) + pos, Dropout
//   ln_fwd / ln_bwd  builders/layers/transformer.py:217-222, 247-260   LayerNormalization(1e-6)(x + Dropout(y))
//   softmax_ce       builders/losses.py:26-41, builders/keras_metrics.py:25  masked token CE + accuracy + in-place gradient
//   pool_fwd / bwd   builders/layers/transformer.py:70-73              SelfAttnV1 after u = tanh(xW + b)
//   expander         builders/layers/transformer.py:370-376            DenseExpander
//   weight images    fp32 master [in][out] -> bf16 [in][out] and [out][in] (the operands of skf_gemm_bf16)
#include <stdlib.h>
#include "skf_common.h"
#include "skf_bf16.h"

namespace {

// VPL consecutive bf16 values of a row <-> fp32 registers (VPL = d / 64 in {2, 4, 8, 16})
template <int VPL> __device__ __forceinline__ void ldrow(const skf_bf16* p, float (&v)[VPL]);
template <> __device__ __forceinline__ void ldrow<2>(const skf_bf16* p, float (&v)[2]) {
  skf_unpack2(*reinterpret_cast<const uint32_t*>(p), v[0], v[1]);
}
template <> __device__ __forceinline__ void ldrow<4>(const skf_bf16* p, float (&v)[4]) { skf_unpack4(*reinterpret_cast<const uint2*>(p), v); }
template <> __device__ __forceinline__ void ldrow<8>(const skf_bf16* p, float (&v)[8]) { skf_unpack8(*reinterpret_cast<const uint4*>(p), v); }
template <> __device__ __forceinline__ void ldrow<16>(const skf_bf16* p, float (&v)[16]) {
  float a[8], b[8];
  skf_unpack8(*reinterpret_cast<const uint4*>(p), a); skf_unpack8(*reinterpret_cast<const uint4*>(p + 8), b);
#pragma unroll
  for (int e = 0; e < 8; ++e) { v[e] = a[e]; v[8 + e] = b[e]; }
}
template <int VPL> __device__ __forceinline__ void strow(skf_bf16* p, const float (&v)[VPL]);
template <> __device__ __forceinline__ void strow<2>(skf_bf16* p, const float (&v)[2]) { *reinterpret_cast<uint32_t*>(p) = skf_pack2(v[0], v[1]); }
template <> __device__ __forceinline__ void strow<4>(skf_bf16* p, const float (&v)[4]) { *reinterpret_cast<uint2*>(p) = skf_pack4(v); }
template <> __device__ __forceinline__ void strow<8>(skf_bf16* p, const float (&v)[8]) { *reinterpret_cast<uint4*>(p) = skf_pack8(v); }
template <> __device__ __forceinline__ void strow<16>(skf_bf16* p, const float (&v)[16]) {
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = v[e]; b[e] = v[8 + e]; }
  *reinterpret_cast<uint4*>(p) = skf_pack8(a); *reinterpret_cast<uint4*>(p + 8) = skf_pack8(b);
}
template <int VPL> __device__ __forceinline__ void ldf32(const float* p, float (&v)[VPL]) {
#pragma unroll
  for (int e = 0; e < VPL; e += 2) { const float2 t = *reinterpret_cast<const float2*>(p + e); v[e] = t.x; v[e + 1] = t.y; }
}
__device__ __forceinline__ float bf_round(float x) { return skf_bf2f(skf_f2bf(x)); }

// ------------------------------------------------------------------ embedding stage
template <int VPL>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ tok, int tok_ld, int Lrows, int rows,
                                                        const float* __restrict__ table, int vocab, const float* __restrict__ pos,
                                                        skf_bf16* __restrict__ out, float rate, uint32_t site, const SkfStepState* st) {
  constexpr int D = VPL * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float sq = sqrtf((float)D);
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const int b = row / Lrows, t = row % Lrows;
    long long tk = tok[(size_t)b * tok_ld + t];
    const bool oob = tk < 0 || tk >= vocab;          // zero vector like tf.gather on a GPU (see skf_rowops.hip)
    if (oob) tk = 0;
    float e[VPL], pe[VPL];
    ldf32<VPL>(table + (size_t)tk * D + lane * VPL, e);
    ldf32<VPL>(pos + (size_t)t * D + lane * VPL, pe);
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
      float v = (oob ? 0.f : e[c]) * sq + pe[c];
      if (rate > 0.f) v *= skf_keep(sk, (uint32_t)row * (uint32_t)D + lane * VPL + c, thresh) ? inv_keep : 0.f;
      e[c] = v;
    }
    strow<VPL>(out + (size_t)row * D + lane * VPL, e);
  }
}

// ------------------------------------------------------------------ residual + LayerNorm
// z = x + drop(y) is rounded to bf16 first (it is what the backward reads back) and the moments are those of the rounded z
template <int VPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const skf_bf16* __restrict__ x, skf_bf16* __restrict__ y,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     skf_bf16* __restrict__ out, float* __restrict__ stats, int rows, float rate,
                                                     uint32_t site, const SkfStepState* st) {
  constexpr int D = VPL * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  float g[VPL], bt[VPL];
  ldf32<VPL>(gamma + lane * VPL, g);
  ldf32<VPL>(beta + lane * VPL, bt);
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const size_t off = (size_t)row * D + lane * VPL;
    float xv[VPL], yv[VPL];
    ldrow<VPL>(x + off, xv);
    ldrow<VPL>(y + off, yv);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
      float dy = yv[c];
      if (rate > 0.f) dy *= skf_keep(sk, (uint32_t)row * (uint32_t)D + lane * VPL + c, thresh) ? inv_keep : 0.f;
      xv[c] = bf_round(xv[c] + dy);
      s += xv[c];
    }
    strow<VPL>(y + off, xv);
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < VPL; ++c) { const float dlt = xv[c] - mean; q += dlt * dlt; }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + 1e-6f);
#pragma unroll
    for (int c = 0; c < VPL; ++c) yv[c] = (xv[c] - mean) * rstd * g[c] + bt[c];
    strow<VPL>(out + off, yv);
    if (lane == 0) *reinterpret_cast<float2*>(stats + 2 * (size_t)row) = make_float2(mean, rstd);
  }
}

// dz = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dout * gamma; dy = dz * dropout mask; per-workgroup partial
// dgamma = sum dout * xhat, dbeta = sum dout  -> part[block][2][D]
template <int VPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const skf_bf16* __restrict__ dout, const skf_bf16* __restrict__ z,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     skf_bf16* __restrict__ dz, skf_bf16* __restrict__ dy, float* __restrict__ part,
                                                     int rows, float rate, uint32_t site, const SkfStepState* st,
                                                     const int* __restrict__ live_len, int rps) {
  constexpr int D = VPL * 64;
  __shared__ float red[3][2][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  float g[VPL], dg[VPL], db[VPL];
  ldf32<VPL>(gamma + lane * VPL, g);
#pragma unroll
  for (int c = 0; c < VPL; ++c) { dg[c] = 0.f; db[c] = 0.f; }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const size_t off = (size_t)row * D + lane * VPL;
    if (live_len) {
      // row t of sample b with t >= live_len[b]: dout is exactly zero (skf_target_live_len), so are dz, dy and the row's
      // share of dgamma / dbeta - stored, not computed
      const int b = row / rps;
      if (row - b * rps >= live_len[b]) {
        float o[VPL];
#pragma unroll
        for (int c = 0; c < VPL; ++c) o[c] = 0.f;
        strow<VPL>(dz + off, o);
        if (rate > 0.f || dy != dz) strow<VPL>(dy + off, o);
        continue;
      }
    }
    float dv[VPL], zv[VPL];
    ldrow<VPL>(dout + off, dv);
    ldrow<VPL>(z + off, zv);
    const float2 ms = *reinterpret_cast<const float2*>(stats + 2 * (size_t)row);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
      const float xh = (zv[c] - ms.x) * ms.y;
      zv[c] = xh;
      dg[c] += dv[c] * xh; db[c] += dv[c];
      dv[c] *= g[c];
      s1 += dv[c]; s2 += dv[c] * xh;
    }
    s1 = wave_sum(s1) * (1.0f / D); s2 = wave_sum(s2) * (1.0f / D);
    float o[VPL];
#pragma unroll
    for (int c = 0; c < VPL; ++c) o[c] = ms.y * (dv[c] - s1 - zv[c] * s2);
    strow<VPL>(dz + off, o);
    if (rate > 0.f) {
#pragma unroll
      for (int c = 0; c < VPL; ++c) o[c] *= skf_keep(sk, (uint32_t)row * (uint32_t)D + lane * VPL + c, thresh) ? inv_keep : 0.f;
      strow<VPL>(dy + off, o);
    } else if (dy != dz) {
      strow<VPL>(dy + off, o);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < VPL; ++c) { red[wave - 1][0][lane * VPL + c] = dg[c]; red[wave - 1][1][lane * VPL + c] = db[c]; }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < VPL; ++c) {
      const int col = lane * VPL + c;
      part[(size_t)blockIdx.x * 2 * D + col] = dg[c] + red[0][0][col] + red[1][0][col] + red[2][0][col];
      part[(size_t)blockIdx.x * 2 * D + D + col] = db[c] + red[0][1][col] + red[1][1][col] + red[2][1][col];
    }
  }
}

// ------------------------------------------------------------------ masked token cross-entropy (one wave per row)
// logits (rows, ld) bf16, ncls <= 64 * 32; loss = (lse - logit[target]) * (target != 0 | !mask_pad); hit = first-index
// argmax == target; in place: (softmax - onehot) * mask * scale, pad columns [ncls, ld) = 0 (they are contraction
// columns of the following dgrad / wgrad GEMMs)
constexpr int CE_MAX = 32;
__global__ __launch_bounds__(256) void softmax_ce_kernel(skf_bf16* __restrict__ logits, int ld, int rows, int ncls,
                                                         const long long* __restrict__ target, int tgt_ld, int tgt_cols,
                                                         int tgt_off, int mask_pad, float scale, float* __restrict__ row_loss,
                                                         float* __restrict__ row_hit, int write_grad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  skf_bf16* lp = logits + (size_t)row * ld;
  const long long tg = target[(size_t)(row / tgt_cols) * tgt_ld + (row % tgt_cols) + tgt_off];
  float v[CE_MAX];
  float mx = -INFINITY;
  int am = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < CE_MAX / 2; ++j) {
    const int c = 2 * (lane + 64 * j);
    v[2 * j] = -INFINITY; v[2 * j + 1] = -INFINITY;
    if (c < ncls) {                                            // ncls is even (multiple of 4)
      skf_unpack2(*reinterpret_cast<const uint32_t*>(lp + c), v[2 * j], v[2 * j + 1]);
      if (v[2 * j] > mx) { mx = v[2 * j]; am = c; }
      if (v[2 * j + 1] > mx) { mx = v[2 * j + 1]; am = c + 1; }
    }
  }
  // wave argmax, first index on ties
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < CE_MAX; ++j) { v[j] = __expf(v[j] - mx); s += v[j]; }          // exp(-inf) = 0 for absent columns
  s = wave_sum(s);
  float tv = 0.f;                                               // exp(logit[target] - mx), found by its owner
  {
    const int tc = (int)tg, owner = (tc >> 1) & 63, slot = ((tc >> 1) >> 6) * 2 + (tc & 1);
    float mine = 0.f;
#pragma unroll
    for (int j = 0; j < CE_MAX; ++j) mine = (j == slot) ? v[j] : mine;
    tv = __shfl(mine, owner, 64);
  }
  const float maskv = (mask_pad && tg == 0) ? 0.f : 1.f;
  if (lane == 0) {
    row_loss[row] = (logf(s) - logf(tv)) * maskv;               // lse - logit[target]
    row_hit[row] = am == (int)tg ? 1.f : 0.f;
  }
  if (!write_grad) return;
  const float inv = 1.0f / s, gs = maskv * scale;
#pragma unroll
  for (int j = 0; j < CE_MAX / 2; ++j) {
    const int c = 2 * (lane + 64 * j);
    if (c < ncls) {
      const float g0 = (v[2 * j] * inv - (c == (int)tg ? 1.f : 0.f)) * gs;
      const float g1 = (v[2 * j + 1] * inv - (c + 1 == (int)tg ? 1.f : 0.f)) * gs;
      *reinterpret_cast<uint32_t*>(lp + c) = skf_pack2(g0, g1);
    } else if (c < ld) {
      *reinterpret_cast<uint32_t*>(lp + c) = 0u;
    }
  }
}

// ------------------------------------------------------------------ SelfAttnV1 pooling (one workgroup per sample)
// scores[t] = u[t,:] . V ; a = softmax over ALL t (no padding mask, like the reference) ; emb[c] = sum_t a[t] x[t][c]
__global__ __launch_bounds__(256) void pool_fwd_kernel(const skf_bf16* __restrict__ u, const float* __restrict__ Vw,
                                                       const skf_bf16* __restrict__ x, int L, int U, int d,
                                                       float* __restrict__ a_out, float* __restrict__ emb) {
  extern __shared__ float sm[];        // [L] scores -> weights, [8] reduction scratch
  float* sc = sm;
  float* red = sm + L;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const skf_bf16* ub = u + (size_t)b * L * U;
  const skf_bf16* xb = x + (size_t)b * L * d;
  for (int t = wave; t < L; t += 4) {
    float s = 0.f;
    for (int c = lane * 2; c < U; c += 128) {
      float u0, u1;
      skf_unpack2(*reinterpret_cast<const uint32_t*>(ub + (size_t)t * U + c), u0, u1);
      s += u0 * Vw[c] + u1 * Vw[c + 1];
    }
    s = wave_sum(s);
    if (lane == 0) sc[t] = s;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = tid; t < L; t += 256) mx = fmaxf(mx, sc[t]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int t = tid; t < L; t += 256) { const float e = __expf(sc[t] - mx); sc[t] = e; s += e; }
  s = wave_sum(s);
  if (lane == 0) red[4 + wave] = s;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int t = tid; t < L; t += 256) { const float a = sc[t] * inv; sc[t] = a; a_out[(size_t)b * L + t] = a; }
  __syncthreads();
  for (int c = tid * 2; c < d; c += 512) {
    float e0 = 0.f, e1 = 0.f;
    for (int t = 0; t < L; ++t) {
      float x0, x1;
      skf_unpack2(*reinterpret_cast<const uint32_t*>(xb + (size_t)t * d + c), x0, x1);
      e0 += sc[t] * x0; e1 += sc[t] * x1;
    }
    emb[(size_t)b * d + c] = e0; emb[(size_t)b * d + c + 1] = e1;
  }
}

// da[t] = demb . x[t,:] ; dx[t][c] = a[t] demb[c] ; dscore = a o (da - sum a da) ; dpre[t][u] = dscore[t] V[u] (1 - u^2)
// (written over u) ; dV partial of this sample: sum_t dscore[t] u[t][u] -> dV_part[b][U]
__global__ __launch_bounds__(256) void pool_bwd_kernel(skf_bf16* __restrict__ u, const float* __restrict__ Vw,
                                                       const skf_bf16* __restrict__ x, const float* __restrict__ a_in,
                                                       const float* __restrict__ demb, int L, int U, int d,
                                                       skf_bf16* __restrict__ dx, float* __restrict__ dV_part) {
  extern __shared__ float sm[];        // [L] da -> dscore, [L] a, [8] scratch
  float* ds = sm;
  float* av = sm + L;
  float* red = sm + 2 * L;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  skf_bf16* ub = u + (size_t)b * L * U;
  const skf_bf16* xb = x + (size_t)b * L * d;
  skf_bf16* dxb = dx + (size_t)b * L * d;
  const float* de = demb + (size_t)b * d;
  for (int t = tid; t < L; t += 256) av[t] = a_in[(size_t)b * L + t];
  __syncthreads();
  for (int t = wave; t < L; t += 4) {
    float s = 0.f;
    const float at = av[t];
    for (int c = lane * 2; c < d; c += 128) {
      float x0, x1;
      skf_unpack2(*reinterpret_cast<const uint32_t*>(xb + (size_t)t * d + c), x0, x1);
      const float d0 = de[c], d1 = de[c + 1];
      s += x0 * d0 + x1 * d1;
      *reinterpret_cast<uint32_t*>(dxb + (size_t)t * d + c) = skf_pack2(at * d0, at * d1);
    }
    s = wave_sum(s);
    if (lane == 0) ds[t] = s;
  }
  __syncthreads();
  float dot = 0.f;
  for (int t = tid; t < L; t += 256) dot += av[t] * ds[t];
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = red[0] + red[1] + red[2] + red[3];
  for (int t = tid; t < L; t += 256) ds[t] = av[t] * (ds[t] - dot);
  __syncthreads();
  for (int c = tid * 2; c < U; c += 512) {
    float g0 = 0.f, g1 = 0.f;
    const float v0 = Vw[c], v1 = Vw[c + 1];
    for (int t = 0; t < L; ++t) {
      float u0, u1;
      skf_bf16* up = ub + (size_t)t * U + c;
      skf_unpack2(*reinterpret_cast<const uint32_t*>(up), u0, u1);
      const float dt = ds[t];
      g0 += dt * u0; g1 += dt * u1;
      *reinterpret_cast<uint32_t*>(up) = skf_pack2(dt * v0 * (1.f - u0 * u0), dt * v1 * (1.f - u1 * u1));
    }
    dV_part[(size_t)b * U + c] = g0; dV_part[(size_t)b * U + c + 1] = g1;
  }
}

// ------------------------------------------------------------------ DenseExpander
__global__ __launch_bounds__(256) void expander_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int B, int L, int d,
                                                           skf_bf16* __restrict__ pre) {
  const size_t total2 = (size_t)B * L * d / 2;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total2; e += (size_t)gridDim.x * 256) {
    const size_t i = e * 2;
    const int c = (int)(i % d), t = (int)((i / d) % L), b = (int)(i / ((size_t)d * L));
    const float wt = w[t], bt = bias[t];
    *reinterpret_cast<uint32_t*>(pre + i) = skf_pack2(emb[(size_t)b * d + c] * wt + bt, emb[(size_t)b * d + c + 1] * wt + bt);
  }
}
// one workgroup per sample: demb[c] (+)= sum_t dpre[t][c] w[t]; dw_part[b][t] = sum_c dpre[t][c] emb[c]; db_part[b][t] = sum_c dpre[t][c]
__global__ __launch_bounds__(256) void expander_bwd_kernel(const skf_bf16* __restrict__ dpre, const float* __restrict__ emb,
                                                           const float* __restrict__ w, int L, int d, float* __restrict__ demb,
                                                           int demb_accumulate, float* __restrict__ dw_part, float* __restrict__ db_part) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const skf_bf16* pb = dpre + (size_t)b * L * d;
  const float* eb = emb + (size_t)b * d;
  for (int t = wave; t < L; t += 4) {
    float s0 = 0.f, s1 = 0.f;
    for (int c = lane * 2; c < d; c += 128) {
      float p0, p1;
      skf_unpack2(*reinterpret_cast<const uint32_t*>(pb + (size_t)t * d + c), p0, p1);
      s0 += p0 * eb[c] + p1 * eb[c + 1];
      s1 += p0 + p1;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (lane == 0) { dw_part[(size_t)b * L + t] = s0; db_part[(size_t)b * L + t] = s1; }
  }
  for (int c = tid * 2; c < d; c += 512) {
    float g0 = 0.f, g1 = 0.f;
    for (int t = 0; t < L; ++t) {
      float p0, p1;
      skf_unpack2(*reinterpret_cast<const uint32_t*>(pb + (size_t)t * d + c), p0, p1);
      g0 += p0 * w[t]; g1 += p1 * w[t];
    }
    float* dst = demb + (size_t)b * d + c;
    if (demb_accumulate) { dst[0] += g0; dst[1] += g1; } else { dst[0] = g0; dst[1] = g1; }
  }
}

// ------------------------------------------------------------------ weight images
// src fp32 [R][C] (row stride lds) -> dst bf16 [R][ldd] (pad columns zero) and dstT bf16 [C][ldt] (pad columns zero)
__global__ __launch_bounds__(256) void cast_weight_kernel(const float* __restrict__ src, int R, int C, int lds,
                                                          skf_bf16* __restrict__ dst, int ldd, skf_bf16* __restrict__ dstT, int ldt) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + ty + 8 * j, c = c0 + tx;
    const float v = (r < R && c < C) ? src[(size_t)r * lds + c] : 0.f;
    tile[ty + 8 * j][tx] = v;
    if (dst && r < R && c < ldd) reinterpret_cast<uint16_t*>(dst)[(size_t)r * ldd + c] = (uint16_t)skf_f2bf(v);
  }
  __syncthreads();
  if (dstT) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + ty + 8 * j, r = r0 + tx;          // transposed: row index c of dstT, column r
      if (c < C && r < ldt) reinterpret_cast<uint16_t*>(dstT)[(size_t)c * ldt + r] = (uint16_t)skf_f2bf(tile[tx][ty + 8 * j]);
    }
  }
}
// the same for MANY weights in one launch: descs[i].block_begin is the running sum of the 32 x 32 blocks of the problems before
// i (the ~90 images of a step were 90 launches of a few microseconds each)
__global__ __launch_bounds__(256) void cast_weight_batch_kernel(const SkfCastDesc* __restrict__ descs, int n) {
  __shared__ float tile[32][33];
  // binary search over scalar loads (constant address space: the table is uploaded once, no kernel writes it); the linear scan over
  // plain global loads was up to ~90 dependent memory round trips for the last workgroups
  typedef const __attribute__((address_space(4))) SkfCastDesc* const_descp;
  const const_descp cd = (const_descp)descs;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cd[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  SkfCastDesc d;                                         // (member-wise: no copy constructor across address spaces)
  d.src = cd[lo].src; d.dst = cd[lo].dst; d.dst_t = cd[lo].dst_t; d.R = cd[lo].R; d.C = cd[lo].C; d.ld_src = cd[lo].ld_src;
  d.ld_dst = cd[lo].ld_dst; d.ld_t = cd[lo].ld_t; d.block_begin = cd[lo].block_begin; d.blocks_x = cd[lo].blocks_x; d.pad = 0;
  const int local = blockIdx.x - d.block_begin;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = (local / d.blocks_x) * 32, c0 = (local % d.blocks_x) * 32;
  const float* src = d.src;
  uint16_t* dst = reinterpret_cast<uint16_t*>(d.dst);
  uint16_t* dstT = reinterpret_cast<uint16_t*>(d.dst_t);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + ty + 8 * j, c = c0 + tx;
    const float v = (r < d.R && c < d.C) ? src[(size_t)r * d.ld_src + c] : 0.f;
    tile[ty + 8 * j][tx] = v;
    if (dst && r < d.R && c < d.ld_dst) dst[(size_t)r * d.ld_dst + c] = (uint16_t)skf_f2bf(v);
  }
  __syncthreads();
  if (dstT) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = c0 + ty + 8 * j, r = r0 + tx;
      if (c < d.C && r < d.ld_t) dstT[(size_t)c * d.ld_t + r] = (uint16_t)skf_f2bf(tile[tx][ty + 8 * j]);
    }
  }
}
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, skf_bf16* __restrict__ dst, size_t n) {
  for (size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2; e < n; e += (size_t)gridDim.x * 512) {
    const float a = src[e], b = e + 1 < n ? src[e + 1] : 0.f;
    if (e + 1 < n) *reinterpret_cast<uint32_t*>(dst + e) = skf_pack2(a, b);
    else reinterpret_cast<uint16_t*>(dst)[e] = (uint16_t)skf_f2bf(a);
  }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const skf_bf16* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
    dst[e] = skf_bf2f(reinterpret_cast<const uint16_t*>(src)[e]);
}

template <typename F> int dispatch_d(int d, F f) {
  switch (d) {
    case 128: return f(std::integral_constant<int, 2>());
    case 256: return f(std::integral_constant<int, 4>());
    case 512: return f(std::integral_constant<int, 8>());
    case 1024: return f(std::integral_constant<int, 16>());
    default: skf_set_error("bf16 row kernels: feature width %d not in {128, 256, 512, 1024}", d); return SKF_EUNSUPPORTED;
  }
}
inline int row_grid(int rows) { int g = (rows + 3) / 4; return g > 4096 ? 4096 : g; }

}  // namespace

extern "C" int skf_embed_fwd_bf16(const long long* tokens, int tok_ld, int B, int L, const float* table, int vocab, int d,
                                  const float* pos, void* out, float rate, unsigned site, const void* step_state,
                                  skf_stream_t stream) {
  SKF_CHECK_ARG(tokens && table && pos && out && B > 0 && L > 0 && vocab > 0, "bad argument");
  SKF_CHECK_ARG(rate >= 0.f && rate < 1.f && (rate == 0.f || step_state), "bad dropout arguments");
  const int rows = B * L;
  hipStream_t st = (hipStream_t)stream;
  SkfProfScope ps(st, "embed_fwd_bf16", 0.0, (double)rows * d * (4.0 + 4.0 + 2.0));
  int rc = dispatch_d(d, [&](auto vpl) {
    hipLaunchKernelGGL((embed_fwd_kernel<decltype(vpl)::value>), dim3(row_grid(rows)), dim3(256), 0, st, tokens, tok_ld, L, rows,
                       table, vocab, pos, (skf_bf16*)out, rate, site, (const SkfStepState*)step_state);
    return SKF_OK;
  });
  if (rc) return rc;
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_layernorm_residual_fwd_bf16(const void* x, void* y_inout_z, const float* gamma, const float* beta, void* out,
                                               float* stats, int rows, int d, float rate, unsigned site, const void* step_state,
                                               skf_stream_t stream) {
  SKF_CHECK_ARG(x && y_inout_z && gamma && beta && out && stats && rows > 0, "bad argument");
  SKF_CHECK_ARG(rate >= 0.f && rate < 1.f && (rate == 0.f || step_state), "bad dropout arguments");
  hipStream_t st = (hipStream_t)stream;
  SkfProfScope ps(st, "ln_fwd_bf16", 0.0, (double)rows * d * 2.0 * 4.0);
  int rc = dispatch_d(d, [&](auto vpl) {
    hipLaunchKernelGGL((ln_fwd_kernel<decltype(vpl)::value>), dim3(row_grid(rows)), dim3(256), 0, st, (const skf_bf16*)x,
                       (skf_bf16*)y_inout_z, gamma, beta, (skf_bf16*)out, stats, rows, rate, site, (const SkfStepState*)step_state);
    return SKF_OK;
  });
  if (rc) return rc;
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

static int ln_bwd_blocks(int rows) { int g = (rows + 3) / 4; return g > 512 ? 512 : g; }
extern "C" size_t skf_layernorm_bwd_bf16_workspace_bytes(int rows, int d) { return (size_t)ln_bwd_blocks(rows) * 2 * d * sizeof(float); }

// dgamma / dbeta: per-workgroup partials [g][2][d] in `workspace` (g = workspace_bytes / (8 d)), column-summed into
// dgamma (and dbeta = dgamma + d when dbeta == dgamma + d) by skf_colsum
extern "C" int skf_layernorm_residual_bwd_bf16(const void* dout, const void* z, const float* stats, const float* gamma, void* dz,
                                               void* dy, float* dgamma, float* dbeta, int rows, int d, float rate, unsigned site,
                                               const void* step_state, void* workspace, size_t workspace_bytes,
                                               skf_stream_t stream) {
  return skf_layernorm_residual_bwd_bf16_rows(dout, z, stats, gamma, dz, dy, dgamma, dbeta, rows, d, rate, site, step_state, workspace,
                                              workspace_bytes, nullptr, 0, stream);
}

extern "C" int skf_layernorm_residual_bwd_bf16_rows(const void* dout, const void* z, const float* stats, const float* gamma, void* dz,
                                                    void* dy, float* dgamma, float* dbeta, int rows, int d, float rate, unsigned site,
                                                    const void* step_state, void* workspace, size_t workspace_bytes,
                                                    const int* live_len, int rows_per_sample, skf_stream_t stream) {
  SKF_CHECK_ARG(!live_len || (rows_per_sample > 0 && rows % rows_per_sample == 0), "live_len needs rows = B * rows_per_sample");
  SKF_CHECK_ARG(dout && z && stats && gamma && dz && dy && rows > 0 && ((dgamma == nullptr) == (dbeta == nullptr)), "bad argument");
  SKF_CHECK_ARG(rate >= 0.f && rate < 1.f && (rate == 0.f || step_state), "bad dropout arguments");
  SKF_CHECK_ARG(workspace && workspace_bytes >= skf_layernorm_bwd_bf16_workspace_bytes(rows, d), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int g = ln_bwd_blocks(rows);
  {
    SkfProfScope ps(st, "ln_bwd_bf16", 0.0, (double)rows * d * 2.0 * (rate > 0.f ? 4.0 : 3.0));
    int rc = dispatch_d(d, [&](auto vpl) {
      hipLaunchKernelGGL((ln_bwd_kernel<decltype(vpl)::value>), dim3(g), dim3(256), 0, st, (const skf_bf16*)dout, (const skf_bf16*)z,
                         stats, gamma, (skf_bf16*)dz, (skf_bf16*)dy, (float*)workspace, rows, rate, site, (const SkfStepState*)step_state,
                         live_len, rows_per_sample);
      return SKF_OK;
    });
    if (rc) return rc;
    SKF_LAUNCH_CHECK();
  }
  if (!dgamma) return SKF_OK;             // the caller sums the [g][2d] partial rows of `workspace` itself (batched with others)
  if (dbeta == dgamma + d) return skf_colsum((const float*)workspace, g, 2 * d, 2 * d, dgamma, 0, stream);
  int rc = skf_colsum((const float*)workspace, g, 2 * d, d, dgamma, 0, stream);
  if (rc) return rc;
  return skf_colsum((const float*)workspace + d, g, 2 * d, d, dbeta, 0, stream);
}

extern "C" int skf_softmax_ce_bf16(void* logits, int ld, int rows, int ncls, const long long* target, int tgt_ld, int tgt_cols,
                                   int tgt_off, int mask_pad, float scale, float* row_loss, float* row_hit, int write_grad,
                                   skf_stream_t stream) {
  SKF_CHECK_ARG(logits && target && row_loss && row_hit && rows > 0 && ncls > 0 && tgt_cols > 0, "bad argument");
  SKF_CHECK_ARG((ncls & 1) == 0 && (ld & 1) == 0 && ld >= ncls && ld <= 64 * CE_MAX, "bf16 CE: even ncls <= ld <= 2048");
  hipStream_t st = (hipStream_t)stream;
  SkfProfScope ps(st, "softmax_ce_bf16", 0.0, (double)rows * ncls * 2.0 * (write_grad ? 2.0 : 1.0));
  hipLaunchKernelGGL(softmax_ce_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, (skf_bf16*)logits, ld, rows, ncls, target, tgt_ld,
                     tgt_cols, tgt_off, mask_pad, scale, row_loss, row_hit, write_grad);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

constexpr int WNT = 1024;
__device__ __forceinline__ float group_sum(float v, int lanes) {     // sum over `lanes` adjacent lanes (power of two <= 64)
  for (int o = 1; o < lanes; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
static bool wide_ok(int w) { return w == 128 || w == 256 || w == 512; }

// pooling forward, 1024 threads per sample (see the backward kernels below for the access pattern)
__global__ __launch_bounds__(WNT) void pool_fwd_wide_kernel(const skf_bf16* __restrict__ u, const float* __restrict__ Vw,
                                                           const skf_bf16* __restrict__ x, int L, int U, int d,
                                                           float* __restrict__ a_out, float* __restrict__ emb) {
  extern __shared__ float sm[];        // [L] scores -> weights, [32] reduction scratch, [TG][d] column partials
  float* sc = sm;
  float* red = sm + L;
  float* colp = red + 32;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    const int LPU = U >> 3, TGU = WNT / LPU, c8 = (tid % LPU) * 8, tg = tid / LPU;
    const skf_bf16* ub = u + (size_t)b * L * U + c8;
    float vw[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) vw[e] = Vw[c8 + e];
    for (int t0 = tg; t0 < L; t0 += 4 * TGU) {
      uint4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int t = t0 + k * TGU; v[k] = *reinterpret_cast<const uint4*>(ub + (size_t)(t < L ? t : L - 1) * U); }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = t0 + k * TGU;
        float p[8];
        skf_unpack8(v[k], p);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += p[e] * vw[e];
        s = group_sum(s, LPU);
        if (c8 == 0 && t < L) sc[t] = s;
      }
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = tid; t < L; t += WNT) mx = fmaxf(mx, sc[t]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int q = 1; q < WNT / 64; ++q) mx = fmaxf(mx, red[q]);
  float s = 0.f;
  for (int t = tid; t < L; t += WNT) { const float e = __expf(sc[t] - mx); sc[t] = e; s += e; }
  s = wave_sum(s);
  if (lane == 0) red[16 + wave] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int q = 0; q < WNT / 64; ++q) tot += red[16 + q];
  const float inv = 1.0f / tot;
  for (int t = tid; t < L; t += WNT) { const float a = sc[t] * inv; sc[t] = a; a_out[(size_t)b * L + t] = a; }
  __syncthreads();
  {
    const int LPRW = d >> 3, TG = WNT / LPRW, c8 = (tid % LPRW) * 8, tg = tid / LPRW;
    const skf_bf16* xb = x + (size_t)b * L * d + c8;
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = 0.f;
    for (int t0 = tg; t0 < L; t0 += 4 * TG) {
      uint4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int t = t0 + k * TG; v[k] = *reinterpret_cast<const uint4*>(xb + (size_t)(t < L ? t : L - 1) * d); }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = t0 + k * TG;
        float p[8];
        skf_unpack8(v[k], p);
        const float at = t < L ? sc[t] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] += at * p[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) colp[tg * d + c8 + e] = g[e];
    __syncthreads();
    for (int c = tid; c < d; c += WNT) {
      float t = 0.f;
      for (int q = 0; q < TG; ++q) t += colp[q * d + c];
      emb[(size_t)b * d + c] = t;
    }
  }
}

extern "C" int skf_pool_fwd_bf16(const void* u, const float* Vw, const void* x, int B, int L, int U, int d, float* a_out, float* emb,
                                 skf_stream_t stream) {
  SKF_CHECK_ARG(u && Vw && x && a_out && emb && B > 0 && L > 0 && (U & 1) == 0 && (d & 1) == 0, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  SkfProfScope ps(st, "pool_fwd_bf16", 0.0, (double)B * L * (U + d) * 2.0);
  if (wide_ok(d) && wide_ok(U) && ((((uintptr_t)u | (uintptr_t)x) & 15) == 0))
    hipLaunchKernelGGL(pool_fwd_wide_kernel, dim3(B), dim3(WNT), (size_t)(L + 32 + 8 * WNT) * sizeof(float), st, (const skf_bf16*)u, Vw,
                       (const skf_bf16*)x, L, U, d, a_out, emb);
  else
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(B), dim3(256), (size_t)(L + 8) * sizeof(float), st, (const skf_bf16*)u, Vw, (const skf_bf16*)x,
                     L, U, d, a_out, emb);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

// ---- the same two kernels for d / 8 <= 64 lanes per row (d = 128 / 256 / 512): 1024 threads per sample, a row is one
// 16-byte access per lane of a (d/8)-lane group, four row groups of loads are in flight before the first is used, and the
// column reductions over the L rows are split over the row groups and folded through LDS in a fixed order.  (The kernels
// above walk a sample's rows one memory round trip at a time: 450 / 385 us per launch at cfg 5.)

__global__ __launch_bounds__(WNT) void expander_bwd_wide_kernel(const skf_bf16* __restrict__ dpre, const float* __restrict__ emb,
                                                               const float* __restrict__ w, int L, int d, float* __restrict__ demb,
                                                               int demb_accumulate, float* __restrict__ dw_part, float* __restrict__ db_part) {
  extern __shared__ float sm[];                 // [TG][d] column partials
  const int b = blockIdx.x, tid = threadIdx.x;
  const int LPRW = d >> 3, TG = WNT / LPRW, c8 = (tid % LPRW) * 8, tg = tid / LPRW;
  const skf_bf16* pb = dpre + (size_t)b * L * d + c8;
  float e8[8], g[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { e8[e] = emb[(size_t)b * d + c8 + e]; g[e] = 0.f; }
  for (int t0 = tg; t0 < L; t0 += 4 * TG) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int t = t0 + u * TG; v[u] = *reinterpret_cast<const uint4*>(pb + (size_t)(t < L ? t : L - 1) * d); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * TG;
      float p[8];
      skf_unpack8(v[u], p);
      const float wt = t < L ? w[t] : 0.f;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { s0 += p[e] * e8[e]; s1 += p[e]; g[e] += p[e] * wt; }
      s0 = group_sum(s0, LPRW); s1 = group_sum(s1, LPRW);
      if (c8 == 0 && t < L) { dw_part[(size_t)b * L + t] = s0; db_part[(size_t)b * L + t] = s1; }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sm[tg * d + c8 + e] = g[e];
  __syncthreads();
  for (int c = tid; c < d; c += WNT) {
    float t = 0.f;
    for (int q = 0; q < TG; ++q) t += sm[q * d + c];
    float* dst = demb + (size_t)b * d + c;
    *dst = demb_accumulate ? *dst + t : t;
  }
}

__global__ __launch_bounds__(WNT) void pool_bwd_wide_kernel(skf_bf16* __restrict__ u, const float* __restrict__ Vw,
                                                           const skf_bf16* __restrict__ x, const float* __restrict__ a_in,
                                                           const float* __restrict__ demb, int L, int U, int d,
                                                           skf_bf16* __restrict__ dx, float* __restrict__ dV_part) {
  extern __shared__ float sm[];                 // [L] da -> dscore, [L] a, [16] scratch, [TGU][U] column partials
  float* ds = sm;
  float* av = sm + L;
  float* red = sm + 2 * L;
  float* colp = red + 16;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int t = tid; t < L; t += WNT) av[t] = a_in[(size_t)b * L + t];
  __syncthreads();
  {
    // da[t] = x[t] . demb ; dx[t] = a[t] * demb
    const int LPRW = d >> 3, TG = WNT / LPRW, c8 = (tid % LPRW) * 8, tg = tid / LPRW;
    const skf_bf16* xb = x + (size_t)b * L * d + c8;
    skf_bf16* dxb = dx + (size_t)b * L * d + c8;
    float de[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) de[e] = demb[(size_t)b * d + c8 + e];
    for (int t0 = tg; t0 < L; t0 += 4 * TG) {
      uint4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int t = t0 + k * TG; v[k] = *reinterpret_cast<const uint4*>(xb + (size_t)(t < L ? t : L - 1) * d); }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = t0 + k * TG;
        float p[8], o[8];
        skf_unpack8(v[k], p);
        const float at = t < L ? av[t] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s += p[e] * de[e]; o[e] = at * de[e]; }
        s = group_sum(s, LPRW);
        if (t < L) {
          *reinterpret_cast<uint4*>(dxb + (size_t)t * d) = skf_pack8(o);
          if (c8 == 0) ds[t] = s;
        }
      }
    }
  }
  __syncthreads();
  float dot = 0.f;
  for (int t = tid; t < L; t += WNT) dot += av[t] * ds[t];
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = 0.f;
#pragma unroll
  for (int q = 0; q < WNT / 64; ++q) dot += red[q];
  __syncthreads();
  for (int t = tid; t < L; t += WNT) ds[t] = av[t] * (ds[t] - dot);
  __syncthreads();
  {
    // dV = sum_t dscore[t] u[t] ; u <- dscore[t] * V * (1 - u^2)   (the gradient at the tanh projection's output)
    const int LPU = U >> 3, TGU = WNT / LPU, c8 = (tid % LPU) * 8, tg = tid / LPU;
    skf_bf16* ub = u + (size_t)b * L * U + c8;
    float vw[8], g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { vw[e] = Vw[c8 + e]; g[e] = 0.f; }
    for (int t0 = tg; t0 < L; t0 += 4 * TGU) {
      uint4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int t = t0 + k * TGU; v[k] = *reinterpret_cast<const uint4*>(ub + (size_t)(t < L ? t : L - 1) * U); }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = t0 + k * TGU;
        if (t >= L) continue;
        float p[8], o[8];
        skf_unpack8(v[k], p);
        const float dt = ds[t];
#pragma unroll
        for (int e = 0; e < 8; ++e) { g[e] += dt * p[e]; o[e] = dt * vw[e] * (1.f - p[e] * p[e]); }
        *reinterpret_cast<uint4*>(ub + (size_t)t * U) = skf_pack8(o);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) colp[tg * U + c8 + e] = g[e];
    __syncthreads();
    for (int c = tid; c < U; c += WNT) {
      float t = 0.f;
      for (int q = 0; q < TGU; ++q) t += colp[q * U + c];
      dV_part[(size_t)b * U + c] = t;
    }
  }
}

// workspace >= B*U floats (per-sample dV partials); dV receives their sum
extern "C" int skf_pool_bwd_bf16(void* u_inout_dpre, const float* Vw, const void* x, const float* a, const float* demb, int B, int L,
                                 int U, int d, void* dx, float* dV, void* workspace, size_t workspace_bytes, skf_stream_t stream) {
  SKF_CHECK_ARG(u_inout_dpre && Vw && x && a && demb && dx && dV && B > 0 && L > 0 && (U & 1) == 0 && (d & 1) == 0, "bad argument");
  SKF_CHECK_ARG(workspace && workspace_bytes >= (size_t)B * U * sizeof(float), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  {
    SkfProfScope ps(st, "pool_bwd_bf16", 0.0, (double)B * L * (2.0 * U + 2.0 * d) * 2.0);
    const bool al = ((((uintptr_t)u_inout_dpre | (uintptr_t)x | (uintptr_t)dx) & 15) == 0);
    if (wide_ok(d) && wide_ok(U) && al)
      hipLaunchKernelGGL(pool_bwd_wide_kernel, dim3(B), dim3(WNT), (size_t)(2 * L + 16 + 8 * WNT) * sizeof(float), st, (skf_bf16*)u_inout_dpre, Vw,
                         (const skf_bf16*)x, a, demb, L, U, d, (skf_bf16*)dx, (float*)workspace);
    else
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(B), dim3(256), (size_t)(2 * L + 8) * sizeof(float), st, (skf_bf16*)u_inout_dpre, Vw,
                       (const skf_bf16*)x, a, demb, L, U, d, (skf_bf16*)dx, (float*)workspace);
    SKF_LAUNCH_CHECK();
  }
  return skf_colsum((const float*)workspace, B, U, U, dV, 0, stream);
}

extern "C" int skf_expander_fwd_bf16(const float* emb, const float* w, const float* bias, int B, int L, int d, void* pre,
                                     skf_stream_t stream) {
  SKF_CHECK_ARG(emb && w && bias && pre && B > 0 && L > 0 && (d & 1) == 0, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  SkfProfScope ps(st, "expander_fwd_bf16", 0.0, (double)B * L * d * 2.0);
  hipLaunchKernelGGL(expander_fwd_kernel, dim3(2048), dim3(256), 0, st, emb, w, bias, B, L, d, (skf_bf16*)pre);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

// workspace >= 2*B*L floats
extern "C" int skf_expander_bwd_bf16(const void* dpre, const float* emb, const float* w, int B, int L, int d, float* demb,
                                     int demb_accumulate, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                                     skf_stream_t stream) {
  SKF_CHECK_ARG(dpre && emb && w && demb && dw && dbias && B > 0 && L > 0 && (d & 1) == 0, "bad argument");
  SKF_CHECK_ARG(workspace && workspace_bytes >= (size_t)2 * B * L * sizeof(float), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* dwp = (float*)workspace;
  float* dbp = dwp + (size_t)B * L;
  {
    SkfProfScope ps(st, "expander_bwd_bf16", 0.0, (double)B * L * d * 2.0 * 2.0);
    if (wide_ok(d) && ((uintptr_t)dpre & 15) == 0)
      hipLaunchKernelGGL(expander_bwd_wide_kernel, dim3(B), dim3(WNT), (size_t)8 * WNT * sizeof(float), st, (const skf_bf16*)dpre, emb, w, L, d, demb,
                         demb_accumulate, dwp, dbp);
    else
    hipLaunchKernelGGL(expander_bwd_kernel, dim3(B), dim3(256), 0, st, (const skf_bf16*)dpre, emb, w, L, d, demb, demb_accumulate, dwp, dbp);
    SKF_LAUNCH_CHECK();
  }
  int rc = skf_colsum(dwp, B, L, L, dw, 0, stream);
  if (rc) return rc;
  return skf_colsum(dbp, B, L, L, dbias, 0, stream);
}

// fp32 master weight [R][C] (row stride ld_src) -> bf16 images: dst [R][ld_dst] and / or dst_t [C][ld_t] (either may be NULL);
// pad columns of both images are written as zeros
extern "C" int skf_cast_weight_bf16(const float* src, int R, int C, int ld_src, void* dst, int ld_dst, void* dst_t, int ld_t,
                                    skf_stream_t stream) {
  SKF_CHECK_ARG(src && R > 0 && C > 0 && ld_src >= C && (dst || dst_t), "bad argument");
  SKF_CHECK_ARG((!dst || ld_dst >= C) && (!dst_t || ld_t >= R), "image pitch smaller than the matrix");
  const int cols = dst ? (ld_dst > C ? ld_dst : C) : C, rws = dst_t ? (ld_t > R ? ld_t : R) : R;
  hipLaunchKernelGGL(cast_weight_kernel, dim3((cols + 31) / 32, (rws + 31) / 32), dim3(256), 0, (hipStream_t)stream, src, R, C, ld_src,
                     (skf_bf16*)dst, ld_dst, (skf_bf16*)dst_t, ld_t);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_cast_weight_bf16_blocks(int R, int C, int ld_dst, int ld_t, int* blocks_x) {
  const int cols = ld_dst > C ? ld_dst : C, rws = ld_t > R ? ld_t : R;
  if (blocks_x) *blocks_x = (cols + 31) / 32;
  return ((cols + 31) / 32) * ((rws + 31) / 32);
}
extern "C" int skf_cast_weight_bf16_batch(const SkfCastDesc* descs_dev, int n, int total_blocks, skf_stream_t stream) {
  SKF_CHECK_ARG(descs_dev && n > 0 && total_blocks > 0, "bad argument");
  hipLaunchKernelGGL(cast_weight_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, n);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_cast_f32_to_bf16(const float* src, void* dst, size_t n, skf_stream_t stream) {
  SKF_CHECK_ARG(src && dst && n > 0 && ((uintptr_t)dst & 3) == 0, "bad argument");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, src, (skf_bf16*)dst, n);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
extern "C" int skf_cast_bf16_to_f32(const void* src, float* dst, size_t n, skf_stream_t stream) {
  SKF_CHECK_ARG(src && dst && n > 0, "bad argument");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (const skf_bf16*)src, dst, n);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
