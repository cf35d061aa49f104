// Any-shape fallbacks of the fp32 path: LayerNorm / DenseExpander backward for a d_model that is not one of {64,128,256,512}
// and scaled-dot-product attention for a head size that is not one of {16,32,64}.
//
// The reference takes any d_model % num_heads == 0 (builders/layers/transformer.py:150-152, MultiHeadAttention); the MFMA
// kernels of skf_rowops.hip / skf_attention.hip are built for the sizes above (every BASELINE config).  Everything else runs
// here: plain fp32 FMA kernels, one wave per row / per (sample, head, query) / per (sample, head, key), same semantics
// (LayerNormalization(1e-6) over z = x + Dropout(y) with the counter-hash masks of skf_common.h; logits (q.k)/sqrt(dh) with
// masked entries SET to -1e9, softmax over keys, saved statistics (base-2 row maximum, 1 / row sum) like the MFMA kernels),
// tested against the same oracle.  They are correct, not fast - a drop-in must accept the shapes the reference accepts.
#include "skf_common.h"
#include "skf_attention_params.h"

namespace {

constexpr int kMaxV = 16;        // columns per lane: d_model <= 64 * kMaxV = 1024

__global__ __launch_bounds__(256) void ln_fwd_any_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ out, float* __restrict__ stats,
                                                         int rows, int D, float rate, uint32_t site, const SkfStepState* st) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const size_t off = (size_t)row * D;
    float z[kMaxV];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < kMaxV; ++v) {
      const int c = lane + 64 * v;
      z[v] = 0.f;
      if (c < D) {
        float yy = y[off + c];
        if (rate > 0.f) yy *= skf_keep(sk, (uint32_t)(off + c), thresh) ? inv_keep : 0.f;
        z[v] = x[off + c] + yy;
        s += z[v];
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < kMaxV; ++v)
      if (lane + 64 * v < D) { const float c = z[v] - mean; q += c * c; }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-6f);
#pragma unroll
    for (int v = 0; v < kMaxV; ++v) {
      const int c = lane + 64 * v;
      if (c < D) {
        y[off + c] = z[v];
        out[off + c] = (z[v] - mean) * rstd * gamma[c] + beta[c];
      }
    }
    if (lane == 0) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
  }
}

// part[block][2][D] = per-workgroup (dgamma, dbeta) partials, like the templated kernels of skf_rowops.hip
__global__ __launch_bounds__(256) void ln_bwd_any_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                         const float* __restrict__ stats, const float* __restrict__ gamma,
                                                         float* __restrict__ dz, float* __restrict__ dy, float* __restrict__ part, int rows,
                                                         int D, float rate, uint32_t site, const SkfStepState* st) {
  extern __shared__ float red_any[];      // [3 waves][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  float dg[kMaxV], db[kMaxV];
#pragma unroll
  for (int v = 0; v < kMaxV; ++v) { dg[v] = 0.f; db[v] = 0.f; }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const size_t off = (size_t)row * D;
    const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
    float xh[kMaxV], gg[kMaxV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < kMaxV; ++v) {
      const int c = lane + 64 * v;
      xh[v] = 0.f; gg[v] = 0.f;
      if (c < D) {
        const float d = dout[off + c];
        xh[v] = (z[off + c] - mean) * rstd;
        gg[v] = d * gamma[c];
        dg[v] += d * xh[v]; db[v] += d;
        s1 += gg[v]; s2 += gg[v] * xh[v];
      }
    }
    s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int v = 0; v < kMaxV; ++v) {
      const int c = lane + 64 * v;
      if (c < D) {
        const float g = rstd * (gg[v] - s1 - xh[v] * s2);
        dz[off + c] = g;
        if (dy) dy[off + c] = rate > 0.f ? g * (skf_keep(sk, (uint32_t)(off + c), thresh) ? inv_keep : 0.f) : g;
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int v = 0; v < kMaxV; ++v) {
      const int c = lane + 64 * v;
      if (c < D) { red_any[((wave - 1) * 2 + 0) * D + c] = dg[v]; red_any[((wave - 1) * 2 + 1) * D + c] = db[v]; }
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int v = 0; v < kMaxV; ++v) {
      const int c = lane + 64 * v;
      if (c < D) {
        part[(size_t)blockIdx.x * 2 * D + c] = dg[v] + red_any[0 * D + c] + red_any[2 * D + c] + red_any[4 * D + c];
        part[(size_t)blockIdx.x * 2 * D + D + c] = db[v] + red_any[1 * D + c] + red_any[3 * D + c] + red_any[5 * D + c];
      }
    }
  }
}

// DenseExpander backward (builders/layers/transformer.py:370-376), one workgroup per sample:
// demb[c] (+)= sum_t dpre[t][c] w[t]; dw_part[b][t] = sum_c dpre[t][c] emb[c]; db_part[b][t] = sum_c dpre[t][c]
__global__ __launch_bounds__(256) void expander_bwd_any_kernel(const float* __restrict__ dpre, const float* __restrict__ emb,
                                                               const float* __restrict__ w, int L, int d, float* __restrict__ demb,
                                                               int demb_accumulate, float* __restrict__ dw_part, float* __restrict__ db_part) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* pb = dpre + (size_t)b * L * d;
  const float* eb = emb + (size_t)b * d;
  for (int t = wave; t < L; t += 4) {
    float s0 = 0.f, s1 = 0.f;
    for (int c = lane; c < d; c += 64) { const float v = pb[(size_t)t * d + c]; s0 += v * eb[c]; s1 += v; }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (lane == 0) { dw_part[(size_t)b * L + t] = s0; db_part[(size_t)b * L + t] = s1; }
  }
  for (int c = tid; c < d; c += 256) {
    float g = 0.f;
    for (int t = 0; t < L; ++t) g += pb[(size_t)t * d + c] * w[t];
    float* dst = demb + (size_t)b * d + c;
    *dst = demb_accumulate ? *dst + g : g;
  }
}

// ------------------------------------------------------------------ attention, any head size <= 128, any Lk <= 1024
constexpr int kMaxDh = 128, kMaxKeysPerLane = 16;

__device__ __forceinline__ float masked_score2(float dot, float c2, const unsigned char* km, int key, int q, int causal) {
  const bool m = (km && km[key]) || (causal && key > q);
  return m ? -1e9f : dot * c2;
}

// one wave per (sample, head, query): scores of keys lane, lane + 64, ... in registers
__global__ __launch_bounds__(256) void attn_fwd_any_kernel(AttnParams p, int DH) {
  __shared__ float qs[4][kMaxDh];
  __shared__ float ps[4][64 * kMaxKeysPerLane];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row_ = blockIdx.x * 4 + wave;                  // (b, h, q) flattened
  const bool active = row_ < p.B * p.H * p.Lq;             // (inactive waves of the last workgroup redo row 0 and store nothing)
  const int row = active ? row_ : 0;
  const int q = row % p.Lq, bh = row / p.Lq, h = bh % p.H, b = bh / p.H;
  const float c2 = 1.44269504088896340736f / sqrtf((float)DH);
  const float* qp = p.Q + (size_t)(b * p.Lq + q) * p.ldq + h * DH;
  for (int d = lane; d < DH; d += 64) qs[wave][d] = qp[d];
  __syncthreads();
  const unsigned char* km = p.key_mask ? p.key_mask + (size_t)b * p.key_mask_ld : nullptr;
  float s[kMaxKeysPerLane];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < kMaxKeysPerLane; ++j) {
    const int key = lane + 64 * j;
    s[j] = -INFINITY;
    if (key < p.Lk) {
      const float* kp = p.K + (size_t)(b * p.Lk + key) * p.ldk + h * DH;
      float dot = 0.f;
      for (int d = 0; d < DH; ++d) dot += qs[wave][d] * kp[d];
      s[j] = masked_score2(dot, c2, km, key, q, p.causal);
      mx = fmaxf(mx, s[j]);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxKeysPerLane; ++j) {
    const int key = lane + 64 * j;
    const float e = key < p.Lk ? exp2f(s[j] - mx) : 0.f;
    if (key < p.Lk) ps[wave][key] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float rinv = 1.0f / sum;
  __syncthreads();
  if (active && lane == 0 && p.stats) { p.stats[2 * (size_t)row] = mx; p.stats[2 * (size_t)row + 1] = rinv; }
  float* op = p.O + (size_t)(b * p.Lq + q) * p.ldo + h * DH;
  for (int d = lane; d < DH; d += 64) {
    float acc = 0.f;
    for (int key = 0; key < p.Lk; ++key) acc += ps[wave][key] * p.V[(size_t)(b * p.Lk + key) * p.ldv + h * DH + d];
    if (active) op[d] = acc * rinv;
  }
}

// The attention weights themselves, softmax(q.k/sqrt(dh) + mask*-1e9) as a (B, H, Lq, Lk) tensor: what
// builders/utils.py:105 returns beside the output.  Never used by the train step (models/sketchformer.py:140-145 drops them);
// materialised only when a caller of the builders front-end asks.  One wave per (sample, head, query), the arithmetic of
// attn_fwd_any_kernel.
__global__ __launch_bounds__(256) void attn_weights_any_kernel(AttnParams p, int DH, float* __restrict__ W) {
  __shared__ float qs[4][kMaxDh];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row_ = blockIdx.x * 4 + wave;
  const bool active = row_ < p.B * p.H * p.Lq;
  const int row = active ? row_ : 0;
  const int q = row % p.Lq, bh = row / p.Lq, h = bh % p.H, b = bh / p.H;
  const float c2 = 1.44269504088896340736f / sqrtf((float)DH);
  const float* qp = p.Q + (size_t)(b * p.Lq + q) * p.ldq + h * DH;
  for (int d = lane; d < DH; d += 64) qs[wave][d] = qp[d];
  __syncthreads();
  const unsigned char* km = p.key_mask ? p.key_mask + (size_t)b * p.key_mask_ld : nullptr;
  float s[kMaxKeysPerLane];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < kMaxKeysPerLane; ++j) {
    const int key = lane + 64 * j;
    s[j] = -INFINITY;
    if (key < p.Lk) {
      const float* kp = p.K + (size_t)(b * p.Lk + key) * p.ldk + h * DH;
      float dot = 0.f;
      for (int d = 0; d < DH; ++d) dot += qs[wave][d] * kp[d];
      s[j] = masked_score2(dot, c2, km, key, q, p.causal);
      mx = fmaxf(mx, s[j]);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxKeysPerLane; ++j) {
    s[j] = lane + 64 * j < p.Lk ? exp2f(s[j] - mx) : 0.f;
    sum += s[j];
  }
  sum = wave_sum(sum);
  const float rinv = 1.0f / sum;
  float* wp = W + (size_t)row * p.Lk;
#pragma unroll
  for (int j = 0; j < kMaxKeysPerLane; ++j) {
    const int key = lane + 64 * j;
    if (active && key < p.Lk) wp[key] = s[j] * rinv;
  }
}

// scaled_dot_product_attention with an ARBITRARY float mask (builders/utils.py:90-105): logits = q.k / sqrt(dh) + mask * -1e9 - the mask is
// ADDED (a value of 0.5 lowers a logit by 5e8, it does not remove the key), broadcast over any of (sample, head, query) through zero
// strides.  One wave per (sample, head, query): softmax over the keys, O = weights . V, optionally the weights.  The padding / look-ahead
// masks the model itself builds never come here (the MFMA kernels take them as bytes); this is the reference's signature for everything else.
__global__ __launch_bounds__(256) void attn_fwd_fmask_kernel(AttnParams p, int DH, const float* __restrict__ mask, long ms_b, long ms_h, long ms_q,
                                                             float* __restrict__ W) {
  __shared__ float qs[4][kMaxDh];
  __shared__ float ws[4][64 * kMaxKeysPerLane];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row_ = blockIdx.x * 4 + wave;
  const bool active = row_ < p.B * p.H * p.Lq;
  const int row = active ? row_ : 0;
  const int q = row % p.Lq, bh = row / p.Lq, h = bh % p.H, b = bh / p.H;
  const float sq = sqrtf((float)DH);
  const float* qp = p.Q + (size_t)(b * p.Lq + q) * p.ldq + h * DH;
  for (int d = lane; d < DH; d += 64) qs[wave][d] = qp[d];
  __syncthreads();
  const float* mrow = mask ? mask + b * ms_b + h * ms_h + q * ms_q : nullptr;
  float s[kMaxKeysPerLane];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < kMaxKeysPerLane; ++j) {
    const int key = lane + 64 * j;
    s[j] = -INFINITY;
    if (key < p.Lk) {
      const float* kp = p.K + (size_t)(b * p.Lk + key) * p.ldk + h * DH;
      float dot = 0.f;
      for (int d = 0; d < DH; ++d) dot += qs[wave][d] * kp[d];
      s[j] = dot / sq;                                   // (divide after the matmul, then add: builders/utils.py:92-97)
      if (mrow) s[j] += mrow[key] * -1e9f;
      mx = fmaxf(mx, s[j]);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxKeysPerLane; ++j) {
    s[j] = lane + 64 * j < p.Lk ? __expf(s[j] - mx) : 0.f;
    sum += s[j];
  }
  sum = wave_sum(sum);
  const float rinv = 1.0f / sum;
#pragma unroll
  for (int j = 0; j < kMaxKeysPerLane; ++j) {
    const int key = lane + 64 * j;
    if (key < p.Lk) {
      ws[wave][key] = s[j] * rinv;
      if (W && active) W[(size_t)row * p.Lk + key] = s[j] * rinv;
    }
  }
  __syncthreads();
  float* op = p.O + (size_t)(b * p.Lq + q) * p.ldo + h * DH;
  for (int d = lane; d < DH; d += 64) {
    float acc = 0.f;
    for (int key = 0; key < p.Lk; ++key) acc += ws[wave][key] * p.V[(size_t)(b * p.Lk + key) * p.ldv + h * DH + d];
    if (active) op[d] = acc;
  }
}

// Row reductions behind LossManager.add_mae_loss / add_mse_loss / add_mean_loss (builders/losses.py:68-75): out[r] = mean over the
// last axis of |a - b| (mode 1), (a - b)^2 (mode 2) or a (mode 0, b ignored).  One wave per row.
__global__ __launch_bounds__(256) void row_mean_kernel(const float* __restrict__ a, const float* __restrict__ b, long rows, int cols, int mode,
                                                       float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float acc = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const float x = a[row * cols + c];
    if (mode == 0) acc += x;
    else { const float d = x - b[row * cols + c]; acc += mode == 1 ? fabsf(d) : d * d; }
  }
  acc = wave_sum(acc);
  if (lane == 0) out[row] = acc / (float)cols;
}

// dQ: one wave per (sample, head, query).  P from the saved statistics, delta = dO . O, dS = P o (dP - delta)
__global__ __launch_bounds__(256) void attn_bwd_q_any_kernel(AttnParams p, int DH) {
  __shared__ float qs[4][kMaxDh], dos[4][kMaxDh];
  __shared__ float dss[4][64 * kMaxKeysPerLane];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row_ = blockIdx.x * 4 + wave;
  const bool active = row_ < p.B * p.H * p.Lq;
  const int row = active ? row_ : 0;
  const int q = row % p.Lq, bh = row / p.Lq, h = bh % p.H, b = bh / p.H;
  const float c2 = 1.44269504088896340736f / sqrtf((float)DH), inv_sqrt = 1.0f / sqrtf((float)DH);
  const size_t qoff = (size_t)(b * p.Lq + q);
  float dl = 0.f;
  for (int d = lane; d < DH; d += 64) {
    qs[wave][d] = p.Q[qoff * p.ldq + h * DH + d];
    const float dv = p.dO[qoff * p.lddo + h * DH + d];
    dos[wave][d] = dv;
    dl += dv * p.O[qoff * p.ldo + h * DH + d];
  }
  const float delta = wave_sum(dl);
  __syncthreads();
  const float mx = p.stats[2 * (size_t)row], rinv = p.stats[2 * (size_t)row + 1];
  const unsigned char* km = p.key_mask ? p.key_mask + (size_t)b * p.key_mask_ld : nullptr;
  for (int key = lane; key < p.Lk; key += 64) {
    const float* kp = p.K + (size_t)(b * p.Lk + key) * p.ldk + h * DH;
    const float* vp = p.V + (size_t)(b * p.Lk + key) * p.ldv + h * DH;
    float dot = 0.f, dp = 0.f;
    for (int d = 0; d < DH; ++d) { dot += qs[wave][d] * kp[d]; dp += dos[wave][d] * vp[d]; }
    const float pv = exp2f(masked_score2(dot, c2, km, key, q, p.causal) - mx) * rinv;
    dss[wave][key] = pv * (dp - delta);
  }
  __syncthreads();
  float* dq = p.dQ + qoff * p.lddq + h * DH;
  for (int d = lane; d < DH; d += 64) {
    float acc = 0.f;
    for (int key = 0; key < p.Lk; ++key) acc += dss[wave][key] * p.K[(size_t)(b * p.Lk + key) * p.ldk + h * DH + d];
    if (active) dq[d] = acc * inv_sqrt;
  }
}

// dK, dV: one wave per (sample, head, key); lanes over the queries, delta of a query recomputed (dO . O)
__global__ __launch_bounds__(256) void attn_bwd_kv_any_kernel(AttnParams p, int DH) {
  __shared__ float ks[4][kMaxDh], vs[4][kMaxDh];
  __shared__ float pq[4][64 * kMaxKeysPerLane], dsq[4][64 * kMaxKeysPerLane];     // P and dS of this key for every query (Lq <= 1024)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row_ = blockIdx.x * 4 + wave;                  // (b, h, key)
  const bool active = row_ < p.B * p.H * p.Lk;
  const int row = active ? row_ : 0;
  const int key = row % p.Lk, bh = row / p.Lk, h = bh % p.H, b = bh / p.H;
  const float c2 = 1.44269504088896340736f / sqrtf((float)DH), inv_sqrt = 1.0f / sqrtf((float)DH);
  const size_t koff = (size_t)(b * p.Lk + key);
  for (int d = lane; d < DH; d += 64) { ks[wave][d] = p.K[koff * p.ldk + h * DH + d]; vs[wave][d] = p.V[koff * p.ldv + h * DH + d]; }
  __syncthreads();
  const unsigned char* km = p.key_mask ? p.key_mask + (size_t)b * p.key_mask_ld : nullptr;
  for (int q = lane; q < p.Lq; q += 64) {
    const size_t qoff = (size_t)(b * p.Lq + q);
    const float* qp = p.Q + qoff * p.ldq + h * DH;
    const float* dop = p.dO + qoff * p.lddo + h * DH;
    const float* op = p.O + qoff * p.ldo + h * DH;
    float dot = 0.f, dp = 0.f, delta = 0.f;
    for (int d = 0; d < DH; ++d) { dot += qp[d] * ks[wave][d]; dp += dop[d] * vs[wave][d]; delta += dop[d] * op[d]; }
    const size_t srow = (size_t)bh * p.Lq + q;
    const float pv = exp2f(masked_score2(dot, c2, km, key, q, p.causal) - p.stats[2 * srow]) * p.stats[2 * srow + 1];
    pq[wave][q] = pv;
    dsq[wave][q] = pv * (dp - delta);
  }
  __syncthreads();
  float* dk = p.dK + koff * p.lddk + h * DH;
  float* dv = p.dV + koff * p.lddv + h * DH;
  for (int d = lane; d < DH; d += 64) {
    float ak = 0.f, av = 0.f;
    for (int q = 0; q < p.Lq; ++q) {
      const size_t qoff = (size_t)(b * p.Lq + q);
      ak += dsq[wave][q] * p.Q[qoff * p.ldq + h * DH + d];
      av += pq[wave][q] * p.dO[qoff * p.lddo + h * DH + d];
    }
    if (active) { dk[d] = ak * inv_sqrt; dv[d] = av; }
  }
}

}  // namespace

int skf_ln_fwd_any(const float* x, float* y_z, const float* gamma, const float* beta, float* out, float* stats, int rows, int d, float rate,
                   unsigned site, const void* st, int grid, hipStream_t s) {
  SKF_CHECK_ARG(d > 0 && d <= 64 * kMaxV, "d_model > 1024 not supported");
  hipLaunchKernelGGL(ln_fwd_any_kernel, dim3(grid), dim3(256), 0, s, x, y_z, gamma, beta, out, stats, rows, d, rate, site, (const SkfStepState*)st);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

int skf_ln_bwd_any(const float* dout, const float* z, const float* stats, const float* gamma, float* dz, float* dy, float* part, int rows, int d,
                   float rate, unsigned site, const void* st, int grid, hipStream_t s) {
  SKF_CHECK_ARG(d > 0 && d <= 64 * kMaxV, "d_model > 1024 not supported");
  hipLaunchKernelGGL(ln_bwd_any_kernel, dim3(grid), dim3(256), (size_t)6 * d * sizeof(float), s, dout, z, stats, gamma, dz, dy, part, rows, d, rate,
                     site, (const SkfStepState*)st);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

int skf_expander_bwd_any(const float* dpre, const float* emb, const float* w, int B, int L, int d, float* demb, int demb_accumulate, float* p1,
                         float* p2, hipStream_t s) {
  hipLaunchKernelGGL(expander_bwd_any_kernel, dim3(B), dim3(256), 0, s, dpre, emb, w, L, d, demb, demb_accumulate, p1, p2);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

int skf_attention_any_supported(int dh, int Lq, int Lk) { return dh > 0 && dh <= kMaxDh && Lq <= 64 * kMaxKeysPerLane && Lk <= 64 * kMaxKeysPerLane; }

int skf_attention_fwd_any(const AttnParams& p, int dh, hipStream_t s) {
  SKF_CHECK_ARG(skf_attention_any_supported(dh, p.Lq, p.Lk), "head size > 128 or sequence > 1024");
  SkfProfScope ps(s, "attn_fwd<any>", 4.0 * p.B * p.H * (double)p.Lq * p.Lk * dh, 4.0 * p.B * p.H * dh * (2.0 * p.Lq + 2.0 * p.Lk));
  hipLaunchKernelGGL(attn_fwd_any_kernel, dim3(skf_cdiv((long)p.B * p.H * p.Lq, 4)), dim3(256), 0, s, p, dh);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

int skf_attention_bwd_any(const AttnParams& p, int dh, hipStream_t s) {
  SKF_CHECK_ARG(skf_attention_any_supported(dh, p.Lq, p.Lk), "head size > 128 or sequence > 1024");
  SkfProfScope ps(s, "attn_bwd<any>", 8.0 * p.B * p.H * (double)p.Lq * p.Lk * dh, 4.0 * p.B * p.H * dh * (4.0 * p.Lq + 4.0 * p.Lk));
  hipLaunchKernelGGL(attn_bwd_q_any_kernel, dim3(skf_cdiv((long)p.B * p.H * p.Lq, 4)), dim3(256), 0, s, p, dh);
  hipLaunchKernelGGL(attn_bwd_kv_any_kernel, dim3(skf_cdiv((long)p.B * p.H * p.Lk, 4)), dim3(256), 0, s, p, dh);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_attention_weights(const float* Q, int ldq, const float* K, int ldk, const unsigned char* key_mask, int key_mask_ld,
                                     int causal, int B, int H, int Lq, int Lk, int dh, float* W, skf_stream_t stream) {
  SKF_CHECK_ARG(Q && K && W, "null operand");
  SKF_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && Lk > 0, "empty problem");
  SKF_CHECK_ARG(skf_attention_any_supported(dh, Lq, Lk), "head size > 128 or sequence > 1024");
  SKF_CHECK_ARG(!causal || Lq == Lk, "causal attention needs Lq == Lk");
  AttnParams p{};
  p.Q = Q; p.K = K; p.ldq = ldq; p.ldk = ldk; p.key_mask = key_mask; p.key_mask_ld = key_mask_ld; p.causal = causal;
  p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk;
  hipLaunchKernelGGL(attn_weights_any_kernel, dim3(skf_cdiv((long)B * H * Lq, 4)), dim3(256), 0, (hipStream_t)stream, p, dh, W);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_attention_fwd_float_mask(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* mask,
                                            long mask_stride_b, long mask_stride_h, long mask_stride_q, int B, int H, int Lq, int Lk, int dh,
                                            float* O, int ldo, float* W, skf_stream_t stream) {
  SKF_CHECK_ARG(Q && K && V && O, "null operand");
  SKF_CHECK_ARG(B > 0 && H > 0 && Lq > 0 && Lk > 0, "empty problem");
  SKF_CHECK_ARG(skf_attention_any_supported(dh, Lq, Lk), "head size > 128 or sequence > 1024");
  SKF_CHECK_ARG(mask_stride_b >= 0 && mask_stride_h >= 0 && mask_stride_q >= 0, "mask strides are element counts >= 0 (0 = broadcast)");
  AttnParams p{};
  p.Q = Q; p.K = K; p.V = V; p.O = O; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk;
  SkfProfScope ps((hipStream_t)stream, "attn_fwd<float mask>", 4.0 * B * H * (double)Lq * Lk * dh, 4.0 * B * H * dh * (2.0 * Lq + 2.0 * Lk));
  hipLaunchKernelGGL(attn_fwd_fmask_kernel, dim3(skf_cdiv((long)B * H * Lq, 4)), dim3(256), 0, (hipStream_t)stream, p, dh, mask, mask_stride_b,
                     mask_stride_h, mask_stride_q, W);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_row_mean(const float* a, const float* b, long rows, int cols, int mode, float* out, skf_stream_t stream) {
  SKF_CHECK_ARG(a && out && rows > 0 && cols > 0, "null operand / empty problem");
  SKF_CHECK_ARG(mode >= 0 && mode <= 2, "mode must be 0 (mean), 1 (mean absolute error) or 2 (mean squared error)");
  SKF_CHECK_ARG(mode == 0 || b, "the error modes need both operands");
  hipLaunchKernelGGL(row_mean_kernel, dim3(skf_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, a, b, rows, cols, mode, out);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
