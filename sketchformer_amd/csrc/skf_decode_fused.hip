// One kernel per emitted position of the KV-cached greedy reconstruction (models/sketchformer.py:255-311
// predict_from_embedding; decoder layer = builders/layers/transformer.py:245-262, 325-344).
//
// The samples of a batch never interact during decoding, so one workgroup owns one sample and walks the whole step:
//   x = embed(token_i) * sqrt(d) + pos[i]
//   per layer: [q|k|v] = x Wqkv ; k|v appended to the cache ; attention of q over keys 0..i (target padding mask) ;
//              o-projection ; LN ; q2 = . Wq ; attention over the cached K|V of pre_decoder ; o-projection ; LN ; FFN ; LN
//   logits -> argmax (first index on ties) / stroke-5 row -> appended ; EOS bookkeeping
// The round-1 path issued this as 51 dependent launches of ~7 us each (0.35 ms per position, all latency); here the
// activations of the position never leave LDS, and what the kernel waits for is the weight stream: every workgroup reads
// the decoder's weights (0.9 MB per layer at d = 128) from the L2 once per position - 16-byte loads, 16 in flight per thread
// (measured with per-phase clock stamps, cfg-2 dimensions, position 199: the six GEMVs of a layer 18 us = 25-30 B/cycle/CU,
// the two attentions 8 + 6 us (four dependent K / V round trips each), three LayerNorms 3 us, logits 6 us; 151 us in all).
// Each Dense is a GEMV: thread = 4 output columns x one slice of the contraction, partial sums folded through LDS in a
// fixed order (deterministic).
// The last workgroup to finish a position (ticket counter) does the cross-sample part: "all n_valid samples have emitted
// an EOS" -> done_step, and advances the device-side step index, so a position is ONE launch (captured once, replayed).
#include "skf_common.h"
#include "skf_decode_fused.h"

namespace {

constexpr int NT = 512;   // threads per workgroup (8 waves)
#ifndef SKF_DEC_UNROLL
#define SKF_DEC_UNROLL 16
#endif
constexpr int UN = SKF_DEC_UNROLL;   // weight rows requested per thread before the first is used

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
  v = wave_sum(v);
  __syncthreads();                       // red is reused from the previous reduction
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) t += red[w];
  return t;
}

// y[n] = act(bias[n] + sum_k x[k] W[k * ldw + n]), n < N.  x, y, part in LDS (y != x).  Ends with a barrier.
template <int VEC>
__device__ __forceinline__ void gemv(const float* __restrict__ W, int ldw, const float* __restrict__ bias, const float* x, float* y, int N,
                     int K, int act, float* part, int tid) {
  const int ncg = (N + VEC - 1) / VEC;
  int ks = 1;
  while (ks * 2 * ncg <= NT && K / (ks * 2) >= UN) ks *= 2;
  const int gp = ncg < NT / ks ? ncg : NT / ks;        // column groups per pass
  const int kch = (K + ks - 1) / ks;
  const int ksl = tid / gp, cgi = tid - ksl * gp;
  for (int base = 0; base < ncg; base += gp) {
    const int cg = base + cgi;
    if (ksl < ks && cg < ncg) {
      const int k0 = ksl * kch, k1 = k0 + kch < K ? k0 + kch : K;
      float acc[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
      const float* wp = W + (size_t)k0 * ldw + cg * VEC;
      int k = k0;
      for (; k + UN <= k1; k += UN) {
        if constexpr (VEC == 4) {
          f32x4 w[UN];
#pragma unroll
          for (int u = 0; u < UN; ++u) w[u] = *reinterpret_cast<const f32x4*>(wp + (size_t)u * ldw);
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const float xv = x[k + u];
            acc[0] += xv * w[u][0]; acc[1] += xv * w[u][1]; acc[2] += xv * w[u][2]; acc[3] += xv * w[u][3];
          }
        } else {
          float w[UN];
#pragma unroll
          for (int u = 0; u < UN; ++u) w[u] = wp[(size_t)u * ldw];
#pragma unroll
          for (int u = 0; u < UN; ++u) acc[0] += x[k + u] * w[u];
        }
        wp += (size_t)UN * ldw;
      }
      for (; k < k1; ++k) {
        const float xv = x[k];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += xv * wp[e];
        wp += ldw;
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) part[(ksl * gp + cgi) * VEC + e] = acc[e];
    }
    __syncthreads();
    for (int idx = tid; idx < gp * VEC; idx += NT) {
      const int n = base * VEC + idx;
      if (n < N) {
        float s = bias ? bias[n] : 0.f;
        for (int q = 0; q < ks; ++q) s += part[q * gp * VEC + idx];
        if (act == 1) s = fmaxf(s, 0.f);
        y[n] = s;
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void dense(const SkfDecDense& w, const float* x, float* y, int act, float* part, int tid) {
  if (w.vec4) gemv<4>(w.w, w.ld, w.b, x, y, w.out, w.in, act, part, tid);
  else gemv<1>(w.w, w.ld, w.b, x, y, w.out, w.in, act, part, tid);
}

// out = LayerNorm(a + r) (eps 1e-6, biased variance; builders/layers/transformer.py:217-222).  d <= NT.  Ends with a barrier.
__device__ __forceinline__ void add_ln(const float* a, const float* r, const float* __restrict__ g, const float* __restrict__ bt, float* out, int d,
                       float* red, int tid) {
  const float z = tid < d ? a[tid] + r[tid] : 0.f;
  const float mean = block_sum(z, red, tid) / (float)d;
  const float c = tid < d ? z - mean : 0.f;
  const float rstd = rsqrtf(block_sum(c * c, red, tid) / (float)d + 1e-6f);
  if (tid < d) out[tid] = c * rstd * g[tid] + bt[tid];
  __syncthreads();
}

// scaled_dot_product_attention (builders/utils.py:71-105) for one query row over the rows j < Lk of a (Lk, ld_kv) K / V image
// of the sample.  A thread owns one 16-byte column group c4 of the d columns and every (NT / (d/4))-th key, so a wave
// instruction reads whole contiguous rows (a lane per key would touch 64 cache lines per instruction: measured 10 of the
// 11 us of this function).  Three phases through LDS: scores[h][j] (dot product folded over the DH/4 adjacent lanes of a
// head), softmax per head (wave = head), weighted sum of V (partials per key slice, folded in a fixed order).
template <int DH>
__device__ __forceinline__ void attend(const float* q_lds, const float* __restrict__ Kb, const float* __restrict__ Vb, int ld_kv, int Lk,
                       const unsigned char* __restrict__ mask, int limit, int d, int H, float* sc, int LkP, float* part,
                       float* o_lds, int tid) {
  const int nc4 = d >> 2, jl = tid / nc4, c4 = tid - jl * nc4, JP = NT / nc4;
  const int h = (4 * c4) / DH;
  const f32x4 q4 = *reinterpret_cast<const f32x4*>(q_lds + 4 * c4);
  const float scale_div = sqrtf((float)DH);
  // All rows of a thread are requested before the first is used (a row-at-a-time loop costs one L2 / HBM round trip per
  // row), V together with K: up to NR rows of each stay in registers; longer key ranges go round the loop again.
  constexpr int NR = 8;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const bool one_pass = Lk <= NR * JP;
  f32x4 kv[NR], vv[NR];
#define SKF_LOAD_ROWS(base, r, j0)                                                                              \
  _Pragma("unroll") for (int u = 0; u < NR; ++u) {                                                              \
    const int j = (j0) + u * JP;                                                                                \
    r[u] = *reinterpret_cast<const f32x4*>((base) + (size_t)(j < Lk ? j : Lk - 1) * ld_kv + 4 * c4);           \
  }
#define SKF_SCORES(j0)                                                                                          \
  _Pragma("unroll") for (int u = 0; u < NR; ++u) {                                                              \
    const int j = (j0) + u * JP;                                                                                \
    float dot = q4[0] * kv[u][0] + q4[1] * kv[u][1] + q4[2] * kv[u][2] + q4[3] * kv[u][3];                      \
    _Pragma("unroll") for (int o = 1; o < DH / 4; o <<= 1) dot += __shfl_xor(dot, o, 64);                       \
    if (j < Lk && (c4 & (DH / 4 - 1)) == 0) {                                                                   \
      const bool masked = (mask && mask[j]) || j >= limit;                                                      \
      sc[h * LkP + j] = dot / scale_div + (masked ? -1e9f : 0.f);                                               \
    }                                                                                                           \
  }
#define SKF_WEIGHTED(j0)                                                                                        \
  _Pragma("unroll") for (int u = 0; u < NR; ++u) {                                                              \
    const int j = (j0) + u * JP;                                                                                \
    const float pj = j < Lk ? sc[h * LkP + j] : 0.f;                                                            \
    acc[0] += pj * vv[u][0]; acc[1] += pj * vv[u][1]; acc[2] += pj * vv[u][2]; acc[3] += pj * vv[u][3];         \
  }
  SKF_LOAD_ROWS(Kb, kv, jl)
  if (one_pass) { SKF_LOAD_ROWS(Vb, vv, jl) }
  SKF_SCORES(jl)
  for (int j0 = jl + NR * JP; j0 < Lk; j0 += NR * JP) { SKF_LOAD_ROWS(Kb, kv, j0) SKF_SCORES(j0) }
  __syncthreads();
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int hh = wave; hh < H; hh += NT / 64) {
      float* row = sc + hh * LkP;
      float mx = -INFINITY;
      for (int j = lane; j < Lk; j += 64) mx = fmaxf(mx, row[j]);
      mx = wave_max(mx);
      float se = 0.f;
      for (int j = lane; j < Lk; j += 64) { const float e = __expf(row[j] - mx); row[j] = e; se += e; }
      se = wave_sum(se);
      const float rinv = 1.0f / se;
      for (int j = lane; j < Lk; j += 64) row[j] *= rinv;
    }
  }
  __syncthreads();
  if (one_pass) { SKF_WEIGHTED(jl) }
  else for (int j0 = jl; j0 < Lk; j0 += NR * JP) { SKF_LOAD_ROWS(Vb, vv, j0) SKF_WEIGHTED(j0) }
#undef SKF_LOAD_ROWS
#undef SKF_SCORES
#undef SKF_WEIGHTED
  *reinterpret_cast<f32x4*>(part + jl * d + 4 * c4) = acc;
  __syncthreads();
  for (int c = tid; c < d; c += NT) {
    float t = 0.f;
    for (int q = 0; q < JP; ++q) t += part[q * d + c];
    o_lds[c] = t;
  }
  __syncthreads();
}

template <int DH>
__global__ __launch_bounds__(NT) void decode_position_kernel(SkfDecodeFused p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int d = p.d;
  float* xs = lds;                 // [d]  layer input
  float* o1 = xs + d;              // [d]  after LN1
  float* o2 = o1 + d;              // [d]  after LN2
  float* ys = o2 + d;              // [d]  Dense output awaiting its residual
  float* os = ys + d;              // [d]  attention output
  float* qkv = os + d;             // [3d]
  float* hs = qkv + 3 * d;         // [max(F, Vout)] FFN hidden / logits
  float* part = hs + p.hs_len;     // [4 NT] GEMV partial sums
  float* red = part + 4 * NT;      // [16]
  float* sc = red + 16;            // [H][LkP] attention scores / probabilities
  const int LkP = p.Le + 1;
  const int step = *p.step_dev;
  const int n_valid = (int)p.dyn[0];
  const long long eos = p.dyn[1];

  // decoder input of the position (builders/layers/transformer.py:325-334, dropout off)
  if (tid < d) {
    float v;
    if (p.tokens) {
      long long tk = p.tokens[(size_t)b * p.Ti + step];
      if (tk < 0 || tk >= p.vocab) tk = 0;
      v = p.emb_table[(size_t)tk * d + tid];
    } else {
      const float* x = p.cont + ((size_t)b * p.Ti + step) * 5;
      const float* W = p.embd_w;
      v = x[0] * W[tid] + x[1] * W[d + tid] + x[2] * W[2 * d + tid] + x[3] * W[3 * d + tid] + x[4] * W[4 * d + tid] + p.embd_b[tid];
    }
    xs[tid] = v * sqrtf((float)d) + p.pos[(size_t)step * d + tid];
  }
  __syncthreads();

  int limit = 0x7fffffff;          // cross attention: keys >= limit are masked (models/sketchformer.py:172,279-283)
  if (!p.blind) { limit = p.limit ? p.limit[b] : -1; if (limit < 0) limit = step + 1; }
  const unsigned char* smask = p.selfmask + (size_t)b * p.mask_ld;

  for (int l = 0; l < p.N; ++l) {
    const SkfDecLayer& w = p.layer[l];
    dense(w.qkv, xs, qkv, 0, part, tid);
    float* cache = w.cache + (size_t)b * p.Le * 2 * d;             // (Le, 2d): K | V of the positions so far
    for (int c = tid; c < 2 * d; c += NT) cache[(size_t)step * 2 * d + c] = qkv[d + c];
    __syncthreads();               // the appended row is read back below: workgroup-scope visibility of the global stores
    attend<DH>(qkv, cache, cache + d, 2 * d, step + 1, smask, 0x7fffffff, d, p.H, sc, LkP, part, os, tid);
    dense(w.o, os, ys, 0, part, tid);
    add_ln(xs, ys, w.ln1_g, w.ln1_b, o1, d, red, tid);
    dense(w.q2, o1, qkv, 0, part, tid);
    const float* kv2 = w.kv2 + (size_t)b * p.Le * 2 * d;
    attend<DH>(qkv, kv2, kv2 + d, 2 * d, p.Le, nullptr, limit, d, p.H, sc, LkP, part, os, tid);
    dense(w.o2, os, ys, 0, part, tid);
    add_ln(o1, ys, w.ln2_g, w.ln2_b, o2, d, red, tid);
    dense(w.f1, o2, hs, 1, part, tid);
    dense(w.f2, hs, ys, 0, part, tid);
    add_ln(o2, ys, w.ln3_g, w.ln3_b, xs, d, red, tid);
  }
  dense(p.out, xs, hs, 0, part, tid);          // logits of the position (F >= Vout is not assumed: hs holds max(F, Vout))

  __shared__ float s_mx[NT / 64];
  __shared__ int s_am[NT / 64];
  if (p.tokens) {
    // models/sketchformer.py:285-301: next = argmax (first index on ties, tf.argmax); PAD masks the key in later steps;
    // EOS flags are sticky
    float mx = -INFINITY; int am = 0x7fffffff;
    for (int j = tid; j < p.Vout; j += NT) {
      const float v = hs[j];
      if (v > mx) { mx = v; am = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o, 64); const int oa = __shfl_xor(am, o, 64);
      if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    if ((tid & 63) == 0) { s_mx[tid >> 6] = mx; s_am[tid >> 6] = am; }
    __syncthreads();
    if (tid == 0) {
      for (int wv = 1; wv < NT / 64; ++wv)
        if (s_mx[wv] > mx || (s_mx[wv] == mx && s_am[wv] < am)) { mx = s_mx[wv]; am = s_am[wv]; }
      p.tokens[(size_t)b * p.Ti + step + 1] = am;
      p.selfmask[(size_t)b * p.mask_ld + step + 1] = am == 0 ? 1 : 0;
      if ((long long)am == eos) p.eos_seen[b] = 1;
    }
  } else if (tid == 0) {
    // continuous: appended row = (x, y, softmax(pen logits)); mask byte = (row[4] == 1); "finished" = argmax(pen) == 2 (not sticky)
    const float* x = hs;
    const float m = fmaxf(x[2], fmaxf(x[3], x[4]));
    const float e0 = __expf(x[2] - m), e1 = __expf(x[3] - m), e2 = __expf(x[4] - m);
    const float r = 1.0f / (e0 + e1 + e2);
    float* o = p.cont + ((size_t)b * p.Ti + step + 1) * 5;
    const float q0 = e0 * r, q1 = e1 * r, q2 = e2 * r;
    o[0] = x[0]; o[1] = x[1]; o[2] = q0; o[3] = q1; o[4] = q2;
    p.selfmask[(size_t)b * p.mask_ld + step + 1] = (q2 == 1.0f) ? 1 : 0;
    const int am = (q0 >= q1 && q0 >= q2) ? 0 : (q1 >= q2 ? 1 : 2);
    p.eos_seen[b] = am == 2 ? 1 : 0;
  }
  // ---- cross-sample part, by whichever workgroup finishes the position last
  if (tid == 0) {
    __threadfence();
    const int t = atomicAdd(p.ticket, 1);
    if (t == p.B - 1) {
      __threadfence();
      int seen = 0;
      for (int i = 0; i < n_valid; ++i) seen += __hip_atomic_load(p.eos_seen + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen >= n_valid && *p.done_step < 0) *p.done_step = step;
      *p.step_dev = step + 1;
      *p.ticket = 0;
    }
  }
}

}  // namespace

size_t skf_decode_fused_lds_bytes(const SkfDecodeFused& p) {
  return (size_t)(8 * p.d + p.hs_len + 4 * NT + 16 + p.H * (p.Le + 1)) * sizeof(float);
}

bool skf_decode_fused_supported(int d, int H, int F, int Le, int N, int Vout) {
  const int dh = H > 0 ? d / H : 0;
  return d <= NT && (d & 3) == 0 && NT % (d / 4) == 0 && (dh == 16 || dh == 32 || dh == 64) && N <= SKF_DEC_MAX_LAYERS && F > 0 && Vout > 0 &&
         (size_t)(8 * d + (F > Vout ? F : Vout) + 4 * NT + 16 + H * (Le + 1)) * sizeof(float) <= 159 * 1024;
}

int skf_decode_fused_launch(const SkfDecodeFused& p, hipStream_t st) {
  const int dh = p.d / p.H;
  const size_t smem = skf_decode_fused_lds_bytes(p);
  SkfProfScope ps(st, "decode_position", 0.0, 0.0);
#define SKF_DF(DHV)                                                                                                  \
  {                                                                                                                  \
    SKF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_position_kernel<DHV>),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)); /* per launch: per-device attribute */ \
    hipLaunchKernelGGL((decode_position_kernel<DHV>), dim3(p.B), dim3(NT), smem, st, p);                             \
  }
  if (dh == 16) SKF_DF(16) else if (dh == 32) SKF_DF(32) else SKF_DF(64)
#undef SKF_DF
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
