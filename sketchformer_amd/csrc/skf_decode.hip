// Kernels of the KV-cached greedy reconstruction (models/sketchformer.py:255-311 predict_from_embedding).
// The reference re-runs the whole decoder on the growing prefix for every emitted token; here one step only
// processes the newest position: its K/V rows are appended to a per-layer cache by the projection GEMM itself,
// and attention is one query row per (sample, head).
#include "skf_common.h"
#include "../../include/skf.h"

namespace {

struct AttnDecodeParams {
  const float* Q; int ldq;
  const float* K; const float* V; int ld_kv; long long kv_bs;   // row stride, per-sample stride (floats)
  const unsigned char* key_mask; int key_mask_ld;               // (B, key_mask_ld) 1 = masked key, or null
  const int* key_limit; int key_limit_all;                      // keys >= limit are masked (per sample / all samples; 0 = none)
  int B, H, Lk;
  float* O; int ldo;
  // graph-replayable decode: the step index lives in device memory
  const int* step_dev;            // non-null: Lk = *step_dev + 1 (the Lk field is the cache capacity)
  const float* K_new; const float* V_new; int ld_new;   // non-null: row Lk-1 comes from here and is appended to the cache
  float* K_cache; float* V_cache;                       // writable aliases of K / V for the append
  int limit_from_step;            // cross attention without per-sample limits: keys >= step + 1 are masked
};

// One wave per (sample, head): lane j owns keys j, j+64, ... (Lk <= 64*MAXJ).
// scaled_dot_product_attention (builders/utils.py:71-105) for a single query row: logits = q.k / sqrt(dk),
// += mask * -1e9, softmax over the keys, weighted sum of V.
template <int DH, int MAXJ>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnDecodeParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.x * 4 + wave;
  if (bh >= p.B * p.H) return;
  const int b = bh / p.H, h = bh % p.H;
  const int step = p.step_dev ? *p.step_dev : 0;
  const int Lk = p.K_new ? step + 1 : p.Lk;            // self attention over the cache grows with the step
  float q[DH];
#pragma unroll
  for (int c4 = 0; c4 < DH / 4; ++c4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p.Q + (size_t)b * p.ldq + h * DH + 4 * c4);
    q[4 * c4] = v[0]; q[4 * c4 + 1] = v[1]; q[4 * c4 + 2] = v[2]; q[4 * c4 + 3] = v[3];
  }
  const float* Kb = p.K + (size_t)b * p.kv_bs + h * DH;
  const float* Vb = p.V + (size_t)b * p.kv_bs + h * DH;
  int limit = p.key_limit ? p.key_limit[b] : (p.key_limit_all > 0 ? p.key_limit_all : 0x7fffffff);
  if (p.limit_from_step && (!p.key_limit || limit < 0)) limit = step + 1;   // make_dummy_input(nattn = i + 1)
  // the newest key / value row: taken from the projection output (no read-after-write through the cache) and appended
  const int jn = p.K_new ? Lk - 1 : -1;
  if (p.K_new && lane < DH / 4) {
    const f32x4 kn = *reinterpret_cast<const f32x4*>(p.K_new + (size_t)b * p.ld_new + h * DH + 4 * lane);
    const f32x4 vn = *reinterpret_cast<const f32x4*>(p.V_new + (size_t)b * p.ld_new + h * DH + 4 * lane);
    *reinterpret_cast<f32x4*>(p.K_cache + (size_t)b * p.kv_bs + (size_t)jn * p.ld_kv + h * DH + 4 * lane) = kn;
    *reinterpret_cast<f32x4*>(p.V_cache + (size_t)b * p.kv_bs + (size_t)jn * p.ld_kv + h * DH + 4 * lane) = vn;
  }
  const float* Kn = p.K_new ? p.K_new + (size_t)b * p.ld_new + h * DH : nullptr;
  const float* Vn = p.V_new ? p.V_new + (size_t)b * p.ld_new + h * DH : nullptr;
  const float scale_div = sqrtf((float)DH);
  float s[MAXJ];
  float mx = -INFINITY;
#pragma unroll
  for (int jj = 0; jj < MAXJ; ++jj) {
    const int j = lane + 64 * jj;
    float v = -INFINITY;
    if (j < Lk) {
      const float* kr = j == jn ? Kn : Kb + (size_t)j * p.ld_kv;
      float dot = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < DH / 4; ++c4) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + 4 * c4);
        dot += q[4 * c4] * kv[0] + q[4 * c4 + 1] * kv[1] + q[4 * c4 + 2] * kv[2] + q[4 * c4 + 3] * kv[3];
      }
      const bool masked = (p.key_mask && p.key_mask[(size_t)b * p.key_mask_ld + j]) || j >= limit;
      v = dot / scale_div + (masked ? -1e9f : 0.f);
    }
    s[jj] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  float se = 0.f;
#pragma unroll
  for (int jj = 0; jj < MAXJ; ++jj) { s[jj] = __expf(s[jj] - mx); se += s[jj]; }   // exp(-inf) = 0 past Lk
  se = wave_sum(se);
  const float rinv = 1.0f / se;
  float acc[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) acc[c] = 0.f;
#pragma unroll
  for (int jj = 0; jj < MAXJ; ++jj) {
    const int j = lane + 64 * jj;
    if (j < Lk) {
      const float pj = s[jj] * rinv;
      const float* vr = j == jn ? Vn : Vb + (size_t)j * p.ld_kv;
#pragma unroll
      for (int c4 = 0; c4 < DH / 4; ++c4) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + 4 * c4);
        acc[4 * c4] += pj * vv[0]; acc[4 * c4 + 1] += pj * vv[1]; acc[4 * c4 + 2] += pj * vv[2]; acc[4 * c4 + 3] += pj * vv[3];
      }
    }
  }
  float mine = 0.f;
#pragma unroll
  for (int c = 0; c < DH; ++c) {
    const float t = wave_sum(acc[c]);
    if (lane == c) mine = t;
  }
  if (lane < DH) p.O[(size_t)b * p.ldo + h * DH + lane] = mine;
}

// The same for any head size % 4 == 0 up to 128 (skf_generic.hip serves such models; every BASELINE config has 16 / 32 / 64):
// query and probabilities in wave-private LDS, lanes over the keys for the scores and over the head columns for the output.
__global__ __launch_bounds__(256) void attn_decode_any_kernel(AttnDecodeParams p, int DH) {
  __shared__ float qs[4][128];
  __shared__ float ps[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh_ = blockIdx.x * 4 + wave;
  const bool active = bh_ < p.B * p.H;
  const int bh = active ? bh_ : 0;
  const int b = bh / p.H, h = bh % p.H;
  const int step = p.step_dev ? *p.step_dev : 0;
  const int Lk = p.K_new ? step + 1 : p.Lk;
  for (int c = lane; c < DH; c += 64) qs[wave][c] = p.Q[(size_t)b * p.ldq + h * DH + c];
  const float* Kb = p.K + (size_t)b * p.kv_bs + h * DH;
  const float* Vb = p.V + (size_t)b * p.kv_bs + h * DH;
  int limit = p.key_limit ? p.key_limit[b] : (p.key_limit_all > 0 ? p.key_limit_all : 0x7fffffff);
  if (p.limit_from_step && (!p.key_limit || limit < 0)) limit = step + 1;
  const int jn = p.K_new ? Lk - 1 : -1;
  const float* Kn = p.K_new ? p.K_new + (size_t)b * p.ld_new + h * DH : nullptr;
  const float* Vn = p.V_new ? p.V_new + (size_t)b * p.ld_new + h * DH : nullptr;
  if (active && p.K_new)
    for (int c = lane; c < DH; c += 64) {
      p.K_cache[(size_t)b * p.kv_bs + (size_t)jn * p.ld_kv + h * DH + c] = Kn[c];
      p.V_cache[(size_t)b * p.kv_bs + (size_t)jn * p.ld_kv + h * DH + c] = Vn[c];
    }
  __syncthreads();
  const float scale_div = sqrtf((float)DH);
  float mx = -INFINITY;
  for (int j = lane; j < Lk; j += 64) {
    const float* kr = j == jn ? Kn : Kb + (size_t)j * p.ld_kv;
    float dot = 0.f;
    for (int c = 0; c < DH; ++c) dot += qs[wave][c] * kr[c];
    const bool masked = (p.key_mask && p.key_mask[(size_t)b * p.key_mask_ld + j]) || j >= limit;
    const float v = dot / scale_div + (masked ? -1e9f : 0.f);
    ps[wave][j] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  float se = 0.f;
  for (int j = lane; j < Lk; j += 64) { const float e = __expf(ps[wave][j] - mx); ps[wave][j] = e; se += e; }
  se = wave_sum(se);
  const float rinv = 1.0f / se;
  __syncthreads();
  for (int c = lane; c < DH; c += 64) {
    float acc = 0.f;
    for (int j = 0; j < Lk; ++j) acc += ps[wave][j] * (j == jn ? Vn[c] : Vb[(size_t)j * p.ld_kv + c]);
    if (active) p.O[(size_t)b * p.ldo + h * DH + c] = acc * rinv;
  }
}

// Token mode, after step `step` produced the logits of position `step`:
//   next = argmax (first index on ties, tf.argmax) -> tokens[b][step+1]; self-mask byte = (next == PAD);
//   EOS flags are sticky; done_step = first step after which every one of the n_valid samples has emitted an EOS.
// One 1024-thread workgroup (16 waves, each walking samples).
__global__ __launch_bounds__(1024) void decode_select_tokens_kernel(const float* __restrict__ logits, int ld, int B, int V,
                                                                     int n_valid, int step, long long eos,
                                                                     long long* __restrict__ tokens, int tok_ld,
                                                                     unsigned char* __restrict__ selfmask, int mask_ld,
                                                                     int* __restrict__ eos_seen, int* __restrict__ done_step,
                                                                     int* __restrict__ step_dev, const long long* __restrict__ dyn) {
  __shared__ int cnt[16];
  if (step_dev) { step = *step_dev; n_valid = (int)dyn[0]; eos = dyn[1]; }   // graph replay: per-call values in device memory
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int seen = 0;
  for (int b = wave; b < B; b += 16) {
    const float* x = logits + (size_t)b * ld;
    float mx = -INFINITY; int am = 0x7fffffff;
    for (int j = lane; j < V; j += 64) {
      const float v = x[j];
      if (v > mx) { mx = v; am = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o, 64); const int oa = __shfl_xor(am, o, 64);
      if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    if (lane == 0) {
      tokens[(size_t)b * tok_ld + step + 1] = am;
      selfmask[(size_t)b * mask_ld + step + 1] = am == 0 ? 1 : 0;
      int e = eos_seen[b];
      if ((long long)am == eos) { e = 1; eos_seen[b] = 1; }
      if (b < n_valid) seen += e;
    }
  }
  if (lane == 0) cnt[wave] = seen;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int k = 0; k < 16; ++k) t += cnt[k];
    if (t >= n_valid && *done_step < 0) *done_step = step;
    if (step_dev) *step_dev = step + 1;
  }
}

// Continuous mode: appended row = (x, y, softmax(pen logits)); self-mask byte = (row[4] == 1);
// done_step = first step in which argmax(pen) == 2 for all n_valid samples at once (not sticky).
__global__ __launch_bounds__(256) void decode_select_continuous_kernel(const float* __restrict__ pred, int ld, int B,
                                                                        int n_valid, int step, float* __restrict__ out,
                                                                        int out_ld_rows, unsigned char* __restrict__ selfmask,
                                                                        int mask_ld, int* __restrict__ done_step,
                                                                        int* __restrict__ step_dev, const long long* __restrict__ dyn) {
  __shared__ int cnt[256];
  if (step_dev) { step = *step_dev; n_valid = (int)dyn[0]; }
  int fin = 0;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* x = pred + (size_t)b * ld;
    const float m = fmaxf(x[2], fmaxf(x[3], x[4]));
    const float e0 = __expf(x[2] - m), e1 = __expf(x[3] - m), e2 = __expf(x[4] - m);
    const float r = 1.0f / (e0 + e1 + e2);
    float* o = out + ((size_t)b * out_ld_rows + step + 1) * 5;
    o[0] = x[0]; o[1] = x[1]; o[2] = e0 * r; o[3] = e1 * r; o[4] = e2 * r;
    selfmask[(size_t)b * mask_ld + step + 1] = (e2 * r == 1.0f) ? 1 : 0;
    const int am = (o[2] >= o[3] && o[2] >= o[4]) ? 0 : (o[3] >= o[4] ? 1 : 2);
    if (b < n_valid && am == 2) fin += 1;
  }
  cnt[threadIdx.x] = fin;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int k = 0; k < 256; ++k) t += cnt[k];
    if (t >= n_valid && *done_step < 0) *done_step = step;
    if (step_dev) *step_dev = step + 1;
  }
}

// Decoder input of the current step (builders/layers/transformer.py:325-334, dropout off): embedding of the newest
// token (or Dense(5->d) of the newest stroke-5 row) * sqrt(d) + pos[step]; the step index is read from device memory.
__global__ __launch_bounds__(256) void decode_embed_kernel(const long long* __restrict__ tokens, const float* __restrict__ cont,
                                                           int ld, int B, const float* __restrict__ table, int vocab,
                                                           const float* __restrict__ W, const float* __restrict__ bias, int d,
                                                           const float* __restrict__ pos, const int* __restrict__ step_dev,
                                                           float* __restrict__ out) {
  const int step = *step_dev;
  const float sq = sqrtf((float)d);
  const float* pe = pos + (size_t)step * d;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < B * d; e += gridDim.x * 256) {
    const int b = e / d, c = e % d;
    float v;
    if (tokens) {
      long long tk = tokens[(size_t)b * ld + step];
      if (tk < 0 || tk >= vocab) tk = 0;
      v = table[(size_t)tk * d + c];
    } else {
      const float* x = cont + ((size_t)b * ld + step) * 5;
      v = x[0] * W[c] + x[1] * W[d + c] + x[2] * W[2 * d + c] + x[3] * W[3 * d + c] + x[4] * W[4 * d + c] + bias[c];
    }
    out[e] = v * sq + pe[c];
  }
}

__global__ void decode_init_kernel(long long* tokens, int tok_ld, float* cont, int cont_ld_rows, unsigned char* selfmask,
                                   int mask_ld, int* eos_seen, int* done_step, int B, long long sos, int* step_dev) {
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    if (tokens) { tokens[(size_t)b * tok_ld] = sos; selfmask[(size_t)b * mask_ld] = sos == 0 ? 1 : 0; }
    if (cont) {
      float* o = cont + (size_t)b * cont_ld_rows * 5;
      o[0] = 0.f; o[1] = 0.f; o[2] = 1.f; o[3] = 0.f; o[4] = 0.f;     // models/sketchformer.py:268
      selfmask[(size_t)b * mask_ld] = 0;
    }
    eos_seen[b] = 0;
  }
  if (threadIdx.x == 0) { *done_step = -1; if (step_dev) *step_dev = 0; }
}

}  // namespace

extern "C" int skf_attention_decode(const float* Q, int ldq, const float* K, const float* V, int ld_kv,
                                    long long kv_batch_stride, const unsigned char* key_mask, int key_mask_ld,
                                    const int* key_limit, int key_limit_all, int B, int H, int Lk, int dh, float* O,
                                    int ldo, const int* step_dev, const float* K_new, const float* V_new, int ld_new,
                                    int limit_from_step, skf_stream_t stream) {
  SKF_CHECK_ARG(Q && K && V && O, "null operand");
  SKF_CHECK_ARG(B > 0 && H > 0 && Lk > 0 && Lk <= 512, "need 0 < Lk <= 512");
  SKF_CHECK_ARG(dh > 0 && dh <= 128 && (dh & 3) == 0, "head size must be a multiple of 4, at most 128");
  SKF_CHECK_ARG((ldq & 3) == 0 && (ld_kv & 3) == 0 && (kv_batch_stride & 3) == 0 &&
                (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) == 0, "Q/K/V must allow 16-byte row loads");
  SKF_CHECK_ARG((K_new == nullptr) == (V_new == nullptr), "K_new and V_new go together");
  SKF_CHECK_ARG(!K_new || (step_dev && (ld_new & 3) == 0 && (((uintptr_t)K_new | (uintptr_t)V_new) & 15) == 0),
                "appending needs the device step index and 16-byte aligned new rows");
  SKF_CHECK_ARG(!limit_from_step || step_dev, "limit_from_step needs the device step index");
  AttnDecodeParams p{Q, ldq, K, V, ld_kv, kv_batch_stride, key_mask, key_mask_ld, key_limit, key_limit_all, B, H, Lk, O, ldo,
                     step_dev, K_new, V_new, ld_new, const_cast<float*>(K), const_cast<float*>(V), limit_from_step};
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(skf_cdiv(B * H, 4)), block(256);
  SkfProfScope ps(st, "attn_decode", 4.0 * B * H * (double)Lk * dh, 8.0 * B * H * (double)Lk * dh);
#define SKF_AD(DHV)                                                                         \
  { if (Lk <= 256) hipLaunchKernelGGL((attn_decode_kernel<DHV, 4>), grid, block, 0, st, p);   \
    else hipLaunchKernelGGL((attn_decode_kernel<DHV, 8>), grid, block, 0, st, p); }
  if (dh == 16) SKF_AD(16) else if (dh == 32) SKF_AD(32) else if (dh == 64) SKF_AD(64)
  else hipLaunchKernelGGL(attn_decode_any_kernel, grid, block, 0, st, p, dh);
#undef SKF_AD
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_decode_init(long long* tokens, int tok_ld, float* cont, int cont_ld_rows, unsigned char* selfmask,
                               int mask_ld, int* eos_seen, int* done_step, int B, long long sos, int* step_dev,
                               skf_stream_t stream) {
  SKF_CHECK_ARG((tokens || cont) && selfmask && eos_seen && done_step && B > 0, "bad argument");
  hipLaunchKernelGGL(decode_init_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tokens, tok_ld, cont, cont_ld_rows,
                     selfmask, mask_ld, eos_seen, done_step, B, sos, step_dev);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_decode_embed(const long long* tokens, const float* cont, int ld, int B, const float* table, int vocab,
                                const float* W, const float* bias, int d, const float* pos, const int* step_dev, float* out,
                                skf_stream_t stream) {
  SKF_CHECK_ARG((tokens != nullptr) != (cont != nullptr), "exactly one of tokens / cont");
  SKF_CHECK_ARG((tokens ? table != nullptr : (W && bias)) && pos && step_dev && out && B > 0 && d > 0, "null operand");
  hipLaunchKernelGGL(decode_embed_kernel, dim3(skf_cdiv(B * d, 256)), dim3(256), 0, (hipStream_t)stream, tokens, cont, ld, B,
                     table, vocab, W, bias, d, pos, step_dev, out);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_decode_select_tokens(const float* logits, int ld, int B, int V, int n_valid, int step, long long eos,
                                        long long* tokens, int tok_ld, unsigned char* selfmask, int mask_ld,
                                        int* eos_seen, int* done_step, int* step_dev, const long long* dyn,
                                        skf_stream_t stream) {
  SKF_CHECK_ARG(logits && tokens && selfmask && eos_seen && done_step, "null operand");
  SKF_CHECK_ARG((step_dev == nullptr) == (dyn == nullptr), "step_dev and dyn go together");
  SKF_CHECK_ARG(B > 0 && V > 0 && (step_dev || (n_valid > 0 && n_valid <= B && step >= 0 && step + 1 < tok_ld && step + 1 < mask_ld)), "bad shape");
  hipLaunchKernelGGL(decode_select_tokens_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, ld, B, V, n_valid,
                     step, eos, tokens, tok_ld, selfmask, mask_ld, eos_seen, done_step, step_dev, dyn);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_decode_select_continuous(const float* pred, int ld, int B, int n_valid, int step, float* out,
                                            int out_ld_rows, unsigned char* selfmask, int mask_ld, int* done_step,
                                            int* step_dev, const long long* dyn, skf_stream_t stream) {
  SKF_CHECK_ARG(pred && out && selfmask && done_step, "null operand");
  SKF_CHECK_ARG((step_dev == nullptr) == (dyn == nullptr), "step_dev and dyn go together");
  SKF_CHECK_ARG(B > 0 && (step_dev || (n_valid > 0 && n_valid <= B && step >= 0 && step + 1 < out_ld_rows && step + 1 < mask_ld)), "bad shape");
  hipLaunchKernelGGL(decode_select_continuous_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, ld, B, n_valid,
                     step, out, out_ld_rows, selfmask, mask_ld, done_step, step_dev, dyn);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
