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
  float q[DH];
#pragma unroll
  for (int c4 = 0; c4 < DH / 4; ++c4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p.Q + (size_t)b * p.ldq + h * DH + 4 * c4);
    q[4 * c4] = v[0]; q[4 * c4 + 1] = v[1]; q[4 * c4 + 2] = v[2]; q[4 * c4 + 3] = v[3];
  }
  const float* Kb = p.K + (size_t)b * p.kv_bs + h * DH;
  const float* Vb = p.V + (size_t)b * p.kv_bs + h * DH;
  const int limit = p.key_limit ? p.key_limit[b] : (p.key_limit_all > 0 ? p.key_limit_all : 0x7fffffff);
  const float scale_div = sqrtf((float)DH);
  float s[MAXJ];
  float mx = -INFINITY;
#pragma unroll
  for (int jj = 0; jj < MAXJ; ++jj) {
    const int j = lane + 64 * jj;
    float v = -INFINITY;
    if (j < p.Lk) {
      const float* kr = Kb + (size_t)j * p.ld_kv;
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
    if (j < p.Lk) {
      const float pj = s[jj] * rinv;
      const float* vr = Vb + (size_t)j * p.ld_kv;
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

// Token mode, after step `step` produced the logits of position `step`:
//   next = argmax (first index on ties, tf.argmax) -> tokens[b][step+1]; self-mask byte = (next == PAD);
//   EOS flags are sticky; done_step = first step after which every one of the n_valid samples has emitted an EOS.
// One 1024-thread workgroup (16 waves, each walking samples).
__global__ __launch_bounds__(1024) void decode_select_tokens_kernel(const float* __restrict__ logits, int ld, int B, int V,
                                                                     int n_valid, int step, long long eos,
                                                                     long long* __restrict__ tokens, int tok_ld,
                                                                     unsigned char* __restrict__ selfmask, int mask_ld,
                                                                     int* __restrict__ eos_seen, int* __restrict__ done_step) {
  __shared__ int cnt[16];
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
  }
}

// Continuous mode: appended row = (x, y, softmax(pen logits)); self-mask byte = (row[4] == 1);
// done_step = first step in which argmax(pen) == 2 for all n_valid samples at once (not sticky).
__global__ __launch_bounds__(256) void decode_select_continuous_kernel(const float* __restrict__ pred, int ld, int B,
                                                                        int n_valid, int step, float* __restrict__ out,
                                                                        int out_ld_rows, unsigned char* __restrict__ selfmask,
                                                                        int mask_ld, int* __restrict__ done_step) {
  __shared__ int cnt[256];
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
  }
}

__global__ void decode_init_kernel(long long* tokens, int tok_ld, float* cont, int cont_ld_rows, unsigned char* selfmask,
                                   int mask_ld, int* eos_seen, int* done_step, int B, long long sos) {
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    if (tokens) { tokens[(size_t)b * tok_ld] = sos; selfmask[(size_t)b * mask_ld] = sos == 0 ? 1 : 0; }
    if (cont) {
      float* o = cont + (size_t)b * cont_ld_rows * 5;
      o[0] = 0.f; o[1] = 0.f; o[2] = 1.f; o[3] = 0.f; o[4] = 0.f;     // models/sketchformer.py:268
      selfmask[(size_t)b * mask_ld] = 0;
    }
    eos_seen[b] = 0;
  }
  if (threadIdx.x == 0) *done_step = -1;
}

}  // namespace

extern "C" int skf_attention_decode(const float* Q, int ldq, const float* K, const float* V, int ld_kv,
                                    long long kv_batch_stride, const unsigned char* key_mask, int key_mask_ld,
                                    const int* key_limit, int key_limit_all, int B, int H, int Lk, int dh, float* O,
                                    int ldo, skf_stream_t stream) {
  SKF_CHECK_ARG(Q && K && V && O, "null operand");
  SKF_CHECK_ARG(B > 0 && H > 0 && Lk > 0 && Lk <= 512, "need 0 < Lk <= 512");
  SKF_CHECK_ARG(dh == 16 || dh == 32 || dh == 64, "head size must be 16, 32 or 64");
  SKF_CHECK_ARG((ldq & 3) == 0 && (ld_kv & 3) == 0 && (kv_batch_stride & 3) == 0 &&
                (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) == 0, "Q/K/V must allow 16-byte row loads");
  AttnDecodeParams p{Q, ldq, K, V, ld_kv, kv_batch_stride, key_mask, key_mask_ld, key_limit, key_limit_all, B, H, Lk, O, ldo};
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(skf_cdiv(B * H, 4)), block(256);
  SkfProfScope ps(st, "attn_decode", 4.0 * B * H * (double)Lk * dh, 8.0 * B * H * (double)Lk * dh);
#define SKF_AD(DHV)                                                                         \
  { if (Lk <= 256) hipLaunchKernelGGL((attn_decode_kernel<DHV, 4>), grid, block, 0, st, p);   \
    else hipLaunchKernelGGL((attn_decode_kernel<DHV, 8>), grid, block, 0, st, p); }
  if (dh == 16) SKF_AD(16) else if (dh == 32) SKF_AD(32) else SKF_AD(64)
#undef SKF_AD
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_decode_init(long long* tokens, int tok_ld, float* cont, int cont_ld_rows, unsigned char* selfmask,
                               int mask_ld, int* eos_seen, int* done_step, int B, long long sos, skf_stream_t stream) {
  SKF_CHECK_ARG((tokens || cont) && selfmask && eos_seen && done_step && B > 0, "bad argument");
  hipLaunchKernelGGL(decode_init_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tokens, tok_ld, cont, cont_ld_rows,
                     selfmask, mask_ld, eos_seen, done_step, B, sos);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_decode_select_tokens(const float* logits, int ld, int B, int V, int n_valid, int step, long long eos,
                                        long long* tokens, int tok_ld, unsigned char* selfmask, int mask_ld,
                                        int* eos_seen, int* done_step, skf_stream_t stream) {
  SKF_CHECK_ARG(logits && tokens && selfmask && eos_seen && done_step, "null operand");
  SKF_CHECK_ARG(B > 0 && V > 0 && n_valid > 0 && n_valid <= B && step >= 0 && step + 1 < tok_ld && step + 1 < mask_ld, "bad shape");
  hipLaunchKernelGGL(decode_select_tokens_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, ld, B, V, n_valid,
                     step, eos, tokens, tok_ld, selfmask, mask_ld, eos_seen, done_step);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_decode_select_continuous(const float* pred, int ld, int B, int n_valid, int step, float* out,
                                            int out_ld_rows, unsigned char* selfmask, int mask_ld, int* done_step,
                                            skf_stream_t stream) {
  SKF_CHECK_ARG(pred && out && selfmask && done_step, "null operand");
  SKF_CHECK_ARG(B > 0 && n_valid > 0 && n_valid <= B && step >= 0 && step + 1 < out_ld_rows && step + 1 < mask_ld, "bad shape");
  hipLaunchKernelGGL(decode_select_continuous_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, ld, B, n_valid,
                     step, out, out_ld_rows, selfmask, mask_ld, done_step);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
