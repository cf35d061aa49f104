// HBM-bound row kernels: embedding (+scale +positional encoding +dropout),
// residual + dropout + LayerNorm (fwd / bwd), softmax cross-entropy heads, the
// SelfAttnV1 bottleneck pool, the DenseExpander, column reductions.
// One 64-lane wave owns one (B*L) row; rows are read/written once, fully coalesced
// (a 128-float row = one 512-byte wave access).
#include "skf_common.h"
#include "skf_bf16.h"

int skf_pool_bwd_partials(float* u_inout_dpre, const float* Vw, const float* x, const float* a, const float* demb, int B, int L, int U, int d,
                          float* dx, float* dV_part, hipStream_t s);
int skf_expander_bwd_partials(const float* dpre, const float* emb, const float* w, int B, int L, int d, float* demb, int demb_accumulate,
                              float* p1, float* p2, hipStream_t s);
namespace {

// two consecutive activation values as fp32 (the bf16 path reuses the sorted embedding gradient kernel)
__device__ __forceinline__ float2 skf_ld2(const float* p, size_t i) { return *reinterpret_cast<const float2*>(p + i); }
__device__ __forceinline__ float2 skf_ld2(const skf_bf16* p, size_t i) {
  float2 r;
  skf_unpack2(*reinterpret_cast<const uint32_t*>(p + i), r.x, r.y);
  return r;
}

#ifndef SKF_LN_FWD_GRID
#define SKF_LN_FWD_GRID 2048
#endif
#ifndef SKF_LN_BWD_GRID
#define SKF_LN_BWD_GRID 640    // more workgroups shorten the kernel (15.4 -> 14.7 us) but lengthen the batched reduction of their partials
#endif
#ifndef SKF_LN_UR
#define SKF_LN_UR 1            // row groups in flight per wave iteration (2 measured 0.3-0.5 us slower per launch in the step)
#endif
#ifndef SKF_LN_BWD_THREADS
#define SKF_LN_BWD_THREADS 256 // threads per workgroup of the LayerNorm backward: more waves per workgroup = more bytes in flight per CU
#endif                         // at the same number of dgamma / dbeta partial rows (one per workgroup)
#ifndef SKF_LN_BWD_UR
#define SKF_LN_BWD_UR SKF_LN_UR
#endif
constexpr int kMaxGrid = SKF_LN_FWD_GRID;
constexpr int kLnBwdGrid = SKF_LN_BWD_GRID;    // workgroups of the LayerNorm backward (each leaves one [2][d] partial for the column sums)

__global__ void padding_mask_kernel(const long long* __restrict__ tok, int tok_ld, int B, int L,
                                    unsigned char* __restrict__ out) {
  const int n = B * L;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256)
    out[e] = tok[(size_t)(e / L) * tok_ld + (e % L)] == 0 ? 1 : 0;
}

// ------------------------------------------------------------------ embedding
// builders/layers/transformer.py:288-296 / 325-334:
//   x = Embedding(tok); x *= sqrt(d_model); x += pos[:, :L]; x = Dropout(x)
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* __restrict__ tok, int tok_ld, int Lrows,
                                                        int rows, const float* __restrict__ table, int vocab, int d,
                                                        const float* __restrict__ pos, float* __restrict__ out,
                                                        float rate, uint32_t site, const SkfStepState* st) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float sq = sqrtf((float)d);
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const int b = row / Lrows, t = row % Lrows;
    long long tk = tok[(size_t)b * tok_ld + t];
    // out-of-range ids: a zero embedding vector like tf.gather on a GPU (the backward kernels skip them, so the
    // pair is consistent); TrainEngine rejects them on the host before they get here (engine._dev_tokens)
    const bool oob = tk < 0 || tk >= vocab;
    if (oob) tk = 0;
    const float* e = table + (size_t)tk * d;
    const float* pe = pos + (size_t)t * d;
    float* o = out + (size_t)row * d;
    for (int c = lane * 2; c < d; c += 128) {
      float2 v = *reinterpret_cast<const float2*>(e + c);
      if (oob) v = make_float2(0.f, 0.f);
      const float2 pp = *reinterpret_cast<const float2*>(pe + c);
      v.x = v.x * sq + pp.x; v.y = v.y * sq + pp.y;
      if (rate > 0.f) {
        const uint32_t idx = (uint32_t)row * (uint32_t)d + c;
        v.x *= skf_keep(sk, idx, thresh) ? inv_keep : 0.f;
        v.y *= skf_keep(sk, idx + 1, thresh) ? inv_keep : 0.f;
      }
      *reinterpret_cast<float2*>(o + c) = v;
    }
  }
}

// Embedding gradient (dense (V,d) buffer, pre-zeroed): scatter-add of
// dx * dropout * sqrt(d).  A wave takes 64 consecutive rows, groups equal token
// ids with ballots and issues one atomic row-add per distinct id (PAD rows, ~60 %
// of a QuickDraw batch, collapse to one add per wave).
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long long* __restrict__ tok, int tok_ld, int Lrows,
                                                        int rows, const float* __restrict__ dx, int vocab, int d,
                                                        float* __restrict__ dtable, float rate, uint32_t site,
                                                        const SkfStepState* st) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float sq = sqrtf((float)d);
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  for (int base = (blockIdx.x * 4 + wave) * 64; base < rows; base += gridDim.x * 4 * 64) {
    const int row = base + lane;
    long long tk = -1;
    if (row < rows) {
      tk = tok[(size_t)(row / Lrows) * tok_ld + (row % Lrows)];
      if (tk < 0 || tk >= vocab) tk = -1;
    }
    unsigned long long todo = __ballot(tk >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const long long ltk = __shfl(tk, leader, 64);
      unsigned long long grp = __ballot(tk == ltk) & todo;
      todo &= ~grp;
      for (int c = lane * 2; c < d; c += 128) {
        float2 acc = make_float2(0.f, 0.f);
        unsigned long long m = grp;
        while (m) {
          const int src = __ffsll((long long)m) - 1;
          m &= m - 1;
          const int r = base + src;
          float2 v = *reinterpret_cast<const float2*>(dx + (size_t)r * d + c);
          if (rate > 0.f) {
            const uint32_t idx = (uint32_t)r * (uint32_t)d + c;
            v.x *= skf_keep(sk, idx, thresh) ? inv_keep : 0.f;
            v.y *= skf_keep(sk, idx + 1, thresh) ? inv_keep : 0.f;
          }
          acc.x += v.x; acc.y += v.y;
        }
        atomicAdd(dtable + (size_t)ltk * d + c, acc.x * sq);
        atomicAdd(dtable + (size_t)ltk * d + c + 1, acc.y * sq);
      }
    }
  }
}

// Atomic-free, run-to-run deterministic embedding gradient in two launches.
//  (1) embed_sort_kernel (ONE workgroup; depends on the tokens only, so the train step runs it on the side stream under the
//      forward): STABLE counting sort of the B*L token positions by id -> order[] (the positions of an id ascend), and a chunk
//      list {id, first, count, n, j}: every id gets n = ceil(count / 256) chunks, at least one (count 0 = "store zeros").
//      Stability without a histogram per thread: wave w owns a contiguous range of rows and its own histogram [vocab] in
//      LDS; an exclusive scan over the 16 waves gives each wave its base inside every id's segment, and inside a 64-row
//      step the rank of a lane among the lanes with the same id comes from ballots (one per distinct id of the step).
//      (vocab > kEmbOrderedVocab does not fit 16 histograms in LDS: one count table, and a single wave walks all rows in order
//      for the scatter - still stable.)
//  (2) embed_bwd_sorted_kernel: one workgroup per chunk (a wave per 64 positions, summed through LDS in wave order) sums the
//      dx rows of its positions (512-byte coalesced rows, dropout mask and sqrt(d) as in embed_bwd_kernel).  Single-chunk ids
//      STORE the table row; the chunks of a split id (PAD, frequent tokens) leave partial rows in the workspace and whichever
//      finishes last (a ticket per id) adds them up IN CHUNK ORDER and stores the row - no float atomics anywhere, so the
//      result does not depend on which workgroup runs when.
constexpr int kEmbChunk = 256;   // positions per chunk = per workgroup of the gradient kernel (4 waves x 64)
constexpr int kEmbOrderedVocab = 2032;   // 19 tables of vocab ints + the 8 KB of scan scratch must fit 160 KB of LDS
constexpr int kEmbMaxD = 512;    // widest row the partial slab of the workspace is sized for
constexpr int kEmbCursorBits = 25;   // ordered scatter: cursor bits of a (wave, id) word; the other 7 hold the step's group count (<= 64)
struct EmbChunk { int id, first, count, n, j, pad; };   // n chunks of this id, this one is the j-th

// NWH = number of per-wave histograms: 16 (vocab <= kEmbOrderedVocab: every wave scatters its own contiguous range of rows) or
// 1 (larger vocabularies, e.g. the grid tokenizer's 10004 ids: 16 histograms do not fit in LDS, so all waves count into one table
// and ONE wave walks the rows in order for the scatter - still stable, i.e. run-to-run deterministic; the cursor table reuses
// the count table once the chunk descriptors are written).
template <int NWH>
__global__ __launch_bounds__(1024) void embed_sort_kernel(const long long* __restrict__ tok, int tok_ld, int Lrows, int rows,
                                                          int vocab, int* __restrict__ hdr, EmbChunk* __restrict__ chunks,
                                                          int* __restrict__ order, int* __restrict__ done) {
  extern __shared__ int smem_i[];
  int* cnt = smem_i;                     // [vocab]      positions per id (NWH == 1: later the cursor table of the scatter)
  int* pst = smem_i + vocab;             // [vocab]      first position of the id in order[]
  int* cst = smem_i + 2 * vocab;         // [vocab + 1]  first chunk of the id (exclusive scan; [vocab] = number of chunks)
  int* hist = NWH == 1 ? cnt : cst + vocab + 1;   // NWH == 16: [16][vocab] positions per (wave, id) -> the wave's cursor inside the id's segment
  __shared__ int part_tok[1024], part_chk[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int RW = ((rows + 15) / 16 + 63) & ~63;      // rows per wave, a multiple of the 64-row step
  for (int v = tid; v < vocab; v += 1024) { cnt[v] = 0; done[v] = 0; }
  if (NWH == 16) for (int e = tid; e < 16 * vocab; e += 1024) hist[e] = 0;
  __syncthreads();
  // token of row r, or -1 (row past the end / id out of range).  The passes below request TB steps of tokens before using
  // the first: one at a time, every 64-row step was a memory round trip of its own (2 x 25 of them at the cfg-2 size)
  constexpr int TB = 8;
  auto token_of = [&](int r) -> int {
    if (r >= rows) return -1;
    const long long t = tok[(size_t)(r / Lrows) * tok_ld + (r % Lrows)];
    return (t >= 0 && t < vocab) ? (int)t : -1;
  };
  const int rend = min(rows, (wave + 1) * RW);
  int* my_hist = NWH == 16 ? hist + wave * vocab : cnt;
  for (int r0 = wave * RW; r0 < rend; r0 += 64 * TB) {
    int tk[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) { const int r = r0 + 64 * u + lane; tk[u] = r < rend ? token_of(r) : -1; }
#pragma unroll
    for (int u = 0; u < TB; ++u)
      if (tk[u] >= 0) atomicAdd(&my_hist[tk[u]], 1);
  }
  __syncthreads();
  if (NWH == 16) {
    for (int v = tid; v < vocab; v += 1024) {          // counts per id; hist -> exclusive prefix over the waves
      int run = 0;
      for (int w = 0; w < 16; ++w) { const int c = hist[w * vocab + v]; hist[w * vocab + v] = run; run += c; }
      cnt[v] = run;
    }
    __syncthreads();
  }
  // exclusive scans over ids: positions and chunks.  Thread t owns ids [t*per, (t+1)*per)
  const int per = (vocab + 1023) / 1024;
  int st = 0, sc = 0;
  for (int k = 0; k < per; ++k) {
    const int v = tid * per + k;
    if (v < vocab) { const int c = cnt[v]; st += c; sc += c == 0 ? 1 : (c + kEmbChunk - 1) / kEmbChunk; }
  }
  part_tok[tid] = st; part_chk[tid] = sc;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {      // Hillis-Steele inclusive scan (two arrays)
    const int a = tid >= off ? part_tok[tid - off] : 0, b = tid >= off ? part_chk[tid - off] : 0;
    __syncthreads();
    part_tok[tid] += a; part_chk[tid] += b;
    __syncthreads();
  }
  int pos = part_tok[tid] - st, chk = part_chk[tid] - sc;
  const int nchunks = part_chk[1023];
  if (tid == 1023) { hdr[0] = nchunks; hdr[1] = part_tok[1023]; cst[vocab] = nchunks; }
  for (int k = 0; k < per; ++k) {
    const int v = tid * per + k;
    if (v >= vocab) break;
    const int c = cnt[v];
    pst[v] = pos; cst[v] = chk;
    pos += c; chk += c == 0 ? 1 : (c + kEmbChunk - 1) / kEmbChunk;
  }
  __syncthreads();
  // chunk descriptors, one per thread and round: the id of chunk ci = last v with cst[v] <= ci (binary search)
  for (int ci = tid; ci < nchunks; ci += 1024) {
    int lo = 0, hi = vocab - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (cst[mid] <= ci) lo = mid; else hi = mid - 1; }
    const int v = lo, j = ci - cst[v], c = cnt[v];
    const int left = c - j * kEmbChunk;
    chunks[ci] = EmbChunk{v, pst[v] + j * kEmbChunk, left < kEmbChunk ? (left > 0 ? left : 0) : kEmbChunk, cst[v + 1] - cst[v], j, 0};
  }
  __syncthreads();
  if (NWH == 1) {                                                   // the counts are no longer needed: cursors start at 0
    for (int v = tid; v < vocab; v += 1024) cnt[v] = 0;
    __syncthreads();
  }
  // stable scatter: a scattering wave walks its rows in order; inside a 64-row step the lanes with equal ids are ranked by lane
  const int sbeg = NWH == 16 ? wave * RW : 0, send = NWH == 16 ? rend : (wave == 0 ? rows : 0);
  for (int rb = sbeg; rb < send; rb += 64 * TB) {
    int tkb[TB];
#pragma unroll
    for (int u = 0; u < TB; ++u) { const int rr = rb + 64 * u + lane; tkb[u] = rr < send ? token_of(rr) : -1; }
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      const int r = rb + 64 * u + lane;
      const int tk = tkb[u];
      // Size of the lane's group (lanes of the step with the same id) from ONE LDS atomic on the high bits of the cursor word
      // (only the count is used, not the order in which the adds land); ids that occur once - most of them - are done.  The
      // others (PAD, repeats) get their rank among the equal lanes from ballots, one round per such id.
      // word = cursor (low 25 bits: after the scan over the waves it is the offset inside the id's WHOLE segment, i.e. up to
      // rows - 1) | group count of the step (high 7 bits: at most 64 lanes)
      const int slot = (NWH == 16 ? wave * vocab : 0) + (tk >= 0 ? tk : 0);
      if (tk >= 0) atomicAdd(reinterpret_cast<unsigned*>(hist) + slot, 1u << kEmbCursorBits);
      const unsigned wv = tk >= 0 ? reinterpret_cast<volatile unsigned*>(hist)[slot] : 0u;       // LDS operations of a wave execute in order
      const int cur = (int)(wv & ((1u << kEmbCursorBits) - 1u)), grp = (int)(wv >> kEmbCursorBits);
      int rank = 0;
      bool lead = grp == 1;
      unsigned long long active = __ballot(grp > 1);
      while (active) {
        const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)active) - 1);
        const int id0 = __builtin_amdgcn_readlane(tk, leader);
        const unsigned long long m = __ballot(tk == id0);
        if (tk == id0) { rank = __popcll(m & ((1ull << lane) - 1ull)); lead = lane == leader; }
        active &= ~m;
      }
      if (tk >= 0) {
        order[pst[tk] + cur + rank] = r;
        if (lead) hist[slot] = cur + grp;                            // cursor advanced, count cleared
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_sorted_kernel(const int* __restrict__ hdr, const EmbChunk* __restrict__ chunks,
                                                               const int* __restrict__ order, const T* __restrict__ dx, int d,
                                                               float* __restrict__ dtable, float rate, uint32_t site,
                                                               const SkfStepState* st, float* __restrict__ partial,
                                                               int* __restrict__ done) {
  extern __shared__ float emb_red[];       // [3 waves][d] partial rows of waves 1..3
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ci = blockIdx.x;
  if (ci >= hdr[0]) return;
  const EmbChunk ch = chunks[ci];
  const float sq = sqrtf((float)d);
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  // wave w owns positions [64w, 64w + 64) of the chunk; lane j holds one of them
  const int wfirst = ch.first + 64 * wave;
  const int wcount = ch.count - 64 * wave < 0 ? 0 : (ch.count - 64 * wave > 64 ? 64 : ch.count - 64 * wave);
  int mine = lane < wcount ? order[wfirst + lane] : 0x7fffffff;
  // The scatter of the sort is unordered (LDS atomics): sort the wave's positions (bitonic network over the 64 lanes) so
  // that the summation order - and with it the result of every id with at most 64 positions - does not change between runs.
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (int j2 = k >> 1; j2 > 0; j2 >>= 1) {
      const int other = __shfl_xor(mine, j2, 64);
      const bool up = (lane & k) == 0, lower = (lane & j2) == 0;
      mine = (lower == up) ? min(mine, other) : max(mine, other);
    }
  for (int c0 = 0; c0 < d; c0 += 128) {              // uniform trip count: every lane takes part in the shuffles
    const int c = c0 + lane * 2;
    const bool on = c < d;
    const int cc = on ? c : 0;
    float2 acc0 = make_float2(0.f, 0.f), acc1 = acc0;
    // eight rows in flight per round (a wave's share is a latency chain of up to 64 row reads otherwise); rows past its
    // count re-read its first row with weight 0
    for (int j = 0; j < wcount; j += 8) {
      int r[8];
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int src = j + u < wcount ? j + u : 0;
        r[u] = __shfl(mine, src, 64);
        v[u] = skf_ld2(dx, (size_t)r[u] * d + cc);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float wx = j + u < wcount ? 1.f : 0.f, wy = wx;
        if (rate > 0.f) {
          const uint32_t i0 = (uint32_t)r[u] * (uint32_t)d + cc;
          wx *= skf_keep(sk, i0, thresh) ? inv_keep : 0.f;
          wy *= skf_keep(sk, i0 + 1, thresh) ? inv_keep : 0.f;
        }
        if (u & 1) { acc1.x += v[u].x * wx; acc1.y += v[u].y * wy; }
        else { acc0.x += v[u].x * wx; acc0.y += v[u].y * wy; }
      }
    }
    if (on && wave > 0) *reinterpret_cast<float2*>(&emb_red[(wave - 1) * d + c]) = make_float2(acc0.x + acc1.x, acc0.y + acc1.y);
    __syncthreads();
    if (on && wave == 0) {
      float gx = acc0.x + acc1.x, gy = acc0.y + acc1.y;
#pragma unroll
      for (int w = 0; w < 3; ++w) { const float2 t = *reinterpret_cast<const float2*>(&emb_red[w * d + c]); gx += t.x; gy += t.y; }
      gx *= sq; gy *= sq;
      float* dst = ch.n == 1 ? dtable + (size_t)ch.id * d + c : partial + (size_t)ci * d + c;
      *reinterpret_cast<float2*>(dst) = make_float2(gx, gy);
    }
    __syncthreads();
  }
  if (ch.n > 1) {
    // split id: the chunk that finishes last adds the partial rows of all chunks in chunk order
    if (threadIdx.x == 0) {
      __threadfence();
      s_last = atomicAdd(&done[ch.id], 1) == ch.n - 1;
    }
    __syncthreads();
    if (s_last) {
      __threadfence();
      const float* p0 = partial + (size_t)(ci - ch.j) * d;
      for (int c = threadIdx.x; c < d; c += 256) {
        float sum = 0.f;
        for (int j0 = 0; j0 < ch.n; j0 += 16) {                      // 16 partial rows requested at once, added in chunk order
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u)
            v[u] = j0 + u < ch.n ? __hip_atomic_load(p0 + (size_t)(j0 + u) * d + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
          for (int u = 0; u < 16; ++u) sum += v[u];
        }
        dtable[(size_t)ch.id * d + c] = sum;
      }
      if (threadIdx.x == 0) done[ch.id] = 0;          // ready for another gradient pass over the same sort
    }
  }
}

// ------------------------------------------------------------------ layernorm
// out = LayerNorm(x + Dropout(y)), eps=1e-6, biased variance
// (builders/layers/transformer.py:217-222, 247-260).  z = x + drop(y) is written
// over y (it is what the backward needs); stats = (mean, rstd) per row.
template <int VPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ out, float* __restrict__ stats, int rows,
                                                     float rate, uint32_t site, const SkfStepState* st) {
  constexpr int D = VPL * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  float gm[VPL], bt[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) { gm[v] = gamma[lane * VPL + v]; bt[v] = beta[lane * VPL + v]; }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const size_t off = (size_t)row * D + lane * VPL;
    float z[VPL];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      float yv = y[off + v];
      if (rate > 0.f) yv *= skf_keep(sk, (uint32_t)off + v, thresh) ? inv_keep : 0.f;
      z[v] = x[off + v] + yv;
      sum += z[v];
    }
    const float mean = wave_sum(sum) * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) { const float c = z[v] - mean; sq += c * c; }
    const float rstd = rsqrtf(wave_sum(sq) * (1.0f / D) + 1e-6f);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      y[off + v] = z[v];
      out[off + v] = (z[v] - mean) * rstd * gm[v] + bt[v];
    }
    if (lane == 0) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
  }
}

// Same, D = 4*LPR <= 256: a row is read by LPR lanes with 16-byte loads, 64/LPR rows per wave instruction and two
// such groups in flight per iteration (the kernel is bound by bytes in flight, not by arithmetic).
template <int LPR>
__global__ __launch_bounds__(256) void ln_fwd_v4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ out, float* __restrict__ stats, int rows,
                                                        float rate, uint32_t site, const SkfStepState* st) {
  constexpr int D = 4 * LPR, RPW = 64 / LPR, UR = SKF_LN_UR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rsel = lane / LPR;
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * sub), bt = *reinterpret_cast<const f32x4*>(beta + 4 * sub);
  const int stride = gridDim.x * 4 * RPW * UR;
  for (int row0 = (blockIdx.x * 4 + wave) * RPW * UR; row0 < rows; row0 += stride) {
    f32x4 xv[UR], yv[UR];
    size_t off[UR];
    bool ok[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int row = row0 + u * RPW + rsel;
      ok[u] = row < rows;
      off[u] = (size_t)(ok[u] ? row : rows - 1) * D + 4 * sub;
      xv[u] = *reinterpret_cast<const f32x4*>(x + off[u]);
      yv[u] = *reinterpret_cast<const f32x4*>(y + off[u]);
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      f32x4 z;
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float yy = yv[u][e];
        if (rate > 0.f) yy *= skf_keep(sk, (uint32_t)off[u] + e, thresh) ? inv_keep : 0.f;
        z[e] = xv[u][e] + yy;
        sum += z[e];
      }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      const float mean = sum * (1.0f / D);
      float sq = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float c = z[e] - mean; sq += c * c; }
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
      const float rstd = rsqrtf(sq * (1.0f / D) + 1e-6f);
      if (ok[u]) {
        *reinterpret_cast<f32x4*>(y + off[u]) = z;
        *reinterpret_cast<f32x4*>(out + off[u]) = (z - mean) * rstd * gm + bt;
        if (sub == 0) { const size_t row = off[u] / D; stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
      }
    }
  }
}

// dz = LN'(dout); dy = dz * dropmask (written to dy when dy != null, i.e. rate > 0);
// partial dgamma/dbeta per workgroup -> part[block][2][D].
template <int VPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     float* __restrict__ dz, float* __restrict__ dy,
                                                     float* __restrict__ part, int rows, float rate, uint32_t site,
                                                     const SkfStepState* st) {
  constexpr int D = VPL * 64;
  constexpr int UR = 2;            // rows per wave iteration: twice the bytes in flight (the kernel is latency bound)
  __shared__ float red[4][2][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  float gm[VPL], dg[VPL], db[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) { gm[v] = gamma[lane * VPL + v]; dg[v] = 0.f; db[v] = 0.f; }
  const int stride = gridDim.x * 4;
  for (int row0 = blockIdx.x * 4 + wave; row0 < rows; row0 += UR * stride) {
    float dv[UR][VPL], zv[UR][VPL], mean[UR], rstd[UR];
    bool ok[UR];
    size_t off[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int row = row0 + u * stride;
      ok[u] = row < rows;
      const int rr = ok[u] ? row : row0;             // clamped: loads stay in bounds, nothing is stored / summed
      off[u] = (size_t)rr * D + lane * VPL;
      mean[u] = stats[2 * (size_t)rr]; rstd[u] = stats[2 * (size_t)rr + 1];
#pragma unroll
      for (int v = 0; v < VPL; ++v) { dv[u][v] = dout[off[u] + v]; zv[u][v] = z[off[u] + v]; }
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      if (!ok[u]) continue;                          // wave-uniform
      float xh[VPL], gg[VPL];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const float d = dv[u][v];
        xh[v] = (zv[u][v] - mean[u]) * rstd[u];
        gg[v] = d * gm[v];
        dg[v] += d * xh[v];
        db[v] += d;
        s1 += gg[v];
        s2 += gg[v] * xh[v];
      }
      s1 = wave_sum(s1) * (1.0f / D);
      s2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const float g = rstd[u] * (gg[v] - s1 - xh[v] * s2);
        dz[off[u] + v] = g;
        if (dy) dy[off[u] + v] = g * (skf_keep(sk, (uint32_t)off[u] + v, thresh) ? inv_keep : 0.f);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VPL; ++v) { red[wave][0][lane * VPL + v] = dg[v]; red[wave][1][lane * VPL + v] = db[v]; }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * D; e += 256) {
    const int w = e / D, c = e % D;
    part[(size_t)blockIdx.x * 2 * D + e] = red[0][w][c] + red[1][w][c] + red[2][w][c] + red[3][w][c];
  }
}

// Same, D = 4*LPR <= 256: 16-byte loads, 64/LPR rows per wave instruction, SKF_LN_UR groups in flight (see ln_fwd_v4_kernel).
template <int LPR>
__global__ __launch_bounds__(SKF_LN_BWD_THREADS) void ln_bwd_v4_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                        const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        float* __restrict__ dz, float* __restrict__ dy,
                                                        float* __restrict__ part, int rows, float rate, uint32_t site,
                                                        const SkfStepState* st, const int* __restrict__ live_len, int rps) {
  constexpr int D = 4 * LPR, RPW = 64 / LPR, UR = SKF_LN_BWD_UR, NWV = SKF_LN_BWD_THREADS / 64;
  __shared__ float red[NWV][2][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rsel = lane / LPR;
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * sub);
  f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = dg;
  const int stride = gridDim.x * NWV * RPW * UR;
  const int first = (blockIdx.x * NWV + wave) * RPW * UR;
  // The loop is software-pipelined: the rows of iteration k + 1 are requested before iteration k is computed, and the live flags of
  // the first PRE iterations (row t of sample b with t >= live_len[b]: dout is exactly zero, skf_target_live_len - neither it nor z is
  // read) come from ONE batch of live_len loads in front of the loop.  Before, every iteration was two dependent memory round trips
  // (live_len[b], then the rows it gates) followed by the arithmetic: ~5 iterations x 2 x 1.5 us of a 15-18 us launch.
  constexpr int PRE = 8;
  unsigned lvmask = 0xffffffffu;
  if (live_len) {
    int ll[PRE][UR], tt[PRE][UR];
#pragma unroll
    for (int k = 0; k < PRE; ++k)
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const int row = first + k * stride + u * RPW + rsel, rr = row < rows ? row : rows - 1, b = rr / rps;
        tt[k][u] = rr - b * rps;
        ll[k][u] = live_len[b];
      }
    lvmask = 0u;
#pragma unroll
    for (int k = 0; k < PRE; ++k)
#pragma unroll
      for (int u = 0; u < UR; ++u) lvmask |= (tt[k][u] < ll[k][u] ? 1u : 0u) << (k * UR + u);
  }
  struct Rows { f32x4 dv[UR], zv[UR]; float mean[UR], rstd[UR], keep[UR]; size_t off[UR]; bool ok[UR]; };
  auto fetch = [&](int row0, int k, Rows& r) {
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int row = row0 + u * RPW + rsel;
      r.ok[u] = row < rows;
      const int rr = r.ok[u] ? row : rows - 1;
      r.off[u] = (size_t)rr * D + 4 * sub;
      r.mean[u] = stats[2 * (size_t)rr]; r.rstd[u] = stats[2 * (size_t)rr + 1];
      // dead rows re-read the first row of the tensors (a cache hit) and are zeroed: EVERY lane issues both loads on every path, so
      // the compiler can count them (guarded loads forced s_waitcnt vmcnt(0) right behind the prefetch); beyond PRE sweeps of the grid
      // (very long inputs) every row is read - dout of a dead row is exactly zero, so that is only slower, not different
      const bool lv = k < PRE ? ((lvmask >> (k * UR + u)) & 1u) != 0u : true;
      const size_t src = lv ? r.off[u] : (size_t)(4 * sub);
      r.dv[u] = *reinterpret_cast<const f32x4*>(dout + src);
      r.zv[u] = *reinterpret_cast<const f32x4*>(z + src);
      r.keep[u] = lv ? 1.f : 0.f;                                 // applied when the rows are consumed, not here: nothing may touch the loads yet
    }
  };
  // dz / dy leave through buffer descriptors (rows past the end fall outside; no dy: an empty descriptor) and the prefetch runs on
  // every iteration (clamped rows): with nothing conditional between a row's loads and its use the compiler waits with a counted
  // vmcnt instead of vmcnt(0), which had put the wait for the PREFETCHED rows in front of the arithmetic of the current ones
  typedef unsigned lnb_u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t dz_rsrc = __builtin_amdgcn_make_buffer_rsrc(dz, 0, (unsigned)rows * (unsigned)D * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(dy ? dy : dz, 0, dy ? (unsigned)rows * (unsigned)D * 4u : 0u, 0x00020000);
  Rows cur, nxt;
  fetch(first, 0, cur);
  int k = 0;
  for (int row0 = first; row0 < rows; row0 += stride, ++k) {
    fetch(row0 + stride, k + 1, nxt);
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const float live = cur.ok[u] ? 1.f : 0.f;             // rows past the end: loads were clamped, contribute nothing
      const f32x4 dvu = cur.dv[u] * cur.keep[u], zvu = cur.zv[u] * cur.keep[u];
      const f32x4 xh = (zvu - cur.mean[u]) * cur.rstd[u];
      const f32x4 gg = dvu * gm;
      dg += (dvu * xh) * live;
      db += dvu * live;
      float s1 = (gg[0] + gg[1]) + (gg[2] + gg[3]);
      float s2 = (gg[0] * xh[0] + gg[1] * xh[1]) + (gg[2] * xh[2] + gg[3] * xh[3]);
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
      s1 *= (1.0f / D); s2 *= (1.0f / D);
      const f32x4 g = cur.rstd[u] * (gg - s1 - xh * s2);
      const unsigned voff = cur.ok[u] ? (unsigned)cur.off[u] * 4u : 0x7ffffff0u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lnb_u32x4, g), dz_rsrc, voff, 0, 0);
      f32x4 gy = g;
      if (rate > 0.f) {                                           // (uniform; dy != null exactly when rate > 0)
#pragma unroll
        for (int e = 0; e < 4; ++e) gy[e] = g[e] * (skf_keep(sk, (uint32_t)cur.off[u] + e, thresh) ? inv_keep : 0.f);
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lnb_u32x4, gy), dy_rsrc, voff, 0, 0);
    }
    cur = nxt;
  }
  // fold the 64/LPR row groups of the wave, then the 4 waves
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) { dg[e] += __shfl_xor(dg[e], o, 64); db[e] += __shfl_xor(db[e], o, 64); }
  }
  if (rsel == 0) {
    *reinterpret_cast<f32x4*>(&red[wave][0][4 * sub]) = dg;
    *reinterpret_cast<f32x4*>(&red[wave][1][4 * sub]) = db;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * D; e += SKF_LN_BWD_THREADS) {
    const int w = e / D, c = e % D;
    float t = red[0][w][c];
#pragma unroll
    for (int k = 1; k < NWV; ++k) t += red[k][w][c];
    part[(size_t)blockIdx.x * 2 * D + e] = t;
  }
}

// out[j] (=|+=) sum_i in[i*ld + j].  64 columns x 16 row groups per 1024-thread workgroup.
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ in, int nrows, int ld, int ncols,
                                                      float* __restrict__ out, int accumulate) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < ncols) {
    int i = rg;
    for (; i + 48 < nrows; i += 64) {
      s0 += in[(size_t)i * ld + col];
      s1 += in[(size_t)(i + 16) * ld + col];
      s2 += in[(size_t)(i + 32) * ld + col];
      s3 += in[(size_t)(i + 48) * ld + col];
    }
    for (; i < nrows; i += 16) s0 += in[(size_t)i * ld + col];
  }
  red[rg][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && col < ncols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][lane];
    out[col] = accumulate ? out[col] + t : t;
  }
}

// ------------------------------------------------------------------ softmax CE
// Sparse softmax cross-entropy from logits for one row per wave, fused with the
// accuracy test (first-index argmax == target) and the gradient, written in place:
//   g = (softmax - onehot) * mask * scale,  mask = (target != 0) when mask_pad.
// builders/losses.py:26-41 (recon), :21-24 (class), builders/keras_metrics.py:25.
__global__ __launch_bounds__(256) void softmax_ce_kernel(float* __restrict__ logits, int ld, int rows, int ncls,
                                                         const long long* __restrict__ target, int tgt_ld, int tgt_cols,
                                                         int tgt_off, int mask_pad, float scale,
                                                         float* __restrict__ row_loss, float* __restrict__ row_hit,
                                                         float* __restrict__ probs_out, int write_grad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    float* x = logits + (size_t)row * ld;
    const long long tg = target[(size_t)(row / tgt_cols) * tgt_ld + (row % tgt_cols) + tgt_off];
    float mx = -INFINITY; int am = 0x7fffffff;
    for (int j = lane; j < ncls; j += 64) {
      const float v = x[j];
      if (v > mx) { mx = v; am = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o, 64); const int oa = __shfl_xor(am, o, 64);
      if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    float se = 0.f;
    for (int j = lane; j < ncls; j += 64) se += __expf(x[j] - mx);
    se = wave_sum(se);
    const float lse = mx + __logf(se);
    const bool valid = tg >= 0 && tg < ncls;
    const float m = (mask_pad && tg == 0) ? 0.f : 1.f;
    if (lane == 0) {
      row_loss[row] = valid ? (lse - x[tg]) * m : 0.f;
      row_hit[row] = (am == (int)tg) ? 1.f : 0.f;
    }
    const float gs = m * scale, rse = 1.0f / se;
    for (int j = lane; j < ncls; j += 64) {
      const float pj = __expf(x[j] - mx) * rse;
      if (probs_out) probs_out[(size_t)row * ncls + j] = pj;
      if (write_grad) x[j] = (pj - ((long long)j == tg ? 1.f : 0.f)) * gs;
    }
  }
}

// Same, for rows that fit in registers (ncls <= 256*NV4, ncls % 4 == 0, 16-byte aligned rows): one read and
// one write of the logits instead of three reads and one write.
template <int NV4>
__global__ __launch_bounds__(256) void softmax_ce_reg_kernel(float* __restrict__ logits, int ld, int rows, int ncls,
                                                             const long long* __restrict__ target, int tgt_ld, int tgt_cols,
                                                             int tgt_off, int mask_pad, float scale,
                                                             float* __restrict__ row_loss, float* __restrict__ row_hit,
                                                             float* __restrict__ probs_out, int write_grad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = ncls >> 2;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    float* x = logits + (size_t)row * ld;
    const long long tg = target[(size_t)(row / tgt_cols) * tgt_ld + (row % tgt_cols) + tgt_off];
    f32x4 v[NV4];
    float mx = -INFINITY; int am = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int j4 = lane + 64 * k;
      v[k] = j4 < n4 ? *reinterpret_cast<const f32x4*>(x + 4 * j4) : (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[k][e] > mx) { mx = v[k][e]; am = 4 * j4 + e; }   // ascending index inside a lane: first maximum wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o, 64); const int oa = __shfl_xor(am, o, 64);
      if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    const bool valid = tg >= 0 && tg < ncls;
    float se = 0.f, xt = 0.f;
#pragma unroll
    for (int k = 0; k < NV4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if ((long long)(4 * (lane + 64 * k) + e) == tg) xt = v[k][e];
        v[k][e] = __expf(v[k][e] - mx);            // exp(-inf) = 0 for the padding slots
        se += v[k][e];
      }
    se = wave_sum(se);
    xt = wave_sum(xt);                              // exactly one lane holds the target logit
    const float lse = mx + __logf(se);
    const float m = (mask_pad && tg == 0) ? 0.f : 1.f;
    if (lane == 0) {
      row_loss[row] = valid ? (lse - xt) * m : 0.f;
      row_hit[row] = (am == (int)tg) ? 1.f : 0.f;
    }
    const float gs = m * scale, rse = 1.0f / se;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int j4 = lane + 64 * k;
      if (j4 < n4) {
        const f32x4 pj = v[k] * rse;
        if (probs_out) *reinterpret_cast<f32x4*>(probs_out + (size_t)row * ncls + 4 * j4) = pj;
        if (write_grad) {
          f32x4 gq;
#pragma unroll
          for (int e = 0; e < 4; ++e) gq[e] = (pj[e] - ((long long)(4 * j4 + e) == tg ? 1.f : 0.f)) * gs;
          *reinterpret_cast<f32x4*>(x + 4 * j4) = gq;
        }
      }
    }
  }
}

// Same for wide rows (2048 < ncls <= 1024*NV4: the grid tokenizer's 10004 classes, utils/tokenizer.py:104-198): ONE WORKGROUP per
// row keeps it in registers (NV4 float4 per thread), maximum / first argmax / sum / target logit folded over the four waves
// through LDS - still one read and one write of the logits (1.02 GB at the cfg-2 size instead of four passes).
template <int NV4>
__global__ __launch_bounds__(256) void softmax_ce_wide_kernel(float* __restrict__ logits, int ld, int rows, int ncls,
                                                              const long long* __restrict__ target, int tgt_ld, int tgt_cols,
                                                              int tgt_off, int mask_pad, float scale,
                                                              float* __restrict__ row_loss, float* __restrict__ row_hit,
                                                              float* __restrict__ probs_out, int write_grad) {
  __shared__ float r_mx[4], r_se[4], r_xt[4];
  __shared__ int r_am[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n4 = ncls >> 2;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    float* x = logits + (size_t)row * ld;
    const long long tg = target[(size_t)(row / tgt_cols) * tgt_ld + (row % tgt_cols) + tgt_off];
    f32x4 v[NV4];
    float mx = -INFINITY; int am = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int j4 = tid + 256 * k;
      v[k] = j4 < n4 ? *reinterpret_cast<const f32x4*>(x + 4 * j4) : (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[k][e] > mx) { mx = v[k][e]; am = 4 * j4 + e; }   // ascending index inside a thread: first maximum wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o, 64); const int oa = __shfl_xor(am, o, 64);
      if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    __syncthreads();                                  // (the previous row's readers are done with the slots)
    if (lane == 0) { r_mx[wave] = mx; r_am[wave] = am; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float om = r_mx[w]; const int oa = r_am[w];
      if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    float se = 0.f, xt = 0.f;
#pragma unroll
    for (int k = 0; k < NV4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if ((long long)(4 * (tid + 256 * k) + e) == tg) xt = v[k][e];
        v[k][e] = __expf(v[k][e] - mx);            // exp(-inf) = 0 for the padding slots
        se += v[k][e];
      }
    se = wave_sum(se);
    xt = wave_sum(xt);                              // exactly one thread holds the target logit
    if (lane == 0) { r_se[wave] = se; r_xt[wave] = xt; }
    __syncthreads();
    se = (r_se[0] + r_se[1]) + (r_se[2] + r_se[3]);
    xt = (r_xt[0] + r_xt[1]) + (r_xt[2] + r_xt[3]);
    const bool valid = tg >= 0 && tg < ncls;
    const float lse = mx + __logf(se);
    const float m = (mask_pad && tg == 0) ? 0.f : 1.f;
    if (tid == 0) {
      row_loss[row] = valid ? (lse - xt) * m : 0.f;
      row_hit[row] = (am == (int)tg) ? 1.f : 0.f;
    }
    const float gs = m * scale, rse = 1.0f / se;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int j4 = tid + 256 * k;
      if (j4 < n4) {
        const f32x4 pj = v[k] * rse;
        if (probs_out) *reinterpret_cast<f32x4*>(probs_out + (size_t)row * ncls + 4 * j4) = pj;
        if (write_grad) {
          f32x4 gq;
#pragma unroll
          for (int e = 0; e < 4; ++e) gq[e] = (pj[e] - ((long long)(4 * j4 + e) == tg ? 1.f : 0.f)) * gs;
          *reinterpret_cast<f32x4*>(x + 4 * j4) = gq;
        }
      }
    }
  }
}

// Step metrics + running Keras metrics (builders/keras_metrics.py:19-42).
// metrics layout (floats): [0..4]  this step: recon_loss, recon_acc, class_loss, class_acc, total_loss
//                          [8..12] running totals, [16..20] running counts
__global__ __launch_bounds__(1024) void metrics_kernel(const float* __restrict__ recon_loss, const float* __restrict__ recon_hit,
                                                       int recon_rows, float recon_weight,
                                                       const float* __restrict__ class_loss, const float* __restrict__ class_hit,
                                                       int class_rows, float class_weight, const float* __restrict__ recon_scalar,
                                                       float* __restrict__ metrics) {
  // one workgroup on the critical path between the loss and the backward: 1024 threads, 16-byte loads, four requests in flight
  // per thread (the 256-thread scalar loop took 31 us for the 25.5 k rows of cfg 2); fixed summation order -> deterministic
  __shared__ float red[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const int n4 = ((((uintptr_t)recon_loss | (uintptr_t)recon_hit) & 15) == 0) ? recon_rows >> 2 : 0;
  for (int i = tid; i < n4; i += 4096) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = i + 1024 * u;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      a[u] = j < n4 ? reinterpret_cast<const f32x4*>(recon_loss)[j] : z;
      b[u] = j < n4 ? reinterpret_cast<const f32x4*>(recon_hit)[j] : z;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { s[0] += (a[u][0] + a[u][1]) + (a[u][2] + a[u][3]); s[1] += (b[u][0] + b[u][1]) + (b[u][2] + b[u][3]); }
  }
  for (int i = 4 * n4 + tid; i < recon_rows; i += 1024) { s[0] += recon_loss[i]; s[1] += recon_hit[i]; }
  for (int i = tid; i < class_rows; i += 1024) { s[2] += class_loss[i]; s[3] += class_hit[i]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float v = wave_sum(s[k]);
    if (lane == 0) red[k][wave] = v;
  }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) t += red[k][w];
      red[k][0] = t;
    }
  }
  if (threadIdx.x == 0) {
    const float rl = recon_scalar ? *recon_scalar : (recon_rows ? recon_weight * red[0][0] / recon_rows : 0.f);
    if (recon_scalar) { red[1][0] = 0.f; recon_rows = 0; }   // no token accuracy in continuous mode
    const float cl = class_rows ? class_weight * red[2][0] / class_rows : 0.f;
    metrics[0] = rl; metrics[1] = recon_rows ? red[1][0] / recon_rows : 0.f;
    metrics[2] = cl; metrics[3] = class_rows ? red[3][0] / class_rows : 0.f;
    metrics[4] = rl + cl;
    metrics[8] += rl;  metrics[16] += 1.f;
    metrics[9] += red[1][0]; metrics[17] += (float)recon_rows;
    metrics[10] += cl; metrics[18] += 1.f;
    metrics[11] += red[3][0]; metrics[19] += (float)class_rows;
    metrics[12] += rl + cl; metrics[20] += 1.f;
  }
}

// Sum / max over the 16 waves of a 1024-thread workgroup (red: 16 floats of LDS; result broadcast to all threads).
__device__ __forceinline__ float block16_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += red[k];
  return t;
}
__device__ __forceinline__ float block16_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int k = 1; k < 16; ++k) t = fmaxf(t, red[k]);
  return t;
}
__device__ __forceinline__ float group16_sum(float v) {   // over the 16 lanes that share lane>>4
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

// ------------------------------------------------------------------ SelfAttnV1 pool
// builders/layers/transformer.py:70-73 after u = tanh(xW+b) (a GEMM):
//   s[t] = u[b,t,:].V ; a = softmax_t(s) (no padding mask) ; emb[b,:] = sum_t a[t] x[b,t,:]
// One 1024-thread workgroup per sample.  Rows are read by groups of 16 lanes with 16-byte loads (64 rows in
// flight per pass); the weighted sum splits the time steps over 1024/(d/4) thread groups and folds through LDS.
// U % 4 == 0, d % 4 == 0, d <= 4096.
__global__ __launch_bounds__(1024) void pool_fwd_kernel(const float* __restrict__ u, const float* __restrict__ Vw,
                                                        const float* __restrict__ x, int L, int U, int d,
                                                        float* __restrict__ a_out, float* __restrict__ emb) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* part = sm;            // [4096] partial sums of the weighted sum
  float* sc = sm + 4096;       // [L] scores -> attention weights
  float* red = sc + L;         // [16]
  const int b = blockIdx.x, tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
  const int U4 = U >> 2, d4 = d >> 2;
  for (int t = rg; t < L; t += 64) {
    const float* ur = u + ((size_t)b * L + t) * U;
    float s = 0.f;
    for (int j4 = l16; j4 < U4; j4 += 16)
      s += dot4(*reinterpret_cast<const f32x4*>(ur + 4 * j4), *reinterpret_cast<const f32x4*>(Vw + 4 * j4));
    s = group16_sum(s);
    if (l16 == 0) sc[t] = s;
  }
  __syncthreads();
  const float mx = block16_max(tid < L ? sc[tid] : -INFINITY, red);   // L <= 1024
  const float e = tid < L ? __expf(sc[tid] - mx) : 0.f;
  const float se = block16_sum(e, red);
  if (tid < L) { const float a = e / se; sc[tid] = a; a_out[(size_t)b * L + tid] = a; }
  __syncthreads();
  const int ngrp = 1024 / d4, c4 = tid % d4, tg = tid / d4;
  if (tg < ngrp) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = tg; t < L; t += ngrp) acc += sc[t] * *reinterpret_cast<const f32x4*>(x + ((size_t)b * L + t) * d + 4 * c4);
    *reinterpret_cast<f32x4*>(part + tg * d + 4 * c4) = acc;
  }
  __syncthreads();
  for (int c = tid; c < d; c += 1024) {
    float acc = 0.f;
    for (int k = 0; k < ngrp; ++k) acc += part[k * d + c];
    emb[(size_t)b * d + c] = acc;
  }
}

// backward of the pool: dx_direct[b,t,c] = a[t] demb[c]; da[t] = demb . x[b,t,:];
// ds = a (da - sum a da); dpre[b,t,j] = ds[t] V[j] (1 - u^2) (in place over u);
// dV partial[b][j] = sum_t ds[t] u[b,t,j]
__global__ __launch_bounds__(1024) void pool_bwd_kernel(float* __restrict__ u, const float* __restrict__ Vw,
                                                        const float* __restrict__ x, const float* __restrict__ a_in,
                                                        const float* __restrict__ demb, int L, int U, int d,
                                                        float* __restrict__ dx, float* __restrict__ dV_part) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* part = sm;            // [4096]
  float* dem = sm + 4096;      // [d]
  float* ds = dem + d;         // [L]
  float* av = ds + L;          // [L]
  float* red = av + L;         // [16]
  const int b = blockIdx.x, tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
  const int U4 = U >> 2, d4 = d >> 2;
  for (int t = tid; t < L; t += 1024) av[t] = a_in[(size_t)b * L + t];
  for (int c = tid; c < d; c += 1024) dem[c] = demb[(size_t)b * d + c];
  __syncthreads();
  for (int t = rg; t < L; t += 64) {
    const float* xr = x + ((size_t)b * L + t) * d;
    float* dxr = dx + ((size_t)b * L + t) * d;
    const float at = av[t];
    float s = 0.f;
    for (int c4 = l16; c4 < d4; c4 += 16) {
      const f32x4 de = *reinterpret_cast<const f32x4*>(dem + 4 * c4);
      s += dot4(de, *reinterpret_cast<const f32x4*>(xr + 4 * c4));
      *reinterpret_cast<f32x4*>(dxr + 4 * c4) = at * de;
    }
    s = group16_sum(s);
    if (l16 == 0) ds[t] = s;
  }
  __syncthreads();
  const float dot = block16_sum(tid < L ? av[tid] * ds[tid] : 0.f, red);   // L <= 1024
  if (tid < L) ds[tid] = av[tid] * (ds[tid] - dot);
  __syncthreads();
  const int ngrp = 1024 / U4, j4 = tid % U4, tg = tid / U4;
  if (tg < ngrp) {
    const f32x4 vj = *reinterpret_cast<const f32x4*>(Vw + 4 * j4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = tg; t < L; t += ngrp) {
      f32x4* up = reinterpret_cast<f32x4*>(u + ((size_t)b * L + t) * U + 4 * j4);
      const f32x4 uv = *up;
      const float dst = ds[t];
      acc += dst * uv;
      *up = dst * vj * (1.f - uv * uv);
    }
    *reinterpret_cast<f32x4*>(part + tg * U + 4 * j4) = acc;
  }
  __syncthreads();
  for (int j = tid; j < U; j += 1024) {
    float acc = 0.f;
    for (int k = 0; k < ngrp; ++k) acc += part[k * U + j];
    dV_part[(size_t)b * U + j] = acc;
  }
}

// ------------------------------------------------------------------ DenseExpander
// builders/layers/transformer.py:370-376: pre[b,t,c] = emb[b,c]*w[t] + bias[t]
__global__ __launch_bounds__(256) void expander_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int B, int L, int d,
                                                           float* __restrict__ pre) {
  const size_t total = (size_t)B * L * d;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % d);
    const size_t bt = e / d;
    const int t = (int)(bt % L), b = (int)(bt / L);
    pre[e] = emb[(size_t)b * d + c] * w[t] + bias[t];
  }
}

// demb[b,c] = sum_t dpre[b,t,c] w[t]; per-b partials dw[b][t] = sum_c dpre*emb, dbias[b][t] = sum_c dpre.
// One pass over dpre: a row is read by 16 lanes (KC float4 each, d = 64*KC), which keep the column sums of
// "their" columns in registers; 1024-thread workgroup per sample, column sums folded wave -> LDS -> global.
template <int KC>
__global__ __launch_bounds__(1024) void expander_bwd_kernel(const float* __restrict__ dpre, const float* __restrict__ emb,
                                                            const float* __restrict__ w, int L,
                                                            float* __restrict__ demb, int demb_accumulate,
                                                            float* __restrict__ dw_part, float* __restrict__ db_part) {
  constexpr int d = 64 * KC;
  __shared__ __attribute__((aligned(16))) float part[16 * d];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = tid & 15, rg = tid >> 4;
  f32x4 em[KC], acc[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    em[k] = *reinterpret_cast<const f32x4*>(emb + (size_t)b * d + 4 * (l16 + 16 * k));
    acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int t = rg; t < L; t += 64) {
    const float* r = dpre + ((size_t)b * L + t) * d;
    const float wt = w[t];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(r + 4 * (l16 + 16 * k));
      s1 += dot4(v, em[k]);
      s2 += (v[0] + v[1]) + (v[2] + v[3]);
      acc[k] += wt * v;
    }
    s1 = group16_sum(s1); s2 = group16_sum(s2);
    if (l16 == 0) { dw_part[(size_t)b * L + t] = s1; db_part[(size_t)b * L + t] = s2; }
  }
#pragma unroll
  for (int k = 0; k < KC; ++k) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[k][e];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      acc[k][e] = v;
    }
    if (lane < 16) *reinterpret_cast<f32x4*>(part + wave * d + 4 * (l16 + 16 * k)) = acc[k];
  }
  __syncthreads();
  for (int c = tid; c < d; c += 1024) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k * d + c];
    float* dst = demb + (size_t)b * d + c;
    *dst = demb_accumulate ? *dst + t : t;
  }
}

// Inverted dropout of a flat tensor with the counter-hash mask of (step key, site, element index); the same call
// with the upstream gradient is its backward.  y may alias x.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                      float rate, uint32_t site, const SkfStepState* st) {
  const uint32_t thresh = skf_drop_thresh(rate);
  const float inv_keep = 1.0f / (1.0f - rate);
  const uint32_t sk = rate > 0.f ? skf_site_key(st->drop_key, site) : 0u;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    y[i] = rate > 0.f ? x[i] * (skf_keep(sk, (uint32_t)i, thresh) ? inv_keep : 0.f) : x[i];
}

}  // namespace
// any-width fallbacks (skf_generic.hip)
int skf_ln_fwd_any(const float* x, float* y_z, const float* gamma, const float* beta, float* out, float* stats, int rows, int d, float rate,
                   unsigned site, const void* st, int grid, hipStream_t s);
int skf_ln_bwd_any(const float* dout, const float* z, const float* stats, const float* gamma, float* dz, float* dy, float* part, int rows, int d,
                   float rate, unsigned site, const void* st, int grid, hipStream_t s);
int skf_expander_bwd_any(const float* dpre, const float* emb, const float* w, int B, int L, int d, float* demb, int demb_accumulate, float* p1,
                         float* p2, hipStream_t s);
namespace {
int grid_for_rows(int rows) { int g = skf_cdiv(rows, 4); return g > kMaxGrid ? kMaxGrid : g; }

}  // namespace

// =========================================================================== C ABI
extern "C" int skf_embed_fwd(const long long* tokens, int tok_ld, int B, int L, const float* table, int vocab, int d,
                             const float* pos, float* out, float rate, unsigned site, const void* step_state,
                             skf_stream_t stream) {
  SKF_CHECK_ARG(tokens && table && pos && out, "null operand");
  SKF_CHECK_ARG((d & 1) == 0, "d_model must be even");
  SKF_CHECK_ARG(rate == 0.f || step_state, "dropout needs the step state");
  const int rows = B * L;
  SkfProfScope ps((hipStream_t)stream, "embed_fwd", 0.0, 8.0 * rows * d);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for_rows(rows)), dim3(256), 0, (hipStream_t)stream, tokens, tok_ld, L,
                     rows, table, vocab, d, pos, out, rate, site, (const SkfStepState*)step_state);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_padding_mask(const long long* tokens, int tok_ld, int B, int L, unsigned char* out,
                                skf_stream_t stream) {
  SKF_CHECK_ARG(tokens && out && B > 0 && L > 0, "bad argument");
  int grid = skf_cdiv(B * L, 256); if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(padding_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, tokens, tok_ld, B, L, out);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

// The three input copies of a step (inputs, targets with their own pitch, labels or zeros) as ONE launch instead of three
// copy-engine kernels (6 us each, back to back at the head of every step); 4-byte words, internal to the library (skf_model.hip).
// mask_L > 0 (token mode: 8-byte ids): the two padding masks of builders/utils.py:35-43 ride along - emask[b][t] = inp[b][t] == 0 (t < mask_L),
// dmask[b][t] = tar[b][t] == 0 (t < mask_L - 1): the side stream's sample ordering no longer waits for two mask launches of its own
__global__ void stage_inputs_kernel(const unsigned* __restrict__ inp, unsigned* __restrict__ dinp, const unsigned* __restrict__ tar,
                                    unsigned* __restrict__ dtar, int row_w, int src_row_w, int copy_w, int batch,
                                    const unsigned* __restrict__ labels, unsigned* __restrict__ dlabels,
                                    unsigned char* __restrict__ emask, unsigned char* __restrict__ dmask, int mask_L) {
  const int n_inp = row_w * batch, n_tar = copy_w * batch, n_lab = 2 * batch;
  const int n_em = mask_L * batch, n_dm = (mask_L > 0 ? mask_L - 1 : 0) * batch;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_inp + n_tar + n_lab + n_em + n_dm; e += gridDim.x * blockDim.x) {
    if (e < n_inp) dinp[e] = inp[e];
    else if (e < n_inp + n_tar) { const int t = e - n_inp, b = t / copy_w, w = t % copy_w; dtar[(size_t)b * row_w + w] = tar[(size_t)b * src_row_w + w]; }
    else if (e < n_inp + n_tar + n_lab) { const int t = e - n_inp - n_tar; dlabels[t] = labels ? labels[t] : 0u; }
    else if (e < n_inp + n_tar + n_lab + n_em) {
      const int t = e - n_inp - n_tar - n_lab, b = t / mask_L, k = t % mask_L;
      const size_t w = (size_t)b * row_w + 2 * k;
      emask[t] = (inp[w] | inp[w + 1]) == 0u ? 1 : 0;
    } else {
      const int t = e - n_inp - n_tar - n_lab - n_em, b = t / (mask_L - 1), k = t % (mask_L - 1);
      const size_t w = (size_t)b * src_row_w + 2 * k;
      dmask[t] = (tar[w] | tar[w + 1]) == 0u ? 1 : 0;
    }
  }
}
// row / src_row / copy in bytes (multiples of 4); returns SKF_EUNSUPPORTED when an operand is not 4-byte aligned (caller copies)
int skf_stage_inputs_launch(const void* inp, void* dinp, const void* tar, void* dtar, size_t row, size_t src_row, size_t copy, int batch,
                            const void* labels, void* dlabels, hipStream_t st, unsigned char* emask, unsigned char* dmask, int mask_L) {
  if (((uintptr_t)inp | (uintptr_t)dinp | (uintptr_t)tar | (uintptr_t)dtar | (uintptr_t)labels | (uintptr_t)dlabels | row | src_row | copy) & 3)
    return SKF_EUNSUPPORTED;
  if ((double)(row + copy) * batch / 4 + 2.0 * batch + 2.0 * mask_L * batch >= 2147483648.0) return SKF_EUNSUPPORTED;
  // the masks read whole 8-byte ids: mask_L of them per row of inp, mask_L - 1 per row of tar
  if (mask_L > 0 && (!emask || !dmask || (size_t)mask_L * 8 > row || (size_t)(mask_L - 1) * 8 > src_row)) mask_L = 0;
  const int total = (int)((row + copy) / 4) * batch + 2 * batch + (mask_L > 0 ? (2 * mask_L - 1) * batch : 0);
  int grid = skf_cdiv(total, 256); if (grid > 512) grid = 512;
  SKF_LAUNCH_TAIL(stage_inputs_kernel, dim3(grid), dim3(256), 0, st, (const unsigned*)inp, (unsigned*)dinp, (const unsigned*)tar,
                     (unsigned*)dtar, (int)(row / 4), (int)(src_row / 4), (int)(copy / 4), batch, (const unsigned*)labels, (unsigned*)dlabels,
                     emask, dmask, mask_L);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_embed_bwd(const long long* tokens, int tok_ld, int B, int L, const float* dx, int vocab, int d,
                             float* dtable, float rate, unsigned site, const void* step_state, skf_stream_t stream) {
  SKF_CHECK_ARG(tokens && dx && dtable, "null operand");
  SKF_CHECK_ARG((d & 1) == 0, "d_model must be even");
  const int rows = B * L;
  int grid = skf_cdiv(rows, 256);
  SkfProfScope ps((hipStream_t)stream, "embed_bwd", 0.0, 8.0 * rows * d);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, tokens, tok_ld, L, rows, dx, vocab,
                     d, dtable, rate, site, (const SkfStepState*)step_state);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

struct EmbWs { int* hdr; EmbChunk* chunks; int* order; int* done; float* partial; size_t bytes; };
static EmbWs emb_ws_layout(void* ws, int B, int L, int vocab) {
  const size_t rows = (size_t)B * L, maxc = (size_t)vocab + rows / kEmbChunk + 1;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  EmbWs w;
  char* base = (char*)ws;
  w.hdr = (int*)(base + off); off += 256;
  w.chunks = (EmbChunk*)(base + off); off += up(maxc * sizeof(EmbChunk));
  w.order = (int*)(base + off); off += up(rows * sizeof(int));
  w.done = (int*)(base + off); off += up((size_t)vocab * sizeof(int));
  w.partial = (float*)(base + off); off += up(maxc * kEmbMaxD * sizeof(float));
  w.bytes = off;
  return w;
}
extern "C" size_t skf_embed_sort_workspace_bytes(int B, int L, int vocab) { return emb_ws_layout(nullptr, B, L, vocab).bytes; }
extern "C" int skf_embed_sort(const long long* tokens, int tok_ld, int B, int L, int vocab, float* zero_table, int d,
                              void* ws, size_t ws_bytes, skf_stream_t stream) {
  (void)zero_table; (void)d;              // (the gradient kernel no longer accumulates into the table: nothing to pre-zero)
  SKF_CHECK_ARG(tokens && ws, "null operand");
  SKF_CHECK_ARG(ws_bytes >= skf_embed_sort_workspace_bytes(B, L, vocab) && (((uintptr_t)ws) & 15) == 0, "workspace too small or misaligned");
  SKF_CHECK_ARG(vocab > 0 && vocab <= 12288, "vocabulary does not fit the sort kernel's LDS tables");
  const int rows = B * L;
  const EmbWs w = emb_ws_layout(ws, B, L, vocab);
  SKF_CHECK_ARG(rows < (1 << kEmbCursorBits), "too many token positions for the scatter's cursor word");
  const bool per_wave = vocab <= kEmbOrderedVocab;      // 16 histograms fit in LDS; otherwise one table and a one-wave scatter
  const size_t smem = ((size_t)3 * vocab + 1 + (per_wave ? (size_t)16 * vocab : 0)) * sizeof(int);
  static SkfOncePerDevice attr_done;
  if (attr_done.needed()) {
    SKF_HIP(hipFuncSetAttribute((const void*)embed_sort_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (3 * 12288 + 1) * 4));
    SKF_HIP(hipFuncSetAttribute((const void*)embed_sort_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (19 * kEmbOrderedVocab + 1) * 4));
    attr_done.mark();
  }
  SkfProfScope ps((hipStream_t)stream, "embed_sort", 0.0, 12.0 * rows);
  if (per_wave)
    hipLaunchKernelGGL(embed_sort_kernel<16>, dim3(1), dim3(1024), smem, (hipStream_t)stream, tokens, tok_ld, L, rows, vocab, w.hdr, w.chunks,
                       w.order, w.done);
  else
    hipLaunchKernelGGL(embed_sort_kernel<1>, dim3(1), dim3(1024), smem, (hipStream_t)stream, tokens, tok_ld, L, rows, vocab, w.hdr, w.chunks,
                       w.order, w.done);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
static int embed_bwd_sorted_launch(const void* ws, int B, int L, const void* dx, int dx_bf16, int vocab, int d, float* dtable,
                                   float rate, unsigned site, const void* step_state, skf_stream_t stream) {
  SKF_CHECK_ARG(ws && dx && dtable, "null operand");
  SKF_CHECK_ARG((d & 1) == 0 && d <= kEmbMaxD, "d_model must be even and <= 512");
  SKF_CHECK_ARG(rate == 0.f || step_state, "dropout needs the step state");
  const int rows = B * L;
  const size_t maxc = (size_t)vocab + (size_t)rows / kEmbChunk + 1;
  const EmbWs w = emb_ws_layout(const_cast<void*>(ws), B, L, vocab);
  const int* hdr = w.hdr;
  const EmbChunk* chunks = w.chunks;
  const int* order = w.order;
  SkfProfScope ps((hipStream_t)stream, dx_bf16 ? "embed_bwd_sorted_bf16" : "embed_bwd_sorted", 0.0,
                  (dx_bf16 ? 2.0 : 4.0) * rows * d + 4.0 * (double)vocab * d);
  if (dx_bf16)
    SKF_LAUNCH_TAIL(embed_bwd_sorted_kernel<skf_bf16>, dim3((unsigned)maxc), dim3(256), (size_t)3 * d * sizeof(float), (hipStream_t)stream,
                       hdr, chunks, order, (const skf_bf16*)dx, d, dtable, rate, site, (const SkfStepState*)step_state, w.partial, w.done);
  else
    SKF_LAUNCH_TAIL(embed_bwd_sorted_kernel<float>, dim3((unsigned)maxc), dim3(256), (size_t)3 * d * sizeof(float), (hipStream_t)stream,
                       hdr, chunks, order, (const float*)dx, d, dtable, rate, site, (const SkfStepState*)step_state, w.partial, w.done);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
extern "C" int skf_embed_bwd_sorted(const void* ws, int B, int L, const float* dx, int vocab, int d, float* dtable, float rate,
                                    unsigned site, const void* step_state, skf_stream_t stream) {
  return embed_bwd_sorted_launch(ws, B, L, dx, 0, vocab, d, dtable, rate, site, step_state, stream);
}
extern "C" int skf_embed_bwd_sorted_bf16(const void* ws, int B, int L, const void* dx, int vocab, int d, float* dtable, float rate,
                                         unsigned site, const void* step_state, skf_stream_t stream) {
  return embed_bwd_sorted_launch(ws, B, L, dx, 1, vocab, d, dtable, rate, site, step_state, stream);
}

extern "C" int skf_layernorm_residual_fwd(const float* x, float* y_inout_z, const float* gamma, const float* beta,
                                          float* out, float* stats, int rows, int d, float rate, unsigned site,
                                          const void* step_state, skf_stream_t stream) {
  SKF_CHECK_ARG(x && y_inout_z && gamma && beta && out && stats, "null operand");
  SKF_CHECK_ARG(rate == 0.f || step_state, "dropout needs the step state");
  const SkfStepState* st = (const SkfStepState*)step_state;
  dim3 grid(grid_for_rows(rows)), block(256);
  hipStream_t s = (hipStream_t)stream;
  SkfProfScope ps(s, "ln_fwd", 0.0, 16.0 * rows * d);
  static const bool v4 = !(skf_knob("SKF_LN_V4") && skf_knob("SKF_LN_V4")[0] == '0');
  const bool al = ((((uintptr_t)x | (uintptr_t)y_inout_z | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0);
  if (v4 && al && (d == 64 || d == 128 || d == 256)) {
    const int rpi = (64 / (d / 4)) * SKF_LN_UR * 4;               // rows per workgroup iteration
    int g = skf_cdiv(rows, rpi); if (g > kMaxGrid) g = kMaxGrid;
    if (d == 64) hipLaunchKernelGGL(ln_fwd_v4_kernel<16>, dim3(g), block, 0, s, x, y_inout_z, gamma, beta, out, stats, rows, rate, site, st);
    else if (d == 128) hipLaunchKernelGGL(ln_fwd_v4_kernel<32>, dim3(g), block, 0, s, x, y_inout_z, gamma, beta, out, stats, rows, rate, site, st);
    else hipLaunchKernelGGL(ln_fwd_v4_kernel<64>, dim3(g), block, 0, s, x, y_inout_z, gamma, beta, out, stats, rows, rate, site, st);
    SKF_LAUNCH_CHECK();
    return SKF_OK;
  }
  switch (d) {
    case 128: hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, block, 0, s, x, y_inout_z, gamma, beta, out, stats, rows, rate, site, st); break;
    case 256: hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, block, 0, s, x, y_inout_z, gamma, beta, out, stats, rows, rate, site, st); break;
    case 512: hipLaunchKernelGGL(ln_fwd_kernel<8>, grid, block, 0, s, x, y_inout_z, gamma, beta, out, stats, rows, rate, site, st); break;
    case 64:  hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, block, 0, s, x, y_inout_z, gamma, beta, out, stats, rows, rate, site, st); break;
    default: return skf_ln_fwd_any(x, y_inout_z, gamma, beta, out, stats, rows, d, rate, site, st, grid_for_rows(rows), s);   // any other width (skf_generic.hip)
  }
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" size_t skf_layernorm_bwd_workspace_bytes(int rows, int d) {
  return (size_t)(grid_for_rows(rows) > kLnBwdGrid ? kLnBwdGrid : grid_for_rows(rows)) * 2 * d * sizeof(float);
}

extern "C" int skf_layernorm_residual_bwd(const float* dout, const float* z, const float* stats, const float* gamma,
                                          float* dz, float* dy, float* dgamma, float* dbeta, int rows, int d, float rate,
                                          unsigned site, const void* step_state, void* workspace, size_t workspace_bytes,
                                          skf_stream_t stream) {
  return skf_layernorm_residual_bwd_rows(dout, z, stats, gamma, dz, dy, dgamma, dbeta, rows, d, rate, site, step_state, workspace,
                                         workspace_bytes, nullptr, 0, stream);
}

extern "C" int skf_layernorm_residual_bwd_rows(const float* dout, const float* z, const float* stats, const float* gamma,
                                               float* dz, float* dy, float* dgamma, float* dbeta, int rows, int d, float rate,
                                               unsigned site, const void* step_state, void* workspace, size_t workspace_bytes,
                                               const int* live_len, int rows_per_sample, skf_stream_t stream) {
  SKF_CHECK_ARG(!live_len || (rows_per_sample > 0 && rows % rows_per_sample == 0), "live_len needs rows = B * rows_per_sample");
  SKF_CHECK_ARG(dout && z && stats && gamma && dz, "null operand");
  SKF_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "dgamma and dbeta must both be given or both be NULL");
  SKF_CHECK_ARG(workspace && workspace_bytes >= skf_layernorm_bwd_workspace_bytes(rows, d), "workspace too small");
  SKF_CHECK_ARG(rate == 0.f || (step_state && dy), "dropout needs the step state and a dy buffer");
  SKF_CHECK_ARG(rate > 0.f || !dy || dy == dz || step_state, "a separate dy buffer needs the step state");
  const SkfStepState* st = (const SkfStepState*)step_state;
  int g = grid_for_rows(rows); if (g > kLnBwdGrid) g = kLnBwdGrid;
  dim3 grid(g), block(256);
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* dyp = (dy && (rate > 0.f || dy != dz)) ? dy : nullptr;   // rate 0 with a separate dy buffer: dy = dz
  SkfProfScope ps(s, "ln_bwd", 0.0, (rate > 0.f ? 16.0 : 12.0) * rows * d);
  static const bool v4 = !(skf_knob("SKF_LN_V4") && skf_knob("SKF_LN_V4")[0] == '0');
  const bool al = ((((uintptr_t)dout | (uintptr_t)z | (uintptr_t)dz | (uintptr_t)dyp | (uintptr_t)gamma) & 15) == 0) &&
                  (double)rows * d * 4 < 2147483648.0;             // (the v4 kernel stores through 32-bit buffer offsets)
  const dim3 block4(SKF_LN_BWD_THREADS);
  if (v4 && al && d == 64) hipLaunchKernelGGL(ln_bwd_v4_kernel<16>, grid, block4, 0, s, dout, z, stats, gamma, dz, dyp, part, rows, rate, site, st, live_len, rows_per_sample);
  else if (v4 && al && d == 128) hipLaunchKernelGGL(ln_bwd_v4_kernel<32>, grid, block4, 0, s, dout, z, stats, gamma, dz, dyp, part, rows, rate, site, st, live_len, rows_per_sample);
  else if (v4 && al && d == 256) hipLaunchKernelGGL(ln_bwd_v4_kernel<64>, grid, block4, 0, s, dout, z, stats, gamma, dz, dyp, part, rows, rate, site, st, live_len, rows_per_sample);
  else
  switch (d) {
    case 128: hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, block, 0, s, dout, z, stats, gamma, dz, dyp, part, rows, rate, site, st); break;
    case 256: hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, block, 0, s, dout, z, stats, gamma, dz, dyp, part, rows, rate, site, st); break;
    case 512: hipLaunchKernelGGL(ln_bwd_kernel<8>, grid, block, 0, s, dout, z, stats, gamma, dz, dyp, part, rows, rate, site, st); break;
    case 64:  hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, block, 0, s, dout, z, stats, gamma, dz, dyp, part, rows, rate, site, st); break;
    default: { const int rc = skf_ln_bwd_any(dout, z, stats, gamma, dz, dyp, part, rows, d, rate, site, st, g, s); if (rc) return rc; }   // skf_generic.hip
  }
  SKF_LAUNCH_CHECK();
  // part is [g][2][d] : columns 0..d-1 = dgamma, d..2d-1 = dbeta
  if (!dgamma) return SKF_OK;   // the caller sums the g = workspace_bytes/(8d) partial rows itself (batched with other reductions)
  if (dbeta == dgamma + d) {   // adjacent in the flat gradient buffer: one launch over 2d columns
    hipLaunchKernelGGL(colsum_kernel, dim3(skf_cdiv(2 * d, 64)), dim3(1024), 0, s, part, g, 2 * d, 2 * d, dgamma, 0);
  } else {
    hipLaunchKernelGGL(colsum_kernel, dim3(skf_cdiv(d, 64)), dim3(1024), 0, s, part, g, 2 * d, d, dgamma, 0);
    hipLaunchKernelGGL(colsum_kernel, dim3(skf_cdiv(d, 64)), dim3(1024), 0, s, part + d, g, 2 * d, d, dbeta, 0);
  }
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_colsum(const float* in, int nrows, int ld, int ncols, float* out, int accumulate, skf_stream_t stream) {
  SKF_CHECK_ARG(in && out && nrows > 0 && ncols > 0, "bad argument");
  hipLaunchKernelGGL(colsum_kernel, dim3(skf_cdiv(ncols, 64)), dim3(1024), 0, (hipStream_t)stream, in, nrows, ld, ncols, out, accumulate);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_softmax_ce(float* logits, int ld, int rows, int ncls, const long long* target, int tgt_ld,
                              int tgt_cols, int tgt_off, int mask_pad, float scale, float* row_loss, float* row_hit,
                              float* probs_out, int write_grad, skf_stream_t stream) {
  SKF_CHECK_ARG(logits && target && row_loss && row_hit, "null operand");
  SKF_CHECK_ARG(rows > 0 && ncls > 0 && tgt_cols > 0, "empty problem");
  SkfProfScope ps((hipStream_t)stream, "softmax_ce", 0.0, 8.0 * rows * ncls);
  const bool al = (ncls & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)logits & 15) == 0 && (!probs_out || ((uintptr_t)probs_out & 15) == 0);
  const bool vec = al && ncls <= 2048;
  if (al && ncls > 2048 && ncls <= 16384) {        // wide rows: one workgroup per row, the row in registers
    int g = rows < 8192 ? rows : 8192;
#define SKF_CE_WIDE(NV4) hipLaunchKernelGGL(softmax_ce_wide_kernel<NV4>, dim3(g), dim3(256), 0, (hipStream_t)stream, \
    logits, ld, rows, ncls, target, tgt_ld, tgt_cols, tgt_off, mask_pad, scale, row_loss, row_hit, probs_out, write_grad)
    if (ncls <= 4096) SKF_CE_WIDE(4); else if (ncls <= 8192) SKF_CE_WIDE(8); else if (ncls <= 12288) SKF_CE_WIDE(12); else SKF_CE_WIDE(16);
#undef SKF_CE_WIDE
    SKF_LAUNCH_CHECK();
    return SKF_OK;
  }
#define SKF_CE_GO(NV4) hipLaunchKernelGGL(softmax_ce_reg_kernel<NV4>, dim3(grid_for_rows(rows)), dim3(256), 0, (hipStream_t)stream, \
    logits, ld, rows, ncls, target, tgt_ld, tgt_cols, tgt_off, mask_pad, scale, row_loss, row_hit, probs_out, write_grad)
  if (vec && ncls <= 256) SKF_CE_GO(1);
  else if (vec && ncls <= 512) SKF_CE_GO(2);
  else if (vec && ncls <= 1024) SKF_CE_GO(4);
  else if (vec) SKF_CE_GO(8);
  else
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(grid_for_rows(rows)), dim3(256), 0, (hipStream_t)stream, logits, ld, rows, ncls,
                     target, tgt_ld, tgt_cols, tgt_off, mask_pad, scale, row_loss, row_hit, probs_out, write_grad);
#undef SKF_CE_GO
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_metrics_update(const float* recon_loss, const float* recon_hit, int recon_rows, float recon_weight,
                                  const float* class_loss, const float* class_hit, int class_rows, float class_weight,
                                  const float* recon_scalar, float* metrics, skf_stream_t stream) {
  SKF_CHECK_ARG(metrics, "null metrics");
  hipLaunchKernelGGL(metrics_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, recon_loss, recon_hit, recon_rows,
                     recon_weight, class_loss, class_hit, class_rows, class_weight, recon_scalar, metrics);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_pool_fwd(const float* u, const float* Vw, const float* x, int B, int L, int U, int d, float* a_out,
                            float* emb, skf_stream_t stream) {
  SKF_CHECK_ARG(u && Vw && x && a_out && emb, "null operand");
  SkfProfScope ps((hipStream_t)stream, "pool_fwd", 0.0, 4.0 * B * L * (U + d));
  SKF_CHECK_ARG(L <= 1024 && (U & 3) == 0 && (d & 3) == 0 && d <= 4096 && U <= 4096, "pool: need L <= 1024, U % 4 == d % 4 == 0, U,d <= 4096");
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(B), dim3(1024), (4096 + L + 16) * sizeof(float), (hipStream_t)stream, u, Vw, x, L, U, d, a_out, emb);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_pool_bwd(float* u_inout_dpre, const float* Vw, const float* x, const float* a, const float* demb,
                            int B, int L, int U, int d, float* dx, float* dV, void* workspace, size_t workspace_bytes,
                            skf_stream_t stream) {
  SKF_CHECK_ARG(u_inout_dpre && Vw && x && a && demb && dx && dV, "null operand");
  SKF_CHECK_ARG(workspace && workspace_bytes >= (size_t)B * U * sizeof(float), "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  const int rc = skf_pool_bwd_partials(u_inout_dpre, Vw, x, a, demb, B, L, U, d, dx, part, s);
  if (rc != SKF_OK) return rc;
  hipLaunchKernelGGL(colsum_kernel, dim3(skf_cdiv(U, 64)), dim3(1024), 0, s, part, B, U, U, dV, 0);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
// internal (skf_model.hip): the launch without the column sum - dV_part[B][U] is left for a batched reduction (a "slab" of B splits of a 1 x U matrix)
int skf_pool_bwd_partials(float* u_inout_dpre, const float* Vw, const float* x, const float* a, const float* demb, int B, int L, int U, int d,
                          float* dx, float* dV_part, hipStream_t s) {
  SkfProfScope ps(s, "pool_bwd", 0.0, 8.0 * B * L * (U + d));
  SKF_CHECK_ARG(L <= 1024 && (U & 3) == 0 && (d & 3) == 0 && d <= 4096 && U <= 4096, "pool: need L <= 1024, U % 4 == d % 4 == 0, U,d <= 4096");
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(B), dim3(1024), (4096 + d + 2 * L + 16) * sizeof(float), s, u_inout_dpre, Vw, x, a, demb, L, U, d, dx, dV_part);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_expander_fwd(const float* emb, const float* w, const float* bias, int B, int L, int d, float* pre,
                                skf_stream_t stream) {
  SKF_CHECK_ARG(emb && w && bias && pre, "null operand");
  const size_t total = (size_t)B * L * d;
  int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
  SkfProfScope ps((hipStream_t)stream, "expander_fwd", 0.0, 4.0 * total);
  SKF_LAUNCH_TAIL(expander_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, emb, w, bias, B, L, d, pre);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_expander_bwd(const float* dpre, const float* emb, const float* w, int B, int L, int d, float* demb,
                                int demb_accumulate, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                                skf_stream_t stream) {
  SKF_CHECK_ARG(dpre && emb && w && demb && dw && dbias, "null operand");
  SKF_CHECK_ARG(workspace && workspace_bytes >= (size_t)2 * B * L * sizeof(float), "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* p1 = (float*)workspace;
  float* p2 = p1 + (size_t)B * L;
  const int rc = skf_expander_bwd_partials(dpre, emb, w, B, L, d, demb, demb_accumulate, p1, p2, s);
  if (rc != SKF_OK) return rc;
  hipLaunchKernelGGL(colsum_kernel, dim3(skf_cdiv(L, 64)), dim3(1024), 0, s, p1, B, L, L, dw, 0);
  hipLaunchKernelGGL(colsum_kernel, dim3(skf_cdiv(L, 64)), dim3(1024), 0, s, p2, B, L, L, dbias, 0);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
// internal (skf_model.hip): the launch without the two column sums - p1 / p2 [B][L] (dw / dbias partials) are left for a batched reduction
int skf_expander_bwd_partials(const float* dpre, const float* emb, const float* w, int B, int L, int d, float* demb, int demb_accumulate,
                              float* p1, float* p2, hipStream_t s) {
  SkfProfScope ps(s, "expander_bwd", 0.0, 8.0 * B * L * d);
  switch (d) {
    case 64:  hipLaunchKernelGGL(expander_bwd_kernel<1>, dim3(B), dim3(1024), 0, s, dpre, emb, w, L, demb, demb_accumulate, p1, p2); break;
    case 128: hipLaunchKernelGGL(expander_bwd_kernel<2>, dim3(B), dim3(1024), 0, s, dpre, emb, w, L, demb, demb_accumulate, p1, p2); break;
    case 256: hipLaunchKernelGGL(expander_bwd_kernel<4>, dim3(B), dim3(1024), 0, s, dpre, emb, w, L, demb, demb_accumulate, p1, p2); break;
    case 512: hipLaunchKernelGGL(expander_bwd_kernel<8>, dim3(B), dim3(1024), 0, s, dpre, emb, w, L, demb, demb_accumulate, p1, p2); break;
    default: { const int rc = skf_expander_bwd_any(dpre, emb, w, B, L, d, demb, demb_accumulate, p1, p2, s); if (rc) return rc; }   // skf_generic.hip
  }
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_dropout(const float* x, float* y, size_t n, float rate, unsigned site, const void* step_state,
                           skf_stream_t stream) {
  SKF_CHECK_ARG(x && y, "null operand");
  SKF_CHECK_ARG(rate >= 0.f && rate < 1.f, "rate out of range");
  SKF_CHECK_ARG(rate == 0.f || step_state, "dropout needs the step state");
  if (n == 0 || (rate == 0.f && x == y)) return SKF_OK;
  size_t blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n, rate, site,
                     (const SkfStepState*)step_state);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_dropout_keep_mask(unsigned drop_key, unsigned site, float rate, size_t n, unsigned char* out_host) {
  SKF_CHECK_ARG(out_host, "null output");
  const uint32_t sk = skf_site_key(drop_key, site), th = skf_drop_thresh(rate);
  for (size_t i = 0; i < n; ++i) out_host[i] = skf_keep(sk, (uint32_t)i, th) ? 1 : 0;
  return SKF_OK;
}
