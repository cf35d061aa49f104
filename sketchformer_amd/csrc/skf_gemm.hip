// fp32 GEMM family on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak on gfx950).
//
// Replaces the tf.keras.layers.Dense forward (x.W+b, optional relu/tanh) and the
// dgrad / wgrad+bias-grad that tf.GradientTape derives for it
// (reference: builders/layers/transformer.py:154-158,196-197, models/sketchformer.py:85-104,347).
//
//   C[M,N] (+)= opA(A)[M,K] . opB(B)[K,N]  (+ bias[N]) (act) (relu-grad mask)
//
//   a_kcontig=1 : A is [M][K], K contiguous   (forward x, dgrad dY)
//   a_kcontig=0 : A is [K][M], M contiguous   (wgrad: X viewed as X^T)
//   b_kcontig=0 : B is [K][N], N contiguous   (forward W, wgrad dY)
//   b_kcontig=1 : B is [N][K], K contiguous   (dgrad: W viewed as W^T)
//
// Work decomposition: 256-thread workgroups (4 waves, one per SIMD), BMxBN output
// tile, BK=32 k-slab double-buffered in LDS with register-staged prefetch (global
// loads for slab t+1 are issued before the MFMAs of slab t and written to LDS
// after them).  Both operands are stored k-major in LDS (S[k][mn]) so each MFMA
// fragment read is one conflict-free ds_read_b32 per lane; k-contiguous operands
// are transposed on the LDS store with a +1 pad that makes the 4 scalar stores
// conflict-free.  blockIdx is remapped so that consecutive logical tiles (which
// share an A row-panel) land on the same XCD / L2.
#include <stdlib.h>
#include <string>
#include "skf_common.h"
#include "skf_gemm_params.h"

namespace {

constexpr int BK = 32;


template <int BX, bool KC>
struct TileLD { static constexpr int value = KC ? BX + 1 : BX + 4; };

// ---- global -> register staging of one BX x BK operand slab.
// VEC: branch-free path (clamped 16-byte loads + select) - needs 16-byte aligned rows and a
// contiguous extent that is a multiple of 4, so a float4 is either fully inside or fully outside.
// Keeping the loads unconditional lets the compiler leave them in flight across the MFMA loop.
template <int BX, bool KC, bool VEC>
__device__ __forceinline__ void load_slab(const float* __restrict__ P, int ld, int mn0, int mn_max,
                                          int k0, int k_end, float4 (&r)[BX * BK / 1024]) {
  const int t = threadIdx.x;
  constexpr int NV = BX * BK / 1024;
#pragma unroll
  for (int p = 0; p < NV; ++p) {
    int mn, k;
    if (KC) { mn = mn0 + (t >> 3) + p * 32; k = k0 + (t & 7) * 4; }
    else    { k = k0 + t / (BX / 4) + p * (1024 / BX); mn = mn0 + (t % (BX / 4)) * 4; }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VEC) {
      const bool ok = KC ? (mn < mn_max && k < k_end) : (k < k_end && mn < mn_max);
      const int mc = ok ? mn : 0, kc = ok ? k : 0;
      const float* src = KC ? P + (size_t)mc * ld + kc : P + (size_t)kc * ld + mc;
      v = *reinterpret_cast<const float4*>(src);   // zeroing of out-of-range vectors happens in store_slab
    } else if (KC) {
      if (mn < mn_max) {
        const float* src = P + (size_t)mn * ld + k;
        if (k + 0 < k_end) v.x = src[0];
        if (k + 1 < k_end) v.y = src[1];
        if (k + 2 < k_end) v.z = src[2];
        if (k + 3 < k_end) v.w = src[3];
      }
    } else {
      if (k < k_end) {
        const float* src = P + (size_t)k * ld + mn;
        if (mn + 0 < mn_max) v.x = src[0];
        if (mn + 1 < mn_max) v.y = src[1];
        if (mn + 2 < mn_max) v.z = src[2];
        if (mn + 3 < mn_max) v.w = src[3];
      }
    }
    r[p] = v;
  }
}

// ---- register -> LDS (k-major image S[k][mn]).  On the VEC path the out-of-range select is applied here,
// after the MFMAs of the previous slab, so the global loads stay in flight across them.
template <int BX, bool KC, bool VEC>
__device__ __forceinline__ void store_slab(float* __restrict__ S, const float4 (&r)[BX * BK / 1024], int mn0, int mn_max,
                                           int k0, int k_end) {
  const int t = threadIdx.x;
  constexpr int LD = TileLD<BX, KC>::value;
  constexpr int NV = BX * BK / 1024;
#pragma unroll
  for (int p = 0; p < NV; ++p) {
    float4 v = r[p];
    if (VEC) {
      int mn, k;
      if (KC) { mn = mn0 + (t >> 3) + p * 32; k = k0 + (t & 7) * 4; }
      else    { k = k0 + t / (BX / 4) + p * (1024 / BX); mn = mn0 + (t % (BX / 4)) * 4; }
      const bool ok = mn < mn_max && k < k_end;
      v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    }
    if (KC) {
      const int row = (t >> 3) + p * 32, kq = (t & 7) * 4;
      S[(kq + 0) * LD + row] = v.x;
      S[(kq + 1) * LD + row] = v.y;
      S[(kq + 2) * LD + row] = v.z;
      S[(kq + 3) * LD + row] = v.w;
    } else {
      const int kr = t / (BX / 4) + p * (1024 / BX), c = (t % (BX / 4)) * 4;
      *reinterpret_cast<float4*>(&S[kr * LD + c]) = v;
    }
  }
}

template <int BM, int BN, int WGM, bool A_KC, bool B_KC, bool SPLITK, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  constexpr int WGN = 4 / WGM;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDA_S = TileLD<BM, A_KC>::value, LDB_S = TileLD<BN, B_KC>::value;
  constexpr int A_SZ = BK * LDA_S, B_SZ = BK * LDB_S;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [2][BK][LDA_S]
  float* Bs = smem + 2 * A_SZ;       // [2][BK][LDB_S]

  // XCD-aware bijective remap of the 1-D grid (block b runs on XCD b%8).
  const int nwg = p.tiles_m * p.tiles_n;
  const int orig = blockIdx.x;
  const int xcd = orig & 7, q = nwg >> 3, rr = nwg & 7;
  const int logical = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
  const int tile_m = logical / p.tiles_n, tile_n = logical % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  if constexpr (!SPLITK && A_KC) {
    // Row-block list of 16-row blocks (skf_row_blocks_build): a tile none of whose blocks is live has all-zero rows of A -
    // its C rows are zeros (stored unless the call accumulates), nothing is loaded or multiplied
    if (p.row_blocks && p.row_block_rows == 16) {
      const int nb = p.row_blocks[1];
      const int* flags = p.row_blocks + 2 + nb;
      int live = 0;
#pragma unroll
      for (int j = 0; j < BM / 16; ++j) { const int k = tile_m * (BM / 16) + j; live |= k < nb ? flags[k] : 0; }
      if (!live) {
        if (!p.accumulate)
          for (int e = threadIdx.x; e < BM * BN; e += 256) {
            const int m = m0 + e / BN, n = n0 + e % BN;
            if (m < p.M && n < p.N) p.C[(size_t)m * p.ldc + n] = 0.f;
          }
        return;
      }
    }
  }
  int kb = 0, ke = p.K;
  if (SPLITK) { kb = blockIdx.z * p.k_chunk; ke = min(p.K, kb + p.k_chunk); }
  const int nk = (ke - kb + BK - 1) / BK;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[BM * BK / 1024], rb[BN * BK / 1024];
  float colsum = 0.f;
  const bool do_colsum = SPLITK && p.colsum_slab != nullptr && tile_m == 0;

  if (nk > 0) {
    load_slab<BM, A_KC, VEC>(p.A, p.lda, m0, p.M, kb, ke, ra);
    load_slab<BN, B_KC, VEC>(p.B, p.ldb, n0, p.N, kb, ke, rb);
    store_slab<BM, A_KC, VEC>(As, ra, m0, p.M, kb, ke);
    store_slab<BN, B_KC, VEC>(Bs, rb, n0, p.N, kb, ke);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more && p.ablate != 3) {
      load_slab<BM, A_KC, VEC>(p.A, p.lda, m0, p.M, kb + (kt + 1) * BK, ke, ra);
      load_slab<BN, B_KC, VEC>(p.B, p.ldb, n0, p.N, kb + (kt + 1) * BK, ke, rb);
    }
    const float* Ac = As + cur * A_SZ;
    const float* Bc = Bs + cur * B_SZ;
    if (p.ablate != 1)
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
      const int k = 2 * kp + lhi;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Ac[k * LDA_S + wm0 + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bc[k * LDB_S + wn0 + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (do_colsum && threadIdx.x < BN) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) colsum += Bc[k * LDB_S + threadIdx.x];
    }
    if (more) {
      store_slab<BM, A_KC, VEC>(As + (cur ^ 1) * A_SZ, ra, m0, p.M, kb + (kt + 1) * BK, ke);
      store_slab<BN, B_KC, VEC>(Bs + (cur ^ 1) * B_SZ, rb, n0, p.N, kb + (kt + 1) * BK, ke);
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (SPLITK) {
    float* slab = p.slab + (size_t)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (row < p.M && col < p.N) slab[(size_t)row * p.N + col] = acc[i][j][r];
        }
      }
    if (do_colsum && threadIdx.x < BN && n0 + (int)threadIdx.x < p.N)
      p.colsum_slab[(size_t)blockIdx.z * p.N + n0 + threadIdx.x] = colsum;
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn0 + j * 32 + l31;
      if (col >= p.N) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row >= p.M) continue;
        float v = acc[i][j][r] + bv;
        if (p.ablate == 2 && v != 12345.678f) continue;
        if (p.act == 1) v = fmaxf(v, 0.f);
        else if (p.act == 2) v = tanhf(v);
        if (p.relu_src && !(p.relu_src[(size_t)row * p.ld_relu + col] > 0.f)) v = 0.f;
        float* dst = p.C + (size_t)row * p.ldc + col;
        if (p.accumulate) v += *dst;
        *dst = v;
      }
    }
}

// C[m][n] (=|+=) sum_z slab[z][m][n];  bias_grad[n] = sum_z colsum_slab[z][n].
// 64 float4 columns x 4 split groups per workgroup: the split loop is 4-way parallel and unrolled.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int splits, int M, int N,
                                                            float* __restrict__ C, int ldc, int accumulate,
                                                            const float* __restrict__ colsum_slab,
                                                            float* __restrict__ bias_grad, int bias_accumulate) {
  __shared__ f32x4 red[4][64];
  const size_t total = (size_t)M * N;            // multiple of 4 is NOT required: tail handled per element
  const size_t total4 = (total + N + 3) / 4;     // [M*N | N] viewed as float4 groups (colsum slab follows the tiles)
  const int zg = threadIdx.x >> 6;
  const size_t e4 = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const size_t base = e4 * 4;
  const bool vec_ok = ((total & 3) == 0) && ((N & 3) == 0);
  if (e4 < total4) {
    if (vec_ok) {
      // tiles: slab[z][base..], column sums: colsum_slab[z][base - total ..]
      const bool in_tiles = base < total;
      const float* src = in_tiles ? slab + base : (colsum_slab ? colsum_slab + (base - total) : nullptr);
      const size_t zstride = in_tiles ? total : (size_t)N;
      if (src) {
        int z = zg;
        for (; z + 12 < splits; z += 16) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(src + (size_t)z * zstride);
          const f32x4 b = *reinterpret_cast<const f32x4*>(src + (size_t)(z + 4) * zstride);
          const f32x4 c = *reinterpret_cast<const f32x4*>(src + (size_t)(z + 8) * zstride);
          const f32x4 d = *reinterpret_cast<const f32x4*>(src + (size_t)(z + 12) * zstride);
          s += (a + b) + (c + d);
        }
        for (; z < splits; z += 4) s += *reinterpret_cast<const f32x4*>(src + (size_t)z * zstride);
      }
    } else {
      for (int c = 0; c < 4; ++c) {
        const size_t i = base + c;
        if (i < total) { for (int z = zg; z < splits; z += 4) s[c] += slab[(size_t)z * total + i]; }
        else if (i < total + N && colsum_slab) { for (int z = zg; z < splits; z += 4) s[c] += colsum_slab[(size_t)z * N + (i - total)]; }
      }
    }
  }
  red[zg][threadIdx.x & 63] = s;
  __syncthreads();
  if (zg != 0 || e4 >= total4) return;
  const f32x4 t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const size_t i = base + c;
    if (i < total) {
      const int m = (int)(i / N), n = (int)(i % N);
      float* dst = C + (size_t)m * ldc + n;
      *dst = accumulate ? *dst + t[c] : t[c];
    } else if (i < total + N && colsum_slab) {
      const int n = (int)(i - total);
      bias_grad[n] = bias_accumulate ? bias_grad[n] + t[c] : t[c];
    }
  }
}

template <int BM, int BN, int WGM, bool SPLITK>
int launch_variant(const GemmParams& p, int a_kc, int b_kc, int splits, hipStream_t st) {
  dim3 grid(p.tiles_m * p.tiles_n, 1, splits), block(256);
#define SKF_GEMM_GO2(AK, BKC, VECV)                                                                            \
  {                                                                                                            \
    constexpr size_t smem = 2 * BK * (TileLD<BM, AK>::value + TileLD<BN, BKC>::value) * sizeof(float);         \
    auto kfn = gemm_kernel<BM, BN, WGM, AK, BKC, SPLITK, VECV>;                                                \
    SKF_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); /* per launch: per-device attribute */ \
    static const std::string tag = std::string(SPLITK ? "gemm_splitk" : "gemm") + "<" + std::to_string(BM) + "x" + \
        std::to_string(BN) + "," + (AK ? "Ak" : "Am") + (BKC ? "Bk" : "Bn") + (VECV ? "" : ",scalar") + ">";  \
    SkfProfScope ps(st, tag.c_str(), 2.0 * p.M * p.N * p.K,                                                    \
                    4.0 * ((double)p.M * p.K + (double)p.K * p.N + (double)p.M * p.N * (p.accumulate ? 2 : 1))); \
    hipLaunchKernelGGL(kfn, grid, block, smem, st, p);                                                         \
  }
#define SKF_GEMM_GO(AK, BKC) { if (vec) SKF_GEMM_GO2(AK, BKC, true) else SKF_GEMM_GO2(AK, BKC, false) }
  // vector path: 16-byte aligned rows and contiguous extents that are multiples of 4
  const bool vec = p.a_vec && p.b_vec && ((a_kc ? p.K : p.M) % 4 == 0) && ((b_kc ? p.K : p.N) % 4 == 0);
  if (a_kc && !b_kc) SKF_GEMM_GO(true, false)
  else if (a_kc && b_kc) SKF_GEMM_GO(true, true)
  else if (!a_kc && !b_kc) SKF_GEMM_GO(false, false)
  else SKF_GEMM_GO(false, true)
#undef SKF_GEMM_GO
#undef SKF_GEMM_GO2
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

}  // namespace

extern "C" size_t skf_gemm_workspace_bytes(int M, int N, int K, int splits, int with_bias_grad) {
  (void)K;
  if (splits <= 1 && !with_bias_grad) return 0;
  if (splits < 1) splits = 1;
  return ((size_t)M * N + N) * (size_t)splits * sizeof(float);
}

// split-K problems (wgrad: small output, long contraction) use 64x64 tiles: the partial-tile slab
// traffic is (#workgroups x tile bytes), so small tiles + ~1 workgroup per CU keep it at ~4 MB.
extern "C" int skf_gemm_default_splits(int M, int N, int K) {
  const int tiles = skf_cdiv(M, 64) * skf_cdiv(N, 64);
  if (K <= 512) return 1;
  static const int wgs = skf_knob("SKF_WGRAD_WGS") ? atoi(skf_knob("SKF_WGRAD_WGS")) : 256;
  int splits = wgs / tiles;
  const int max_splits = skf_cdiv(K, 8 * BK);   // at least 8 slabs per split
  if (splits > max_splits) splits = max_splits;
  return splits < 1 ? 1 : splits;
}

// ---- deferred split-K reduction of MANY wgrads in one launch (the train step runs ~47 wgrads per step)
namespace {
__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(const SkfReduceDesc* __restrict__ descs, int ndesc) {
  __shared__ f32x4 red[4][64];
  // descriptor of this workgroup: the last one with block_begin <= blockIdx.x.  Binary search over scalar loads (constant address
  // space: the table is uploaded once and never written by a kernel) - the linear scan over plain global loads was up to ~50
  // DEPENDENT memory round trips (`global_load_dword; s_waitcnt vmcnt(0)`) before a late workgroup issued its first slab load
  typedef const __attribute__((address_space(4))) SkfReduceDesc* const_descp;
  const const_descp cd = (const_descp)descs;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= cd[mid].block_begin) lo = mid; else hi = mid - 1;
  }
  SkfReduceDesc d;
  d.slab = cd[lo].slab; d.C = cd[lo].C; d.bias_grad = cd[lo].bias_grad;
  d.splits = cd[lo].splits; d.M = cd[lo].M; d.N = cd[lo].N; d.ldc = cd[lo].ldc; d.block_begin = cd[lo].block_begin; d.pad = 0;
  const int M = d.M, N = d.N, splits = d.splits;
  const size_t total = (size_t)M * N, total4 = (total + N + 3) / 4;
  const int zg = threadIdx.x >> 6;
  const size_t e4 = (size_t)(blockIdx.x - d.block_begin) * 64 + (threadIdx.x & 63);
  const size_t base = e4 * 4;
  const float* colsum_slab = d.bias_grad ? d.slab + (size_t)splits * total : nullptr;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (e4 < total4) {
    const bool in_tiles = base < total;
    const float* src = in_tiles ? d.slab + base : (colsum_slab ? colsum_slab + (base - total) : nullptr);
    const size_t zstride = in_tiles ? total : (size_t)N;
    if (src) {
      // (measured: all <= 16 slabs of a wave group requested before the first add - one round trip instead of four - made the launch
      //  SLOWER, 95 vs 52 us for ~95 MB: the slabs of one output tile lie 64-262 KB apart, and more strided requests in flight
      //  only deepen the translation / DRAM-page misses; the cure would be a [tile][split] slab layout)
      int z = zg;
      for (; z + 12 < splits; z += 16) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + (size_t)z * zstride);
        const f32x4 b = *reinterpret_cast<const f32x4*>(src + (size_t)(z + 4) * zstride);
        const f32x4 c = *reinterpret_cast<const f32x4*>(src + (size_t)(z + 8) * zstride);
        const f32x4 e = *reinterpret_cast<const f32x4*>(src + (size_t)(z + 12) * zstride);
        s += (a + b) + (c + e);
      }
      for (; z < splits; z += 4) s += *reinterpret_cast<const f32x4*>(src + (size_t)z * zstride);
    }
  }
  red[zg][threadIdx.x & 63] = s;
  __syncthreads();
  if (zg != 0 || e4 >= total4) return;
  const f32x4 t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const size_t i = base + c;
    if (i < total) d.C[(size_t)(i / N) * d.ldc + (i % N)] = t[c];
    else if (i < total + N && d.bias_grad) d.bias_grad[i - total] = t[c];
  }
}
}  // namespace

// one slab -> C (+ bias gradient): the reduction a split-K skf_gemm_f32 call ends with, for callers that produced the slab
// themselves (the bf16 weight gradient)
extern "C" int skf_splitk_reduce(const float* slab, int splits, int M, int N, float* C, int ldc, int accumulate,
                                 float* bias_grad, int bias_grad_accumulate, skf_stream_t stream) {
  SKF_CHECK_ARG(slab && C && splits > 0 && M > 0 && N > 0, "bad argument");
  SKF_CHECK_ARG((((size_t)M * N) & 3) == 0 && ((uintptr_t)slab & 15) == 0, "slab must be 16-byte aligned with M*N a multiple of 4");
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)M * N + N;
  const int blocks = (int)(((total + 3) / 4 + 63) / 64);
  SkfProfScope ps(st, "splitk_reduce", 0.0, 4.0 * ((double)splits + 1) * total);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, const_cast<float*>(slab), splits, M, N, C, ldc, accumulate,
                     bias_grad ? const_cast<float*>(slab) + (size_t)splits * M * N : nullptr, bias_grad, bias_grad_accumulate);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_splitk_reduce_blocks(int M, int N) { return (int)((((size_t)M * N + N + 3) / 4 + 63) / 64); }

extern "C" int skf_splitk_reduce_batch(const SkfReduceDesc* descs_dev, int ndesc, int total_blocks, skf_stream_t stream) {
  SKF_CHECK_ARG(descs_dev && ndesc > 0 && total_blocks > 0, "bad argument");
  SkfProfScope ps((hipStream_t)stream, "splitk_reduce_batch", 0.0, 0.0);
  SKF_LAUNCH_TAIL(splitk_reduce_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, ndesc);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

// wgrad main kernel only: dW partials (+ column sums of B when with_bias_grad) into `slab`
// ([splits][M][N] then [splits][N]); *splits_used receives the effective split count.  Needs M, N, lda, ldb % 4 == 0.
extern "C" int skf_gemm_wgrad_partial_rows(int M, int N, int K, const float* A, int lda, const float* B, int ldb, int splits,
                                           int with_bias_grad, float* slab, size_t slab_bytes, int* splits_used,
                                           int precision, const int* row_blocks, int row_block_rows, skf_stream_t stream) {
  SKF_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && slab && splits_used, "bad argument");
  SKF_CHECK_ARG(precision == SKF_PREC_F32 || precision == SKF_PREC_BF16X3 || precision == SKF_PREC_BF16X6, "precision must be 0 (fp32 MFMA), 6 (bf16x6) or 3 (bf16x3)");
  if (splits < 1) splits = 1;
  int chunk = skf_cdiv(K, splits);
  chunk = skf_cdiv(chunk, 64) * 64;
  splits = skf_cdiv(K, chunk);
  SKF_CHECK_ARG(slab_bytes >= skf_gemm_workspace_bytes(M, N, K, splits, 1), "slab too small");
  GemmParams p{};
  p.A = A; p.B = B; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
  p.a_vec = ((lda & 3) == 0) && (((uintptr_t)A & 15) == 0);
  p.b_vec = ((ldb & 3) == 0) && (((uintptr_t)B & 15) == 0);
  p.k_chunk = chunk; p.slab = slab; p.precision = precision;
  p.row_blocks = row_blocks; p.row_block_rows = row_blocks ? row_block_rows : 0;
  p.colsum_slab = with_bias_grad ? slab + (size_t)splits * M * N : nullptr;
  p.tiles_m = skf_cdiv(M, 64); p.tiles_n = skf_cdiv(N, 64);
  *splits_used = splits;
  int handled = 0;
  int rc = skf_gemm_wgrad_dispatch(p, 0, 0, splits, (hipStream_t)stream, &handled);
  if (rc != SKF_OK || handled) return rc;
  return launch_variant<64, 64, 2, true>(p, 0, 0, splits, (hipStream_t)stream);
}

// the partial-tile kernels of several weight gradients in ONE launch (falls back to one launch each when a problem is not
// one for the split-arithmetic fast path); probs[i].splits_used is written
extern "C" int skf_gemm_wgrad_partial_group(SkfWgradProblem* probs, int n, int precision, skf_stream_t stream) {
  SKF_CHECK_ARG(probs && n > 0, "bad argument");
  SKF_CHECK_ARG(precision == SKF_PREC_F32 || precision == SKF_PREC_BF16X3 || precision == SKF_PREC_BF16X6, "precision must be 0 (fp32 MFMA), 6 (bf16x6) or 3 (bf16x3)");
  if (n <= 8) {
    GemmParams ps[8];
    int splits[8];
    bool ok = true;
    for (int i = 0; i < n; ++i) {
      SkfWgradProblem& w = probs[i];
      SKF_CHECK_ARG(w.M > 0 && w.N > 0 && w.K > 0 && w.A && w.B && w.slab, "bad problem");
      int sp = w.splits < 1 ? 1 : w.splits;
      int chunk = skf_cdiv(w.K, sp);
      chunk = skf_cdiv(chunk, 64) * 64;
      sp = skf_cdiv(w.K, chunk);
      SKF_CHECK_ARG(w.slab_bytes >= skf_gemm_workspace_bytes(w.M, w.N, w.K, sp, 1), "slab too small");
      GemmParams p{};
      p.A = w.A; p.B = w.B; p.M = w.M; p.N = w.N; p.K = w.K; p.lda = w.lda; p.ldb = w.ldb;
      p.a_vec = ((w.lda & 3) == 0) && (((uintptr_t)w.A & 15) == 0);
      p.b_vec = ((w.ldb & 3) == 0) && (((uintptr_t)w.B & 15) == 0);
      p.k_chunk = chunk; p.slab = w.slab; p.precision = precision;
      p.row_blocks = w.row_blocks; p.row_block_rows = w.row_blocks ? w.row_block_rows : 0;
      p.colsum_slab = w.with_bias_grad ? w.slab + (size_t)sp * w.M * w.N : nullptr;
      ps[i] = p; splits[i] = sp;
      ok = ok && precision != SKF_PREC_F32;
    }
    int handled = 0;
    if (ok) {
      int rc = skf_gemm_wgrad_group_dispatch(ps, splits, n, (hipStream_t)stream, &handled);
      if (rc != SKF_OK) return rc;
    }
    if (handled) {
      for (int i = 0; i < n; ++i) probs[i].splits_used = splits[i];
      return SKF_OK;
    }
  }
  for (int i = 0; i < n; ++i) {
    SkfWgradProblem& w = probs[i];
    int used = 0;
    int rc = skf_gemm_wgrad_partial_rows(w.M, w.N, w.K, w.A, w.lda, w.B, w.ldb, w.splits, w.with_bias_grad, w.slab, w.slab_bytes, &used,
                                         precision, w.row_blocks, w.row_block_rows, stream);
    if (rc != SKF_OK) return rc;
    w.splits_used = used;
  }
  return SKF_OK;
}

extern "C" int skf_gemm_wgrad_partial(int M, int N, int K, const float* A, int lda, const float* B, int ldb, int splits,
                                      int with_bias_grad, float* slab, size_t slab_bytes, int* splits_used,
                                      int precision, skf_stream_t stream) {
  return skf_gemm_wgrad_partial_rows(M, N, K, A, lda, B, ldb, splits, with_bias_grad, slab, slab_bytes, splits_used, precision,
                                     nullptr, 0, stream);
}

extern "C" int skf_gemm_f32(int a_kcontig, int b_kcontig, int M, int N, int K,
                            const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                            const float* bias, int act, const float* relu_src, int ld_relu, int accumulate,
                            int splits, float* bias_grad, int bias_grad_accumulate,
                            void* workspace, size_t workspace_bytes, int precision, skf_stream_t stream) {
  return skf_gemm_f32_rows(a_kcontig, b_kcontig, M, N, K, A, lda, B, ldb, C, ldc, bias, act, relu_src, ld_relu, accumulate, splits,
                           bias_grad, bias_grad_accumulate, workspace, workspace_bytes, precision, nullptr, 0, stream);
}

extern "C" int skf_gemm_f32_rows(int a_kcontig, int b_kcontig, int M, int N, int K,
                                 const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                 const float* bias, int act, const float* relu_src, int ld_relu, int accumulate,
                                 int splits, float* bias_grad, int bias_grad_accumulate,
                                 void* workspace, size_t workspace_bytes, int precision, const int* row_blocks,
                                 int row_block_rows, skf_stream_t stream) {
  return skf_gemm_f32_bits(a_kcontig, b_kcontig, M, N, K, A, lda, B, ldb, C, ldc, bias, act, relu_src, ld_relu, accumulate, splits,
                           bias_grad, bias_grad_accumulate, workspace, workspace_bytes, precision, row_blocks, row_block_rows,
                           nullptr, nullptr, stream);
}

// Dense + residual + dropout + LayerNorm in one launch (the attention output projection of every layer): exists where one
// workgroup of the split-arithmetic weight-stationary kernel owns whole output rows, i.e. K = N = 128.
extern "C" int skf_gemm_ln_residual_supported(int M, int N, int K, int precision) {
  return precision != SKF_PREC_F32 && K == 128 && N == 128 && M > 0 && (double)M * 128 * 4 < 2147483648.0;
}
extern "C" int skf_gemm_ln_residual_f32(int M, int N, int K, const float* A, int lda, const float* W, int ldw, const float* bias,
                                        const float* x, const float* gamma, const float* beta, float* z, float* out, float* stats,
                                        float rate, unsigned site, const void* step_state, int precision, skf_stream_t stream) {
  SKF_CHECK_ARG(skf_gemm_ln_residual_supported(M, N, K, precision), "skf_gemm_ln_residual_f32: K = N = 128 in a split-arithmetic mode only (skf_gemm_ln_residual_supported)");
  SKF_CHECK_ARG(A && W && x && gamma && beta && z && out && stats, "null operand");
  SKF_CHECK_ARG(rate >= 0.f && rate < 1.f && (rate == 0.f || step_state), "dropout needs 0 <= rate < 1 and the step state");
  SKF_CHECK_ARG((lda & 3) == 0 && lda >= K && (ldw & 1) == 0 && ldw >= N, "bad pitch");
  SKF_CHECK_ARG((((uintptr_t)A | (uintptr_t)x | (uintptr_t)z | (uintptr_t)out) & 15) == 0 && (((uintptr_t)W | (uintptr_t)stats) & 7) == 0, "operands must be 16-byte aligned");
  SKF_CHECK_ARG((double)M * lda * 4 < 2147483648.0, "A exceeds 32-bit byte offsets");
  GemmParams p{};
  p.A = A; p.B = W; p.C = z; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = N; p.bias = bias;
  p.precision = precision;
  p.a_vec = 1; p.b_vec = ((ldw & 3) == 0) && (((uintptr_t)W & 15) == 0);
  p.ln_x = x; p.ln_gamma = gamma; p.ln_beta = beta; p.ln_out = out; p.ln_stats = stats;
  p.ln_rate = rate; p.ln_site = site; p.ln_state = step_state;
#if SKF_MEASURE
  { const char* db = skf_knob("SKF_GEMM_DBG"); p.dbg = db ? (long long*)strtoull(db, nullptr, 0) : nullptr; }
#endif
  return skf_gemm_wsx_launch(p, 0, precision == SKF_PREC_BF16X3 ? 2 : 3, (hipStream_t)stream);
}

// The sign-bit path exists where the split-arithmetic weight-stationary kernel takes the launch (skf_gemm_ws_dispatch +
// ws_launch_one): A [M][K] with K in {128,256,384,512}, M >= 1024, a problem above the small-GEMM size, 32-bit byte offsets.
static bool relu_bits_shape(int M, int N, int K, int precision) {
  return precision != SKF_PREC_F32 && (K == 128 || K == 256 || K == 384 || K == 512) && M >= 1024 && (N & 3) == 0 &&
         (double)M * N * K > 33554432.0 && (double)M * (N > K ? N : K) * 4 < 2147483648.0;
}
extern "C" size_t skf_gemm_relu_bits_bytes(int M, int N, int K, int precision) {
  return relu_bits_shape(M, N, K, precision) ? skf_gemm_wsx_relu_bits_bytes(M, N, K) : 0;
}

extern "C" int skf_gemm_f32_bits(int a_kcontig, int b_kcontig, int M, int N, int K,
                                 const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                 const float* bias, int act, const float* relu_src, int ld_relu, int accumulate,
                                 int splits, float* bias_grad, int bias_grad_accumulate,
                                 void* workspace, size_t workspace_bytes, int precision, const int* row_blocks,
                                 int row_block_rows, void* relu_bits_out, const void* relu_bits_in, skf_stream_t stream) {
  SKF_CHECK_ARG(M > 0 && N > 0 && K > 0, "empty problem");
  if (relu_bits_out || relu_bits_in) {
    SKF_CHECK_ARG(relu_bits_shape(M, N, K, precision) && a_kcontig && splits <= 1 && !bias_grad && lda == K && ldc == N &&
                  (((uintptr_t)A | (uintptr_t)C | (uintptr_t)B) & 15) == 0 && (ldb & 3) == 0,
                  "ReLU sign bits: not a launch of the split-arithmetic weight-stationary kernel (skf_gemm_relu_bits_bytes == 0)");
    SKF_CHECK_ARG(!relu_bits_out || (act == 1 && !relu_src && !accumulate), "sign bits are written by a plain relu forward launch");
    SKF_CHECK_ARG(!relu_bits_in || act == 0, "sign bits are read by an input-gradient launch (no activation)");
    SKF_CHECK_ARG((((uintptr_t)relu_bits_out | (uintptr_t)relu_bits_in) & 63) == 0, "sign-bit buffers must be 64-byte aligned");
  }
  SKF_CHECK_ARG(precision == SKF_PREC_F32 || precision == SKF_PREC_BF16X3 || precision == SKF_PREC_BF16X6, "precision must be 0 (fp32 MFMA), 6 (bf16x6) or 3 (bf16x3)");
  SKF_CHECK_ARG(A && B && C, "null operand");
  SKF_CHECK_ARG(act >= 0 && act <= 2, "bad activation");
  hipStream_t st = (hipStream_t)stream;
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.bias = bias; p.act = act; p.relu_src = relu_src; p.ld_relu = ld_relu; p.accumulate = accumulate;
  p.precision = precision;
  SKF_CHECK_ARG(!row_blocks || (a_kcontig && row_block_rows > 0 && !bias && splits <= 1 && !bias_grad),
                "a row-block list goes with the dgrad form: A [M][K], no bias, no split");
  p.row_blocks = row_blocks; p.row_block_rows = row_blocks ? row_block_rows : 0;
  p.relu_bits_out = (unsigned long long*)relu_bits_out; p.relu_bits_in = (const unsigned long long*)relu_bits_in;
  p.a_vec = ((lda & 3) == 0) && (((uintptr_t)A & 15) == 0);
  p.b_vec = ((ldb & 3) == 0) && (((uintptr_t)B & 15) == 0);
  p.tiles_m = skf_cdiv(M, 128); p.tiles_n = skf_cdiv(N, 128);
  { const char* ab = skf_knob("SKF_GEMM_ABLATE"); p.ablate = ab ? atoi(ab) : 0; }
  { const char* xr = skf_knob("SKF_WS_XCD"); p.xcd_remap = xr ? atoi(xr) : 0; }
#if SKF_MEASURE
  { const char* db = skf_knob("SKF_GEMM_DBG"); p.dbg = db ? (long long*)strtoull(db, nullptr, 0) : nullptr; }
#endif
  SKF_CHECK_ARG(!bias_grad || !b_kcontig, "bias_grad needs B as [K][N]");
  {
    int handled = 0;
    int rc = skf_gemm_small_dispatch(p, a_kcontig, b_kcontig, bias_grad, bias_grad_accumulate, st, &handled);
    if (rc != SKF_OK || handled) return rc;
  }
  if (splits <= 1 && !bias_grad) {
    int handled = 0;
    int rc = skf_gemm_ws_dispatch(p, a_kcontig, b_kcontig, st, &handled);
    if (rc != SKF_OK || handled) return rc;
    if (relu_bits_out || relu_bits_in) { skf_set_error("skf_gemm_f32_bits: the weight-stationary path is switched off"); return SKF_EUNSUPPORTED; }
    // 64-row tiles when 128-row tiles would leave most CUs idle
    if (p.tiles_m * p.tiles_n < 400 && M > 64) {
      p.tiles_m = skf_cdiv(M, 64);
      return launch_variant<64, 128, 1, false>(p, a_kcontig, b_kcontig, 1, st);
    }
    return launch_variant<128, 128, 2, false>(p, a_kcontig, b_kcontig, 1, st);
  }
  SKF_CHECK_ARG(!bias && act == 0 && !relu_src, "split-K supports no fused epilogue");
  if (splits < 1) splits = 1;
  SKF_CHECK_ARG(workspace && workspace_bytes >= skf_gemm_workspace_bytes(M, N, K, splits, 1), "workspace too small");
  SKF_CHECK_ARG(!bias_grad || !b_kcontig, "bias_grad needs B as [K][N]");
  int chunk = skf_cdiv(K, splits);
  chunk = skf_cdiv(chunk, 64) * 64;
  splits = skf_cdiv(K, chunk);
  p.k_chunk = chunk;
  p.slab = (float*)workspace;
  p.colsum_slab = bias_grad ? p.slab + (size_t)splits * M * N : nullptr;
  p.tiles_m = skf_cdiv(M, 64); p.tiles_n = skf_cdiv(N, 64);
  int handled = 0;
  int rc = skf_gemm_wgrad_dispatch(p, a_kcontig, b_kcontig, splits, st, &handled);
  if (rc != SKF_OK) return rc;
  if (!handled) rc = launch_variant<64, 64, 2, true>(p, a_kcontig, b_kcontig, splits, st);
  if (rc != SKF_OK) return rc;
  const size_t total = (size_t)M * N + N;
  const int blocks = (int)(((total + 3) / 4 + 63) / 64);
  SkfProfScope ps(st, "splitk_reduce", 0.0, 4.0 * ((double)splits + 1) * total);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.slab, splits, M, N, C, ldc, accumulate,
                     p.colsum_slab, bias_grad, bias_grad_accumulate);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
