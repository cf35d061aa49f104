// Weight-stationary fp32 GEMM for the Dense forward / dgrad shapes of this model:
//   C[M,N] (+)= A[M,K] . opB(B)[K,N]  with a SMALL contraction K in {128,256,384,512} and M ~ 25k rows.
//
// Why a second kernel: with K this short the generic LDS-tiled kernel is all prologue/epilogue
// (ablation on MI355X: 25600x128x128 takes 15.8 us of 23.2 us with the MFMAs removed).  Here
//   * every wave keeps ITS slice of the weight matrix in registers for the whole kernel
//     (CW columns x K values = K/4 * CW/16 VGPRs per lane; gfx950 has 512 per lane),
//   * workgroups are persistent: 256 of them (one per CU) walk 16-row tiles of A, so tile
//     quantisation is 1600 tiles / 256 CUs instead of 200..400 big tiles / 256 CUs,
//   * A tiles stream global -> registers -> LDS with the loads of tile t+1 in flight during the
//     MFMAs + stores of tile t (one barrier per tile); A rows land in LDS as full 512-byte lines,
//   * the per-lane k index set of an MFMA step is free as long as A and B agree, so lane group g
//     owns k = 16j + 4g + e (j < K/16, e < 4): A fragments are ds_read_b128 (4 MFMA steps per read,
//     near conflict-free with a +4 float row pad) and are shared by all column blocks of the wave,
//   * the C tile goes back through a wave-private LDS patch so global stores are float4 rows.
// MFMA: v_mfma_f32_16x16x4_f32 (A: lane (i=l&15,g=l>>4) holds A[i][k=g], B[k=g][j=i]; C: col=i,row=4g+r).
#include "skf_common.h"
#include "skf_gemm_params.h"

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int TR = 16;   // rows per tile

template <int K>
__device__ __forceinline__ void ws_load_tile(const float* __restrict__ A, int lda, int M, int tile,
                                             f32x4 (&ra)[TR * K / 1024]) {
#pragma unroll
  for (int v = 0; v < TR * K / 1024; ++v) {
    const int e = threadIdx.x + v * 256, row = e / (K / 4), c4 = (e % (K / 4)) * 4;
    int grow = tile * TR + row;
    grow = grow < M ? grow : M - 1;             // clamped: rows past M are never stored
    ra[v] = *reinterpret_cast<const f32x4*>(A + (size_t)grow * lda + c4);   // ext_vector load: stays in VGPRs
  }
}

template <int K>
__device__ __forceinline__ void ws_store_tile(float* __restrict__ dst, const f32x4 (&ra)[TR * K / 1024]) {
#pragma unroll
  for (int v = 0; v < TR * K / 1024; ++v) {
    const int e = threadIdx.x + v * 256, row = e / (K / 4), c4 = (e % (K / 4)) * 4;
    *reinterpret_cast<f32x4*>(&dst[row * (K + 4) + c4]) = ra[v];
  }
}

template <int K, int CW, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_ws_kernel(GemmParams p, int groups, int workers) {
  constexpr int NB = CW / 16;            // 16-column blocks per wave
  constexpr int KQ = K / 4;              // k values per lane group
  constexpr int LDA_S = K + 4;           // +16 B: the 16 rows of a ds_read_b128 group land on 16 different slots
  constexpr int LDC_S = CW + 4;
  constexpr int NV = TR * K / 1024;      // float4 per thread per A tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                  // [2][TR][LDA_S]
  float* Cs = smem + 2 * TR * LDA_S;                 // [4 waves][TR][LDC_S]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  // the column groups of one worker read the same A tiles: consecutive logical ids -> same XCD / L2
  const int logical = p.xcd_remap ? skf_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int group = logical % groups, worker = logical / groups;
  const int n_wave = group * 4 * CW + wave * CW;     // first output column of this wave
  const int ntiles = (p.M + TR - 1) / TR;

  long long* dbg = (p.dbg && lane == 0 && wave == 0 && (blockIdx.x % 97) == 0 && blockIdx.x / 97 < 8) ? p.dbg + (blockIdx.x / 97) * 32 : nullptr;
  int dbi = 0;
#define SKF_STAMP() do { if (dbg && dbi < 32) dbg[dbi++] = clock64(); } while (0)
  SKF_STAMP();
  // ---- this wave's weight slice -> registers (once)
  float breg[NB][KQ];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n_wave + nb * 16 + i;
    const bool nok = n < p.N;
    if (B_KC) {                                       // B stored [N][K]: 16-byte loads along k
#pragma unroll
      for (int j = 0; j < KQ / 4; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nok) v = *reinterpret_cast<const float4*>(p.B + (size_t)n * p.ldb + 16 * j + 4 * g);
        breg[nb][4 * j + 0] = v.x; breg[nb][4 * j + 1] = v.y; breg[nb][4 * j + 2] = v.z; breg[nb][4 * j + 3] = v.w;
      }
    } else {                                          // B stored [K][N]
#pragma unroll
      for (int s = 0; s < KQ; ++s) breg[nb][s] = nok ? p.B[(size_t)(16 * (s >> 2) + 4 * g + (s & 3)) * p.ldb + n] : 0.f;
    }
  }

  f32x4 ra[NV];

  // bias of this wave's columns: loop invariant, but the compiler cannot hoist the load past the C stores
  float bias_r[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n_wave + nb * 16 + i;
    bias_r[nb] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
  }
  SKF_STAMP();   // B loads issued
  int tile = worker;
  ws_load_tile<K>(p.A, p.lda, p.M, tile < ntiles ? tile : 0, ra);
  ws_store_tile<K>(As, ra);
  __syncthreads();
  SKF_STAMP();   // first A tile in LDS
  float* cs = Cs + wave * TR * LDC_S;
  int cur = 0;
  for (; tile < ntiles; tile += workers) {
    const int next = tile + workers;
    if (p.ablate != 3) ws_load_tile<K>(p.A, p.lda, p.M, next < ntiles ? next : tile, ra);
    const float* At = As + cur * TR * LDA_S + i * LDA_S + 4 * g;
    f32x4 acc[NB][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { acc[nb][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[nb][1] = acc[nb][0]; }
    if (p.ablate != 1)
#pragma unroll
    for (int j = 0; j < KQ / 4; ++j) {
      const float4 a = *reinterpret_cast<const float4*>(At + 16 * j);
      // consecutive MFMAs never share an accumulator (dependent latency 40 cycles > 32-cycle issue)
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[nb][e & 1] = mfma16(av[e], breg[nb][4 * j + e], acc[nb][e & 1]);
    }
    SKF_STAMP();   // MFMAs issued (first use of acc below waits for them)
    // ---- epilogue: lane (i,g) holds C[row 4g+r][col nb*16+i]; bias/act, then through the wave's LDS patch
    // one wave-uniform switch per tile (a per-element switch costs ~40 scalar branches per tile)
    float vals[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) vals[nb][r] = acc[nb][0][r] + acc[nb][1][r] + bias_r[nb];
    if (p.act == 1) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) vals[nb][r] = fmaxf(vals[nb][r], 0.f);
    } else if (p.act == 2) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) vals[nb][r] = tanhf(vals[nb][r]);
    }
    if (p.direct_store) {
      // straight from the MFMA C layout: each store covers 4 rows x 64 contiguous bytes (L2 merges the two column blocks)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = n_wave + nb * 16 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int grow = tile * TR + 4 * g + r;
          if (grow < p.M && n < p.N) {
            float v = vals[nb][r];
            if (p.relu_src) v = p.relu_src[(size_t)grow * p.ld_relu + n] > 0.f ? v : 0.f;
            float* dst = p.C + (size_t)grow * p.ldc + n;
            if (p.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
    } else {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[(4 * g + r) * LDC_S + nb * 16 + i] = vals[nb][r];
    __builtin_amdgcn_wave_barrier();
    constexpr int F4_ROW = CW / 4;                     // float4 per row of the wave's patch
#pragma unroll
    for (int e = lane; e < TR * F4_ROW; e += 64) {
      const int row = e / F4_ROW, c4 = (e % F4_ROW) * 4;
      const int grow = tile * TR + row, n = n_wave + c4;
      float4 v = *reinterpret_cast<const float4*>(&cs[row * LDC_S + c4]);
      if (p.ablate == 2 && v.x != 12345.678f) continue;
      if (grow < p.M && n < p.N) {                     // N % 4 == 0: a float4 is fully in or fully out
        if (p.relu_src) {
          const float4 h = *reinterpret_cast<const float4*>(p.relu_src + (size_t)grow * p.ld_relu + n);
          v.x = h.x > 0.f ? v.x : 0.f; v.y = h.y > 0.f ? v.y : 0.f; v.z = h.z > 0.f ? v.z : 0.f; v.w = h.w > 0.f ? v.w : 0.f;
        }
        float4* dst = reinterpret_cast<float4*>(p.C + (size_t)grow * p.ldc + n);
        if (p.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *dst = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    }
    SKF_STAMP();   // C tile stored
    ws_store_tile<K>(As + (cur ^ 1) * TR * LDA_S, ra);
    __syncthreads();
    SKF_STAMP();   // next A tile in LDS + barrier
    cur ^= 1;
  }
}

template <int K, int CW>
int launch_ws(const GemmParams& p, int b_kc, hipStream_t st) {
  const int groups = skf_cdiv(p.N, 4 * CW);
  // two workgroups per CU: one's epilogue / tile hand-over overlaps the other's MFMAs
  static const int wg_target = getenv("SKF_WS_WGS") ? atoi(getenv("SKF_WS_WGS")) : (K == 128 ? 768 : 512);
  int workers = wg_target / groups;
  if (workers < 1) workers = 1;
  const int ntiles = skf_cdiv(p.M, TR);
  if (workers > ntiles) workers = ntiles;
  const size_t smem = (size_t)(2 * TR * (K + 4) + 4 * TR * (CW + 4)) * sizeof(float);
  dim3 grid(groups * workers), block(256);
  static const std::string tag = "gemm_ws<K" + std::to_string(K) + ",CW" + std::to_string(CW) + ">";
  SkfProfScope ps(st, tag.c_str(), 2.0 * p.M * p.N * p.K,
                  4.0 * ((double)p.M * p.K + (double)p.K * p.N + (double)p.M * p.N * (p.accumulate ? 2 : 1)));
  if (b_kc) hipLaunchKernelGGL((gemm_ws_kernel<K, CW, true>), grid, block, smem, st, p, groups, workers);
  else hipLaunchKernelGGL((gemm_ws_kernel<K, CW, false>), grid, block, smem, st, p, groups, workers);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

}  // namespace

// Returns SKF_OK and sets *handled = 1 when the weight-stationary path applies.
int skf_gemm_ws_dispatch(const GemmParams& p, int a_kcontig, int b_kcontig, hipStream_t st, int* handled) {
  *handled = 0;
  const char* off = getenv("SKF_GEMM_NO_WS");
  if (off && off[0] == '1') return SKF_OK;
  if (!a_kcontig || p.M < 1024) return SKF_OK;
  if (!(p.K == 128 || p.K == 256 || p.K == 384 || p.K == 512)) return SKF_OK;
  if ((p.N & 3) || (p.lda & 3) || (p.ldc & 3) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.C & 15)) return SKF_OK;
  if (b_kcontig && ((p.ldb & 3) || ((uintptr_t)p.B & 15))) return SKF_OK;
  if (p.relu_src && ((p.ld_relu & 3) || ((uintptr_t)p.relu_src & 15))) return SKF_OK;
  *handled = 1;
  switch (p.K) {
    case 128: return launch_ws<128, 32>(p, b_kcontig, st);
    case 256: return launch_ws<256, 16>(p, b_kcontig, st);
    case 384: return launch_ws<384, 16>(p, b_kcontig, st);
    default:  return launch_ws<512, 16>(p, b_kcontig, st);
  }
}
