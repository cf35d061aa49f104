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
#include <string.h>

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int TR = 16;   // rows per tile

template <int N> struct VecOf;
template <> struct VecOf<1> { typedef float type; typedef unsigned utype; };
template <> struct VecOf<2> { typedef float __attribute__((ext_vector_type(2))) type; typedef unsigned __attribute__((ext_vector_type(2))) utype; };
template <> struct VecOf<4> { typedef f32x4 type; typedef unsigned __attribute__((ext_vector_type(4))) utype; };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Buffer descriptor over the rows [row0, M) of a row-major matrix: per-lane byte offsets stay constant from
// tile to tile (no 64-bit address VALU in the loop - f32 MFMA and VALU share the issue pipe on gfx950), and the
// hardware range check drops stores / zero-fills loads of rows past M.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rows_rsrc(const float* base, int ld, int M, int row0) {
  long long rem = ((long long)M - row0) * ld * 4;
  rem = rem < 0 ? 0 : (rem > 0xffffffffLL ? 0xffffffffLL : rem);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (long long)row0 * ld), 0, (unsigned)rem, 0x00020000);
}
template <int NB>
__device__ __forceinline__ typename VecOf<NB>::type ws_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  typedef typename VecOf<NB>::type vecn;
  if constexpr (NB == 1) return __builtin_bit_cast(vecn, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
  else if constexpr (NB == 2) return __builtin_bit_cast(vecn, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
  else return __builtin_bit_cast(vecn, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
template <int NB>
__device__ __forceinline__ void ws_buf_store(typename VecOf<NB>::type v, __amdgpu_buffer_rsrc_t r, unsigned voff) {
  typedef typename VecOf<NB>::utype uvec;
  if constexpr (NB == 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uvec, v), r, voff, 0, 0);
  else if constexpr (NB == 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, v), r, voff, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, v), r, voff, 0, 0);
}

template <int K>
__device__ __forceinline__ void ws_load_tile(const float* __restrict__ A, int lda, int M, int tile,
                                             const unsigned (&a_voff)[TR * K / 1024], f32x4 (&ra)[TR * K / 1024]) {
  const __amdgpu_buffer_rsrc_t r = ws_rows_rsrc(A, lda, M, tile * TR);
#pragma unroll
  for (int v = 0; v < TR * K / 1024; ++v)
    ra[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, a_voff[v], 0, 0));
}

template <int K>
__device__ __forceinline__ void ws_store_tile(float* __restrict__ dst, const f32x4 (&ra)[TR * K / 1024]) {
#pragma unroll
  for (int v = 0; v < TR * K / 1024; ++v) {
    const int e = threadIdx.x + v * 256, row = e / (K / 4), c4 = (e % (K / 4)) * 4;
    *reinterpret_cast<f32x4*>(&dst[row * (K + 8) + c4]) = ra[v];
  }
}

// Column ownership: lane i of a wave owns the NB ADJACENT columns n_wave + NB*i + nb (MFMA block nb uses
// column nb of every lane).  The MFMA does not care which actual column sits behind "column i", and with
// this choice (a) a [K][N] weight row is read with one NB-wide vector load per k instead of NB dword loads
// (the dword version spent 4.2k cycles per workgroup in the texture-address unit), (b) the C fragment of a
// lane is NB adjacent floats of one row, so it is stored straight from registers as 16 lanes x NB*4 bytes
// contiguous - no LDS patch, no LDS round trip in the epilogue.
//
// Tile pipeline of one wave (t = tile in LDS buffer cur):
//   after the barrier: ds_read the first A fragments of t | store C of t-1 (kept in registers)   <- fills the LDS latency
//   MFMAs of t; half way: hand tile t+1 (registers) to LDS buffer cur^1, issue the global loads of t+3 into
//   those registers, issue the relu-source / accumulate loads of t
//   barrier
// so the only MFMA-free stretch per tile is the LDS read latency + the barrier.
// EXTRA = the epilogue reads a relu source and/or the old C (kept out of the plain variant: even an unused,
// predicated-off load leaves an s_waitcnt in front of the C stores that also waits for the A prefetch).
template <int K, int NB, bool B_KC, bool EXTRA>
__global__ __launch_bounds__(256, (K * NB <= 512 ? 2 : 1)) void gemm_ws_kernel(GemmParams p, int groups, int workers) {
  constexpr int CW = 16 * NB;            // columns per wave
  constexpr int KQ = K / 4;              // k values per lane group
  constexpr int NJ = KQ / 4;             // A fragments (ds_read_b128) per tile
  constexpr int PF = NJ <= 8 ? NJ : 4;   // fragments read ahead of the C stores
  constexpr int NACC = NB == 1 ? 2 : 1;  // independent accumulators per column block (dependent MFMA latency 40 > 32)
  constexpr int LDA_S = K + 8;           // +32 B: pitch/16 = 2 (mod 4) keeps the four 16-lane groups of ds_read_b128 (which mix the lane
                                         // groups g) free of bank conflicts; +16 B left a 2-way conflict per group
  constexpr int NV = TR * K / 1024;      // float4 per thread per A tile
  constexpr unsigned OOB = 0x7ffffff0u;  // byte offset outside any descriptor: load -> 0, store -> dropped
  typedef typename VecOf<NB>::type vecn;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                  // [2][TR][LDA_S]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  // the column groups of one worker read the same A tiles: consecutive logical ids -> same XCD / L2
  const int logical = p.xcd_remap ? skf_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int group = logical % groups, worker = logical / groups;
  const int n_lane = group * 4 * CW + wave * CW + NB * i;   // first of this lane's NB columns
  const bool nok = n_lane < p.N;                            // N % NB == 0: all NB columns in or all out
  const int n_ld = nok ? n_lane : p.N - NB;                 // clamped: loads stay in bounds, stores are guarded
  const int ntiles = (p.M + TR - 1) / TR;

  long long* dbg = (p.dbg && lane == 0 && wave == 0 && (blockIdx.x % 97) == 0 && blockIdx.x / 97 < 8) ? p.dbg + (blockIdx.x / 97) * 32 : nullptr;
  int dbi = 0;
#if SKF_WS_STAMPS   // per-phase s_memtime stamps (tools/ws_timeline.py); off by default: every conditional memory
                    // operation in the tile loop makes the compiler's s_waitcnt vmcnt counts conservative
#define SKF_STAMP() do { if (dbg && dbi < 32) dbg[dbi++] = clock64(); } while (0)
#else
#define SKF_STAMP() do { (void)dbg; (void)dbi; } while (0)
#endif
  SKF_STAMP();
  // per-lane byte offsets inside a tile (constant for the whole kernel)
  unsigned a_voff[NV], c_voff[4], h_voff[4];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int e = tid + v * 256, row = e / (K / 4), c4 = (e % (K / 4)) * 4;
    a_voff[v] = (unsigned)(row * p.lda + c4) * 4u;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    c_voff[r] = nok ? (unsigned)((4 * g + r) * p.ldc + n_lane) * 4u : OOB;
    h_voff[r] = nok ? (unsigned)((4 * g + r) * p.ld_relu + n_lane) * 4u : OOB;
  }
  // A tiles are prefetched TWO tiles ahead (tile t in LDS, t+1 and t+2 in registers): a global load
  // under a busy chip takes ~3-4k cycles, longer than one tile of MFMAs (2k cycles per wave).
  // The first two go out before the weight loads so that their latency hides behind those.
  f32x4 ra0[NV], ra1[NV];
  int tile = worker;
  ws_load_tile<K>(p.A, p.lda, p.M, tile, a_voff, ra0);
  ws_load_tile<K>(p.A, p.lda, p.M, tile + workers, a_voff, ra1);
  // ---- this wave's weight slice -> registers (once): breg[nb][s] = B[k = 16(s>>2) + 4g + (s&3)][n_lane + nb]
  float breg[NB][KQ];
  if (B_KC) {                                         // B stored [N][K]: 16-byte loads along k
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.B + (size_t)(n_ld + nb) * p.ldb + 16 * j + 4 * g);
        breg[nb][4 * j + 0] = v[0]; breg[nb][4 * j + 1] = v[1]; breg[nb][4 * j + 2] = v[2]; breg[nb][4 * j + 3] = v[3];
      }
  } else {                                            // B stored [K][N]: one NB-wide load per k
#pragma unroll
    for (int s = 0; s < KQ; ++s) {
      const vecn v = *reinterpret_cast<const vecn*>(p.B + (size_t)(16 * (s >> 2) + 4 * g + (s & 3)) * p.ldb + n_ld);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) breg[nb][s] = reinterpret_cast<const float*>(&v)[nb];
    }
  }
  float bias_r[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bias_r[nb] = p.bias ? p.bias[n_ld + nb] : 0.f;


  SKF_STAMP();   // B loads issued
  ws_store_tile<K>(As, ra0);
  __syncthreads();
  ws_load_tile<K>(p.A, p.lda, p.M, tile + 2 * workers, a_voff, ra0);
  SKF_STAMP();   // first A tile in LDS

  vecn cprev[4], hsrc[4], oacc[4];   // finished C fragment of the previous tile (+ its relu source / old C)
  // No branch around any memory operation of the tile loop: "nothing to do" cases use an empty descriptor
  // (first iteration: prev_tile = ntiles -> zero records -> the stores are dropped by the range check).
  int prev_tile = ntiles;
  const bool has_relu = EXTRA && p.relu_src != nullptr;
  auto store_prev = [&]() {
    const __amdgpu_buffer_rsrc_t rc = ws_rows_rsrc(p.C, p.ldc, p.M, prev_tile * TR);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      vecn v = cprev[r];
      if (EXTRA) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          reinterpret_cast<float*>(&v)[nb] = (!has_relu || reinterpret_cast<const float*>(&hsrc[r])[nb] > 0.f) ? reinterpret_cast<const float*>(&v)[nb] : 0.f;
      }
      if (EXTRA) {   // old C (zeros from the empty descriptor when not accumulating)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) reinterpret_cast<float*>(&v)[nb] += reinterpret_cast<const float*>(&oacc[r])[nb];
      }
      ws_buf_store<NB>(v, rc, c_voff[r]);
    }
  };

  // EARLY (every A fragment of a tile fits in registers, K = 128): the workgroup barrier sits at 3/4 of the
  // MFMAs - by then every wave has written its share of the next tile (done at 1/2) - and the next tile's
  // fragments are read right behind it into the other fragment set, so the MFMA stream of a wave never waits
  // for LDS or for the barrier hand-shake.  (All fragments of a tile are in registers before that tile's barrier,
  // so the buffer it leaves is free to be overwritten half a tile later.)
  constexpr bool EARLY = NJ <= 8;
  f32x4 afA[PF], afB[PF];
  if (EARLY) {
#pragma unroll
    for (int j = 0; j < PF; ++j) afA[j] = *reinterpret_cast<const f32x4*>(As + i * LDA_S + 4 * g + 16 * j);
  }

  // one tile: MFMAs on LDS buffer `cur`; hand `rn` (tile+workers) to the other buffer; refill `rn` with tile+3*workers
  auto do_tile = [&](int cur, f32x4 (&rn)[NV], f32x4 (&af)[PF], f32x4 (&afn)[PF]) {
    const float* At = As + cur * TR * LDA_S + i * LDA_S + 4 * g;
    if (!EARLY) {
#pragma unroll
      for (int j = 0; j < PF; ++j) af[j] = *reinterpret_cast<const f32x4*>(At + 16 * j);
    }
    __builtin_amdgcn_sched_barrier(0);
    store_prev();
    __builtin_amdgcn_sched_barrier(0);
    SKF_STAMP();   // previous C tile stored
    f32x4 acc[NB][NACC];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int c = 0; c < NACC; ++c) {
        const float b0 = c == 0 ? bias_r[nb] : 0.f;   // bias folded into the accumulator
        acc[nb][c] = (f32x4){b0, b0, b0, b0};
      }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f32x4 a;
      if (j < PF) a = af[j]; else a = *reinterpret_cast<const f32x4*>(At + 16 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[nb][NACC == 1 ? 0 : (j & 1)] = mfma16(a[e], breg[nb][4 * j + e], acc[nb][NACC == 1 ? 0 : (j & 1)]);
      if (j == NJ / 2 - 1) {
        // half way: tile+workers goes to the other LDS buffer (its readers passed the last barrier), then the
        // registers are refilled with tile+3*workers; loads this tile's epilogue needs are issued now as well
        __builtin_amdgcn_sched_barrier(0);
        ws_store_tile<K>(As + (cur ^ 1) * TR * LDA_S, rn);
        ws_load_tile<K>(p.A, p.lda, p.M, tile + 3 * workers, a_voff, rn);
        if (EXTRA) {
          const __amdgpu_buffer_rsrc_t rh = ws_rows_rsrc(has_relu ? p.relu_src : p.C, p.ld_relu, has_relu ? p.M : 0, tile * TR);
#pragma unroll
          for (int r = 0; r < 4; ++r) hsrc[r] = ws_buf_load<NB>(rh, h_voff[r]);
          const __amdgpu_buffer_rsrc_t ro = ws_rows_rsrc(p.C, p.ldc, p.accumulate ? p.M : 0, tile * TR);
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[r] = ws_buf_load<NB>(ro, c_voff[r]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (EARLY && j == (3 * NJ) / 4 - 1) {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        const float* An = As + (cur ^ 1) * TR * LDA_S + i * LDA_S + 4 * g;
#pragma unroll
        for (int jj = 0; jj < PF; ++jj) afn[jj] = *reinterpret_cast<const f32x4*>(An + 16 * jj);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    SKF_STAMP();   // MFMAs issued
    // lane (i,g) holds C[row 4g+r][n_lane + nb]; one wave-uniform activation switch per tile
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float v = acc[nb][0][r];
        if (NACC == 2) v += acc[nb][1][r];
        reinterpret_cast<float*>(&cprev[r])[nb] = v;
      }
    if (p.act == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) reinterpret_cast<float*>(&cprev[r])[nb] = fmaxf(reinterpret_cast<float*>(&cprev[r])[nb], 0.f);
    } else if (p.act == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) reinterpret_cast<float*>(&cprev[r])[nb] = tanhf(reinterpret_cast<float*>(&cprev[r])[nb]);
    }
    prev_tile = tile;
    if (!EARLY) __syncthreads();
    SKF_STAMP();   // barrier
  };
  while (tile < ntiles) {
    do_tile(0, ra1, afA, afB);
    tile += workers;
    if (tile >= ntiles) break;
    do_tile(1, ra0, afB, afA);
    tile += workers;
  }
  store_prev();
}

template <int K, int NB>
int launch_ws(const GemmParams& p, int b_kc, hipStream_t st) {
  constexpr int CW = 16 * NB;
  const int groups = skf_cdiv(p.N, 4 * CW);
  // two workgroups per CU where the register budget allows it (K*NB <= 512), else one
  static const int wg_target = skf_knob("SKF_WS_WGS") ? atoi(skf_knob("SKF_WS_WGS")) : (K * NB <= 512 ? 512 : 256);
  int workers = wg_target / groups;
  if (workers < 1) workers = 1;
  const int ntiles = skf_cdiv(p.M, TR);
  if (workers > ntiles) workers = ntiles;
  const size_t smem = (size_t)(2 * TR * (K + 8)) * sizeof(float);
  dim3 grid(groups * workers), block(256);
  static const std::string tag = "gemm_ws<K" + std::to_string(K) + ",CW" + std::to_string(CW) + ">";
  SkfProfScope ps(st, tag.c_str(), 2.0 * p.M * p.N * p.K,
                  4.0 * ((double)p.M * p.K + (double)p.K * p.N + (double)p.M * p.N * (p.accumulate ? 2 : 1)));
  const bool extra = p.relu_src || p.accumulate;
  if (b_kc && extra) hipLaunchKernelGGL((gemm_ws_kernel<K, NB, true, true>), grid, block, smem, st, p, groups, workers);
  else if (b_kc) hipLaunchKernelGGL((gemm_ws_kernel<K, NB, true, false>), grid, block, smem, st, p, groups, workers);
  else if (extra) hipLaunchKernelGGL((gemm_ws_kernel<K, NB, false, true>), grid, block, smem, st, p, groups, workers);
  else hipLaunchKernelGGL((gemm_ws_kernel<K, NB, false, false>), grid, block, smem, st, p, groups, workers);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

}  // namespace

// p.precision: 0 = fp32 MFMA, 6 / 3 = fp32 operands split into 3 / 2 bf16 pieces on the bf16 matrix cores (skf_gemm_wsx.hip).
// Returns SKF_OK and sets *handled = 1 when the weight-stationary path applies.
static int ws_launch_one(const GemmParams& p, int b_kcontig, hipStream_t st) {
  const bool fits32 = (double)p.M * p.lda * 4 < 2147483648.0 && (double)p.M * p.ldc * 4 < 2147483648.0 &&
                      (!p.relu_src || (double)p.M * p.ld_relu * 4 < 2147483648.0);   // the split kernels use 32-bit byte offsets
  if (const int prec = p.precision; prec && fits32) return skf_gemm_wsx_launch(p, b_kcontig, prec == 3 ? 2 : 3, st);
  switch (p.K) {
    case 128: return launch_ws<128, 2>(p, b_kcontig, st);
    case 256: return launch_ws<256, 2>(p, b_kcontig, st);
    // K >= 384: one column per lane keeps the weight slice at K/4 registers and two workgroups per CU
    // (two columns per lane = one workgroup per CU measured 5-8 % slower on 25600x128x{384,512})
    case 384: return launch_ws<384, 1>(p, b_kcontig, st);
    default:  return launch_ws<512, 1>(p, b_kcontig, st);
  }
}

int skf_gemm_ws_dispatch(const GemmParams& p, int a_kcontig, int b_kcontig, hipStream_t st, int* handled) {
  *handled = 0;
  const char* off = skf_knob("SKF_GEMM_NO_WS");
  if (off && off[0] == '1') return SKF_OK;
  if (!a_kcontig || p.M < 1024) return SKF_OK;
  // K in {128,256,384,512} = one launch; longer K (a multiple of 128 up to 2048: dff = 1024 / 2048, the 3d-wide qkv dgrad
  // at d = 256) = a chain of <= 512-deep launches over column slices of A / row slices of B, every launch after the first
  // accumulating into C - only without an epilogue that must see the complete sum (activation, relu mask)
  const bool single = p.K == 128 || p.K == 256 || p.K == 384 || p.K == 512;
  const bool chain = !single && p.K > 512 && p.K <= 2048 && (p.K & 127) == 0 && p.act == 0 && !p.relu_src;
  // Input-gradient form with any K % 4 == 0 (the logits layer: K = vocabulary = 1004 / 10004), split arithmetic only: 512-deep
  // slices, the last one masked (GemmParams::k_valid).  SKF_NO_MASKED_CHAIN=1 keeps such shapes on the generic kernel.
  static const bool masked_off = skf_knob("SKF_NO_MASKED_CHAIN") && skf_knob("SKF_NO_MASKED_CHAIN")[0] == '1';
  const bool fits32m = (double)p.M * p.lda * 4 < 2147483648.0 && (double)p.M * p.ldc * 4 < 2147483648.0;
  // (round 6: up to 16384 - the grid tokenizer's vocabulary of 10004 is twenty 512-deep launches, ~0.45 ms instead of ONE 0.85-ms launch of
  //  the generic fp32-MFMA kernel; every launch after the first re-reads and rewrites the 13 MB of C)
  const bool chain_masked = !single && !chain && !masked_off && b_kcontig && p.precision != SKF_PREC_F32 && fits32m && p.K > 512 &&
                            p.K <= 16384 && (p.K & 3) == 0 && p.act == 0 && !p.relu_src && !p.relu_bits_in && !p.relu_bits_out && !p.bias;
  if (!single && !chain && !chain_masked) return SKF_OK;
  if ((p.N & 3) || (p.lda & 3) || (p.ldc & 3) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.C & 15)) return SKF_OK;
  if (b_kcontig && ((p.ldb & 3) || ((uintptr_t)p.B & 15))) return SKF_OK;
  if (!b_kcontig && ((p.ldb & 1) || ((uintptr_t)p.B & 7))) return SKF_OK;   // NB-wide loads along n
  if (p.relu_src && ((p.ld_relu & 3) || ((uintptr_t)p.relu_src & 15))) return SKF_OK;
  *handled = 1;
  if (single) return ws_launch_one(p, b_kcontig, st);
  // a chain of launches: an event parked for "the launch" (skf_common.h: SKF_LAUNCH_TAIL) belongs to the LAST one
  const hipEvent_t tail_event = skf_tls_stop_event;
  skf_tls_stop_event = nullptr;
  if (chain_masked) {
    for (int k0 = 0; k0 < p.K; k0 += 512) {
      if (k0 + 512 >= p.K) skf_tls_stop_event = tail_event;
      GemmParams q = p;
      q.K = 512;
      q.A = p.A + k0;
      q.B = p.B + k0;
      if (k0 > 0) q.accumulate = 1;
      if (p.K - k0 < 512) { q.k_valid = p.K - k0; q.a_cut = k0 * 4; }
      const int rc = skf_gemm_wsx_launch(q, 1, p.precision == 3 ? 2 : 3, st);
      if (rc != SKF_OK) return rc;
    }
    return SKF_OK;
  }
  for (int k0 = 0; k0 < p.K;) {
    const int left = p.K - k0, kc = left >= 512 ? 512 : left;       // left is a multiple of 128 below 512: 128 / 256 / 384
    if (k0 + kc >= p.K) skf_tls_stop_event = tail_event;
    GemmParams q = p;
    q.K = kc;
    q.A = p.A + k0;
    q.B = b_kcontig ? p.B + k0 : p.B + (size_t)k0 * p.ldb;
    if (k0 > 0) { q.bias = nullptr; q.accumulate = 1; }
    const int rc = ws_launch_one(q, b_kcontig, st);
    if (rc != SKF_OK) return rc;
    k0 += kc;
  }
  return SKF_OK;
}
