// Dense layers of the bf16 path (BASELINE cfg 5: bf16 storage + bf16 MFMA, fp32 accumulate, fp32 master weights).
//
// tf.keras.layers.Dense forward / tape dgrad / wgrad (builders/layers/transformer.py:154-158,196-197;
// models/sketchformer.py:85-104) on v_mfma_f32_32x32x16_bf16.  Two kernels:
//
//   gemm_bf16_nt : C[M][N] = A[M][K] . B[N][K]^T   - both operands contraction-contiguous.  Forward uses the [out][in]
//                  image of the weight (kept beside the [in][out] image by the optimizer), dgrad uses the [in][out]
//                  image as it is: dX[m][k] = sum_n dY[m][n] W[k][n].
//   gemm_bf16_tn : C[P][Q] = sum_r A[r][P] . B[r][Q] - both operands contraction-STRIDED (weight gradient
//                  dW = X^T dY over the B*L rows).  The tiles go to LDS as they lie in memory and the MFMA operands are
//                  read with ds_read_b64_tr_b16 (gfx950's transposing LDS read: a 16-lane group fetches a [4][16] block
//                  and every lane receives one column of it), so nothing is ever transposed in HBM or in registers.
//                  Split over r into fp32 partial tiles (+ column sums of B = the bias gradient); the existing split-K
//                  reduction sums them into the fp32 gradient buffer.
//
// Tiles: 128 x 128 outputs per 256-thread workgroup (2 x 2 waves, 2 x 2 MFMA tiles of 32 x 32 per wave), 64-deep
// contraction steps, register-staged double-buffered LDS (one barrier per step), XCD-contiguous tile order with the
// tiles that share an A panel adjacent.  nt computes C^T tiles (mfma(B, A)) so that a lane owns 4 consecutive columns of
// one output row: bias / relu-mask / accumulate operands and the bf16 result move as 8-byte pieces.
#include <stdlib.h>
#include "skf_common.h"
#include "skf_bf16.h"

extern "C" int skf_splitk_reduce(const float* slab, int splits, int M, int N, float* C, int ldc, int accumulate, float* bias_grad,
                                 int bias_grad_accumulate, skf_stream_t stream);

namespace {

struct NtParams {
  const skf_bf16* A; const skf_bf16* B; skf_bf16* C;
  int M, N, K, lda, ldb, ldc;
  const float* bias;            // [N] fp32 or null
  int act;                      // 0 none, 1 relu, 2 tanh
  const skf_bf16* relu_src;     // [M][ld_relu]: C = 0 where relu_src <= 0 (dgrad through a relu), or null
  int ld_relu;
  int accumulate;               // C += result (read-modify-write in bf16)
  float* C32; int ldc32;        // optional fp32 copy of the result (used for the small fp32 heads), or null
  int tiles_m, tiles_n;
  int wide_c;                   // C (and relu_src) rows are 16-byte aligned: the epilogue moves 16-byte pieces
  const int* row_blocks;        // live-ROW list (skf_row_blocks_build with granule 1: {n_live, M, live rows, dead rows, flags}) or null:
  int zero_dead;                //   the m tiles run over the compacted live rows; dead rows of C are zero-filled if zero_dead
  // ReLU sign bits, row-major: bit (n & 7) of byte [m][n >> 3] (row pitch ld_bits bytes) = "C[m][n] > 0".  A relu forward launch
  // writes them (bits_out), the input-gradient launch reads them (bits_in) instead of the 16-fold larger hidden tensor.
  unsigned char* bits_out; const unsigned char* bits_in; int ld_bits;
};

// bit e of the result = element e of the 8 packed bf16 is > 0 (after a relu a stored value is +0 or positive)
__device__ __forceinline__ unsigned nt_sign_byte(const uint4& w) {
  const unsigned d[4] = {w.x, w.y, w.z, w.w};
  unsigned m = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    m |= ((d[j] & 0x7fffu) != 0u && !(d[j] & 0x8000u)) ? (1u << (2 * j)) : 0u;
    m |= ((d[j] & 0x7fff0000u) != 0u && !(d[j] & 0x80000000u)) ? (2u << (2 * j)) : 0u;
  }
  return m;
}

// ---- LDS images.  nt: a tile row = 64 bf16 = 128 B; two rows share a 256-byte super-row and the 16-byte chunk slot is
// XOR-ed with the super-row index, so the 16 lanes of a ds_read_b128 group (16 consecutive rows, one logical chunk) hit
// 16 different bank groups.
template <int BK>
__device__ __forceinline__ int nt_lds_off(int row, int chunk) {       // byte offset of 16-byte chunk `chunk` of tile row `row`
  constexpr int CPR = BK / 8, RPS = 16 / CPR;
  const int s = row / RPS, h = row % RPS;
  return s * 256 + (((h * CPR + chunk) ^ (s & 15)) << 4);
}

// DMA: the tiles travel global -> LDS directly (global_load_lds_dwordx4: the LDS address of a wave instruction is a uniform
// base + 16 B per lane, i.e. a linear 1-KB piece = 4 super-rows, so the chunk swizzle is applied on the SOURCE side: lane l of
// piece P fetches the logical chunk whose swizzled slot is l).  No staging registers, no ds_write pass; needs K % 64 == 0
// (a DMA cannot zero-fill a partial step) - other K take the register-staged variant.
// The loop's barrier also waits for the loads of the NEXT step (vmcnt(0) before s_barrier).  Measured (M = 65536, cfg-5
// shapes): a step costs ~1.9 us whatever the tile, i.e. the K loop runs at the ~14 B/cycle/CU the fabric delivers to a CU when
// all 256 fetch at once - more resident workgroups of a smaller step (BK = 32, four per CU) bought nothing; fewer bytes per
// flop do:
// BIG: 256 x 256 outputs per 512-thread workgroup (2 x 4 waves, 4 x 2 MFMA tiles per wave), 128 KB of LDS, one workgroup per
// CU.  What a step can overlap with its own memory round trip is the MFMA work of the resident workgroups, and per LDS
// byte that is proportional to BM BN / (BM + BN): the large tile doubles it (and halves the L2 -> LDS traffic per flop).
// EXTRA: relu_src / accumulate / fp32 copy, TANH: act == 2 - separate instantiations keep the plain epilogue lean (unrolled
// over the 32 pieces of a lane, a runtime tanh branch made it 50 KB of code that every piece jumps across)
// PH8 (round 5, BIG only): the K loop as FOUR PHASES per 64-deep step (cdna_hip_programming.md section 5, "the 256^2 8-phase template"; the
// schedule below is this file's own - the guide's example source is not in the image).  A phase = one quadrant of the wave's 128 x 64
// block (2 m tiles x 1 n tile x 4 k steps = 8 MFMAs, 256 matrix-pipe cycles) between two raw s_barriers:
//     [fragment reads of the quadrant | 2 DMA pieces of ONE half-tile, six half-tiles ahead | counted s_waitcnt vmcnt(8)]  s_barrier
//     [s_waitcnt lgkmcnt(0) | s_setprio 1 | 8 MFMAs | s_setprio 0]                                                        s_barrier
// The waves of the second wave row run ONE barrier behind the first (an extra s_barrier in front of the loop, its twin behind it): on
// every SIMD one wave multiplies while the other reads and issues - the role split that s_setprio then arbitrates.  The loads of a
// K step are never drained: a half-tile (A rows of one m half, or B rows of one n half, of ALL waves: 16 KB = 2 DMA pieces per wave)
// is requested 6 phases before the phase that first reads it, retired by a counted wait 2 phases before that read (data becomes
// readable one phase after the wait that retires it: the barrier in between publishes the other waves' pieces), and overwrites a
// region whose last reader finished >= 2 phases earlier.  Order of the quadrants (0,0) (0,1) (1,1) (1,0) with BOTH n halves of B kept
// in registers: phase 1 reads A(m half 0) + B(n half 0), phase 2 B(n half 1), phase 3 A(m half 1), phase 4 nothing.
#ifndef SKF_PH8_ABLATE
#define SKF_PH8_ABLATE 0      // timing experiments (results wrong): 1 no stagger, 2 no s_setprio, 4 no fragment reads, 8 no DMA, 16 no MFMA
#endif
template <bool EXTRA, bool DMA, int BK, bool BIG, bool TANH, bool PH8 = false>
__global__ __launch_bounds__(BIG ? 512 : 256, 2) void gemm_bf16_nt_kernel(NtParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WN = BIG ? 4 : 2, MT = BIG ? 4 : 2, NT = 2, NW = 2 * WN;    // waves along n, MFMA tiles per wave, waves
  constexpr int BM = 2 * MT * 32, BN = WN * NT * 32;
  constexpr int CPR = BK / 8;                 // 16-byte chunks per tile row
  constexpr int TBA = BM * BK * 2, TBB = BN * BK * 2, TB2 = TBA + TBB;      // bytes of the tile images / of one buffer
  constexpr int NJ = TBA / 1024 / NW;         // DMA pieces per wave and tile
  static_assert(BM == BN, "the staging below assumes equally tall A and B tiles");
  static_assert(DMA || (BK == 64 && !BIG), "the register-staged variant is written for 128 x 128 tiles and 64-deep steps");
  static_assert(!PH8 || (BIG && DMA && BK == 64), "the phased K loop is written for the 256 x 256 DMA tile");
  // buffer b of the A / B tile images: A at b * TB2, B TBA behind it
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // XCD-contiguous logical ids, n fastest: the tiles_n workgroups that share an A panel run back to back on one XCD
  // (with a live-row list the computing tiles are the first few logical ids: XCD-contiguous ids would put them all on one
  //  or two XCDs - measured 129 us instead of ~80 with 82 live workgroups - so there the ids stay round-robin)
  const int lid = p.row_blocks ? (int)blockIdx.x : skf_xcd_remap(blockIdx.x, gridDim.x);
  const int tm = lid / p.tiles_n, tn = lid % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  // Live-row list: tile rows are positions of the list (live rows first, then the dead ones), prow() maps a position to its
  // row of A / C / relu_src.  A tile that starts behind the live rows has nothing to compute: its rows of C become zeros.
  const int* rl = p.row_blocks;
  const int nlive = rl ? rl[0] : p.M;
  auto prow = [&](int m) -> int { return rl ? rl[2 + m] : m; };
  if (m0 >= nlive) {
    if (p.zero_dead && !p.accumulate) {
      constexpr int CPRow = BN / 8, NTH = BIG ? 512 : 256, RPI = NTH / CPRow, NIT = BM / RPI;
      const int c8 = (threadIdx.x % CPRow) * 8, r0 = threadIdx.x / CPRow, n = n0 + c8;
      int pm[NIT];                                       // the row ids first, in one batch (not one load round trip per row)
#pragma unroll
      for (int it = 0; it < NIT; ++it) { const int m = m0 + r0 + it * RPI; pm[it] = m < p.M ? prow(m) : -1; }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (pm[it] < 0 || n >= p.N) continue;
        skf_bf16* cp = p.C + (size_t)pm[it] * p.ldc + n;
        if (p.wide_c && n + 8 <= p.N) *reinterpret_cast<uint4*>(cp) = make_uint4(0, 0, 0, 0);
        else {
          *reinterpret_cast<uint2*>(cp) = make_uint2(0, 0);
          if (n + 8 <= p.N) *reinterpret_cast<uint2*>(cp + 4) = make_uint2(0, 0);
        }
        if (EXTRA && p.C32) {
          float* c32 = p.C32 + (size_t)pm[it] * p.ldc32 + n;
          *reinterpret_cast<float4*>(c32) = make_float4(0.f, 0.f, 0.f, 0.f);
          if (n + 8 <= p.N) *reinterpret_cast<float4*>(c32 + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    return;
  }
  const int kchunks = (p.K + 7) >> 3;                 // 16-byte chunks along K (row pitches are padded to 8 elements)
  const int nk = (p.K + BK - 1) / BK;

  // staging: chunk id = tid + 256 j -> row = id / 8, chunk = id % 8 (8 consecutive threads = one 128-byte row segment)
  const int srow = tid >> 3, sch = tid & 7;
  const skf_bf16* ag[4]; const skf_bf16* bg[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = srow + 32 * j;
    const int ar = prow(min(m0 + r, p.M - 1)), br = min(n0 + r, p.N - 1);     // rows past the edge re-read the last row (never stored)
    ag[j] = p.A + (size_t)ar * p.lda + sch * 8;
    bg[j] = p.B + (size_t)br * p.ldb + sch * 8;
  }
  uint4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const bool ok = kt * 8 + sch < kchunks;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ra[j] = ok ? *reinterpret_cast<const uint4*>(ag[j] + kt * 64) : make_uint4(0, 0, 0, 0);
      rb[j] = ok ? *reinterpret_cast<const uint4*>(bg[j] + kt * 64) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int off = nt_lds_off<BK>(srow + 32 * j, sch);
      *reinterpret_cast<uint4*>((smem + buf * TB2) + off) = ra[j];
      *reinterpret_cast<uint4*>((smem + buf * TB2 + TBA) + off) = rb[j];
    }
  };

  f32x16 acc[NT][MT];      // [n tile][m tile] of C^T
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // DMA addressing: wave w moves pieces 4w .. 4w+3 of each tile; lane l -> slot l of the piece = super-row S = 4*piece + l/16,
  // slot q = l % 16 -> logical (h, chunk) = q ^ (S & 15), row = 2 S + h
  const skf_bf16* da[NJ]; const skf_bf16* db[NJ];
  if constexpr (DMA) {
#pragma unroll
    for (int j2 = 0; j2 < NJ; ++j2) {
      const int S = 4 * (NJ * wave + j2) + (lane >> 4), hc = (lane & 15) ^ (S & 15), r = (16 / CPR) * S + hc / CPR, c = hc % CPR;
      da[j2] = p.A + (size_t)prow(min(m0 + r, p.M - 1)) * p.lda + c * 8;
      db[j2] = p.B + (size_t)min(n0 + r, p.N - 1) * p.ldb + c * 8;
    }
  }
  auto dma = [&](int kt, int buf) {
#pragma unroll
    for (int j2 = 0; j2 < NJ; ++j2) {
      char* la = smem + buf * TB2 + (NJ * wave + j2) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(da[j2] + kt * BK),
                                       (__attribute__((address_space(3))) void*)la, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db[j2] + kt * BK),
                                       (__attribute__((address_space(3))) void*)(la + TBA), 16, 0, 0);
    }
  };
  const int lrow = lane & 31, lhi = lane >> 5;
  if constexpr (PH8) {
    // ---- half-tile DMA.  kind 0 = A rows of m half 0 (of both wave rows), 1 = B rows of n half 0 (of the four wave columns), 2 = B n half 1,
    // 3 = A m half 1: the order in which a K step first reads them.  A wave moves pieces q = 2 wave + {0,1} of the 16 of a half-tile.
    const skf_bf16* sp[4][2];
    int spo[4][2];                                   // byte offset of the piece in its buffer (wave-uniform)
#pragma unroll
    for (int kind = 0; kind < 4; ++kind)
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int q = 2 * wave + j2, is_a = kind == 0 || kind == 3, half = kind >> 1;     // (kind 0: m half 0, 3: m half 1; 1: n half 0, 2: n half 1)
        const int piece = is_a ? (q < 8 ? 8 * half + q : 16 + 8 * half + (q - 8)) : 8 * (q >> 2) + 4 * half + (q & 3);
        const int S = 4 * piece + (lane >> 4), hc = (lane & 15) ^ (S & 15), r = 2 * S + (hc >> 3), c = hc & 7;
        sp[kind][j2] = is_a ? p.A + (size_t)prow(min(m0 + r, p.M - 1)) * p.lda + c * 8 : p.B + (size_t)min(n0 + r, p.N - 1) * p.ldb + c * 8;
        spo[kind][j2] = (is_a ? 0 : TBA) + piece * 1024;
      }
    const int n_stage = 4 * nk;
    auto stage = [&](int kind, int kts) {            // half-tile `kind` of K step kts -> buffer kts & 1
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sp[kind][j2] + kts * BK),
                                         (__attribute__((address_space(3))) void*)(smem + (kts & 1) * TB2 + spo[kind][j2]), 16, 0, 0);
    };
    // fragment addresses: tile row lrow, chunk 2 ks + lhi -> the XOR-ed slot depends on the lane and on ks only (tile bases are multiples of 32 rows)
    int fo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fo[ks] = nt_lds_off<BK>(lrow, ks * 2 + lhi);
    const int a_base = wm * (MT * 32) * 128, b_base = TBA + wn * (NT * 32) * 128;     // 128 bytes per tile row
    skf_bf16x8 af[2][4], bf[2][4];
    // the first six half-tiles (the loop keeps requesting six ahead); the first K step's A / B halves 0 must have landed
#pragma unroll
    for (int s0 = 0; s0 < 6; ++s0)
      if (s0 < n_stage) stage(s0 & 3, s0 >> 2);
    if (n_stage > 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1 && !(SKF_PH8_ABLATE & 1)) __builtin_amdgcn_s_barrier();       // the second wave row runs one barrier behind
    for (int kt = 0; kt < nk; ++kt) {
      const char* At = smem + (kt & 1) * TB2 + a_base;
      const char* Bt = smem + (kt & 1) * TB2 + b_base;
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int mh = ph >> 1, nh = (ph == 1 || ph == 2) ? 1 : 0;
        if ((ph == 0 || ph == 1) && !((SKF_PH8_ABLATE & 4) && kt > 0)) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) bf[nh][ks] = *reinterpret_cast<const skf_bf16x8*>(Bt + nh * 32 * 128 + fo[ks]);
        }
        if ((ph == 0 || ph == 2) && !((SKF_PH8_ABLATE & 4) && kt > 0)) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) af[t][ks] = *reinterpret_cast<const skf_bf16x8*>(At + (2 * mh + t) * 32 * 128 + fo[ks]);
        }
        const int sn = 4 * kt + ph + 6;               // the half-tile requested in this phase: kind (ph + 2) & 3 of K step kt + 1 (ph < 2) / kt + 2
        if (sn < n_stage && !(SKF_PH8_ABLATE & 8)) { stage((ph + 2) & 3, kt + (ph < 2 ? 1 : 2)); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(SKF_PH8_ABLATE & 2)) __builtin_amdgcn_s_setprio(1);
        if (!(SKF_PH8_ABLATE & 16)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[nh][2 * mh + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[nh][ks], af[t][ks], acc[nh][2 * mh + t], 0, 0, 0);
        }
        if (!(SKF_PH8_ABLATE & 2)) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
    }
    if (wm == 0 && !(SKF_PH8_ABLATE & 1)) __builtin_amdgcn_s_barrier();       // (the twin of the extra barrier in front of the loop)
    __syncthreads();                                  // the tile buffers are dead: the epilogue's images may overwrite them
  } else {
  if constexpr (DMA) {
    dma(0, 0);
  } else {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if constexpr (DMA) { if (kt + 1 < nk) dma(kt + 1, cur ^ 1); }
    else { if (kt + 1 < nk) gload(kt + 1); }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      skf_bf16x8 af[MT], bf[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
        bf[t] = *reinterpret_cast<const skf_bf16x8*>((smem + cur * TB2 + TBA) + nt_lds_off<BK>(wn * (NT * 32) + t * 32 + lrow, ks * 2 + lhi));
#pragma unroll
      for (int t = 0; t < MT; ++t)
        af[t] = *reinterpret_cast<const skf_bf16x8*>((smem + cur * TB2) + nt_lds_off<BK>(wm * (MT * 32) + t * 32 + lrow, ks * 2 + lhi));
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[a], af[b], acc[a][b], 0, 0, 0);
    }
    if constexpr (!DMA) { if (kt + 1 < nk) lstore(cur ^ 1); }
    __syncthreads();
  }
  }

  // ---- epilogue.  acc[a][b][r]: n = n0 + wn*64 + a*32 + (r&3) + 8*(r>>2) + 4*lhi, m = m0 + wm*(MT*32) + b*32 + lrow.
  // A lane owns 4 consecutive columns of 32 different rows: stored from there, a wave instruction touches 32 cache lines
  // with 16 bytes each and the store tail is issue-bound (measured: ~8 us of a 30-us workgroup at K = 512).  So the
  // [MT*32][64] block of a wave goes through a wave-private LDS image (the staging buffers are dead after the loop's last
  // barrier; 144-byte pitch: the 8-byte writes of 32 rows and the 16-byte reads of 8 rows both spread over all banks) and
  // leaves as full 128-byte row segments, 8 rows per instruction.  bias / activation are applied on the way in (fp32),
  // the relu mask and the accumulate operand on the way out (16-byte coalesced reads; the sum is formed in fp32 from the
  // bf16-rounded product).  The fp32 copy needs the unrounded values: that (small-head) case keeps the direct stores.
  constexpr int EP = 144;
  const bool direct = EXTRA && p.C32;
  // the lane's 8 bias pieces in ONE batch of loads (fetched piece by piece inside the branches below, each was a full,
  // serialised L2 round trip: 32 of them = 6 us per workgroup)
  const float act_floor = p.act == 1 ? 0.f : -__builtin_inff();
  float4 bias_r[NT][4];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wn * (NT * 32) + a * 32 + 8 * q + 4 * lhi;
      bias_r[a][q] = (p.bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  char* ew = smem + wave * (MT * 32 * EP);
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int m = m0 + wm * (MT * 32) + b * 32 + lrow;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = a * 32 + 8 * q + 4 * lhi, n = n0 + wn * (NT * 32) + nl;
        const bool in = m < p.M && n < p.N;              // N and the pitches are multiples of 4: a piece is all in or all out
        float v[4] = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
        v[0] += bias_r[a][q].x; v[1] += bias_r[a][q].y; v[2] += bias_r[a][q].z; v[3] += bias_r[a][q].w;
        if constexpr (TANH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], act_floor);      // relu, or a no-op floor
        }
        if constexpr (EXTRA) {
          if (direct) {
            if (!in) continue;
            const int pm = prow(m);
            if (p.relu_src) {
              const uint2 hv = *reinterpret_cast<const uint2*>(p.relu_src + (size_t)pm * p.ld_relu + n);
              float h[4]; skf_unpack4(hv, h);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = h[e] > 0.f ? v[e] : 0.f;
            }
            if (p.accumulate) {
              const uint2 ov = *reinterpret_cast<const uint2*>(p.C + (size_t)pm * p.ldc + n);
              float o[4]; skf_unpack4(ov, o);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += o[e];
            }
            *reinterpret_cast<float4*>(p.C32 + (size_t)pm * p.ldc32 + n) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<uint2*>(p.C + (size_t)pm * p.ldc + n) = skf_pack4(v);
            continue;
          }
        }
        *reinterpret_cast<uint2*>(ew + (b * 32 + lrow) * EP + nl * 2) = skf_pack4(v);
      }
  }
  if (direct) return;
  // way out: lane -> row it*8 + lane/8, 16-byte chunk lane%8 (wave-private image: no workgroup barrier, the compiler's
  // lgkmcnt wait orders the ds_write / ds_read pair of one wave)
  const int orow = lane >> 3, och = lane & 7;
  const int n = n0 + wn * (NT * 32) + och * 8;
  const bool wide = p.wide_c;
  if (wide && m0 + wm * (MT * 32) + MT * 32 <= p.M && n + 8 <= p.N) {
    // interior block (every tile of the cfg-5 shapes): no per-row conditions, and the mask / accumulate operands of all
    // MT*4 rows of the lane are requested in one batch before the first is used (one memory round trip, not MT*4)
    int pr[MT * 4];                                      // rows of C (and of the mask) behind the lane's MT*4 tile rows
#pragma unroll
    for (int it = 0; it < MT * 4; ++it) pr[it] = prow(m0 + wm * (MT * 32) + it * 8 + orow);
    if constexpr (EXTRA) {
      uint4 hv[MT * 4], ov[MT * 4];
      unsigned bv[MT * 4];
      if (p.bits_in) {
#pragma unroll
        for (int it = 0; it < MT * 4; ++it) bv[it] = p.bits_in[(size_t)pr[it] * p.ld_bits + (n >> 3)];
      } else if (p.relu_src) {
#pragma unroll
        for (int it = 0; it < MT * 4; ++it) hv[it] = *reinterpret_cast<const uint4*>(p.relu_src + (size_t)pr[it] * p.ld_relu + n);
      }
      if (p.accumulate) {
#pragma unroll
        for (int it = 0; it < MT * 4; ++it) ov[it] = *reinterpret_cast<const uint4*>(p.C + (size_t)pr[it] * p.ldc + n);
      }
#pragma unroll
      for (int it = 0; it < MT * 4; ++it) {
        uint4 w = *reinterpret_cast<const uint4*>(ew + (it * 8 + orow) * EP + och * 16);
        float v[8];
        skf_unpack8(w, v);
        if (p.bits_in) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ((bv[it] >> e) & 1u) ? v[e] : 0.f;
        } else if (p.relu_src) {
          float h[8]; skf_unpack8(hv[it], h);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = h[e] > 0.f ? v[e] : 0.f;
        }
        if (p.accumulate) {
          float o[8]; skf_unpack8(ov[it], o);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += o[e];
        }
        *reinterpret_cast<uint4*>(p.C + (size_t)pr[it] * p.ldc + n) = skf_pack8(v);
      }
    } else {
#pragma unroll
      for (int it = 0; it < MT * 4; ++it) {
        const uint4 w = *reinterpret_cast<const uint4*>(ew + (it * 8 + orow) * EP + och * 16);
        *reinterpret_cast<uint4*>(p.C + (size_t)pr[it] * p.ldc + n) = w;
        if (p.bits_out) p.bits_out[(size_t)pr[it] * p.ld_bits + (n >> 3)] = (unsigned char)nt_sign_byte(w);
      }
    }
    return;
  }
#pragma unroll 1
  for (int it = 0; it < MT * 4; ++it) {
    const int rloc = it * 8 + orow, m = m0 + wm * (MT * 32) + rloc;
    if (m >= p.M || n >= p.N) continue;
    const int pm = prow(m);
    uint4 w = *reinterpret_cast<const uint4*>(ew + rloc * EP + och * 16);
    const bool full = n + 8 <= p.N;                      // else the chunk's first 4 columns only
    skf_bf16* cp = p.C + (size_t)pm * p.ldc + n;
    if (!EXTRA && p.bits_out) p.bits_out[(size_t)pm * p.ld_bits + (n >> 3)] = (unsigned char)nt_sign_byte(w);   // (columns past N: relu(0 + 0) = 0)
    if constexpr (EXTRA) {
      float v[8];
      skf_unpack8(w, v);
      if (p.bits_in) {
        const unsigned bvv = p.bits_in[(size_t)pm * p.ld_bits + (n >> 3)];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ((bvv >> e) & 1u) ? v[e] : 0.f;
      } else if (p.relu_src) {
        const skf_bf16* hp = p.relu_src + (size_t)pm * p.ld_relu + n;
        uint4 hv;
        if (wide && full) hv = *reinterpret_cast<const uint4*>(hp);
        else {
          const uint2 h0 = *reinterpret_cast<const uint2*>(hp), h1 = full ? *reinterpret_cast<const uint2*>(hp + 4) : make_uint2(0, 0);
          hv = make_uint4(h0.x, h0.y, h1.x, h1.y);
        }
        float h[8]; skf_unpack8(hv, h);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = h[e] > 0.f ? v[e] : 0.f;
      }
      if (p.accumulate) {
        uint4 ov;
        if (wide && full) ov = *reinterpret_cast<const uint4*>(cp);
        else {
          const uint2 o0 = *reinterpret_cast<const uint2*>(cp), o1 = full ? *reinterpret_cast<const uint2*>(cp + 4) : make_uint2(0, 0);
          ov = make_uint4(o0.x, o0.y, o1.x, o1.y);
        }
        float o[8]; skf_unpack8(ov, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += o[e];
      }
      w = skf_pack8(v);
    }
    if (wide && full) *reinterpret_cast<uint4*>(cp) = w;
    else {
      *reinterpret_cast<uint2*>(cp) = make_uint2(w.x, w.y);
      if (full) *reinterpret_cast<uint2*>(cp + 4) = make_uint2(w.z, w.w);
    }
  }
}

// ------------------------------------------------------------------ tn (weight gradient)
struct TnParams {
  const skf_bf16* A; const skf_bf16* B;   // A [R][P] (lda), B [R][Q] (ldb)
  int R, P, Q, lda, ldb;
  int r_chunk;                             // contraction rows per split (multiple of 64)
  float* slab;                             // [splits][P][Q] fp32 partial tiles
  float* colsum_slab;                      // [splits][Q] partial column sums of B, or null
  int tiles_p, tiles_q, splits;
  const int* row_blocks;                   // 256 x 256 kernel: live 64-row blocks of the contraction (dead ones are zero in B) or null
};

// tile row = 128 bf16 = 256 B = four 64-byte segments; segment slot XOR (row & 3): the four rows a transposing read
// touches land in four different bank groups
__device__ __forceinline__ int tn_lds_off(int row, int col) {          // byte offset of element (row, col), col multiple of 4
  const int seg = col >> 5;
  return row * 256 + (((seg ^ row) & 3) << 6) + ((col & 31) << 1);
}

__device__ __forceinline__ skf_bf16x8 tn_read_frag(const char* tile, int r0, int c0, int lane) {
  // operand fragment of v_mfma_f32_32x32x16_bf16 from a [r][c] image, transposed: lane l gets the 8 contraction values
  // r0 + 8*(l>>5) .. +7 of column c0 + (l & 31).  16-lane group g: columns c0 + 16*(g&1).., rows r0 + 8*(g>>1)..; inside a
  // group lane j addresses row (j >> 2) and the four columns 4*(j & 3).. of its block and receives column j of it.
  const int g = lane >> 4, j = lane & 15;
  const int row = r0 + 8 * (g >> 1) + (j >> 2), col = c0 + 16 * (g & 1) + 4 * (j & 3);
  typedef short s4 __attribute__((ext_vector_type(4)));
  const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(tile + tn_lds_off(row, col)));
  const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(tile + tn_lds_off(row + 4, col)));
  typedef short s8 __attribute__((ext_vector_type(8)));
  const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(skf_bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_tn_kernel(TnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // buffer b of the A / B tile images: A at b * 32 KB, B 16 KB behind it
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave >> 1, wq = wave & 1;
  // logical id = (split z, p tile, q tile), q fastest: the q tiles of one (z, p) share the A panel; XCD-contiguous ranges
  const int lid = skf_xcd_remap(blockIdx.x, gridDim.x);
  const int tq = lid % p.tiles_q, tp = (lid / p.tiles_q) % p.tiles_p, z = lid / (p.tiles_q * p.tiles_p);
  const int p0 = tp * 128, q0 = tq * 128;
  const int rbeg = z * p.r_chunk, rend = min(p.R, rbeg + p.r_chunk);
  const int nk = (rend - rbeg + 63) >> 6;

  // staging: chunk id = tid + 256 j -> row = id / 16, chunk16 = id % 16 (16 threads = one 256-byte row segment)
  const int srow = tid >> 4, sch = tid & 15;
  const bool a_ok = p0 + sch * 8 < p.P, b_ok = q0 + sch * 8 < p.Q;      // P, Q multiples of 8: a chunk is all in or all out
  const skf_bf16* ag = p.A + (size_t)rbeg * p.lda + p0 + sch * 8;
  const skf_bf16* bg = p.B + (size_t)rbeg * p.ldb + q0 + sch * 8;
  uint4 ra[4], rb[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = kt * 64 + srow + 16 * j;
      const bool ok = rbeg + r < rend;
      ra[j] = (ok && a_ok) ? *reinterpret_cast<const uint4*>(ag + (size_t)r * p.lda) : make_uint4(0, 0, 0, 0);
      rb[j] = (ok && b_ok) ? *reinterpret_cast<const uint4*>(bg + (size_t)r * p.ldb) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int off = tn_lds_off(srow + 16 * j, sch * 8);
      *reinterpret_cast<uint4*>((smem + buf * 32768) + off) = ra[j];
      *reinterpret_cast<uint4*>((smem + buf * 32768 + 16384) + off) = rb[j];
    }
  };
  // bias gradient: the p == 0 tiles also sum the columns of their B tiles (thread: columns q0 + 8*sch .. +7)
  const bool do_colsum = p.colsum_slab && tp == 0;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto colsum_acc = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f[8];
      skf_unpack8(rb[j], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[e] += f[e];
    }
  };

  f32x16 acc[2][2];      // [p tile][q tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (nk > 0) {
    gload(0);
    if (do_colsum) colsum_acc();
    lstore(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) { gload(kt + 1); if (do_colsum) colsum_acc(); }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      skf_bf16x8 af[2], bf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        af[t] = tn_read_frag(smem + cur * 32768, ks * 16, wp * 64 + t * 32, lane);
        bf[t] = tn_read_frag(smem + cur * 32768 + 16384, ks * 16, wq * 64 + t * 32, lane);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }
  // acc[a][b][r]: row p = p0 + wp*64 + a*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), col q = q0 + wq*64 + b*32 + (lane&31)
  float* out = p.slab + (size_t)z * p.P * p.Q;
  const int lcol = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = q0 + wq * 64 + b * 32 + lcol;
      if (q >= p.Q) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pr = p0 + wp * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (pr < p.P) out[(size_t)pr * p.Q + q] = acc[a][b][r];
      }
    }
  if (do_colsum) {
    float* red = reinterpret_cast<float*>(smem);          // [16 row groups][128 columns]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[srow * 128 + sch * 8 + e] = cs[e];
    __syncthreads();
    if (tid < 128 && q0 + tid < p.Q) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) s += red[g * 128 + tid];
      p.colsum_slab[(size_t)z * p.Q + q0 + tid] = s;
    }
  }
}

// ---- 256 x 256 tiles, direct-to-LDS (the shapes of cfg 5).  Same reasoning as the nt kernel: what bounds the loop is the
// rate at which a CU's LDS can be filled (~15 B/cycle/CU measured), so the tile that needs half the bytes per flop runs
// nearly twice as fast (128 x 128: 576-604 TFLOP/s measured = that rate).  512 threads = 2 x 4 waves, 4 x 2 MFMA tiles per
// wave; a tile row is 256 bf16 = 512 B = eight 64-byte segments, slot = segment XOR (row & 3) in its low two bits.  The DMA
// moves linear 1-KB pieces (two tile rows): lane l lands at slot l % 32 of row 2 piece + l / 32, so it FETCHES the logical
// chunk that belongs there.  Needs whole 64-row steps (R % 64 == 0 - a DMA cannot zero-fill; garbage in the columns past
// P / Q is harmless, those outputs are never stored).
// Bias gradient: the column sums of the dY tile are one more MFMA per step with an all-ones A operand on the B fragments that
// are in registers anyway (p tile 0, waves wp == 0) - exact products, fp32 accumulation, no extra LDS or HBM traffic.
__device__ __forceinline__ int tnb_lds_off(int row, int col) {         // byte offset of element (row, col), col multiple of 4
  const int seg = col >> 5;
  return row * 512 + ((seg ^ (row & 3)) << 6) + ((col & 31) << 1);
}

__device__ __forceinline__ skf_bf16x8 tnb_read_frag(const char* tile, int r0, int c0, int lane) {
  const int g = lane >> 4, j = lane & 15;
  const int row = r0 + 8 * (g >> 1) + (j >> 2), col = c0 + 16 * (g & 1) + 4 * (j & 3);
  typedef short s4 __attribute__((ext_vector_type(4)));
  const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(tile + tnb_lds_off(row, col)));
  const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(tile + tnb_lds_off(row + 4, col)));
  typedef short s8 __attribute__((ext_vector_type(8)));
  const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(skf_bf16x8, v);
}

__global__ __launch_bounds__(512, 2) void gemm_bf16_tn_big_kernel(TnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // buffer b of the A / B tile images: A at b * 64 KB, B 32 KB behind it
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 2, wq = wave & 3;
  const int lid = skf_xcd_remap(blockIdx.x, gridDim.x);
  const int tq = lid % p.tiles_q, tp = (lid / p.tiles_q) % p.tiles_p, z = lid / (p.tiles_q * p.tiles_p);
  const int p0 = tp * 256, q0 = tq * 256;
  // contraction steps of this split: 64-row steps of [rbeg, rend), or (with a live-block list) the entries [eb, ee) of the
  // list - the dead 64-row blocks hold zeros in B (= dY), leaving them out is exact
  // (constant address space: the list is written by an earlier launch and every index is wave-uniform -> s_load; as plain global loads
  //  each step's lookup was `global_load_dword; s_waitcnt vmcnt(0)`, which also drained the tile DMA in flight)
  typedef const __attribute__((address_space(4))) int* const_i32p;
  const const_i32p blk = (const_i32p)p.row_blocks;
  int rbeg = z * p.r_chunk, nk = (min(p.R, rbeg + p.r_chunk) - rbeg) >> 6, eb = 0;
  if (blk) {
    const int nlive = blk[0], per = (nlive + p.splits - 1) / p.splits;
    eb = z * per; nk = max(min(nlive, eb + per) - eb, 0); rbeg = 0;
  }

  const skf_bf16* da[4]; const skf_bf16* db[4];
  const int qpad = (p.Q + 7) & ~7;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 2 * (4 * wave + j) + (lane >> 5), s = lane & 31;
    const int col = ((((s >> 2) ^ (row & 3)) << 2) | (s & 3)) * 8;
    da[j] = p.A + (size_t)(rbeg + row) * p.lda + min(p0 + col, p.P - 8);
    db[j] = p.B + (size_t)(rbeg + row) * p.ldb + min(q0 + col, qpad - 8);
  }
  auto dma = [&](int kt, int buf) {
    const size_t r64 = blk ? (size_t)blk[2 + eb + kt] : (size_t)kt;      // 64-row block of this step (relative to rbeg)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      char* la = smem + buf * 65536 + (4 * wave + j) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(da[j] + r64 * 64 * p.lda),
                                       (__attribute__((address_space(3))) void*)la, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db[j] + r64 * 64 * p.ldb),
                                       (__attribute__((address_space(3))) void*)(la + 32768), 16, 0, 0);
    }
  };

  f32x16 acc[4][2];      // [p tile][q tile]
  f32x16 cs[2];          // column sums of the B tiles (every row of the MFMA result holds the same sums)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int a = 0; a < 4; ++a) { acc[a][0][r] = 0.f; acc[a][1][r] = 0.f; }
    cs[0][r] = 0.f; cs[1][r] = 0.f;
  }
  const bool do_colsum = p.colsum_slab && tp == 0 && wp == 0;       // wave-uniform
  const skf_bf16x8 ones = __builtin_bit_cast(skf_bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));

  if (nk > 0) dma(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) dma(kt + 1, cur ^ 1);
    const char* At = smem + cur * 65536;
    const char* Bt = At + 32768;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      skf_bf16x8 af[4], bf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) bf[t] = tnb_read_frag(Bt, ks * 16, wq * 64 + t * 32, lane);
#pragma unroll
      for (int t = 0; t < 4; ++t) af[t] = tnb_read_frag(At, ks * 16, wp * 128 + t * 32, lane);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
      if (do_colsum) {
        cs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bf[0], cs[0], 0, 0, 0);
        cs[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bf[1], cs[1], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // acc[a][b][r]: row p = p0 + wp*128 + a*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), col q = q0 + wq*64 + b*32 + (lane&31)
  float* out = p.slab + (size_t)z * p.P * p.Q;
  const int lcol = lane & 31, lhi = lane >> 5;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int q = q0 + wq * 64 + b * 32 + lcol;
    if (q >= p.Q) continue;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pr = p0 + wp * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (pr < p.P) out[(size_t)pr * p.Q + q] = acc[a][b][r];
      }
    if (do_colsum && lhi == 0) p.colsum_slab[(size_t)z * p.Q + q] = cs[b][0];
  }
}

__host__ inline bool tn_use_big(int P, int Q, int R) {
  static const int tile_env = skf_knob("SKF_BF16_GEMM_TILE") ? atoi(skf_knob("SKF_BF16_GEMM_TILE")) : 0;
  return tile_env != 128 && (R & 63) == 0 && P >= 256 && Q >= 256 && R >= 16384;
}

template <typename K>
int set_smem(K kfn, size_t bytes) {
  SKF_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return SKF_OK;
}

}  // namespace

// rows per output tile the launcher will use for this problem (a live-row block list must be built for that height)
extern "C" int skf_gemm_bf16_tile_rows(int M, int N, int K, int act) {
  static const bool dma_off = skf_knob("SKF_BF16_GEMM_DMA") && skf_knob("SKF_BF16_GEMM_DMA")[0] == '0';
  static const int tile_env = skf_knob("SKF_BF16_GEMM_TILE") ? atoi(skf_knob("SKF_BF16_GEMM_TILE")) : 0;
  const bool dma = (K & 63) == 0 && !dma_off;
  // the 256 x 256 tile: one workgroup per CU, so only where there are enough tiles to fill the chip
  const bool big = dma && tile_env != 128 && act != 2 && N >= 256 && ((long)skf_cdiv(M, 256) * skf_cdiv(N, 256) >= 256 || tile_env == 256);
  return big ? 256 : 128;
}

extern "C" int skf_gemm_bf16(int M, int N, int K, const void* A, int lda, const void* B_nk, int ldb, void* C, int ldc,
                             const float* bias, int act, const void* relu_src, int ld_relu, int accumulate, float* C_f32,
                             int ldc_f32, skf_stream_t stream) {
  return skf_gemm_bf16_rows(M, N, K, A, lda, B_nk, ldb, C, ldc, bias, act, relu_src, ld_relu, accumulate, C_f32, ldc_f32, nullptr, 0, stream);
}

extern "C" int skf_gemm_bf16_rows(int M, int N, int K, const void* A, int lda, const void* B_nk, int ldb, void* C, int ldc,
                                  const float* bias, int act, const void* relu_src, int ld_relu, int accumulate, float* C_f32,
                                  int ldc_f32, const int* row_list, int zero_dead, skf_stream_t stream) {
  return skf_gemm_bf16_bits(M, N, K, A, lda, B_nk, ldb, C, ldc, bias, act, relu_src, ld_relu, accumulate, C_f32, ldc_f32, row_list, zero_dead,
                            nullptr, nullptr, 0, stream);
}

extern "C" size_t skf_gemm_bf16_relu_bits_bytes(int M, int N) { return (N & 7) ? 0 : (size_t)M * (N >> 3); }

extern "C" int skf_gemm_bf16_bits(int M, int N, int K, const void* A, int lda, const void* B_nk, int ldb, void* C, int ldc,
                                  const float* bias, int act, const void* relu_src, int ld_relu, int accumulate, float* C_f32,
                                  int ldc_f32, const int* row_list, int zero_dead, void* relu_bits_out, const void* relu_bits_in,
                                  int ld_bits, skf_stream_t stream) {
  if (relu_bits_out || relu_bits_in) {
    SKF_CHECK_ARG((N & 7) == 0 && ld_bits >= (N >> 3), "ReLU sign bits: N must be a multiple of 8 and the row pitch >= N / 8 bytes");
    SKF_CHECK_ARG(!relu_bits_out || (act == 1 && !relu_src && !accumulate && !C_f32), "sign bits are written by a plain relu forward launch");
    SKF_CHECK_ARG(!relu_bits_in || (act == 0 && !C_f32), "sign bits are read by an input-gradient launch (no activation, no fp32 copy)");
  }
  SKF_CHECK_ARG(!row_list || !bias, "a live-row list goes with the dgrad form (no bias: dead rows must come out as zeros)");
  SKF_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B_nk && C, "bad problem");
  SKF_CHECK_ARG(act >= 0 && act <= 2, "bad activation");
  SKF_CHECK_ARG((lda & 7) == 0 && (ldb & 7) == 0 && (ldc & 3) == 0 && (N & 3) == 0, "pitches must be multiples of 8 (A, B) / 4 (C, N) elements");
  SKF_CHECK_ARG(lda >= ((K + 7) & ~7) && ldb >= ((K + 7) & ~7), "operand rows must be padded to a multiple of 8 elements along K");
  SKF_CHECK_ARG((((uintptr_t)A | (uintptr_t)B_nk) & 15) == 0 && ((uintptr_t)C & 7) == 0, "operands must be 16-byte aligned");
  SKF_CHECK_ARG(!relu_src || ((ld_relu & 3) == 0 && ((uintptr_t)relu_src & 7) == 0), "relu source must be 8-byte aligned");
  SKF_CHECK_ARG(!C_f32 || ((ldc_f32 & 3) == 0 && ((uintptr_t)C_f32 & 15) == 0), "fp32 copy must be 16-byte aligned");
  SKF_CHECK_ARG(!bias || ((uintptr_t)bias & 15) == 0, "bias must be 16-byte aligned");
  NtParams p{};
  p.A = (const skf_bf16*)A; p.B = (const skf_bf16*)B_nk; p.C = (skf_bf16*)C;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.bias = bias; p.act = act; p.relu_src = (const skf_bf16*)relu_src; p.ld_relu = ld_relu; p.accumulate = accumulate;
  p.C32 = C_f32; p.ldc32 = ldc_f32;
  hipStream_t st = (hipStream_t)stream;
  p.bits_out = (unsigned char*)relu_bits_out; p.bits_in = (const unsigned char*)relu_bits_in; p.ld_bits = ld_bits;
  const bool extra = relu_src || accumulate || C_f32 || relu_bits_in;
  // SKF_BF16_GEMM_DMA=0: register-staged tiles everywhere; SKF_BF16_GEMM_TILE=128: no 256 x 256 tiles (measurement knobs)
  static const bool dma_off = skf_knob("SKF_BF16_GEMM_DMA") && skf_knob("SKF_BF16_GEMM_DMA")[0] == '0';
  const bool dma = (K & 63) == 0 && !dma_off;
  const bool big = skf_gemm_bf16_tile_rows(M, N, K, act) == 256;
  const int tile = big ? 256 : 128;
  p.tiles_m = skf_cdiv(M, tile); p.tiles_n = skf_cdiv(N, tile);
  p.row_blocks = row_list; p.zero_dead = zero_dead;
  p.wide_c = (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 && (!relu_src || ((ld_relu & 7) == 0 && ((uintptr_t)relu_src & 15) == 0));
  const size_t smem = big ? 8 * 128 * 144 : 65536;      // the epilogue's wave-private images (144-byte rows) exceed the 128 KB of tile buffers
  const double live = skf_prof_list_fraction(row_list);      // live rows of A that are loaded and multiplied (C is written in full)
  const double a_c = (double)M * K + (double)M * N * ((accumulate ? 1 : 0) + (relu_src && !relu_bits_in ? 1 : 0) + (relu_bits_in ? 1.0 / 16 : 0.0));
  SkfProfScope ps(st, "gemm_bf16_nt", 2.0 * M * N * K, 2.0 * (a_c + (double)N * K + (double)M * N));
  ps.done(2.0 * M * N * K * live, 2.0 * (a_c * live + (double)N * K + (double)M * N));
  int rc;
#define SKF_NT_GO(EX, DM, BG, TH)                                                                                            \
  {                                                                                                                          \
    if ((rc = set_smem(gemm_bf16_nt_kernel<EX, DM, 64, BG, TH>, smem))) return rc;                                           \
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<EX, DM, 64, BG, TH>), dim3(p.tiles_m * p.tiles_n), dim3(BG ? 512 : 256), smem, st, p); \
  }
  // the phased K loop (PH8) for the 256 x 256 tile with at least two 64-deep steps; SKF_BF16_GEMM_PH8=0 (measurement builds): the two-buffer loop
  static const bool ph8_off = skf_knob("SKF_BF16_GEMM_PH8") && skf_knob("SKF_BF16_GEMM_PH8")[0] == '0';
  const bool ph8 = big && K >= 128 && !ph8_off;
#define SKF_NT_GO8(EX)                                                                                                       \
  {                                                                                                                          \
    if ((rc = set_smem(gemm_bf16_nt_kernel<EX, true, 64, true, false, true>, smem))) return rc;                              \
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<EX, true, 64, true, false, true>), dim3(p.tiles_m * p.tiles_n), dim3(512), smem, st, p); \
  }
#define SKF_NT_GO2(EX, TH)                                                                                         \
  { if (ph8) SKF_NT_GO8(EX) else if (big) SKF_NT_GO(EX, true, true, false) else if (dma) SKF_NT_GO(EX, true, false, TH) else SKF_NT_GO(EX, false, false, TH) }
  if (act == 2) { if (extra) SKF_NT_GO2(true, true) else SKF_NT_GO2(false, true) }
  else { if (extra) SKF_NT_GO2(true, false) else SKF_NT_GO2(false, false) }
#undef SKF_NT_GO2
#undef SKF_NT_GO8
#undef SKF_NT_GO
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_gemm_bf16_wgrad_splits(int P, int Q, int R) {
  const bool big = tn_use_big(P, Q, R);
  const int tiles = big ? skf_cdiv(P, 256) * skf_cdiv(Q, 256) : skf_cdiv(P, 128) * skf_cdiv(Q, 128);
  // two 128 x 128 workgroups per CU, or ONE 256 x 256 - there a 257th workgroup would be a second round: round down
  int splits = big ? 256 / tiles : (512 + tiles - 1) / tiles;
  const int max_splits = skf_cdiv(R, 512);            // at least 8 contraction steps per split
  if (splits > max_splits) splits = max_splits;
  return splits < 1 ? 1 : splits;
}

extern "C" size_t skf_gemm_bf16_wgrad_workspace_bytes(int P, int Q, int R, int splits) {
  (void)R;
  if (splits < 1) splits = 1;
  return ((size_t)P * Q + Q) * (size_t)splits * sizeof(float);
}

// dW[P][Q] = sum_r X[r][P] dY[r][Q] (+ bias_grad[Q] = sum_r dY[r][Q]): partial tiles into `slab`
// ([splits][P][Q] then [splits][Q]); the caller reduces them (skf_splitk_reduce_batch / skf_gemm_bf16_wgrad).
extern "C" int skf_gemm_bf16_wgrad_partial(int P, int Q, int R, const void* X, int ldx, const void* dY, int lddy, int splits,
                                           int with_bias_grad, float* slab, size_t slab_bytes, int* splits_used_host,
                                           skf_stream_t stream) {
  return skf_gemm_bf16_wgrad_partial_rows(P, Q, R, X, ldx, dY, lddy, splits, with_bias_grad, slab, slab_bytes, splits_used_host, nullptr, stream);
}

extern "C" int skf_gemm_bf16_wgrad_partial_rows(int P, int Q, int R, const void* X, int ldx, const void* dY, int lddy, int splits,
                                                int with_bias_grad, float* slab, size_t slab_bytes, int* splits_used_host,
                                                const int* row_blocks_64, skf_stream_t stream) {
  SKF_CHECK_ARG(P > 0 && Q > 0 && R > 0 && X && dY && slab && splits_used_host, "bad argument");
  // Q may stop inside an 8-element chunk (vocabulary 1004): the rows of dY are padded to the pitch, pad columns are never stored
  SKF_CHECK_ARG((P & 7) == 0 && (Q & 3) == 0 && (ldx & 7) == 0 && (lddy & 7) == 0 && lddy >= ((Q + 7) & ~7),
                "P and the pitches must be multiples of 8 elements, Q a multiple of 4 with rows padded to 8");
  SKF_CHECK_ARG((((uintptr_t)X | (uintptr_t)dY) & 15) == 0 && ((uintptr_t)slab & 15) == 0, "operands must be 16-byte aligned");
  if (splits < 1) splits = 1;
  int chunk = skf_cdiv(R, splits);
  chunk = skf_cdiv(chunk, 64) * 64;
  splits = skf_cdiv(R, chunk);
  SKF_CHECK_ARG(slab_bytes >= skf_gemm_bf16_wgrad_workspace_bytes(P, Q, R, splits), "slab too small");
  TnParams p{};
  p.A = (const skf_bf16*)X; p.B = (const skf_bf16*)dY; p.R = R; p.P = P; p.Q = Q; p.lda = ldx; p.ldb = lddy;
  p.r_chunk = chunk; p.slab = slab; p.colsum_slab = with_bias_grad ? slab + (size_t)splits * P * Q : nullptr;
  const bool big = tn_use_big(P, Q, R);
  const int tile = big ? 256 : 128;
  p.tiles_p = skf_cdiv(P, tile); p.tiles_q = skf_cdiv(Q, tile); p.splits = splits;
  *splits_used_host = splits;
  hipStream_t st = (hipStream_t)stream;
  const size_t smem = big ? 131072 : 65536;
  int rc;
  p.row_blocks = big ? row_blocks_64 : nullptr;      // the 128 x 128 kernel contracts over every row
  const double live = skf_prof_list_fraction(p.row_blocks);
  SkfProfScope ps(st, "gemm_bf16_tn(wgrad)", 2.0 * P * Q * R, 2.0 * (double)R * (P + Q) + 4.0 * (double)splits * P * Q);
  ps.done(2.0 * P * Q * R * live, 2.0 * (double)R * (P + Q) * live + 4.0 * (double)splits * P * Q);
  if (big) {
    if ((rc = set_smem(gemm_bf16_tn_big_kernel, smem))) return rc;
    hipLaunchKernelGGL(gemm_bf16_tn_big_kernel, dim3(p.tiles_p * p.tiles_q * splits), dim3(512), smem, st, p);
  } else {
    if ((rc = set_smem(gemm_bf16_tn_kernel, smem))) return rc;
    hipLaunchKernelGGL(gemm_bf16_tn_kernel, dim3(p.tiles_p * p.tiles_q * splits), dim3(256), smem, st, p);
  }
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

// partial tiles + their reduction into the fp32 gradient: dW[P][Q] (row stride ldw) and bias_grad[Q] (may be NULL)
extern "C" int skf_gemm_bf16_wgrad(int P, int Q, int R, const void* X, int ldx, const void* dY, int lddy, float* dW, int ldw,
                                   float* bias_grad, void* workspace, size_t workspace_bytes, skf_stream_t stream) {
  return skf_gemm_bf16_wgrad_rows(P, Q, R, X, ldx, dY, lddy, dW, ldw, bias_grad, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int skf_gemm_bf16_wgrad_rows(int P, int Q, int R, const void* X, int ldx, const void* dY, int lddy, float* dW, int ldw,
                                        float* bias_grad, void* workspace, size_t workspace_bytes, const int* row_blocks_64,
                                        skf_stream_t stream) {
  SKF_CHECK_ARG(dW && workspace, "null operand");
  int splits = skf_gemm_bf16_wgrad_splits(P, Q, R), used = 0;
  while (splits > 1 && skf_gemm_bf16_wgrad_workspace_bytes(P, Q, R, splits) > workspace_bytes) --splits;
  int rc = skf_gemm_bf16_wgrad_partial_rows(P, Q, R, X, ldx, dY, lddy, splits, bias_grad != nullptr, (float*)workspace, workspace_bytes,
                                            &used, row_blocks_64, stream);
  if (rc) return rc;
  return skf_splitk_reduce((const float*)workspace, used, P, Q, dW, ldw, 0, bias_grad, 0, stream);
}
