// Dense weight-gradient (+ bias-gradient) kernel:  dW[Kin,Nout] = X[M,Kin]^T . dY[M,Nout],  M ~ 25k rows.
//
// Both operands are "row = contraction index" matrices, so MFMA fragments can be fed STRAIGHT from global
// memory with fully coalesced 16-byte loads and no LDS staging at all:
//   v_mfma_f32_16x16x4_f32 wants A[i][k=g], B[k=g][j=i] for lane (i = l&15, g = l>>4).  For 4 consecutive
//   rows m0..m0+3 lane (i,g) loads xa = X[m0+g][a0+4i..4i+3] and yb = dY[m0+g][b0+4i..4i+3] (one dwordx4
//   each; the 16 lanes of a group read 256 contiguous bytes).  MFMA (e,f) with a = xa[e], b = yb[f]
//   accumulates C_ef[i][j] = dW[a0+4i+e][b0+4j+f]: 16 independent MFMAs per pair of loads cover a 64x64
//   block of dW, and in the C layout (col = lane&15, row = 4g+r) every lane ends up with float4-contiguous
//   output columns (f = 0..3), so partial tiles are written with dwordx4 stores.
// A workgroup owns one 64x64 block of dW and a range of M; its 4 waves interleave 4-row steps of that
// range (intra-workgroup split-K) and are summed through LDS, so only ONE 16 KB partial per workgroup
// reaches the split-K slab (~4 MB per wgrad).  The bias gradient (column sums of dY) rides along.
#include "skf_common.h"
#include "skf_gemm_params.h"

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

#ifndef SKF_WGRAD_OCC
#define SKF_WGRAD_OCC 1  // workgroups per CU of the split-arithmetic kernels (2: the one-register-set loop below)
#endif
constexpr int UN = 4;    // 4-row steps per unrolled iteration (per wave): 16 rows
#ifndef SKF_WGRAD_PIPE
#define SKF_WGRAD_PIPE 0 // 1: software-pipelined split-arithmetic step (see wgrad_x_body) - measured SLOWER, kept as a build-time experiment
#endif

__global__ __launch_bounds__(256, 2) void wgrad_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][64][64] + [4][64] column sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  // all tiles of one M-range (they read the same X / dY rows) get consecutive logical ids -> same XCD / L2
  const int ntile = p.tiles_m * p.tiles_n;
  const int logical = skf_xcd_remap(blockIdx.x, gridDim.x);
  const int split = logical / ntile, tile = logical % ntile;
  const int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
  const int a0 = tile_m * 64, b0 = tile_n * 64;                   // block origin in dW
  const int kb = split * p.k_chunk, ke = min(p.K, kb + p.k_chunk);
  const bool aok = a0 + 4 * i < p.M, bok = b0 + 4 * i < p.N;      // M = Kin, N = Nout (multiples of 4)
  const float* xp = p.A + (aok ? a0 + 4 * i : 0);
  const float* yp = p.B + (bok ? b0 + 4 * i : 0);
  const bool do_colsum = p.colsum_slab != nullptr && tile_m == 0;

  f32x4 acc[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[e][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};

  // wave w takes rows kb + 16*(4*it + u) + 4*w + g  (u < UN)
  f32x4 xa[UN], yb[UN];
#define row_of(it_, u_) (kb + 16 * (UN * (it_) + (u_)) + 4 * wave + g)
  const int niter = (ke - kb + 16 * UN - 1) / (16 * UN);
#pragma unroll
  for (int u = 0; u < UN; ++u) {
    const int m = row_of(0, u);
    const int mc = m < ke ? m : kb;
    xa[u] = *reinterpret_cast<const f32x4*>(xp + (size_t)mc * p.lda);
    yb[u] = *reinterpret_cast<const f32x4*>(yp + (size_t)mc * p.ldb);
  }
  for (int it = 0; it < niter; ++it) {
    f32x4 xc[UN], yc[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bool ok = row_of(it, u) < ke;
      const float za = (ok && aok) ? 1.f : 0.f, zb = (ok && bok) ? 1.f : 0.f;
      xc[u] = xa[u] * za; yc[u] = yb[u] * zb;          // rows / columns past the end contribute exact zeros
    }
    if (it + 1 < niter) {
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int m = row_of(it + 1, u);
        const int mc = m < ke ? m : kb;
        xa[u] = *reinterpret_cast<const f32x4*>(xp + (size_t)mc * p.lda);
        yb[u] = *reinterpret_cast<const f32x4*>(yp + (size_t)mc * p.ldb);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[e][f] = mfma16(xc[u][e], yc[u][f], acc[e][f]);
      csum += yc[u];
    }
  }

#undef row_of
  // ---- intra-workgroup reduction through LDS in the final row-major [64][64] layout
  float* mine = smem + wave * 4096;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * g + 4 * r + e;                          // dW row a0+row
      *reinterpret_cast<f32x4*>(&mine[row * 64 + 4 * i]) =
          (f32x4){acc[e][0][r], acc[e][1][r], acc[e][2][r], acc[e][3][r]};
    }
  float* cs = smem + 4 * 4096;
  if (do_colsum) {
    // lanes with equal i hold the same columns for different rows g: reduce over g, then over waves in LDS
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = csum[c];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      csum[c] = v;
    }
    if (g == 0) *reinterpret_cast<f32x4*>(&cs[wave * 64 + 4 * i]) = csum;
  }
  __syncthreads();
  float* slab = p.slab + (size_t)split * p.M * p.N;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int e4 = tid + v * 256, row = e4 >> 4, c4 = (e4 & 15) * 4;
    const f32x4 s = *reinterpret_cast<const f32x4*>(&smem[row * 64 + c4]) +
                    *reinterpret_cast<const f32x4*>(&smem[4096 + row * 64 + c4]) +
                    *reinterpret_cast<const f32x4*>(&smem[8192 + row * 64 + c4]) +
                    *reinterpret_cast<const f32x4*>(&smem[12288 + row * 64 + c4]);
    if (a0 + row < p.M && b0 + c4 < p.N)
      *reinterpret_cast<f32x4*>(&slab[(size_t)(a0 + row) * p.N + b0 + c4]) = s;
  }
  if (do_colsum && tid < 64 && b0 + tid < p.N)
    p.colsum_slab[(size_t)split * p.N + b0 + tid] = cs[tid] + cs[64 + tid] + cs[128 + tid] + cs[192 + tid];
}

// ---- the same block decomposition on the bf16 matrix cores (see skf_gemm_wsx.hip for the exact three-piece split):
// one step = 32 rows.  Lane (i, g) loads X[m0 + 8g + j][a0 + 4i .. 4i+3] and dY[m0 + 8g + j][b0 + 4i .. 4i+3], j < 8
// (dwordx4, 256 contiguous bytes per 16 lanes); operand e of v_mfma_f32_16x16x32_bf16 is the column e of those
// eight rows, split in registers - still no LDS in the loop.  Rows outside [kb, ke) and columns outside the matrix
// come back as exact zeros from the buffer range check, so there is no masking arithmetic.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// column e of eight row vectors -> P operands of 8 bf16
template <int P>
__device__ __forceinline__ void wg_split_col(const f32x4 (&rows)[8], int e, u32x4 (&out)[P], const SkfSplitSel& sel) {
#ifdef SKF_WG_ABLATE_SPLIT   // diagnostics (wrong results): what operand PLANES handed over by the producers would leave of the split -
  // 1: one v_perm per pair and plane (row-major bf16 planes transposed in registers), 2: nothing (operand-ordered planes)
#pragma unroll
  for (int q = 0; q < P; ++q) {
#if SKF_WG_ABLATE_SPLIT == 1
#pragma unroll
    for (int d = 0; d < 4; ++d)
      out[q][d] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, rows[2 * d + 1][(e + q) & 3]), __builtin_bit_cast(unsigned, rows[2 * d][(e + q) & 3]), 0x07060302u);
#else
    out[q] = __builtin_bit_cast(u32x4, rows[(2 * e + q) & 7]);
#endif
  }
  return;
#endif
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    unsigned pc[P];
    skf_split2<P>(rows[2 * d][e], rows[2 * d + 1][e], pc, sel);
#pragma unroll
    for (int q = 0; q < P; ++q) out[q][d] = pc[q];
  }
}
// rows [row0, row_end) of a row-major matrix; 32-bit byte arithmetic (the launcher checks rows * ld * 4 < 2^31)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rows_rsrc(const float* base, int ld, int row0, int row_end) {
  const int rows_left = row_end - row0 > 0 ? row_end - row0 : 0;
  const unsigned rem = (unsigned)rows_left * (unsigned)ld * 4u;
  const unsigned off = rows_left > 0 ? (unsigned)row0 * (unsigned)ld * 4u : 0u;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base) + off), 0, rem, 0x00020000);
}

// bid / nblocks: the workgroup's index inside its problem and the problem's workgroup count (the kernel's own grid for a
// single problem; a slice of the grid that starts at a multiple of 8 - so that bid % 8 is still the XCD - in a grouped launch)
template <int P, int NW>   // NW waves per workgroup = intra-workgroup split of the row range
__device__ __forceinline__ void wgrad_x_body(const GemmParams& p, const int bid, const int nblocks, float* smem) {
  const SkfSplitSel sel = skf_split_sel();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int ntile = p.tiles_m * p.tiles_n;
  const int logical = skf_xcd_remap(bid, nblocks);
  const int split = logical / ntile, tile = logical % ntile;
  const int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
  const int a0 = tile_m * 64, b0 = tile_n * 64;
  const int kb = split * p.k_chunk, ke = min(p.K, kb + p.k_chunk);
  const bool aok = a0 + 4 * i < p.M, bok = b0 + 4 * i < p.N;
  const bool do_colsum = p.colsum_slab != nullptr && tile_m == 0;
  constexpr unsigned OOB = 0x7ffffff0u;
  // per-lane byte offsets of the eight rows of a step (row 8g + j of the wave's 32-row block)
  unsigned xo[8], yo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    xo[j] = aok ? (unsigned)((8 * g + j) * p.lda + a0 + 4 * i) * 4u : OOB;
    yo[j] = bok ? (unsigned)((8 * g + j) * p.ldb + b0 + 4 * i) * 4u : OOB;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[e][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};

  constexpr int RI = 32 * NW;                       // rows per iteration of the workgroup
  // With a row-block list (32-row blocks) the contraction runs over the LIVE blocks only: split z takes the list entries
  // [z * per, (z + 1) * per), wave w every NW-th of them - the rows of a dead block are zero in dY, so leaving them out is exact.
  // (constant address space: written by an earlier launch, wave-uniform indices -> s_load instead of global_load + s_waitcnt vmcnt(0),
  //  which also drained the operand rows requested a step ahead)
  typedef const __attribute__((address_space(4))) int* const_i32p;
  const const_i32p blk = (const_i32p)p.row_blocks;
  int eb = 0, ee = 0;
  if (blk) {
    const int nlive = blk[0], per = (nlive + nblocks / ntile - 1) / (nblocks / ntile);
    eb = split * per; ee = min(nlive, eb + per);
  }
  const int niter = blk ? (max(ee - eb, 0) + NW - 1) / NW : (ke - kb + RI - 1) / RI;
  // two register sets of operand rows, used alternately (no copies: the split works in place on the rows)
  f32x4 xa0[8], yb0[8], xa1[8], yb1[8];
  auto load_step = [&](int it, f32x4 (&xa)[8], f32x4 (&yb)[8]) {
    int r0, r1;                                     // the wave's rows of this iteration: [r0, r1)
    if (blk) {
      const int e = eb + NW * it + wave;
      r0 = e < ee ? blk[2 + e] * 32 : p.K; r1 = min(p.K, r0 + 32);
    } else {
      r0 = kb + RI * it + 32 * wave; r1 = ke;
    }
#ifdef SKF_WG_ABLATE_LOAD   // diagnostics: every step re-reads the first rows of its split (cache hits; wrong results)
    if (r0 < p.K) { r0 = kb + 32 * wave; r1 = ke; }
#endif
    const __amdgpu_buffer_rsrc_t rx = wg_rows_rsrc(p.A, p.lda, r0, r1);
    const __amdgpu_buffer_rsrc_t ry = wg_rows_rsrc(p.B, p.ldb, r0, r1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xa[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xo[j], 0, 0));
      yb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, yo[j], 0, 0));
    }
  };
  auto step = [&](const f32x4 (&xa)[8], const f32x4 (&yb)[8]) {
    u32x4 ax[4][P];
#pragma unroll
    for (int e = 0; e < 4; ++e) wg_split_col<P>(xa, e, ax[e], sel);
    if (do_colsum) {                                 // before the split: v_dot2c works in place, the rows die with it
#pragma unroll
      for (int j = 0; j < 8; ++j) csum += yb[j];
    }
    // the split of dY column f+1 is independent of the MFMAs of column f: written next to them so that the scheduler can
    // fill MFMA issue gaps with it (27.5 / 31.9 / 50.5 us instead of 29.1 / 33.8 / 54.4 for 128x384 / 128x512 / 128x1004;
    // also splitting the NEXT step's X operands there was slower again: 28.6 / 33.0 / 51.6)
    u32x4 by[4][P];
    wg_split_col<P>(yb, 0, by[0], sel);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (f < 3) wg_split_col<P>(yb, f + 1, by[f + 1], sel);
#pragma unroll
      for (int d = P - 1; d >= 0; --d)               // small products first
#pragma unroll
        for (int qa = 0; qa <= d; ++qa)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ax[e][qa]),
                                                                __builtin_bit_cast(bf16x8, by[f][d - qa]), acc[e][f], 0, 0, 0);
    }
  };
#if SKF_WGRAD_PIPE
  // MEASURED SLOWER (round 3: 34.5 / 33.2 / 29.4 / 18.1 us against 28.5 / 28.3 / 24.5 / 16.0 for 128x512 / 512x128 / 128x384 / 128x128,
  // 4.40 vs 4.15 ms/step): with every MFMA followed by 1-3 v_perm / v_dot2c the step is longer than with the compiler's own
  // arrangement (a VALU phase, then the MFMAs back to back) - 338 registers (82 of them AGPRs reached through v_accvgpr moves) and a
  // hazard s_nop at most MFMA <-> VALU transitions cost more than the overlap returns.  Off by default (-DSKF_WGRAD_PIPE=1 builds it).
  // Software-pipelined step (P = 3): the 112 VALU of the NEXT step's X split sit under the MFMAs of dY columns 2 and 3 of this step
  // (28 + 56 and 56 VALU for 24 MFMAs each, the interleave pinned with sched_group_barrier), the rows of step it + 2 are requested
  // in column 3, when both halves of their register set are dead.  Without it the step was 190 VALU, then 12 MFMAs with 3 VALU each,
  // 50 VALU, 78 MFMAs back to back = 1536 cycles of MFMA + ~1000 of VALU one after the other (tools/micro/valu_rates.hip: v_perm /
  // v_dot2c issue in ~4.5 cycles; tools/micro/mfma_bf16_valu_overlap.hip: about half of the VALU time hides under interleaved MFMAs).
  auto mfma_col = [&](const u32x4 (&ax)[4][P], const u32x4 (&byf)[P], int f) {
#pragma unroll
    for (int d = P - 1; d >= 0; --d)
#pragma unroll
      for (int qa = 0; qa <= d; ++qa)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ax[e][qa]), __builtin_bit_cast(bf16x8, byf[d - qa]), acc[e][f], 0, 0, 0);
  };
  // yb / xfree: the register set of this step (dY rows; its X rows were split a step ago and are the target of the loads for it + 2);
  // xn: X rows of step it + 1; ax: split X of this step; axn: receives the split X of step it + 1
  // pins a split result inside the scheduling region it was written in: instruction selection otherwise sinks the (side-effect free)
  // split next to its first use, i.e. behind the sched_barrier into the next step
  auto pin = [&](u32x4 (&v)[P]) {
#pragma unroll
    for (int q = 0; q < P; ++q) asm volatile("" : "+v"(v[q]));
  };
  auto pstep = [&](int it, f32x4 (&yb)[8], f32x4 (&xfree)[8], const f32x4 (&xn)[8], const u32x4 (&ax)[4][P], u32x4 (&axn)[4][P]) {
    if (do_colsum) {
#pragma unroll
      for (int j = 0; j < 8; ++j) csum += yb[j];
    }
    u32x4 by[4][P];
    wg_split_col<P>(yb, 0, by[0], sel);
    __builtin_amdgcn_sched_barrier(0);
    wg_split_col<P>(yb, 1, by[1], sel);
    mfma_col(ax, by[0], 0);
#pragma unroll
    for (int k = 0; k < 24; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); }
    __builtin_amdgcn_sched_barrier(0);
    wg_split_col<P>(yb, 2, by[2], sel);
    mfma_col(ax, by[1], 1);
#pragma unroll
    for (int k = 0; k < 24; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); }
    __builtin_amdgcn_sched_barrier(0);
    wg_split_col<P>(yb, 3, by[3], sel);
    wg_split_col<P>(xn, 0, axn[0], sel);
    wg_split_col<P>(xn, 1, axn[1], sel);
    mfma_col(ax, by[2], 2);
    pin(by[3]); pin(axn[0]); pin(axn[1]);
#pragma unroll
    for (int k = 0; k < 24; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
    __builtin_amdgcn_sched_barrier(0);
    load_step(it + 2, xfree, yb);                    // both halves of this set are dead now; past the end: empty descriptor, zeros
    wg_split_col<P>(xn, 2, axn[2], sel);
    wg_split_col<P>(xn, 3, axn[3], sel);
    mfma_col(ax, by[3], 3);
    pin(axn[2]); pin(axn[3]);
#pragma unroll
    for (int k = 0; k < 24; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    __builtin_amdgcn_sched_barrier(0);
  };
  u32x4 axA[4][P], axB[4][P];
  load_step(0, xa0, yb0);
  load_step(1, xa1, yb1);
#pragma unroll
  for (int e = 0; e < 4; ++e) wg_split_col<P>(xa0, e, axA[e], sel);
  for (int it = 0; it < niter; it += 2) {
    pstep(it, yb0, xa0, xa1, axA, axB);
    if (it + 1 >= niter) break;
    pstep(it + 1, yb1, xa1, xa0, axB, axA);
  }
#elif SKF_WGRAD_OCC == 2
  // Two workgroups per CU (<= 256 registers): ONE register set of operand rows.  The X rows of step it + 1 are requested as soon as
  // this step's X columns are split (the rows die with the split), the dY rows behind the split of the last dY column; the loads fly
  // under the step's MFMAs and under the other workgroup's wave on the same SIMD, whose prologue / epilogue (first-load latency, LDS
  // reduction, slab store) this wave covers in turn.
  auto rows_of = [&](int it, int& r0, int& r1) {
    if (blk) {
      const int e = eb + NW * it + wave;
      r0 = e < ee ? blk[2 + e] * 32 : p.K; r1 = min(p.K, r0 + 32);
    } else {
      r0 = kb + RI * it + 32 * wave; r1 = ke;
    }
  };
  load_step(0, xa0, yb0);
  for (int it = 0; it < niter; ++it) {
    int r0, r1;
    rows_of(it + 1, r0, r1);                         // past the end: empty descriptor, zeros
    const __amdgpu_buffer_rsrc_t rx = wg_rows_rsrc(p.A, p.lda, r0, r1);
    const __amdgpu_buffer_rsrc_t ry = wg_rows_rsrc(p.B, p.ldb, r0, r1);
    u32x4 ax[4][P];
#pragma unroll
    for (int e = 0; e < 4; ++e) wg_split_col<P>(xa0, e, ax[e], sel);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) xa0[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xo[j], 0, 0));
    if (do_colsum) {
#pragma unroll
      for (int j = 0; j < 8; ++j) csum += yb0[j];
    }
    u32x4 by[4][P];
    wg_split_col<P>(yb0, 0, by[0], sel);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (f < 3) wg_split_col<P>(yb0, f + 1, by[f + 1], sel);
      if (f == 3) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) yb0[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, yo[j], 0, 0));
      }
#pragma unroll
      for (int d = P - 1; d >= 0; --d)
#pragma unroll
        for (int qa = 0; qa <= d; ++qa)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ax[e][qa]),
                                                                __builtin_bit_cast(bf16x8, by[f][d - qa]), acc[e][f], 0, 0, 0);
    }
  }
#else
  load_step(0, xa0, yb0);
  for (int it = 0; it < niter; it += 2) {
    load_step(it + 1, xa1, yb1);                     // past the end: empty descriptor, zeros
    step(xa0, yb0);
    if (it + 1 >= niter) break;
    load_step(it + 2, xa0, yb0);
    step(xa1, yb1);
  }
#endif

  float* mine = smem + wave * 4096;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * g + 4 * r + e;
      *reinterpret_cast<f32x4*>(&mine[row * 64 + 4 * i]) =
          (f32x4){acc[e][0][r], acc[e][1][r], acc[e][2][r], acc[e][3][r]};
    }
  float* cs = smem + NW * 4096;
  if (do_colsum) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = csum[c];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      csum[c] = v;
    }
    if (g == 0) *reinterpret_cast<f32x4*>(&cs[wave * 64 + 4 * i]) = csum;
  }
  __syncthreads();
  float* slab = p.slab + (size_t)split * p.M * p.N;
#pragma unroll
  for (int v = 0; v < 16 / NW; ++v) {
    const int e4 = tid + v * 64 * NW, row = e4 >> 4, c4 = (e4 & 15) * 4;
    f32x4 s = *reinterpret_cast<const f32x4*>(&smem[row * 64 + c4]);
#pragma unroll
    for (int w = 1; w < NW; ++w) s += *reinterpret_cast<const f32x4*>(&smem[w * 4096 + row * 64 + c4]);
    if (a0 + row < p.M && b0 + c4 < p.N)
      *reinterpret_cast<f32x4*>(&slab[(size_t)(a0 + row) * p.N + b0 + c4]) = s;
  }
  if (do_colsum && tid < 64 && b0 + tid < p.N) {
    float v = cs[tid];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += cs[w * 64 + tid];
    p.colsum_slab[(size_t)split * p.N + b0 + tid] = v;
  }
}

template <int P, int NW>
__global__ __launch_bounds__(64 * NW, SKF_WGRAD_OCC) void wgrad_x_kernel(GemmParams p) {   // grids are ~one workgroup per CU: registers before occupancy
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [NW waves][64][64] + [NW][64] column sums
  wgrad_x_body<P, NW>(p, blockIdx.x, gridDim.x, smem);
}

// Several weight gradients in ONE launch (the 4-8 of a transformer layer, issued together on the side stream): problem g owns
// the workgroups [start[g], start[g + 1]) of the grid, every start a multiple of 8; the padding workgroups exit at once.
constexpr int kWgradGroupMax = 8;
struct WgradGroup { int n; int start[kWgradGroupMax + 1]; int nblocks[kWgradGroupMax]; GemmParams p[kWgradGroupMax]; };
template <int P, int NW>
__global__ __launch_bounds__(64 * NW, SKF_WGRAD_OCC) void wgrad_x_group_kernel(WgradGroup grp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int gi = 0;
  while (gi + 1 < grp.n && (int)blockIdx.x >= grp.start[gi + 1]) ++gi;       // wave-uniform scan of a short table
  const int bid = (int)blockIdx.x - grp.start[gi];
  if (bid >= grp.nblocks[gi]) return;
  wgrad_x_body<P, NW>(grp.p[gi], bid, grp.nblocks[gi], smem);
}

}  // namespace

// wgrad fast path: A = X stored [K=rows][M=Kin], B = dY stored [K=rows][N=Nout]; writes the split-K slab
// (the caller runs the slab reduction).  p.k_chunk must already be set (multiple of 64).
int skf_gemm_wgrad_dispatch(const GemmParams& p, int a_kcontig, int b_kcontig, int splits, hipStream_t st, int* handled) {
  *handled = 0;
  const char* off = skf_knob("SKF_GEMM_NO_WGRAD");
  if (off && off[0] == '1') return SKF_OK;
  if (a_kcontig || b_kcontig) return SKF_OK;
  if ((p.M & 3) || (p.N & 3) || (p.lda & 3) || (p.ldb & 3) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) return SKF_OK;
  if (((uintptr_t)p.slab & 15) || (p.k_chunk & 63)) return SKF_OK;
  *handled = 1;
  GemmParams q = p;
  if (q.row_block_rows != 32) q.row_blocks = nullptr;       // wgrad_x walks 32-row blocks; the fp32 kernel ignores the list
  q.tiles_m = skf_cdiv(p.M, 64); q.tiles_n = skf_cdiv(p.N, 64);
  const size_t smem = (size_t)(4 * 4096 + 4 * 64) * sizeof(float);
  SKF_HIP(hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));   // per launch: the attribute is per device
  const int prec = p.precision;
  const bool fits32 = (double)p.K * p.lda * 4 < 2147483648.0 && (double)p.K * p.ldb * 4 < 2147483648.0;
  if (prec && fits32) {
    // 4 waves per workgroup, one workgroup per CU (~256 workgroups): 8 waves (two per SIMD, 128 KB of LDS for the
    // reduction) measured the same - the loop is issue-bound (6 MFMAs + ~14 VALU per operand pair), not latency-bound -
    // and would halve the register budget the pipelined splits need
    const int nwg = q.tiles_m * q.tiles_n * splits;
    const size_t smem_x = (size_t)4 * (4096 + 64) * sizeof(float);
    static SkfOncePerDevice attr_x;
    if (attr_x.needed()) {
      SKF_HIP(hipFuncSetAttribute((const void*)wgrad_x_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_x));
      SKF_HIP(hipFuncSetAttribute((const void*)wgrad_x_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_x));
      attr_x.mark();
    }
    const double live = skf_prof_list_fraction(q.row_blocks);      // contraction over the live 32-row blocks only
    SkfProfScope ps(st, prec == 3 ? "wgrad<64x64,bf16x3>" : "wgrad<64x64,bf16x6>", 2.0 * p.M * p.N * p.K, 4.0 * (double)p.K * (p.M + p.N));
    ps.done(2.0 * p.M * p.N * p.K * live, 4.0 * (double)p.K * (p.M + p.N) * live);
    const dim3 grid(nwg), block(256);
    if (prec == 3) hipLaunchKernelGGL((wgrad_x_kernel<2, 4>), grid, block, smem_x, st, q);
    else hipLaunchKernelGGL((wgrad_x_kernel<3, 4>), grid, block, smem_x, st, q);
    SKF_LAUNCH_CHECK();
    return SKF_OK;
  }
  SkfProfScope ps(st, "wgrad<64x64>", 2.0 * p.M * p.N * p.K, 4.0 * (double)p.K * (p.M + p.N));
  hipLaunchKernelGGL(wgrad_kernel, dim3(q.tiles_m * q.tiles_n * splits), dim3(256), smem, st, q);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

// Grouped form of the fast path above: every problem must be one that skf_gemm_wgrad_dispatch would hand to wgrad_x_kernel
// (split arithmetic, aligned, 32-bit offsets); *handled = 0 when any is not (the caller then issues them one by one).
int skf_gemm_wgrad_group_dispatch(const GemmParams* ps, const int* splits, int n, hipStream_t st, int* handled) {
  *handled = 0;
  const char* off = skf_knob("SKF_GEMM_NO_WGRAD");
  static const bool group_off = skf_knob("SKF_NO_WGRAD_GROUP") && skf_knob("SKF_NO_WGRAD_GROUP")[0] == '1';
  if ((off && off[0] == '1') || group_off || n < 2 || n > kWgradGroupMax) return SKF_OK;
  WgradGroup grp{};
  grp.n = n;
  int cursor = 0;
  const int prec = ps[0].precision;
  double flops = 0.0, bytes = 0.0, flops_done = 0.0, bytes_done = 0.0;
  for (int g = 0; g < n; ++g) {
    const GemmParams& p = ps[g];
    if ((p.M & 3) || (p.N & 3) || (p.lda & 3) || (p.ldb & 3) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15)) return SKF_OK;
    if (((uintptr_t)p.slab & 15) || (p.k_chunk & 63) || !p.precision || p.precision != prec) return SKF_OK;
    if (!((double)p.K * p.lda * 4 < 2147483648.0 && (double)p.K * p.ldb * 4 < 2147483648.0)) return SKF_OK;
    GemmParams q = p;
    if (q.row_block_rows != 32) q.row_blocks = nullptr;
    q.tiles_m = skf_cdiv(p.M, 64); q.tiles_n = skf_cdiv(p.N, 64);
    grp.p[g] = q;
    grp.start[g] = cursor;
    grp.nblocks[g] = q.tiles_m * q.tiles_n * splits[g];
    cursor += (grp.nblocks[g] + 7) & ~7;
    flops += 2.0 * p.M * p.N * p.K; bytes += 4.0 * (double)p.K * (p.M + p.N);
    const double live = skf_prof_list_fraction(q.row_blocks);
    flops_done += 2.0 * p.M * p.N * p.K * live; bytes_done += 4.0 * (double)p.K * (p.M + p.N) * live;
  }
  grp.start[n] = cursor;
  *handled = 1;
  const size_t smem_x = (size_t)4 * (4096 + 64) * sizeof(float);
  static SkfOncePerDevice attr;
  if (attr.needed()) {
    SKF_HIP(hipFuncSetAttribute((const void*)wgrad_x_group_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_x));
    SKF_HIP(hipFuncSetAttribute((const void*)wgrad_x_group_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_x));
    attr.mark();
  }
  SkfProfScope ps_(st, prec == 3 ? "wgrad_group<64x64,bf16x3>" : "wgrad_group<64x64,bf16x6>", flops, bytes);
  ps_.done(flops_done, bytes_done);
  if (prec == 3) hipLaunchKernelGGL((wgrad_x_group_kernel<2, 4>), dim3(cursor), dim3(256), smem_x, st, grp);
  else hipLaunchKernelGGL((wgrad_x_group_kernel<3, 4>), dim3(cursor), dim3(256), smem_x, st, grp);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
