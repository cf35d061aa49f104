// Attention backward, head size 16, sequences up to 208 (the cfg-1 / cfg-2 shapes), round 5.
//
// tape gradient of builders/utils.py:71-105 (scaled_dot_product_attention) with the masks of builders/utils.py:35-68; same
// semantics, statistics and skipping rules as attn_bwd_kernel (skf_attention.hip) and attn_bwd2_kernel (skf_attention_bwd2.hip),
// bf16x6 arithmetic of skf_common.h (every fp32 operand = three bf16 pieces, six piece products, fp32 accumulation).
//
// What rounds 2-4 and the first measurements of this round established: on a gfx950 SIMD matrix and vector
// instructions share one issue stream - time = 16 cycles per v_mfma_f32_16x16x32_bf16 + ~2.5 per VALU instruction, whichever wave
// they come from (tools/micro/mfma_bf16_valu_overlap.hip) - so a kernel is as fast as its instruction count.  The one-pass kernel
// is fp32-MFMA-bound (20 x 32 cycles per tile pair) behind ~20 us of staging / per-tile barriers; the two-pass kernel spent 36 of
// its 95 us staging twice and ~106 VALU + 8 ds_bpermute per tile pair; the first form of THIS file (each pass its own workgroup,
// the other side streamed from global memory) ran 206 VALU per tile pair (operand assembly moves, nine address registers per
// loop, splits of the streamed rows) and exposed a 40-us chain of memory round trips per round of workgroups.  Hence:
//   * ONE workgroup of eight waves per (sample, head) stages Q, K, V, dO ONCE as three bf16 planes each (Q pre-scaled by
//     log2(e)/sqrt(dh), dO by the 1/sum of its row) plus -max and -delta/sum per row: 64 KB read per head, the minimum.  After
//     one barrier no wave touches global memory again until it stores a result tile.
//   * The two passes of the two-pass kernel run SIDE BY SIDE in that workgroup as work items of one queue (an LDS ticket
//     counter): item = one query tile of pass A (dQ: S^T / dP^T with a lane holding 4 keys of one query, dS^T is the B operand
//     of dQ^T += K^T.dS^T) or one key tile of pass B (dK, dV: S / dP with a lane holding 4 queries of one key, e and dS are the
//     B operands of dV^T += dO^T.e, dK^T += Q^T.dS).  No transposes through memory, no cross-wave sums, no barrier in the
//     loops; padding, the look-ahead mask and dead query tiles only change how many items / iterations there are, and the queue
//     hands the items out in falling order of their cost, so the waves end within one item of each other.
//   * Per-score arithmetic lives in operands and accumulator seeds: the S accumulator starts at -max (exp2 of the MFMA result
//     IS the unnormalised weight e), the dP accumulator at -delta/sum: dS = e * acc_dp.  2 VALU per score; mask arithmetic only
//     on tiles that hold a masked key or the look-ahead diagonal.  dK is rescaled by ln 2, dQ by 1/sqrt(dh) when stored.
//   * Every MFMA operand is read from LDS into the registers it is consumed from: both sides of the contractions over d are
//     16-byte reads of plane rows (lane half selects the plane: [x0|x1] / [x0|x2] against [y0|y0] / [y1|y1] / [y2|y0], two A
//     reads per tensor and tile pair); the row-contraction operands are {x0,x2}, {x1,x1}, {x0,x0} from SIX transposing reads
//     against {y0,y1} (used twice from the same registers) and {y2,y0} from the split - an LDS read issues beside the vector
//     stream, a v_mov assembling an operand does not; plane pitches are compile-time and the dynamic LDS segment starts at 0,
//     so every LDS address is one of two or three per-lane base registers plus an immediate.
//   * Rows of tiles nobody visits (padded keys, dead query tiles) are not even requested: on QuickDraw-shaped batches a
//     launch is a burst of row loads followed by little arithmetic.
// 81.6 KB of LDS, <= 105 VGPRs: two workgroups = 16 waves per CU.  DESIGN.md section 3e has the measurements and the dropped designs.
#include <stdlib.h>
#include "skf_attention_params.h"

namespace {

typedef __bf16 b3_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned b3_u32x8 __attribute__((ext_vector_type(8)));
typedef unsigned b3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned b3_u32x2 __attribute__((ext_vector_type(2)));
typedef short b3_s4 __attribute__((ext_vector_type(4)));

constexpr int RP = 32;                        // bytes per plane row (16 bf16)
constexpr int MAXT = 13;                      // tiles per tensor: sequences up to 208
constexpr int RMAX = MAXT * 16;
constexpr int PB = RMAX * RP;                 // bytes per plane (6656)
// LDS map (bytes): 12 planes [K0 K1 K2 V0 V1 V2 Q0 Q1 Q2 G0 G1 G2] (G = dO / sum), then -max, -delta/sum per query row, control words
constexpr int K_OFF = 0, V_OFF = 3 * PB, Q_OFF = 6 * PB, G_OFF = 9 * PB;
constexpr int NMX_OFF = 12 * PB, NDL_OFF = NMX_OFF + RMAX * 4, KBITS_OFF = NDL_OFF + RMAX * 4, CTL_OFF = KBITS_OFF + 16 * 4;
constexpr int SMEM_BYTES = CTL_OFF + 16;      // 81,616 <= 81,920 = half of a CU's LDS
constexpr float kLog2e = 1.44269504088896340736f, kLn2 = 0.69314718055994530942f;

__device__ __forceinline__ f32x4 mfma_x(b3_u32x4 a, b3_u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b3_bf16x8, a), __builtin_bit_cast(b3_bf16x8, b), c, 0, 0, 0);
}
// LDS is addressed by 32-bit byte offsets (generic pointers cost a null check per access and hide the immediates)
#define SKF_LDS(T, a) (*reinterpret_cast<T __attribute__((address_space(3)))*>(static_cast<uintptr_t>(a)))
__device__ __forceinline__ b3_u32x2 tr_read(unsigned a) {
  const b3_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<b3_s4 __attribute__((address_space(3)))*>(static_cast<uintptr_t>(a)));
  return __builtin_bit_cast(b3_u32x2, v);
}
__device__ __forceinline__ b3_u32x4 rd128(unsigned a) { return SKF_LDS(const b3_u32x4, a); }
__device__ __forceinline__ b3_u32x4 cat(b3_u32x2 a, b3_u32x2 b) { return (b3_u32x4){a[0], a[1], b[0], b[1]}; }

// Contraction over d (16 deep: two piece products per 32-deep MFMA).  Lane (j, g): row j of the tile, k-slots = columns 8(g&1)..
// of the first (g < 2) or second (g >= 2) piece of the pair.  Y side (loop-invariant, registers): [y0|y0], [y1|y1], [y2|y0];
// X side (two reads per tile pair): [x0|x1], [x0|x2].  seed + x2.y0 + x0.y2 + x1.y1 + x0.y1 + x1.y0 + x0.y0, smallest first.
struct YOps { b3_u32x4 y00, y11, y20; };
__device__ __forceinline__ YOps y_ops(unsigned base /* plane 0 + tile + lane_d */, bool first) {
  YOps o;
  o.y00 = rd128(base);
  o.y11 = rd128(base + PB);
  o.y20 = rd128(base + (first ? 2 * PB : 0));
  return o;
}
__device__ __forceinline__ f32x4 dot_d(const b3_u32x4& x01, const b3_u32x4& x02, const YOps& y, f32x4 acc) {
  acc = mfma_x(x02, y.y20, acc);      // x0.y2 + x2.y0
  acc = mfma_x(x01, y.y11, acc);      // x0.y1 + x1.y1
  acc = mfma_x(x01, y.y00, acc);      // x0.y0 + x1.y0
  return acc;
}
// Contraction over the 16 rows of a tile: acc[d][col] += sum_row X^T[d][row] . y[row][col].  y = the lane's 4 rows (C layout), split
// in registers (pieces y0, y1, y2 as pairs of dwords); X^T = transposing reads of plane rows 4g...  The pairing keeps the register
// moves on the cheap side: B operands {y0,y1} (used by two MFMAs from the same registers) and {y2,y0} (one copy of y0), A operands
// {x0,x2}, {x1,x1}, {x0,x0} from SIX transposing reads, each landing in the register pair it is consumed from (an LDS read is
// issued beside the vector instructions, a v_mov competes with the MFMAs for the same issue slot).
struct TOps { b3_u32x4 x02, x11, x00; };
__device__ __forceinline__ TOps tr_ops(unsigned tr /* plane 0 + tile + lane_t */) {
  TOps o;
  o.x02 = cat(tr_read(tr), tr_read(tr + 2 * PB));
  unsigned trb = tr;
  asm volatile("" : "+v"(trb));                       // (opaque second address: the compiler would merge equal reads and copy the result)
  o.x11 = cat(tr_read(tr + PB), tr_read(trb + PB));
  unsigned trc = tr;
  asm volatile("" : "+v"(trc));
  o.x00 = cat(tr_read(trb), tr_read(trc));
  return o;
}
__device__ __forceinline__ f32x4 dot_rows(const TOps& t, const f32x4& y, f32x4 acc, const SkfSplitSel& sel) {
  unsigned lo[3], hi[3];
  skf_split2<3>(y[0], y[1], lo, sel);
  skf_split2<3>(y[2], y[3], hi, sel);
  const b3_u32x4 y01 = {lo[0], hi[0], lo[1], hi[1]}, y20 = {lo[2], hi[2], lo[0], hi[0]};
  acc = mfma_x(t.x02, y20, acc);      // x0.y2 + x2.y0
  acc = mfma_x(t.x11, y01, acc);      // x1.y0 + x1.y1
  acc = mfma_x(t.x00, y01, acc);      // x0.y0 + x0.y1
  return acc;
}

// one 16-byte row chunk (float4) -> its three plane rows
__device__ __forceinline__ void put_planes(unsigned dst, int row, int c4, const float4& v, const SkfSplitSel& sel) {
  unsigned lo[3], hi[3];
  skf_split2<3>(v.x, v.y, lo, sel);
  skf_split2<3>(v.z, v.w, hi, sel);
#pragma unroll
  for (int q = 0; q < 3; ++q) SKF_LDS(b3_u32x2, dst + q * PB + row * RP + c4 * 2) = (b3_u32x2){lo[q], hi[q]};
}
__device__ __forceinline__ float4 scaled(const float4& v, float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }

template <bool CAUSAL>
__global__ __launch_bounds__(512, 4) void attn_bwd3_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nkt = (p.Lk + 15) >> 4, nqt_all = (p.Lq + 15) >> 4;
  // (the dynamic LDS segment starts at LDS address `sbase`; everything below addresses LDS by byte offset)
  // (this kernel declares no static LDS, so the dynamic segment starts at LDS address 0: a literal base lets every address be a
  //  per-lane register plus an immediate; checked once, a wrong assumption traps instead of corrupting)
  constexpr unsigned sbase = 0u;
  if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem != sbase) __builtin_trap();
  // NMX_OFF: -max of the base-2 logits of a query row; NDL_OFF: -rowsum(dO o O) / sum
  unsigned* Kbits = reinterpret_cast<unsigned*>(smem + KBITS_OFF);   // [16] bit i of word t: key 16 t + i is padded (key mask)
  int* ctl = reinterpret_cast<int*>(smem + CTL_OFF);   // [0] last un-padded key, [1] bit t: query tile t has a non-zero dO row, [2] item ticket,
                                                       // [3] bit t: key tile t holds a padded key or reaches past Lk
  const SkfSplitSel sel = skf_split_sel();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const bool first = g < 2;
  const float c2 = kLog2e * 0.25f;                   // log2(e) / sqrt(dh)
  // One workgroup per (sample, head), ids XCD-contiguous: all heads of a sample on one XCD, consecutively in time (the 64-byte head
  // slices of a 512-byte activation row are L2 hits for the neighbours).  Measured and dropped (profiles/r05e_*): two PERSISTENT
  // workgroups per CU walking the heads with the second one started late so that one stages while the other multiplies - static
  // head assignment loses more on padded batches (91 vs 71 us) than the offset gains (100 vs 107 us on full-length rows).
  int bh = p.xcd_remap ? skf_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  if (p.order) { const int k = skf_deal_rank(blockIdx.x, p.H); bh = min(max(p.order[k / p.H], 0), p.B - 1) * p.H + k % p.H; }   // (clamped: a list that is no permutation must not leave the tensors)
  const int b = bh / p.H, h = bh % p.H;
#if SKF_MEASURE     // clock stamps of a few workgroups (tools/attn_bwd3_timeline.py): measurement builds only
  long long* dbg = (p.dbg && lane == 0 && (blockIdx.x % 131) == 0 && blockIdx.x / 131 < 8) ? p.dbg + ((blockIdx.x / 131) * 8 + wave) * 16 : nullptr;
  int dbi = 0;
#define SKF_STAMP3() do { if (dbg && dbi < 16) dbg[dbi++] = wall_clock64(); } while (0)
#else
#define SKF_STAMP3() do { } while (0)
#endif
#if SKF_MEASURE     // experiment (tools/attn_bwd3_stagger.py): the second workgroup of a CU in the first round starts late
  if (p.ablate >= 1000 && (int)blockIdx.x < 512) {
    const unsigned lds_base = __builtin_amdgcn_s_getreg((12 - 1) << 11 | 0 << 6 | 6);      // HW_REG_LDS_ALLOC.LDS_BASE
    if (lds_base != 0u) {
      const long long t0 = wall_clock64();
      while (wall_clock64() - t0 < (long long)(p.ablate / 1000)) __builtin_amdgcn_s_sleep(32);
    }
  }
#endif
  SKF_STAMP3();      // 0: start
  // query tiles behind the sample's last live row have dO == 0 exactly: nothing for dK / dV, dQ = 0 (skf_attention_bwd_rows)
  const int nqt = p.q_live ? min(nqt_all, (max(p.q_live[b], 0) + 15) >> 4) : nqt_all;
  // ---------------- which key tiles are visited: the key mask first (one byte per thread: Lk <= 208 < 512), so that the rows of a
  // head are only REQUESTED where somebody will read them - on QuickDraw-shaped batches (58 % padding) the launch is a burst of
  // row loads followed by little arithmetic, and the burst is what it costs
  const unsigned char mb = p.key_mask ? p.key_mask[(size_t)b * p.key_mask_ld + min(tid, p.Lk - 1)] : (unsigned char)0;
  if (tid < 16) Kbits[tid] = 0u;
  if (tid == 32) { ctl[0] = -1; ctl[1] = 0; ctl[2] = 8; ctl[3] = (p.Lk & 15) ? 1 << (nkt - 1) : 0; }
  __syncthreads();
  if (tid < p.Lk) {
    if (mb) { atomicOr(&Kbits[tid >> 4], 1u << (tid & 15)); atomicOr(reinterpret_cast<unsigned*>(&ctl[3]), 1u << (tid >> 4)); }
    else atomicMax(&ctl[0], tid);
  }
  __syncthreads();
  SKF_STAMP3();      // 1: key mask known
  const int lastk = ctl[0];
  // skipping fully look-ahead-masked tiles is exact only if key 0 is visible; trailing all-padding key tiles have P == 0 exactly
  // unless some row may see no key at all (see skf_attention.hip)
  const bool can_skip = CAUSAL && !(Kbits[0] & 1u);
  const int nkt_eff = (lastk >= 0 && (!CAUSAL || can_skip)) ? (lastk >> 4) + 1 : nkt;
  // ---------------- staging loads: chunk e = (row, 4 columns) of all five tensors, two chunks per thread, every load of the
  // workgroup requested before the first wait; rows of dead tiles are not requested at all
  float4 kv[2], vv[2], qv[2], gv[2], ov[2];
  float2 sv[2];
  {
    const float* Qb = p.Q + (size_t)b * p.Lq * p.ldq + h * 16;
    const float* Kb = p.K + (size_t)b * p.Lk * p.ldk + h * 16;
    const float* Vb = p.V + (size_t)b * p.Lk * p.ldv + h * 16;
    const float* Ob = p.O + (size_t)b * p.Lq * p.ldo + h * 16;
    const float* dOb = p.dO + (size_t)b * p.Lq * p.lddo + h * 16;
    const float2* stats = reinterpret_cast<const float2*>(p.stats) + (size_t)bh * p.Lq;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 512 * u, row = e >> 2, c4 = (e & 3) * 4;
      const int rk = min(row, p.Lk - 1), rq = min(row, p.Lq - 1);
      kv[u] = vv[u] = qv[u] = gv[u] = ov[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      sv[u] = make_float2(0.f, 0.f);
      if (row < nkt_eff * 16) {
        kv[u] = *reinterpret_cast<const float4*>(Kb + (size_t)rk * p.ldk + c4);
        vv[u] = *reinterpret_cast<const float4*>(Vb + (size_t)rk * p.ldv + c4);
      }
      if (row < nqt * 16) {
        qv[u] = *reinterpret_cast<const float4*>(Qb + (size_t)rq * p.ldq + c4);
        gv[u] = *reinterpret_cast<const float4*>(dOb + (size_t)rq * p.lddo + c4);
        ov[u] = *reinterpret_cast<const float4*>(Ob + (size_t)rq * p.ldo + c4);
        sv[u] = stats[rq];
      }
    }
  }
  SKF_STAMP3();      // 2: loads requested
  // live query tiles: the 64 chunks of a wave in pass u are the 16 rows of tile (e >> 6): one ballot, one LDS atomic
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 512 * u, row = e >> 2;
    const bool nz = row < p.Lq && (gv[u].x != 0.f || gv[u].y != 0.f || gv[u].z != 0.f || gv[u].w != 0.f);
    if (__ballot(nz) != 0ull && lane == 0 && (e >> 6) < nqt) atomicOr(reinterpret_cast<unsigned*>(&ctl[1]), 1u << (e >> 6));
  }
  SKF_STAMP3();      // 3: rows arrived (the ballot above waits for dO)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 512 * u, row = e >> 2, c4 = (e & 3) * 4;
    if (row < nkt_eff * 16) {
      const float z = row < p.Lk ? 1.f : 0.f;
      put_planes(sbase + K_OFF, row, c4, scaled(kv[u], z), sel);
      __builtin_amdgcn_sched_barrier(0);          // (one tensor at a time: interleaved, the eight splits of a thread spill)
      put_planes(sbase + V_OFF, row, c4, scaled(vv[u], z), sel);
      __builtin_amdgcn_sched_barrier(0);
    } else if (row < p.Lk) {      // key tiles nobody visits: dK = dV = 0
      *reinterpret_cast<float4*>(p.dK + (size_t)(b * p.Lk + row) * p.lddk + h * 16 + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(p.dV + (size_t)(b * p.Lk + row) * p.lddv + h * 16 + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool qok = row < p.Lq;
    const float ri = qok ? sv[u].y : 0.f;       // rows past Lq: P == 0
    float dl = gv[u].x * ov[u].x + gv[u].y * ov[u].y + gv[u].z * ov[u].z + gv[u].w * ov[u].w;
    dl += __shfl_xor(dl, 1, 64);
    dl += __shfl_xor(dl, 2, 64);
    if (row < nqt * 16) {
      put_planes(sbase + Q_OFF, row, c4, scaled(qv[u], qok ? c2 : 0.f), sel);
      __builtin_amdgcn_sched_barrier(0);
      put_planes(sbase + G_OFF, row, c4, scaled(gv[u], ri), sel);
      __builtin_amdgcn_sched_barrier(0);
      // rows past Lq of the last live tile: seed -inf, so that exp2 gives exactly 0 on the masked and the unmasked path alike (with the
      // clamped row's -max as the seed, exp2(0 - max) overflowed for max < -128 and inf * 0 put NaN into the whole key tile's dK / dV)
      if (c4 == 0) { SKF_LDS(float, sbase + NMX_OFF + row * 4) = qok ? -sv[u].x : -INFINITY; SKF_LDS(float, sbase + NDL_OFF + row * 4) = qok ? -dl * ri : 0.f; }
    } else if (qok) {             // dead query tiles: dQ = 0
      *reinterpret_cast<float4*>(p.dQ + (size_t)(b * p.Lq + row) * p.lddq + h * 16 + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  SKF_STAMP3();      // 4: planes written (this wave)
  __syncthreads();
  SKF_STAMP3();      // 5: planes complete
  const unsigned q_live = (unsigned)ctl[1], k_masked = (unsigned)ctl[3];

  // ---------------- the queue.  Items in falling order of cost: key tiles ascend (under the look-ahead mask key tile t is seen by
  // the query tiles >= t), query tiles descend (tile t sees t + 1 key tiles), a pass-B item (12 MFMAs per pair) before the
  // pass-A item (9) of the same rank.  The first eight tickets are the wave numbers.
  const int nA = nqt, nB = nkt_eff, nm = nA < nB ? nA : nB, n_items = nA + nB;
  const int lane_d = j * RP + (g & 1) * 16;                       // row j of a tile, the lane's 8 columns
  const int x01 = lane_d + (first ? 0 : PB), x02 = lane_d + (first ? 0 : 2 * PB);     // per-lane plane selection of [x0|x1], [x0|x2]
  const int lane_t = (4 * g + (j >> 2)) * RP + (j & 3) * 8;       // transposing reads: plane row 4g + (j >> 2), 4 columns

  for (int item = wave; item < n_items;) {
    int is_b, t;
    if (item < 2 * nm) { is_b = !(item & 1); t = item >> 1; }
    else { is_b = nB > nA; t = nm + (item - 2 * nm); }
    if (!is_b) {
      // ================================================================ pass A item: dQ of query tile qt
      const int qt = nA - 1 - t, q = qt * 16 + j;
      const bool live = ((q_live >> qt) & 1u) != 0u;
      const unsigned qrow = sbase + qt * 16 * RP + lane_d;
      const YOps qo = y_ops(qrow + Q_OFF, first), go = y_ops(qrow + G_OFF, first);
      const float nmx = SKF_LDS(const float, sbase + NMX_OFF + q * 4), ndl = SKF_LDS(const float, sbase + NDL_OFF + q * 4);
      const int nt = (live && !(p.ablate & 1)) ? min(can_skip ? qt + 1 : nkt, nkt_eff) : 0;   // (ablate: diagnostics)
      const f32x4 s_seed = {nmx, nmx, nmx, nmx}, d_seed = {ndl, ndl, ndl, ndl};
      f32x4 dq = {0.f, 0.f, 0.f, 0.f};
      unsigned a1 = sbase + x01, a2 = sbase + x02, tr = sbase + lane_t;
      for (int kt = 0; kt < nt; ++kt, a1 += 16 * RP, a2 += 16 * RP, tr += 16 * RP) {
        const f32x4 sacc = dot_d(rd128(a1 + K_OFF), rd128(a2 + K_OFF), qo, s_seed);    // S^T - max: lane = query j, rows = keys kt*16 + 4g + r
        const f32x4 dpacc = dot_d(rd128(a1 + V_OFF), rd128(a2 + V_OFF), go, d_seed);   // (dP - delta) / sum
        const TOps kw = tr_ops(tr + K_OFF);
        f32x4 ds;
        if (((k_masked >> kt) & 1u) || (CAUSAL && kt >= qt)) {                         // wave-uniform (scalar branch)
          const unsigned bits = __builtin_amdgcn_readfirstlane(Kbits[kt]);
          const float mx = -nmx;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int key = kt * 16 + 4 * g + r4;
            float m = key < p.Lk ? (((bits >> (4 * g + r4)) & 1u) ? -1e9f : 0.f) : -INFINITY;
            if (CAUSAL && key > q) m = fminf(m, -1e9f);
            const float tt = m < 0.f ? m - mx : sacc[r4];
            ds[r4] = __builtin_amdgcn_exp2f(tt) * dpacc[r4];
          }
        } else {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) ds[r4] = __builtin_amdgcn_exp2f(sacc[r4]) * dpacc[r4];
        }
        dq = dot_rows(kw, ds, dq, sel);                                // dQ^T[d][q] += sum_k K[k][d] dS[q][k]
      }
      if (q < p.Lq)
        *reinterpret_cast<float4*>(p.dQ + (size_t)(b * p.Lq + q) * p.lddq + h * 16 + 4 * g) =
            make_float4(dq[0] * 0.25f, dq[1] * 0.25f, dq[2] * 0.25f, dq[3] * 0.25f);
    } else {
      // ================================================================ pass B item: dK, dV of key tile kt
      const int kt = t, key = kt * 16 + j;
      const unsigned krow = sbase + kt * 16 * RP + lane_d;
      const YOps ko = y_ops(krow + K_OFF, first), vo = y_ops(krow + V_OFF, first);
      const unsigned kbits = __builtin_amdgcn_readfirstlane(Kbits[kt]);
      // keys past Lk get -inf (never -1e9): with a fully padded sample the row max itself is -1e9
      const float kadd = key < p.Lk ? (((kbits >> j) & 1u) ? -1e9f : 0.f) : -INFINITY;
      const bool tile_masked = ((k_masked >> kt) & 1u) != 0u;                          // wave-uniform (scalar branch)
      f32x4 dkt = {0.f, 0.f, 0.f, 0.f}, dvt = dkt;
      const int qt0 = (CAUSAL && can_skip) ? kt : 0;
      // (bases inside the Q / G half of the planes: DS immediates are 16 bits)
      unsigned a1 = sbase + Q_OFF + x01 + qt0 * 16 * RP, a2 = sbase + Q_OFF + x02 + qt0 * 16 * RP, tr = sbase + Q_OFF + lane_t + qt0 * 16 * RP;
      unsigned st = sbase + NMX_OFF + (qt0 * 16 + 4 * g) * 4;
      const int qend = (p.ablate & 2) ? 0 : nqt;
      for (int qt = qt0; qt < qend; ++qt, a1 += 16 * RP, a2 += 16 * RP, tr += 16 * RP, st += 64) {
        if (!((q_live >> qt) & 1u)) continue;
        const f32x4 s_seed = SKF_LDS(const f32x4, st), d_seed = SKF_LDS(const f32x4, st + (NDL_OFF - NMX_OFF));
        const f32x4 sacc = dot_d(rd128(a1), rd128(a2), ko, s_seed);                    // S - max: lane = key j, rows = queries 16 qt + 4g + r
        const f32x4 dpacc = dot_d(rd128(a1 + 3 * PB), rd128(a2 + 3 * PB), vo, d_seed); // (dP - delta) / sum
        const TOps gw = tr_ops(tr + 3 * PB), qw = tr_ops(tr);
        f32x4 pr, ds;
        if (tile_masked || (CAUSAL && kt >= qt)) {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const float m = CAUSAL ? fminf(kadd, key > qt * 16 + 4 * g + r4 ? -1e9f : 0.f) : kadd;
            const float tt = m < 0.f ? m + s_seed[r4] : sacc[r4];
            pr[r4] = __builtin_amdgcn_exp2f(tt);
            ds[r4] = pr[r4] * dpacc[r4];
          }
        } else {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) { pr[r4] = __builtin_amdgcn_exp2f(sacc[r4]); ds[r4] = pr[r4] * dpacc[r4]; }
        }
        dvt = dot_rows(gw, pr, dvt, sel);                              // dV^T[d][k] += sum_q (dO[q][d] / sum_q) e[q][k]
        dkt = dot_rows(qw, ds, dkt, sel);                              // dK^T[d][k] += sum_q (Q[q][d] c2) dS[q][k]
      }
      if (key < p.Lk) {
        // (Q c2) / (c2 sqrt(dh)): c2 sqrt(dh) = log2(e)
        *reinterpret_cast<float4*>(p.dK + (size_t)(b * p.Lk + key) * p.lddk + h * 16 + 4 * g) =
            make_float4(dkt[0] * kLn2, dkt[1] * kLn2, dkt[2] * kLn2, dkt[3] * kLn2);
        *reinterpret_cast<float4*>(p.dV + (size_t)(b * p.Lk + key) * p.lddv + h * 16 + 4 * g) = make_float4(dvt[0], dvt[1], dvt[2], dvt[3]);
      }
    }
    // next ticket (one lane asks, the wave shares the answer)
    int nx = 0;
    if (lane == 0) nx = atomicAdd(&ctl[2], 1);
    item = __builtin_amdgcn_readfirstlane(nx);
    SKF_STAMP3();    // 6..: one per item
  }
  // out of items: request the next head's rows (the staging registers are free again) and wait for the other waves to leave the planes
}

}  // namespace

int skf_attention_bwd3_supported(int dh, int Lq, int Lk) { return dh == 16 && Lq <= RMAX && Lk <= RMAX; }

int skf_attention_bwd3_launch(const AttnParams& p_in, hipStream_t st) {
  AttnParams p = p_in;
  SKF_CHECK_ARG(p.Lq <= RMAX && p.Lk <= RMAX, "sequence longer than 208");
  SKF_CHECK_ARG((p.ldq & 3) == 0 && (p.ldk & 3) == 0 && (p.ldv & 3) == 0 && (p.ldo & 3) == 0 && (p.lddo & 3) == 0 && (p.lddq & 3) == 0 &&
                (p.lddk & 3) == 0 && (p.lddv & 3) == 0, "row strides must be multiples of 4");
  const void* kfn = p.causal ? (const void*)attn_bwd3_kernel<true> : (const void*)attn_bwd3_kernel<false>;
  SKF_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
  const int grid = p.B * p.H;
#if SKF_MEASURE
  { const char* db = skf_knob("SKF_ATTN_DBG"); p.dbg = db ? (long long*)strtoull(db, nullptr, 0) : nullptr; }
#endif
  const double visited = skf_prof_attention_fraction(p.key_mask, p.key_mask_ld, p.causal, p.B, p.Lq, p.Lk, p.q_live, 16, 16);
  SkfProfScope ps(st, "attn_bwd<dh16>", 8.0 * p.B * p.H * (double)p.Lq * p.Lk * 16, 4.0 * p.B * p.H * 16 * (4.0 * p.Lq + 4.0 * p.Lk));
  ps.done(8.0 * p.B * p.H * (double)p.Lq * p.Lk * 16 * visited, 4.0 * p.B * p.H * 16 * (4.0 * p.Lq + 4.0 * p.Lk));
  if (p.causal) hipLaunchKernelGGL((attn_bwd3_kernel<true>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  else hipLaunchKernelGGL((attn_bwd3_kernel<false>), dim3(grid), dim3(512), SMEM_BYTES, st, p);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
