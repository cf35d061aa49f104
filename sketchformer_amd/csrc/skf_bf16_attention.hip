// Streaming (online-softmax) attention for the bf16 path, head size 64 (BASELINE cfg 5: L = 512, dh = 64: the K/V of a head
// no longer fit the LDS budget of the fp32 kernels in skf_attention.hip).
//
// builders/utils.py:71-105 scaled_dot_product_attention + the head split / merge of builders/layers/transformer.py:160-186
// + the masks of builders/utils.py:35-68.  Same semantics as the fp32 kernels: logits = q.k / sqrt(dh), masked logits are
// SET to -1e9 (one -1e9 for pad OR look-ahead, never two), keys past Lk get -inf, softmax in base 2
// (s2 = q.k * log2(e)/sqrt(dh)), a fully masked row is uniform over all keys; (B,H,Lq,Lk) tensors never exist.
//
// Storage bf16, arithmetic: v_mfma_f32_32x32x16_bf16 with fp32 accumulators, softmax / dS in fp32 registers, P and dS are
// rounded to bf16 only as MFMA operands (like the bf16 reference path of any mixed-precision transformer).
//
// Layout: Q/K/V/O (B, L, ld) bf16, head h = columns [64h, 64h+64).  All three kernels stream 64-row tiles of the "other"
// sequence through a double-buffered LDS image (row pitch 144 B: conflict-free ds_read_b128 of 16 consecutive rows, at most
// 2-way for the transposing reads) with register-staged prefetch and one barrier per tile:
//   forward  : workgroup = 128 queries of one (b, h) (4 waves x 32), streams K and V.  S^T = K.Q^T so that a lane owns 16
//              keys of ONE query (C layout: col = query, rows = keys): the softmax is in-register + one cross-half shuffle,
//              and P^T is already the B operand of O^T = V^T.P^T; V^T fragments come from ds_read_b64_tr_b16 with the row
//              order of the C layout (so P needs no permutation).
//   backward : two passes, no atomics, no transposes through memory, deterministic.
//     dQ pass   : like the forward (streams K, V): S^T, dP^T = V.dO^T, dS^T = P^T o (dP^T - delta), dQ^T += K^T.dS^T
//                 (K^T by transposing reads of the same K tile); also produces delta = rowsum(dO o O) for the second pass.
//     dK/dV pass: workgroup = 128 keys (4 waves x 32, K / V fragments in registers), streams Q, dO and the row statistics:
//                 S = Q.K^T (lane owns 16 queries of ONE key), dP = dO.V^T, dV^T += dO^T.P, dK^T += Q^T.dS (dO^T / Q^T by
//                 transposing reads).
#include <stdlib.h>
#include "skf_common.h"
#include "skf_bf16.h"

namespace {

constexpr int DH = 64;
constexpr int TP = 144;            // bytes per tile row in LDS: 64 bf16 + 16 B pad
constexpr int TILE = 64 * TP;      // one 64-row tile
constexpr float LOG2E = 1.44269504088896340736f;

struct AP {
  const skf_bf16* Q; const skf_bf16* K; const skf_bf16* V; skf_bf16* O;
  int ldq, ldk, ldv, ldo;
  const unsigned char* key_mask; int key_mask_ld;
  int causal, B, H, Lq, Lk;
  float* stats;                 // (B, H, Lq, 2): row max of the base-2 logits, 1 / row sum
  const skf_bf16* dO; int lddo;
  skf_bf16* dQ; skf_bf16* dK; skf_bf16* dV; int lddq, lddk, lddv;
  float* delta;                 // (B, H, Lq) rowsum(dO o O): written by the dQ pass, read by the dK/dV pass
  const int* q_live;            // optional (B): query rows >= q_live[b] have dO == 0 exactly (skf_target_live_len) - backward only
  const int* order;             // optional (B): the samples sorted by length, longest first (skf_sample_order); needs B % 8 == 0 (the entry checks)
  skf_bf16* Olo;                // optional (same shape / pitch as O): O_fp32 - bf16(O), the rounding residual of the output.
                                // delta = rowsum(dO o O) is subtracted from dP = dO.V^T, which it nearly cancels wherever the
                                // softmax gradient is small: with O at 8 significand bits the error of delta (2^-9 |delta|)
                                // exceeded the whole dS of such rows (last encoder layers of cfg 5: 200 % error in dWq / dWk);
                                // O + Olo carries 16 bits at the price of one more bf16 tensor per attention call
};

typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

// A / B operand of v_mfma_f32_32x32x16_bf16 for a contraction over the tile COLUMNS (d): lane l supplies row `row`,
// columns 16*ks + 8*(l>>5) .. +7
__device__ __forceinline__ skf_bf16x8 read_rows(const char* tile, int row, int ks, int hi) {
  return *reinterpret_cast<const skf_bf16x8*>(tile + row * TP + (ks * 16 + hi * 8) * 2);
}
// A operand for a contraction over the tile ROWS: lane l supplies column d0 + (l & 31) and the 8 rows
// rlo + 4*(l>>5) + {0..3}, rlo + 8 + 4*(l>>5) + {0..3} - the order in which the C layout of a 32x32 MFMA holds 16 rows in a
// lane (regs 8s .. 8s+7), so a C-layout tile converts to the matching B operand without moving data between lanes.
__device__ __forceinline__ skf_bf16x8 read_cols(const char* tile, int rlo, int d0, int lane) {
  const int g = lane >> 4, j = lane & 15;
  const char* a = tile + (rlo + 4 * (g >> 1) + (j >> 2)) * TP + (d0 + 16 * (g & 1) + 4 * (j & 3)) * 2;
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)a);
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a + 8 * TP));
  const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(skf_bf16x8, v);
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
__device__ __forceinline__ f32x16 mfma32(skf_bf16x8 a, skf_bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Register-staged copy of one 64-row x 64-column bf16 tile: thread t moves chunks (row = t>>3 (+32), 16-byte chunk t&7).
struct TileRegs { uint4 v[2]; };
__device__ __forceinline__ void tile_gload(TileRegs& r, const skf_bf16* base, int ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (tid >> 3) + 32 * j;
    r.v[j] = row0 + row < nrows ? *reinterpret_cast<const uint4*>(base + (size_t)(row0 + row) * ld + (tid & 7) * 8)
                                : make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ void tile_lstore(const TileRegs& r, char* tile, int tid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) *reinterpret_cast<uint4*>(tile + ((tid >> 3) + 32 * j) * TP + (tid & 7) * 16) = r.v[j];
}

// last un-padded key of sample b (or -1) and which 64-key blocks hold a padded key (bit min(block, 31)), workgroup-wide; `red` = 8 ints of LDS
__device__ __forceinline__ int last_valid_key(const unsigned char* km, int Lk, int* red, int tid, unsigned* masked_blocks) {
  int lv = -1;
  unsigned nm = 0u;
  for (int key = tid; key < Lk; key += 256) {
    if (!(km && km[key])) lv = key;
    else nm |= 1u << min(key >> 6, 31);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lv = max(lv, __shfl_xor(lv, o, 64)); nm |= (unsigned)__shfl_xor((int)nm, o, 64); }
  if ((tid & 63) == 0) { red[tid >> 6] = lv; red[4 + (tid >> 6)] = (int)nm; }
  __syncthreads();
  *masked_blocks = (unsigned)(red[4] | red[5] | red[6] | red[7]);
  return max(max(red[0], red[1]), max(red[2], red[3]));
}

// mask value of every key of a 64-key block into LDS: 0, -1e9 (padded key) or -inf (key >= Lk); returns nothing
__device__ __forceinline__ void stage_key_mask(float* Ms, const unsigned char* km, int k0, int Lk, int tid) {
  if (tid < 64) {
    const int key = k0 + tid;
    Ms[tid] = key < Lk ? ((km && km[key]) ? -1e9f : 0.f) : -INFINITY;
  }
}

// ============================================================================ forward / dQ pass (stream K, V)
// MODE 0: forward.  MODE 1: dQ pass of the backward.
template <int MODE>
__global__ __launch_bounds__(256, 3) void attn_bf16_q_kernel(AP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kt = smem;                       // [2][TILE]
  char* Vt = smem + 2 * TILE;            // [2][TILE]
  float* Ms = reinterpret_cast<float*>(smem + 4 * TILE);    // [2][64]
  int* red = reinterpret_cast<int*>(Ms + 128);              // [8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, hi = lane >> 5;
  const int nqb = (p.Lq + 127) >> 7;
  // XCD-contiguous ids, query block fastest: the blocks of one (b, h) - and the 8 heads of a sample, whose 128-byte slices
  // interleave in every activation row - run on one XCD back to back (K/V re-reads and neighbouring heads hit that L2)
  // (round 5) the block a workgroup takes is rotated by its (b, h): under the look-ahead mask block nqb-1 visits the most keys, and
  // with 4 blocks per (b, h) numbered in order every such block went to the same shader engine (see skf_part_major; the part-major
  // numbering itself costs this kernel 10 % - measured, profiles/r05n_bf16_attn_dispatch.txt)
  // (round 5) with a sorted sample list: the samples of the list are dealt over the 8 XCDs (skf_deal_rank), the m-th (sample, head) pair
  // of an XCD takes this kernel's blocks in a row - every XCD, and every engine of it, gets the same mix of lengths, longest first
  int bh, qb;
  if (p.order) {
    const int i = blockIdx.x >> 3, m = i / nqb, r = skf_deal_rank((int)(blockIdx.x & 7) + 8 * m, p.H);
    qb = (i - m * nqb + m + (m >> 3)) % nqb;
    bh = min(max(p.order[r / p.H], 0), p.B - 1) * p.H + r % p.H;
  } else {
    const int lid = skf_xcd_remap(blockIdx.x, gridDim.x);
    bh = lid / nqb; qb = (lid - bh * nqb + bh + (bh >> 3)) % nqb;
  }
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = qb * 128 + wave * 32, q = q0 + lq;
  const bool qok = q < p.Lq;
  if constexpr (MODE == 1) {
    // a query block behind the sample's last live row: dO is zero there, so is dQ (and delta, which the dK/dV pass never
    // reads for these rows) - stored, not computed
    if (p.q_live && qb * 128 >= p.q_live[b]) {
      if (qok) {
        skf_bf16* dst = p.dQ + (size_t)(b * p.Lq + q) * p.lddq + h * DH + hi * 32;
#pragma unroll
        for (int c = 0; c < 8; ++c) *reinterpret_cast<uint2*>(dst + 4 * c) = make_uint2(0, 0);      // lddq is a multiple of 4 elements
      }
      return;
    }
  }
  const int qc = qok ? q : p.Lq - 1;
  const unsigned char* km = p.key_mask ? p.key_mask + (size_t)b * p.key_mask_ld : nullptr;
  const float c2 = LOG2E / sqrtf((float)DH);

  // query-side operands: B operand rows (lane = query lq, columns 16 ks + 8 hi ..)
  skf_bf16x8 qf[4], dof[4];
  {
    const skf_bf16* qp = p.Q + (size_t)(b * p.Lq + qc) * p.ldq + h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const skf_bf16x8*>(qp + ks * 16);
  }
  float mrow = 0.f, rinv = 0.f, delta = 0.f;
  if constexpr (MODE == 1) {
    const skf_bf16* dp = p.dO + (size_t)(b * p.Lq + qc) * p.lddo + h * DH + hi * 8;
    const skf_bf16* op = p.O + (size_t)(b * p.Lq + qc) * p.ldo + h * DH + hi * 8;
    const skf_bf16* lp = p.Olo ? p.Olo + (size_t)(b * p.Lq + qc) * p.ldo + h * DH + hi * 8 : nullptr;
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 dv = *reinterpret_cast<const uint4*>(dp + ks * 16), ov = *reinterpret_cast<const uint4*>(op + ks * 16);
      dof[ks] = __builtin_bit_cast(skf_bf16x8, dv);
      float df[8], of[8];
      skf_unpack8(dv, df); skf_unpack8(ov, of);
      if (lp) {
        float lf[8];
        skf_unpack8(*reinterpret_cast<const uint4*>(lp + ks * 16), lf);
#pragma unroll
        for (int e = 0; e < 8; ++e) of[e] += lf[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += df[e] * of[e];
    }
    dl += __shfl_xor(dl, 32, 64);
    delta = dl;
    const float2 st = reinterpret_cast<const float2*>(p.stats)[(size_t)bh * p.Lq + qc];
    mrow = st.x; rinv = qok ? st.y : 0.f;          // rows past Lq: p == 0
    if (qok && hi == 0) p.delta[(size_t)bh * p.Lq + q] = dl;
  }

  unsigned masked_blocks;
  const int lastk = last_valid_key(km, p.Lk, red, tid, &masked_blocks);
  // causal skipping is exact only when key 0 is visible to every query; trailing all-padding blocks contribute exactly 0
  // unless some row may see no key at all (see skf_attention.hip)
  const bool can_skip = p.causal && !(km && km[0]);
  const int nkb = (p.Lk + 63) >> 6;
  int nkb_eff = (lastk >= 0 && (!p.causal || can_skip)) ? (lastk >> 6) + 1 : nkb;
  if (can_skip) nkb_eff = min(nkb_eff, ((qb * 128 + 127) >> 6) + 1);      // blocks above this workgroup's last query

  const skf_bf16* kbase = p.K + (size_t)b * p.Lk * p.ldk + h * DH;
  const skf_bf16* vbase = p.V + (size_t)b * p.Lk * p.ldv + h * DH;
  TileRegs rk, rv;
  tile_gload(rk, kbase, p.ldk, 0, p.Lk, tid);
  tile_gload(rv, vbase, p.ldv, 0, p.Lk, tid);
  tile_lstore(rk, Kt, tid); tile_lstore(rv, Vt, tid);
  stage_key_mask(Ms, km, 0, p.Lk, tid);
  __syncthreads();

  f32x16 acc[2] = {zero16(), zero16()};       // O^T (forward) or dQ^T (dQ pass): [d tile][..]
  float m_run = -INFINITY, l_run = 0.f;
  for (int kb = 0; kb < nkb_eff; ++kb) {
    const int cur = kb & 1;
    const char* kt = Kt + cur * TILE;
    const char* vt = Vt + cur * TILE;
    const float* ms = Ms + cur * 64;
    if (kb + 1 < nkb_eff) {
      tile_gload(rk, kbase, p.ldk, (kb + 1) * 64, p.Lk, tid);
      tile_gload(rv, vbase, p.ldv, (kb + 1) * 64, p.Lk, tid);
    }
    const int k0 = kb * 64;
    // wave-uniform: may this block hold a masked key for this wave's queries?
    // (round 5: per 64-key block, not per sample - on a padded batch every sample has padded keys, but only in its last visited block)
    const bool need_mask = ((masked_blocks >> min(kb, 31)) & 1u) != 0u || k0 + 64 > p.Lk || (p.causal && k0 + 63 > q0);
    // ---- S^T (two 32-key tiles): lane = query lq, regs = keys (r&3) + 8*(r>>2) + 4*hi
    float s[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 a = zero16();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a = mfma32(read_rows(kt, t * 32 + lq, ks, hi), qf[ks], a);
      if (need_mask) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float4 m4 = *reinterpret_cast<const float4*>(ms + t * 32 + 8 * r4 + 4 * hi);
          const float mr[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int key = k0 + t * 32 + 8 * r4 + 4 * hi + e;
            const float mv = fminf(mr[e], (p.causal && key > q) ? -1e9f : 0.f);
            s[t][4 * r4 + e] = mv < 0.f ? mv : a[4 * r4 + e] * c2;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = a[r] * c2;
      }
    }
    float pr[2][16];
    if constexpr (MODE == 0) {
      float mx = s[0][0];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // first block: exp2(-inf) = 0
      float ls = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { pr[t][r] = __builtin_amdgcn_exp2f(s[t][r] - m_new); ls += pr[t][r]; }
      l_run = l_run * alpha + ls;
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
      // O^T += V^T . P^T
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          const skf_bf16x8 pf = skf_cvt8(pr[t][8 * sx], pr[t][8 * sx + 1], pr[t][8 * sx + 2], pr[t][8 * sx + 3],
                                         pr[t][8 * sx + 4], pr[t][8 * sx + 5], pr[t][8 * sx + 6], pr[t][8 * sx + 7]);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) acc[dt] = mfma32(read_cols(vt, t * 32 + 16 * sx, dt * 32, lane), pf, acc[dt]);
        }
    } else {
      // P^T from the saved statistics, dP^T = V . dO^T, dS^T = P^T o (dP^T - delta)
      // (a dP accumulator seeded with -delta costs 16 more live registers here: 7 spilled at the 168 this kernel may use - measured, not kept)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16 dp = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dp = mfma32(read_rows(vt, t * 32 + lq, ks, hi), dof[ks], dp);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(s[t][r] - mrow) * rinv;
          pr[t][r] = pv * (dp[r] - delta);
        }
      }
      // dQ^T += K^T . dS^T
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          const skf_bf16x8 df = skf_cvt8(pr[t][8 * sx], pr[t][8 * sx + 1], pr[t][8 * sx + 2], pr[t][8 * sx + 3],
                                         pr[t][8 * sx + 4], pr[t][8 * sx + 5], pr[t][8 * sx + 6], pr[t][8 * sx + 7]);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) acc[dt] = mfma32(read_cols(kt, t * 32 + 16 * sx, dt * 32, lane), df, acc[dt]);
        }
    }
    if (kb + 1 < nkb_eff) {
      tile_lstore(rk, Kt + (cur ^ 1) * TILE, tid);
      tile_lstore(rv, Vt + (cur ^ 1) * TILE, tid);
      stage_key_mask(Ms + (cur ^ 1) * 64, km, (kb + 1) * 64, p.Lk, tid);
    }
    __syncthreads();
  }
  // ---- epilogue: lane = query lq, acc[dt][r] = column d = 32 dt + (r&3) + 8*(r>>2) + 4*hi
  float scale;
  if constexpr (MODE == 0) {
    l_run += __shfl_xor(l_run, 32, 64);
    scale = 1.0f / l_run;
    if (qok && hi == 0 && p.stats) reinterpret_cast<float2*>(p.stats)[(size_t)bh * p.Lq + q] = make_float2(m_run, scale);
  } else {
    scale = 1.0f / sqrtf((float)DH);
  }
  if (qok) {
    skf_bf16* dst = (MODE == 0 ? p.O + (size_t)(b * p.Lq + q) * p.ldo : p.dQ + (size_t)(b * p.Lq + q) * p.lddq) + h * DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float v[4] = {acc[dt][4 * r4] * scale, acc[dt][4 * r4 + 1] * scale, acc[dt][4 * r4 + 2] * scale, acc[dt][4 * r4 + 3] * scale};
        const uint2 pk = skf_pack4(v);
        *reinterpret_cast<uint2*>(dst + dt * 32 + 8 * r4 + 4 * hi) = pk;
        if (MODE == 0 && p.Olo) {
          float hi4[4];
          skf_unpack4(pk, hi4);
          const float lo[4] = {v[0] - hi4[0], v[1] - hi4[1], v[2] - hi4[2], v[3] - hi4[3]};
          *reinterpret_cast<uint2*>(p.Olo + (size_t)(b * p.Lq + q) * p.ldo + h * DH + dt * 32 + 8 * r4 + 4 * hi) = skf_pack4(lo);
        }
      }
  }
}

// ============================================================================ dK / dV pass (stream Q, dO, row statistics)
__global__ __launch_bounds__(256, 2) void attn_bf16_kv_kernel(AP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qt = smem;                       // [2][TILE]
  char* Dt = smem + 2 * TILE;            // [2][TILE]   dO
  float* St = reinterpret_cast<float*>(smem + 4 * TILE);    // [2][3][64]: row max, 1/sum, delta
  int* red = reinterpret_cast<int*>(St + 2 * 3 * 64);       // [8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane & 31, hi = lane >> 5;
  const int nkb = (p.Lk + 127) >> 7;
  int kbk, bh;
  if (p.order) {       // the sorted (sample, head) pairs dealt over the XCDs (see the forward / dQ kernel), block-major in chunks of 32 inside an XCD
    int m;
    skf_part_major((int)(blockIdx.x >> 3), (p.B * p.H) >> 3, nkb, &m, &kbk);
    const int r = skf_deal_rank((int)(blockIdx.x & 7) + 8 * m, p.H);
    bh = min(max(p.order[r / p.H], 0), p.B - 1) * p.H + r % p.H;
  } else {
    const int lid = skf_xcd_remap(blockIdx.x, gridDim.x);
    skf_part_major(lid, p.B * p.H, nkb, &bh, &kbk);     // key block 0 (never all padding, the most queries under the look-ahead mask) first
  }
  const int b = bh / p.H, h = bh % p.H;
  const int k0 = kbk * 128 + wave * 32, key = k0 + lk;
  const bool kok = key < p.Lk;
  const int kc = kok ? key : p.Lk - 1;
  const unsigned char* km = p.key_mask ? p.key_mask + (size_t)b * p.key_mask_ld : nullptr;
  const float c2 = LOG2E / sqrtf((float)DH);
  const float kadd = kok ? ((km && km[key]) ? -1e9f : 0.f) : -INFINITY;

  // key-side operands: B operand rows (lane = key lk, columns 16 ks + 8 hi ..), zero for keys past Lk
  skf_bf16x8 kf[4], vf[4];
  {
    const skf_bf16* kp = p.K + (size_t)(b * p.Lk + kc) * p.ldk + h * DH + hi * 8;
    const skf_bf16* vp = p.V + (size_t)(b * p.Lk + kc) * p.ldv + h * DH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 kv = *reinterpret_cast<const uint4*>(kp + ks * 16), vv = *reinterpret_cast<const uint4*>(vp + ks * 16);
      if (!kok) { kv = make_uint4(0, 0, 0, 0); vv = kv; }
      kf[ks] = __builtin_bit_cast(skf_bf16x8, kv);
      vf[ks] = __builtin_bit_cast(skf_bf16x8, vv);
    }
  }
  unsigned masked_blocks;
  const int lastk = last_valid_key(km, p.Lk, red, tid, &masked_blocks);
  (void)masked_blocks;
  // (round 5) does any of this wave's 32 keys carry a mask value?  If not - and no query of the block sits before a key under the
  // look-ahead mask - the per-score mask arithmetic (two compares, two selects, a min) is skipped: wave-uniform branch
  const bool wave_masked = __ballot(kadd != 0.f) != 0ull;
  const bool can_skip = p.causal && !(km && km[0]);
  const bool pad_skip = lastk >= 0 && (!p.causal || can_skip);       // same rule as the forward: padded keys have P == 0 exactly
  // query blocks behind the sample's last live row have dO == 0: they contribute nothing to dK / dV
  const int nqb = ((p.q_live ? min(p.Lq, p.q_live[b]) : p.Lq) + 63) >> 6;
  // a workgroup whose keys are all padding (or past Lk) writes zeros; causal: queries before the first key see none of them
  const bool dead = pad_skip && kbk * 128 > lastk;
  const int qb_first = can_skip ? (kbk * 128) >> 6 : 0;

  f32x16 dkt[2] = {zero16(), zero16()}, dvt[2] = {zero16(), zero16()};
  if (!dead && qb_first < nqb) {
    const skf_bf16* qbase = p.Q + (size_t)b * p.Lq * p.ldq + h * DH;
    const skf_bf16* dbase = p.dO + (size_t)b * p.Lq * p.lddo + h * DH;
    const float2* stats = reinterpret_cast<const float2*>(p.stats) + (size_t)bh * p.Lq;
    const float* delta = p.delta + (size_t)bh * p.Lq;
    TileRegs rq, rd;
    float st_m = 0.f, st_r = 0.f, st_d = 0.f;
    auto stats_gload = [&](int qb) {
      if (tid < 64) {
        const int q = qb * 64 + tid;
        st_m = 0.f; st_r = 0.f; st_d = 0.f;
        if (q < p.Lq) { const float2 s2 = stats[q]; st_m = s2.x; st_r = s2.y; st_d = delta[q]; }
      }
    };
    auto stats_lstore = [&](int buf) {
      if (tid < 64) { float* s = St + buf * 192; s[tid] = st_m; s[64 + tid] = st_r; s[128 + tid] = -st_d; }     // (-delta: the dP accumulator's seed)
    };
    tile_gload(rq, qbase, p.ldq, qb_first * 64, p.Lq, tid);
    tile_gload(rd, dbase, p.lddo, qb_first * 64, p.Lq, tid);
    stats_gload(qb_first);
    tile_lstore(rq, Qt, tid); tile_lstore(rd, Dt, tid); stats_lstore(0);
    __syncthreads();
    for (int qb = qb_first; qb < nqb; ++qb) {
      const int cur = (qb - qb_first) & 1;
      const char* qt = Qt + cur * TILE;
      const char* dt_ = Dt + cur * TILE;
      const float* st = St + cur * 192;
      if (qb + 1 < nqb) {
        tile_gload(rq, qbase, p.ldq, (qb + 1) * 64, p.Lq, tid);
        tile_gload(rd, dbase, p.lddo, (qb + 1) * 64, p.Lq, tid);
        stats_gload(qb + 1);
      }
      const int qq0 = qb * 64;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // S and dP tiles: lane = key lk, regs = queries qq0 + 32 t + (r&3) + 8*(r>>2) + 4*hi; the dP accumulator starts at -delta of its rows
        f32x16 sa = zero16(), dp;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float4 d4 = *reinterpret_cast<const float4*>(st + 128 + t * 32 + 8 * r4 + 4 * hi);
          dp[4 * r4] = d4.x; dp[4 * r4 + 1] = d4.y; dp[4 * r4 + 2] = d4.z; dp[4 * r4 + 3] = d4.w;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          sa = mfma32(read_rows(qt, t * 32 + lk, ks, hi), kf[ks], sa);
          dp = mfma32(read_rows(dt_, t * 32 + lk, ks, hi), vf[ks], dp);
        }
        float pr[16], ds[16];
        const bool need_mask = wave_masked || (p.causal && k0 + 31 > qq0 + t * 32);      // wave-uniform
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int qo = t * 32 + 8 * r4 + 4 * hi;
          const float4 m4 = *reinterpret_cast<const float4*>(st + qo);
          const float4 r4v = *reinterpret_cast<const float4*>(st + 64 + qo);
          const float mr[4] = {m4.x, m4.y, m4.z, m4.w}, rr[4] = {r4v.x, r4v.y, r4v.z, r4v.w};
          if (need_mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * r4 + e, q = qq0 + qo + e;
              const float mv = p.causal ? fminf(kadd, key > q ? -1e9f : 0.f) : kadd;
              const float sv = mv < 0.f ? mv : sa[r] * c2;
              const float pv = __builtin_amdgcn_exp2f(sv - mr[e]) * rr[e];       // rows past Lq: rr == 0
              pr[r] = pv;
              ds[r] = pv * dp[r];
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * r4 + e;
              const float pv = __builtin_amdgcn_exp2f(fmaf(sa[r], c2, -mr[e])) * rr[e];
              pr[r] = pv;
              ds[r] = pv * dp[r];
            }
          }
        }
        // dV^T += dO^T . P ; dK^T += Q^T . dS
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          const skf_bf16x8 pf = skf_cvt8(pr[8 * sx], pr[8 * sx + 1], pr[8 * sx + 2], pr[8 * sx + 3], pr[8 * sx + 4],
                                         pr[8 * sx + 5], pr[8 * sx + 6], pr[8 * sx + 7]);
          const skf_bf16x8 df = skf_cvt8(ds[8 * sx], ds[8 * sx + 1], ds[8 * sx + 2], ds[8 * sx + 3], ds[8 * sx + 4],
                                         ds[8 * sx + 5], ds[8 * sx + 6], ds[8 * sx + 7]);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dvt[dt] = mfma32(read_cols(dt_, t * 32 + 16 * sx, dt * 32, lane), pf, dvt[dt]);
            dkt[dt] = mfma32(read_cols(qt, t * 32 + 16 * sx, dt * 32, lane), df, dkt[dt]);
          }
        }
      }
      if (qb + 1 < nqb) {
        tile_lstore(rq, Qt + (cur ^ 1) * TILE, tid);
        tile_lstore(rd, Dt + (cur ^ 1) * TILE, tid);
        stats_lstore(cur ^ 1);
      }
      __syncthreads();
    }
  }
  if (kok) {
    const float inv_sqrt = 1.0f / sqrtf((float)DH);
    skf_bf16* dk = p.dK + (size_t)(b * p.Lk + key) * p.lddk + h * DH;
    skf_bf16* dv = p.dV + (size_t)(b * p.Lk + key) * p.lddv + h * DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float a[4] = {dkt[dt][4 * r4] * inv_sqrt, dkt[dt][4 * r4 + 1] * inv_sqrt, dkt[dt][4 * r4 + 2] * inv_sqrt, dkt[dt][4 * r4 + 3] * inv_sqrt};
        const float c[4] = {dvt[dt][4 * r4], dvt[dt][4 * r4 + 1], dvt[dt][4 * r4 + 2], dvt[dt][4 * r4 + 3]};
        *reinterpret_cast<uint2*>(dk + dt * 32 + 8 * r4 + 4 * hi) = skf_pack4(a);
        *reinterpret_cast<uint2*>(dv + dt * 32 + 8 * r4 + 4 * hi) = skf_pack4(c);
      }
  }
}

int check(const AP& p, int dh) {
  SKF_CHECK_ARG(dh == 64, "the bf16 attention kernels are built for head size 64");
  SKF_CHECK_ARG(p.B > 0 && p.H > 0 && p.Lq > 0 && p.Lk > 0, "empty problem");
  SKF_CHECK_ARG((p.ldq & 7) == 0 && (p.ldk & 7) == 0 && (p.ldv & 7) == 0 && (p.ldo & 3) == 0, "row strides must be multiples of 8 elements");
  SKF_CHECK_ARG(!p.causal || p.Lq == p.Lk, "causal attention needs Lq == Lk");
  return SKF_OK;
}
constexpr size_t Q_SMEM = 4 * TILE + 128 * sizeof(float) + 32;
constexpr size_t KV_SMEM = 4 * TILE + 2 * 3 * 64 * sizeof(float) + 32;

template <typename K>
int set_smem(K kfn, size_t bytes) {
  SKF_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return SKF_OK;
}

}  // namespace

extern "C" int skf_attention_bf16_fwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                      const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk,
                                      int dh, void* O, int ldo, void* O_lo, float* stats, skf_stream_t stream) {
  return skf_attention_bf16_fwd_ordered(Q, ldq, K, ldk, V, ldv, key_mask, key_mask_ld, causal, B, H, Lq, Lk, dh, O, ldo, O_lo, stats, nullptr, stream);
}

extern "C" int skf_attention_bf16_fwd_ordered(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                                              const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk,
                                              int dh, void* O, int ldo, void* O_lo, float* stats, const int* sample_order, skf_stream_t stream) {
  AP p{};
  p.order = (B & 7) == 0 ? sample_order : nullptr;     // whole rounds of the 8 XCDs (a numbering: never a result bit)
  p.Olo = (skf_bf16*)O_lo;
  p.Q = (const skf_bf16*)Q; p.K = (const skf_bf16*)K; p.V = (const skf_bf16*)V; p.O = (skf_bf16*)O;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.key_mask = key_mask; p.key_mask_ld = key_mask_ld; p.causal = causal;
  p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.stats = stats;
  int rc = check(p, dh);
  if (rc) return rc;
  SKF_CHECK_ARG(Q && K && V && O, "null operand");
  SKF_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) == 0 && ((uintptr_t)O & 7) == 0, "operands must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if ((rc = set_smem(attn_bf16_q_kernel<0>, Q_SMEM))) return rc;
  const double visited = skf_prof_attention_fraction(key_mask, key_mask_ld, causal, B, Lq, Lk, nullptr, 128, 64);
  SkfProfScope ps(st, "attn_bf16_fwd<dh64>", 4.0 * B * H * (double)Lq * Lk * dh, 2.0 * B * H * dh * (2.0 * Lq + 2.0 * Lk));
  ps.done(4.0 * B * H * (double)Lq * Lk * dh * visited, 2.0 * B * H * dh * (2.0 * Lq + 2.0 * Lk));
  hipLaunchKernelGGL(attn_bf16_q_kernel<0>, dim3(B * H * ((Lq + 127) / 128)), dim3(256), Q_SMEM, st, p);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" size_t skf_attention_bf16_bwd_workspace_bytes(int B, int H, int Lq) { return (size_t)B * H * Lq * sizeof(float); }

extern "C" int skf_attention_bf16_bwd(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O,
                                      int ldo, const void* O_lo, const void* dO, int lddo, const float* stats, const unsigned char* key_mask,
                                      int key_mask_ld, int causal, int B, int H, int Lq, int Lk, int dh, void* dQ, int lddq,
                                      void* dK, int lddk, void* dV, int lddv, void* workspace, size_t workspace_bytes,
                                      skf_stream_t stream) {
  return skf_attention_bf16_bwd_ordered(Q, ldq, K, ldk, V, ldv, O, ldo, O_lo, dO, lddo, stats, key_mask, key_mask_ld, causal, B, H, Lq, Lk, dh,
                                        dQ, lddq, dK, lddk, dV, lddv, workspace, workspace_bytes, nullptr, nullptr, stream);
}

extern "C" int skf_attention_bf16_bwd_rows(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O,
                                           int ldo, const void* O_lo, const void* dO, int lddo, const float* stats,
                                           const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk,
                                           int dh, void* dQ, int lddq, void* dK, int lddk, void* dV, int lddv, void* workspace,
                                           size_t workspace_bytes, const int* q_live_len, skf_stream_t stream) {
  return skf_attention_bf16_bwd_ordered(Q, ldq, K, ldk, V, ldv, O, ldo, O_lo, dO, lddo, stats, key_mask, key_mask_ld, causal, B, H, Lq, Lk, dh,
                                        dQ, lddq, dK, lddk, dV, lddv, workspace, workspace_bytes, q_live_len, nullptr, stream);
}

extern "C" int skf_attention_bf16_bwd_ordered(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* O,
                                              int ldo, const void* O_lo, const void* dO, int lddo, const float* stats,
                                              const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk,
                                              int dh, void* dQ, int lddq, void* dK, int lddk, void* dV, int lddv, void* workspace,
                                              size_t workspace_bytes, const int* q_live_len, const int* sample_order, skf_stream_t stream) {
  AP p{};
  p.q_live = q_live_len;
  p.order = (B & 7) == 0 ? sample_order : nullptr;
  p.Q = (const skf_bf16*)Q; p.K = (const skf_bf16*)K; p.V = (const skf_bf16*)V; p.O = (skf_bf16*)const_cast<void*>(O);
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.key_mask = key_mask; p.key_mask_ld = key_mask_ld; p.causal = causal;
  p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.stats = const_cast<float*>(stats);
  p.dO = (const skf_bf16*)dO; p.lddo = lddo; p.dQ = (skf_bf16*)dQ; p.dK = (skf_bf16*)dK; p.dV = (skf_bf16*)dV;
  p.lddq = lddq; p.lddk = lddk; p.lddv = lddv; p.delta = (float*)workspace;
  p.Olo = (skf_bf16*)const_cast<void*>(O_lo);
  SKF_CHECK_ARG(!O_lo || ((uintptr_t)O_lo & 15) == 0, "O_lo must be 16-byte aligned");
  int rc = check(p, dh);
  if (rc) return rc;
  SKF_CHECK_ARG(Q && K && V && O && dO && stats && dQ && dK && dV, "null operand");
  SKF_CHECK_ARG((lddo & 7) == 0 && (lddq & 3) == 0 && (lddk & 3) == 0 && (lddv & 3) == 0, "row strides must be multiples of 8 / 4 elements");
  SKF_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)dO) & 15) == 0, "operands must be 16-byte aligned");
  SKF_CHECK_ARG(workspace && workspace_bytes >= skf_attention_bf16_bwd_workspace_bytes(B, H, Lq), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if ((rc = set_smem(attn_bf16_q_kernel<1>, Q_SMEM))) return rc;
  if ((rc = set_smem(attn_bf16_kv_kernel, KV_SMEM))) return rc;
  const double vis_q = skf_prof_attention_fraction(key_mask, key_mask_ld, causal, B, Lq, Lk, q_live_len, 128, 64);
  const double vis_kv = skf_prof_attention_fraction(key_mask, key_mask_ld, causal, B, Lq, Lk, q_live_len, 64, 128);
  {
    SkfProfScope ps(st, "attn_bf16_bwd_dq<dh64>", 6.0 * B * H * (double)Lq * Lk * dh, 2.0 * B * H * dh * (4.0 * Lq + 2.0 * Lk));
    ps.done(6.0 * B * H * (double)Lq * Lk * dh * vis_q, 2.0 * B * H * dh * (4.0 * Lq + 2.0 * Lk));
    hipLaunchKernelGGL(attn_bf16_q_kernel<1>, dim3(B * H * ((Lq + 127) / 128)), dim3(256), Q_SMEM, st, p);
    SKF_LAUNCH_CHECK();
  }
  {
    SkfProfScope ps(st, "attn_bf16_bwd_dkv<dh64>", 8.0 * B * H * (double)Lq * Lk * dh, 2.0 * B * H * dh * (2.0 * Lq + 4.0 * Lk));
    ps.done(8.0 * B * H * (double)Lq * Lk * dh * vis_kv, 2.0 * B * H * dh * (2.0 * Lq + 4.0 * Lk));
    hipLaunchKernelGGL(attn_bf16_kv_kernel, dim3(B * H * ((Lk + 127) / 128)), dim3(256), KV_SMEM, st, p);
    SKF_LAUNCH_CHECK();
  }
  return SKF_OK;
}
