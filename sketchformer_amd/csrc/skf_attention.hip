// Fused scaled-dot-product attention forward / backward on v_mfma_f32_16x16x4_f32.
//
// Replaces builders/utils.py:71-105 (scaled_dot_product_attention) plus the
// split_heads / merge transposes of builders/layers/transformer.py:160-186 and the
// masks of builders/utils.py:35-68, which are never materialised: the key padding
// mask is a (B,Lk) byte array and the look-ahead mask is derived from indices.
// Semantics kept exactly: logits = (q.k)/sqrt(dh) + mask*(-1e9); softmax over keys;
// out = P.V.  The (B,H,Lq,Lk) score / weight tensors never reach HBM.
//
// Layout: Q/K/V/O are row-major (B, L, ld) activations; head h occupies columns
// [h*DH, (h+1)*DH).  One 256-thread workgroup per (b, h).
//
// Forward: K and V of the head are staged once in LDS; each wave owns 16-row query
// tiles.  S^T = K.Q^T is computed per 16x16 tile so that a lane holds 4 keys of ONE
// query (C layout: col = lane&15 = query, row = 4*(lane>>4)+r = key): the softmax
// reductions are in-register + two shuffles, and P^T is already in B-operand layout
// for O^T = V^T.P^T (the k index of MFMA step s in lane group g is 4g+s on both
// operands).  All Lk scores of a query row stay in registers: exact two-pass softmax.
//
// Backward: each wave owns a 16-key tile (K/V fragments live in registers, dK/dV
// accumulate in registers) and walks the query tiles; S / dP are computed untransposed
// (row = query) so they feed dV^T += dO^T.P and dK^T += Q^T.dS directly; dS is
// transposed through a wave-private LDS scratch for dQ^T += K^T.dS^T; each wave writes its
// 16 x DH partial of dQ to its own LDS slot and the workgroup sums the four slots into global
// dQ (no atomics: fixed summation order).  Head sizes 16 and 32 (L <= 256) with the split
// arithmetic take the two-pass kernel of skf_attention_bwd2.hip instead; this one serves the
// fp32-MFMA mode, head size 64 and long sequences.
#include <stdlib.h>
#include "skf_common.h"
#include <type_traits>
#include "skf_attention_params.h"

namespace {


__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Softmax runs in the base-2 domain: s2 = (q.k) * log2(e)/sqrt(dh), p = 2^(s2 - max2) (one v_exp_f32, no extra
// multiply), the saved row statistics are (max2, 1/sum) and the backward recomputes p the same way.  Masked logits are
// SET to -1e9 (pad / look-ahead) or -inf (key >= Lk) instead of added: identical probabilities whenever a row sees at
// least one real key (the reference's fp32 `x + -1e9` already rounds to -1e9), and exactly uniform weights for a
// fully masked row.  f32 MFMA and VALU share the issue pipe on gfx950, so every per-score VALU instruction counts:
// the mask work is only done on key tiles that contain a masked key (or the diagonal tile of a causal row), the
// 1/sum normalisation is applied to the 16 x dh output instead of the L scores, V sits transposed in LDS so that
// one ds_read_b128 feeds four MFMAs.
// SPLIT (dh = 16 only): S^T = K.Q^T on the bf16 matrix cores.  Both operands are split exactly into three bf16 pieces
// (skf_split2) and the contraction is only 16 deep, so the 32-deep v_mfma_f32_16x16x32_bf16 takes TWO piece products per
// instruction: [k1|k0].[q1|q2], [k1|k2].[q0|q0], [k0|k0].[q0|q1] = all six products of the fp32-equivalent sum in
// 3 x 16 cycles instead of 4 x 32 for v_mfma_f32_16x16x4_f32 (lane group g supplies dh 8(g&1)..+7 of the first (g < 2)
// or second (g >= 2) operand of each pair).  K sits in LDS as three bf16 planes of 48-byte rows.
typedef __bf16 attn_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned attn_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned attn_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 mfma_bf16(attn_u32x4 a, attn_u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(attn_bf16x8, a), __builtin_bit_cast(attn_bf16x8, b), c, 0, 0, 0);
}
constexpr int KPP = 32;   // bytes per key row of a bf16 K plane: unpadded (2-way bank conflicts on the fragment reads) so that K planes + V^T of a 208-key head stay under 40 KB = four workgroups per CU; 40 bytes would break the 16-byte alignment of ds_read_b128 (1.6x slower)

template <int DH, int MAXT, bool SPLIT>
__global__ __launch_bounds__(256, MAXT <= 13 ? 4 : 1) void attn_fwd_kernel(AttnParams p) {   // <= 128 VGPRs: 1024 (sample, head) workgroups = one round of 4 per CU
  static_assert(!SPLIT || DH == 16, "split S^T path is written for dh = 16");
  constexpr int NC = DH / 16;
  constexpr int LD = SPLIT ? 3 * KPP / 4 : DH + 4;   // floats per key row of the K image (SPLIT: three 48-byte plane rows)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // A head is a 64..256-byte slice of every activation row, i.e. half (or less) of each 128-byte line it touches; the
  // other half belongs to the neighbouring head.  XCD-contiguous ids put all heads of a sample on ONE XCD (one L2),
  // consecutively in time, so the neighbour's half is an L2 hit instead of a second HBM fetch of the same line.
  int bh = p.xcd_remap ? skf_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  if (p.order) { const int k = skf_deal_rank(blockIdx.x, p.H); bh = min(max(p.order[k / p.H], 0), p.B - 1) * p.H + k % p.H; }   // (clamped: a list that is no permutation must not leave the tensors)
  const int b = bh / p.H, h = bh % p.H;
  const int nkt = (p.Lk + 15) >> 4, nqt = (p.Lq + 15) >> 4;
  const int VP = nkt * 16 + 8;            // pitch of the transposed V image: (VP / 4) mod 4 == 2 keeps the ds_read_b128 of the PV
                                          // operands free of bank conflicts (the four 16-lane groups of the instruction mix lane groups g)
  float* Ks = smem;                       // [nkt*16][LD]
  float* Vt = smem + nkt * 16 * LD;       // [DH][VP]   V transposed: Vt[d][key]
  float* Ms = Vt + DH * VP;               // [nkt*16] key mask: 0, -1e9 (padded key) or -inf (key >= Lk)
  int* Tf = reinterpret_cast<int*>(Ms + nkt * 16);   // [nkt] 1 = the key tile holds a masked key
  int* last_valid = Tf + nkt;             // [4]: per-wave index of the last un-padded key
  const SkfSplitSel sel = skf_split_sel();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: loop bounds stay scalar
  const int i = lane & 15, g = lane >> 4;

  // Query rows of a tile: loaded one tile ahead (the first before the K/V staging) - a global load under a busy chip
  // takes 2-4k cycles and nothing else of a tile can start without them.  Rows past Lq read row Lq-1 and are zeroed.
  constexpr int NQ = SPLIT ? 2 : NC;
  float4 qnext[NQ];
  auto load_q = [&](int qt_, float4 (&dst)[NQ]) {
    const int row = qt_ * 16 + i, rc = row < p.Lq ? row : p.Lq - 1;
    const float* qp = p.Q + (size_t)(b * p.Lq + rc) * p.ldq + h * DH;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(qp + (SPLIT ? (g & 1) * 8 + 4 * c : c * 16 + g * 4));
      dst[c] = row < p.Lq ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
#if SKF_MEASURE     // clock stamps of a few workgroups (tools/attn_fwd_timeline.py): measurement builds only
  long long* fdbg = (p.dbg && lane == 0 && (blockIdx.x % 131) == 0 && blockIdx.x / 131 < 8) ? p.dbg + ((blockIdx.x / 131) * 4 + wave) * 16 : nullptr;
  int fdbi = 0;
#define SKF_FSTAMP() do { if (fdbg && fdbi < 16) fdbg[fdbi++] = wall_clock64(); } while (0)
#else
#define SKF_FSTAMP() do { } while (0)
#endif
  SKF_FSTAMP();      // start
  load_q((wave + bh) & 3, qnext);
  // ---- stage K, V^T (zero-filled tail rows) and the key mask.  All global loads of the prologue are issued before the first wait
  // (clamped, always valid addresses; rows past Lk zeroed afterwards): the guarded form - `if (row < Lk) load` inside a 256-element
  // loop, then the mask bytes - was one serialised memory round trip per loop iteration, five before the first MFMA at L = 200.
  const int last_key = p.Lk - 1;
  const unsigned char* kmp = p.key_mask ? p.key_mask + (size_t)b * p.key_mask_ld
                                        : reinterpret_cast<const unsigned char*>(p.K + (size_t)b * p.Lk * p.ldk);   // no mask: readable bytes, ignored
  const unsigned char mk0 = kmp[min(tid, last_key)];
  constexpr int SB = 4;                                  // staging loads in flight per thread (K and V: 2 x SB float4)
  const int stage_total = nkt * 16 * (DH / 4);
  for (int e0 = tid; e0 < stage_total; e0 += 256 * SB) {
    float4 kvb[SB], vvb[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int e = e0 + u * 256, rc = min(e / (DH / 4), last_key), c4 = (e % (DH / 4)) * 4;
      kvb[u] = *reinterpret_cast<const float4*>(p.K + (size_t)(b * p.Lk + rc) * p.ldk + h * DH + c4);
      vvb[u] = *reinterpret_cast<const float4*>(p.V + (size_t)(b * p.Lk + rc) * p.ldv + h * DH + c4);
    }
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int e = e0 + u * 256, row = e / (DH / 4), c4 = (e % (DH / 4)) * 4;
      if (e >= stage_total) continue;
      float4 kv = kvb[u], vv = vvb[u];
      if (row >= p.Lk) { kv = make_float4(0.f, 0.f, 0.f, 0.f); vv = kv; }
      if constexpr (SPLIT) {
        unsigned lo[3], hi[3];
        skf_split2<3>(kv.x, kv.y, lo, sel);
        skf_split2<3>(kv.z, kv.w, hi, sel);
        char* kp = reinterpret_cast<char*>(Ks);
#pragma unroll
        for (int q = 0; q < 3; ++q)
          *reinterpret_cast<attn_u32x2*>(kp + (size_t)q * nkt * 16 * KPP + row * KPP + c4 * 2) = (attn_u32x2){lo[q], hi[q]};
      } else {
        *reinterpret_cast<float4*>(&Ks[row * LD + c4]) = kv;
      }
      Vt[(c4 + 0) * VP + row] = vv.x; Vt[(c4 + 1) * VP + row] = vv.y; Vt[(c4 + 2) * VP + row] = vv.z; Vt[(c4 + 3) * VP + row] = vv.w;
    }
  }
  {
    int lv = -1;
    auto mask_key = [&](int key, unsigned char mb) {
      float mv = -INFINITY;
      if (key < p.Lk) mv = (p.key_mask && mb) ? -1e9f : 0.f;
      Ms[key] = mv;
      if (mv == 0.f) lv = key;             // keys ascend within a thread
    };
    if (tid < nkt * 16) mask_key(tid, mk0);
    if constexpr (MAXT * 16 > 256)
      for (int key = tid + 256; key < nkt * 16; key += 256) mask_key(key, kmp[min(key, last_key)]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lv = max(lv, __shfl_xor(lv, o, 64));
    if (lane == 0) last_valid[wave] = lv;
  }
  SKF_FSTAMP();      // this wave's share of K / V staged
  __syncthreads();
  if (tid < nkt) {
    int f = 0;
    for (int k = 0; k < 16; ++k) f |= Ms[tid * 16 + k] != 0.f;
    Tf[tid] = f;
  }
  __syncthreads();
  SKF_FSTAMP();      // staging complete

  // Causal tile skipping is exact only when key 0 is visible to every query
  // (then every row max is a real score and masked probabilities are exactly 0).
  const bool can_skip = p.causal && Ms[0] == 0.f;
  // Trailing key tiles that hold only padded keys contribute exactly 0 to every row that sees at least one real
  // key (2^(-1e9 - max) == 0 in fp32), so they are skipped: QuickDraw batches are ~60 % padding.  Not applied when
  // some row may have no visible key at all (then the softmax is uniform over ALL keys).
  const int lastk = max(max(last_valid[0], last_valid[1]), max(last_valid[2], last_valid[3]));
  const int nkt_eff = (lastk >= 0 && (!p.causal || can_skip)) ? (lastk >> 4) + 1 : nkt;
  const float c2 = 1.44269504088896340736f / sqrtf((float)DH);

  // Query tiles are dealt to the waves round-robin starting at a per-workgroup offset: with 13 tiles one wave gets 4 and
  // the others 3, and wave w always runs on SIMD w - without the rotation SIMD 0 of every CU carries the extra tile of
  // every resident workgroup.
  // SPLIT: per-lane plane bases of the three A operands ([k1|k0], [k1|k2], [k0|k0]) inside the K image
  const char* kp = reinterpret_cast<const char*>(Ks);
  const int plane_b = nkt * 16 * KPP, lane_k = i * KPP + (g & 1) * 16;
  const char* ka3 = kp + (g < 2 ? plane_b : 0) + lane_k;
  const char* ka2 = kp + (g < 2 ? plane_b : 2 * plane_b) + lane_k;
  const char* ka1 = kp + lane_k;
  // output rows / row statistics of this (sample, head): 32-bit byte offsets inside the sample (Lq * ldo * 4 < 2^31, checked on the host)
  const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.O + (size_t)b * p.Lq * p.ldo, 0, (unsigned)(p.Lq * p.ldo) * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.stats ? p.stats + (size_t)bh * p.Lq * 2 : p.O, 0,
                                                                           p.stats ? (unsigned)p.Lq * 8u : 0u, 0x00020000);
  for (int qt = (wave + bh) & 3; qt < nqt; qt += 4) {
    const int q0 = qt * 16, qrow = q0 + i;
    const bool qok = qrow < p.Lq;
    float4 qcur[NQ];
#pragma unroll
    for (int c = 0; c < NQ; ++c) qcur[c] = qnext[c];
    load_q(qt + 4, qnext);                // the next tile's rows travel during this tile's MFMAs
    float4 qf[NC];
    attn_u32x4 qb1, qb2, qb3;             // SPLIT: B operands [q0|q1], [q0|q0], [q1|q2]
    if constexpr (SPLIT) {
      // (round 5) the query rows carry the softmax scale log2(e)/sqrt(dh) into the split, like skf_attention_bwd3.hip's: one multiply per
      // query element instead of one per score, and the same split operands as the backward kernel forms
      const float4 qa = make_float4(qcur[0].x * c2, qcur[0].y * c2, qcur[0].z * c2, qcur[0].w * c2);
      const float4 qc = make_float4(qcur[1].x * c2, qcur[1].y * c2, qcur[1].z * c2, qcur[1].w * c2);
      unsigned d0[3], d1[3], d2[3], d3[3];
      skf_split2<3>(qa.x, qa.y, d0, sel); skf_split2<3>(qa.z, qa.w, d1, sel);
      skf_split2<3>(qc.x, qc.y, d2, sel); skf_split2<3>(qc.z, qc.w, d3, sel);
      const attn_u32x4 p0 = {d0[0], d1[0], d2[0], d3[0]}, p1 = {d0[1], d1[1], d2[1], d3[1]}, p2 = {d0[2], d1[2], d2[2], d3[2]};
      const bool first = g < 2;
      qb1 = first ? p0 : p1; qb2 = p0; qb3 = first ? p1 : p2;
    } else {
#pragma unroll
      for (int c = 0; c < NC; ++c) qf[c] = qcur[c];
    }
    const int nt = min(can_skip ? qt + 1 : nkt, nkt_eff);
    float s[MAXT][4];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt) {
      if (kt < nt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (SPLIT) {
          const attn_u32x4 a3 = *reinterpret_cast<const attn_u32x4*>(ka3 + kt * 16 * KPP);
          const attn_u32x4 a2 = *reinterpret_cast<const attn_u32x4*>(ka2 + kt * 16 * KPP);
          const attn_u32x4 a1 = *reinterpret_cast<const attn_u32x4*>(ka1 + kt * 16 * KPP);
          acc = mfma_bf16(a3, qb3, acc);      // k1.q1 + k0.q2   (smallest first)
          acc = mfma_bf16(a2, qb2, acc);      // k1.q0 + k2.q0
          acc = mfma_bf16(a1, qb1, acc);      // k0.q0 + k0.q1
        } else {
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const float4 kf = *reinterpret_cast<const float4*>(&Ks[(kt * 16 + i) * LD + c * 16 + g * 4]);
            acc = mfma16(kf.x, qf[c].x, acc);
            acc = mfma16(kf.y, qf[c].y, acc);
            acc = mfma16(kf.z, qf[c].z, acc);
            acc = mfma16(kf.w, qf[c].w, acc);
          }
        }
        // wave-uniform: does this tile need any masking at all?
        const bool need_mask = __builtin_amdgcn_readfirstlane(Tf[kt]) != 0 || (p.causal && kt >= qt);
        if (need_mask) {
          const float4 m4 = *reinterpret_cast<const float4*>(&Ms[kt * 16 + g * 4]);
          const float mr[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + g * 4 + r;
            const float m = fminf(mr[r], (p.causal && key > qrow) ? -1e9f : 0.f);   // one -1e9, never two
            const float v = m < 0.f ? m : (SPLIT ? acc[r] : acc[r] * c2);
            s[kt][r] = v;
            mx = fmaxf(mx, v);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s[kt][r] = SPLIT ? acc[r] : acc[r] * c2; mx = fmaxf(mx, s[kt][r]); }
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    f32x4 o[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt)
      if (kt < nt) {
        float pe[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { pe[r] = __builtin_amdgcn_exp2f(s[kt][r] - mx); sum += pe[r]; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float4 vt = *reinterpret_cast<const float4*>(&Vt[(c * 16 + i) * VP + kt * 16 + g * 4]);
          o[c] = mfma16(vt.x, pe[0], o[c]);
          o[c] = mfma16(vt.y, pe[1], o[c]);
          o[c] = mfma16(vt.z, pe[2], o[c]);
          o[c] = mfma16(vt.w, pe[3], o[c]);
        }
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float rinv = 1.0f / sum;
    // Stores through buffer descriptors, issued by every lane on every path (rows past Lq / lanes without a statistic fall outside
    // the descriptor): behind `if (qok)` the compiler cannot count them, and the wait for the NEXT tile's query rows at the top of
    // the loop became s_waitcnt vmcnt(0) - i.e. every tile also waited for the stores it had just issued.
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const f32x4 ov = {o[c][0] * rinv, o[c][1] * rinv, o[c][2] * rinv, o[c][3] * rinv};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(attn_u32x4, ov), o_rsrc,
                                             qok ? (unsigned)(qrow * p.ldo + h * DH + c * 16 + g * 4) * 4u : 0x7ffffff0u, 0, 0);
    }
    __builtin_amdgcn_raw_buffer_store_b64((attn_u32x2){__builtin_bit_cast(unsigned, mx), __builtin_bit_cast(unsigned, rinv)}, st_rsrc,
                                          (qok && g == 0) ? (unsigned)qrow * 8u : 0x7ffffff0u, 0, 0);
    SKF_FSTAMP();    // one per query tile
  }
}

// KTW = key tiles owned by one wave at a time (register budget: 20*NC*KTW accumulator/fragment registers)
// CAUSAL=false: the per-slot body is branch-free (key tiles past Lk are zero-filled dummies) so the compiler can overlap
// consecutive key tiles, and when nkt = 4*(KTW-1)+1 (L = 193..208 at dh=16) the odd 13th key tile is SHARED: every wave
// holds its fragments and processes it for the query tiles with (qt & 3) == wave; the four partial dK/dV are summed
// through LDS at the end (work per wave 42.25 pairs instead of 52 / 39 / 39 / 39).
#ifndef SKF_ATTN_BWD_WAVES
#define SKF_ATTN_BWD_WAVES 2   // waves per SIMD the register allocation aims at (3 = 168 VGPRs)
#endif
#ifndef SKF_ATTN_BWD_TRP
#define SKF_ATTN_BWD_TRP 0     // transpose patches per wave: 0 = one per key tile of the wave (KTW); 1 = one shared patch (20 -> 5 KB of LDS at dh = 16:
                               // three workgroups per CU instead of two when SKF_ATTN_BWD_WAVES = 3)
#endif
template <int DH, int KTW, bool CAUSAL>
__global__ __launch_bounds__(256, SKF_ATTN_BWD_WAVES) void attn_bwd_kernel(AttnParams p) {
  constexpr int NC = DH / 16;
  constexpr int LD = DH + 4;
  constexpr int TLD = 20;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // A head is a 64..256-byte slice of every activation row, i.e. half (or less) of each 128-byte line it touches; the
  // other half belongs to the neighbouring head.  XCD-contiguous ids put all heads of a sample on ONE XCD (one L2),
  // consecutively in time, so the neighbour's half is an L2 hit instead of a second HBM fetch of the same line.
  int bh = p.xcd_remap ? skf_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  if (p.order) { const int k = skf_deal_rank(blockIdx.x, p.H); bh = min(max(p.order[k / p.H], 0), p.B - 1) * p.H + k % p.H; }   // (clamped: a list that is no permutation must not leave the tensors)
  const int b = bh / p.H, h = bh % p.H;
  const int nkt = (p.Lk + 15) >> 4, nqt_all = (p.Lq + 15) >> 4;
  // query tiles behind the sample's last live row have dO == 0: they add nothing to dK / dV and their dQ is zero
  const int nqt = p.q_live ? min(nqt_all, (max(p.q_live[b], 0) + 15) >> 4) : nqt_all;
  const int QR = nqt_all * 16;
  constexpr int RLD = DH + 1;          // row pitch of the dQ reduction slots
  float* Qs = smem;                   // [QR][LD]
  float* dOs = Qs + QR * LD;          // [QR][LD]
  float* Red = dOs + QR * LD;         // [2 parities][4 waves][16 q][RLD]: per-query-tile dQ partials of the 4 waves
  float* Mx = Red + 2 * 4 * 16 * RLD; // [QR]
  float* Ri = Mx + QR;                // [QR]
  float* Dl = Ri + QR;                // [QR]  delta = sum_d dO*O
  float* Tr = Dl + QR;                // [4 waves][KTW][16][TLD]: one transpose patch per key tile of the wave, so the
                                      // key tiles of a query tile are independent chains the scheduler may interleave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;

  long long* dbg = (p.dbg && lane == 0 && (blockIdx.x % 131) == 0 && blockIdx.x / 131 < 8) ? p.dbg + ((blockIdx.x / 131) * 4 + wave) * 32 : nullptr;
  int dbi = 0;
#if SKF_MEASURE     // clock stamps of a few workgroups (tools/attn_timeline.py): measurement builds only
#define SKF_STAMP() do { if (dbg && dbi < 32) dbg[dbi++] = clock64(); } while (0)
#else
#define SKF_STAMP() do { (void)dbg; (void)dbi; } while (0)
#endif
  SKF_STAMP();
  // Everything the prologue reads from global memory is requested before its first wait: the key fragments of the first pass,
  // this thread's mask byte and row statistics, then the Q / dO / O rows (four serialised round trips before: rows, mask scan,
  // statistics, fragments - 8.4k + 3.9k cycles of a ~40k-cycle workgroup at one or two waves per SIMD).
  const unsigned char* km = p.key_mask ? p.key_mask + (size_t)b * p.key_mask_ld : nullptr;
  const int last_row = p.Lk - 1;
  const unsigned char* kmp = km ? km : reinterpret_cast<const unsigned char*>(p.K + (size_t)b * p.Lk * p.ldk);   // no mask: any readable bytes, ignored
  // key-tile ownership rotates with the workgroup id (see forward): the wave with the most key tiles (padding- and
  // causal-skipping make the load uneven) lands on a different SIMD for each of the co-resident workgroups
  const int wv = (wave + bh) & 3;
  // B-operand fragments (lane = key i, contraction d = 16c+4g+s), the raw elements of the A-operand (transposed) fragments
  // (lane = d 16c+i, contraction key = k0+4g+s) and the mask bytes of the wave's key tiles; from clamped (always valid) addresses,
  // rows past Lk are zeroed when they are consumed: written as guarded loads (`kr < Lk ? K[..] : 0`) each transposed element and
  // each mask byte became its own `global_load; s_waitcnt vmcnt(0)` - ~22 serialised memory round trips per workgroup.
  float4 kb[KTW][NC], vb[KTW][NC];
  float kraw[KTW][NC][4];
  unsigned char mkb[KTW];
  auto load_frags = [&](const int kg, const bool share) {
    const int kt0 = kg + wv;
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
      const int ktj = (share && j == KTW - 1) ? nkt - 1 : kt0 + 4 * j;
      const int k0 = ktj * 16;
      const int krc = min(k0 + i, last_row);
      mkb[j] = kmp[krc];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        kb[j][c] = *reinterpret_cast<const float4*>(p.K + (size_t)(b * p.Lk + krc) * p.ldk + h * DH + c * 16 + g * 4);
        vb[j][c] = *reinterpret_cast<const float4*>(p.V + (size_t)(b * p.Lk + krc) * p.ldv + h * DH + c * 16 + g * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
          kraw[j][c][s] = p.K[(size_t)(b * p.Lk + min(k0 + g * 4 + s, last_row)) * p.ldk + h * DH + c * 16 + i];
      }
    }
  };
  load_frags(0, !CAUSAL && KTW > 1 && nkt == 4 * (KTW - 1) + 1);
  const unsigned char mk0 = kmp[min(tid, last_row)];
  const float2 st0 = reinterpret_cast<const float2*>(p.stats)[(size_t)bh * p.Lq + min(tid, p.Lq - 1)];
  // Staging: all global loads of a batch are issued before the first LDS store (one HBM latency per batch of
  // 4 float4 x 3 arrays instead of one per element), delta = sum_d dO*O is reduced over the DH/4 lanes of a row.
  {
    constexpr int F4 = DH / 4;
    const int total = nqt * 16 * F4;
    for (int e0 = tid; e0 < total; e0 += 1024) {
      f32x4 qv[4], dv[4], ov[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 256, row = e / F4, c4 = (e % F4) * 4;
        const bool ok = e < total && row < p.Lq;
        const int rr = ok ? row : 0;
        qv[u] = *reinterpret_cast<const f32x4*>(p.Q + (size_t)(b * p.Lq + rr) * p.ldq + h * DH + c4);
        dv[u] = *reinterpret_cast<const f32x4*>(p.dO + (size_t)(b * p.Lq + rr) * p.lddo + h * DH + c4);
        ov[u] = *reinterpret_cast<const f32x4*>(p.O + (size_t)(b * p.Lq + rr) * p.ldo + h * DH + c4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 256, row = e / F4, c4 = (e % F4) * 4;
        const bool ok = e < total && row < p.Lq;
        const float z = ok ? 1.f : 0.f;
        const f32x4 q4 = qv[u] * z, d4 = dv[u] * z, o4 = ov[u] * z;
        float dl = d4[0] * o4[0] + d4[1] * o4[1] + d4[2] * o4[2] + d4[3] * o4[3];
#pragma unroll
        for (int o = 1; o < F4; o <<= 1) dl += __shfl_xor(dl, o, 64);
        if (e < total) {
          *reinterpret_cast<f32x4*>(&Qs[row * LD + c4]) = q4;
          *reinterpret_cast<f32x4*>(&dOs[row * LD + c4]) = d4;
          if (c4 == 0) Dl[row] = dl;
        }
      }
    }
  }
  int* last_valid = reinterpret_cast<int*>(Tr + 4 * (SKF_ATTN_BWD_TRP ? SKF_ATTN_BWD_TRP : KTW) * 16 * TLD);   // [4]: per-wave index of the last un-padded key
  {
    int lv = -1;
    if (tid < p.Lk && !(km && mk0)) lv = tid;
    for (int key = tid + 256; key < p.Lk; key += 256)
      if (!(km && km[key])) lv = key;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lv = max(lv, __shfl_xor(lv, o, 64));
    if (lane == 0) last_valid[wave] = lv;
  }
  for (int row = tid; row < nqt * 16; row += 256) {
    float2 st = make_float2(0.f, 0.f);
    if (row < p.Lq) st = row == tid ? st0 : reinterpret_cast<const float2*>(p.stats)[(size_t)bh * p.Lq + row];
    Mx[row] = st.x; Ri[row] = st.y;
  }
  // dQ of the dead query tiles: zeros (their tiles are never visited below)
  for (int e = nqt * 16 * DH + tid; e < p.Lq * DH; e += 256) p.dQ[(size_t)(b * p.Lq + e / DH) * p.lddq + h * DH + e % DH] = 0.f;
  __syncthreads();
  SKF_STAMP();   // staging done

  // skipping fully look-ahead-masked tiles is exact only if key 0 is visible (see forward)
  const bool can_skip = CAUSAL && !(km && km[0]);      // CAUSAL == p.causal (dispatch)
  // trailing all-padding key tiles have P == 0 exactly: their dK/dV are 0 and they add nothing to dQ (see forward)
  const int lastk = max(max(last_valid[0], last_valid[1]), max(last_valid[2], last_valid[3]));
  const int nkt_eff = (lastk >= 0 && (!CAUSAL || can_skip)) ? (lastk >> 4) + 1 : nkt;
  const float inv_sqrt = 1.0f / sqrtf((float)DH);
  const float c2 = 1.44269504088896340736f / sqrtf((float)DH);
  float* tr0 = Tr + wave * (SKF_ATTN_BWD_TRP ? SKF_ATTN_BWD_TRP : KTW) * 16 * TLD;
  f32x4 dK_shared[NC], dV_shared[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { dK_shared[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; dV_shared[c] = dK_shared[c]; }

  // One pass over the query tiles per group of 4 KTW key tiles.  The first pass STORES dQ, later ones (Lk > 64 KTW) add to it: as a
  // run-time `kg == 0 ? v : *dst + v` the conditional load sat in the query-tile loop of every launch, and the waits the compiler
  // scatters for a load that may be pending (register reuse) are s_waitcnt vmcnt(n) on the dQ STORES when it is not.
  auto kg_pass = [&](const int kg, auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const int kt0 = kg + wv;                   // smallest key tile of this wave in this group
    const bool share = !CAUSAL && FIRST && KTW > 1 && nkt == 4 * (KTW - 1) + 1;
    float kT[KTW][NC][4];
    float kadd[KTW];
    f32x4 dKt[KTW][NC], dVt[KTW][NC];
    if (!FIRST) load_frags(kg, false);         // (the first pass's fragments were requested at the top of the kernel)
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
      const int ktj = (share && j == KTW - 1) ? nkt - 1 : kt0 + 4 * j;
      const int k0 = ktj * 16, krow = k0 + i;
      const bool kok = krow < p.Lk;
      // keys past Lk get -inf (never -1e9): with a fully padded sample the row max itself is -1e9 and exp(x - max) would overflow
      kadd[j] = kok ? ((km && mkb[j]) ? -1e9f : 0.f) : -INFINITY;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (!kok) { kb[j][c] = make_float4(0.f, 0.f, 0.f, 0.f); vb[j][c] = kb[j][c]; }
#pragma unroll
        for (int s = 0; s < 4; ++s)   // pre-scaled by 1/sqrt(dh): dQ = (P o (dP - delta)) . K / sqrt(dh) without a multiply per score
          kT[j][c][s] = k0 + g * 4 + s < p.Lk ? kraw[j][c][s] * inv_sqrt : 0.f;
        dKt[j][c] = (f32x4){0.f, 0.f, 0.f, 0.f}; dVt[j][c] = dKt[j][c];
      }
    }

    SKF_STAMP();   // K/V fragments loaded (issued)
    // every wave walks ALL query tiles in lock step (the trip count must be workgroup-uniform: one barrier per tile)
    for (int qt = 0; qt < nqt; ++qt) {
      SKF_STAMP();
      const int q0 = qt * 16;
      float4 qa[NC], da[NC];
      float qT[NC][4], dT[NC][4];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        qa[c] = *reinterpret_cast<const float4*>(&Qs[(q0 + i) * LD + c * 16 + g * 4]);
        da[c] = *reinterpret_cast<const float4*>(&dOs[(q0 + i) * LD + c * 16 + g * 4]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          qT[c][r] = Qs[(q0 + g * 4 + r) * LD + c * 16 + i] * inv_sqrt;   // likewise for dK = dS^T . Q / sqrt(dh)
          dT[c][r] = dOs[(q0 + g * 4 + r) * LD + c * 16 + i];
        }
      }
      const float4 mx4 = *reinterpret_cast<const float4*>(&Mx[q0 + g * 4]);
      const float4 ri4 = *reinterpret_cast<const float4*>(&Ri[q0 + g * 4]);   // 0 for rows >= Lq
      const float4 dl4 = *reinterpret_cast<const float4*>(&Dl[q0 + g * 4]);
      const float mxr[4] = {mx4.x, mx4.y, mx4.z, mx4.w}, rir[4] = {ri4.x, ri4.y, ri4.z, ri4.w},
                  dlr[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
      f32x4 dq[NC], dq1[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) { dq[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; dq1[c] = dq[c]; }

#pragma unroll
      for (int j = 0; j < KTW; ++j) {
        const int kt = (share && j == KTW - 1) ? nkt - 1 : kt0 + 4 * j;
        if (kt >= nkt_eff) continue;                            // wave-uniform: all-padding (or non-existent) key tile
        if (CAUSAL) {
          if (can_skip && kt > qt) continue;
        } else if (share && j == KTW - 1) {
          if ((qt & 3) != wv) continue;                        // the shared tile: one wave per query tile
        }
        const int krow = kt * 16 + i;
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = sacc;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          sacc = mfma16(qa[c].x, kb[j][c].x, sacc);
          sacc = mfma16(qa[c].y, kb[j][c].y, sacc);
          sacc = mfma16(qa[c].z, kb[j][c].z, sacc);
          sacc = mfma16(qa[c].w, kb[j][c].w, sacc);
          dpacc = mfma16(da[c].x, vb[j][c].x, dpacc);
          dpacc = mfma16(da[c].y, vb[j][c].y, dpacc);
          dpacc = mfma16(da[c].z, vb[j][c].z, dpacc);
          dpacc = mfma16(da[c].w, vb[j][c].w, dpacc);
        }
        // lane holds rows q = q0+4g+r, column key = k0+i
        float pr[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = q0 + g * 4 + r;
          // base-2 logits, masked ones SET (see forward); keys >= Lk carry -inf, so their p is exactly 0
          const float m = CAUSAL ? fminf(kadd[j], krow > q ? -1e9f : 0.f) : kadd[j];
          const float v = m < 0.f ? m : sacc[r] * c2;
          const float pv = __builtin_amdgcn_exp2f(v - mxr[r]) * rir[r];
          pr[r] = pv;
          ds[r] = pv * (dpacc[r] - dlr[r]);                       // the 1/sqrt(dh) factor sits in qT / kT
        }
        // dV^T[d][k] += sum_q dO[q][d] P[q][k];  dK^T[d][k] += sum_q Q[q][d] dS[q][k]
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            dVt[j][c] = mfma16(dT[c][r], pr[r], dVt[j][c]);
            dKt[j][c] = mfma16(qT[c][r], ds[r], dKt[j][c]);
          }
        // transpose dS through the wave-private scratch: write [q][k], read [q=i][k=4g..4g+3]
        constexpr int TRP = SKF_ATTN_BWD_TRP ? SKF_ATTN_BWD_TRP : KTW;     // patches per wave
        float* tr = tr0 + (j % TRP) * 16 * TLD;
#pragma unroll
        for (int r = 0; r < 4; ++r) tr[(g * 4 + r) * TLD + i] = ds[r];
        // (no fence: the LDS executes one wave's instructions in order, and the patch is private to (wave, j))
        const float4 dst = *reinterpret_cast<const float4*>(&tr[i * TLD + g * 4]);
        // dQ^T[d][q] += sum_k K[k][d] dS[q][k]   (lane: d = 16c+4g+r, q = q0+i)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          dq[c] = mfma16(kT[j][c][0], dst.x, dq[c]);
          dq1[c] = mfma16(kT[j][c][1], dst.y, dq1[c]);
          dq[c] = mfma16(kT[j][c][2], dst.z, dq[c]);
          dq1[c] = mfma16(kT[j][c][3], dst.w, dq1[c]);
        }
      }
      // dQ of this query tile = sum of the 4 waves' partials: LDS float atomics cost ~20 cycles per lane on gfx950
      // (they were 35 % of this kernel), so each wave writes its 16 x DH partial to its own slot, one barrier, and
      // the workgroup sums the 4 slots straight into global dQ (double-buffered slots -> one barrier per query tile).
      {
        float* slot = Red + ((qt & 1) * 4 + wave) * 16 * RLD;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) slot[i * RLD + c * 16 + g * 4 + r] = dq[c][r] + dq1[c][r];
        __syncthreads();
        const float* base = Red + (qt & 1) * 4 * 16 * RLD;
        for (int e = tid; e < 16 * DH; e += 256) {
          const int qq = e / DH, dd = e % DH;
          if (q0 + qq < p.Lq) {
            const float v = base[qq * RLD + dd] + base[16 * RLD + qq * RLD + dd] + base[2 * 16 * RLD + qq * RLD + dd] +
                            base[3 * 16 * RLD + qq * RLD + dd];
            float* dst = p.dQ + (size_t)(b * p.Lq + q0 + qq) * p.lddq + h * DH + dd;
            *dst = FIRST ? v : *dst + v;
          }
        }
      }
    }
    if (share) {
#pragma unroll
      for (int c = 0; c < NC; ++c) { dK_shared[c] = dKt[KTW - 1][c]; dV_shared[c] = dVt[KTW - 1][c]; }
    }
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
      if (share && j == KTW - 1) continue;                    // reduced across waves below
      const int krow = (kt0 + 4 * j) * 16 + i;
      if (krow < p.Lk) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          *reinterpret_cast<float4*>(p.dK + (size_t)(b * p.Lk + krow) * p.lddk + h * DH + c * 16 + g * 4) =
              make_float4(dKt[j][c][0], dKt[j][c][1], dKt[j][c][2], dKt[j][c][3]);
          *reinterpret_cast<float4*>(p.dV + (size_t)(b * p.Lk + krow) * p.lddv + h * DH + c * 16 + g * 4) =
              make_float4(dVt[j][c][0], dVt[j][c][1], dVt[j][c][2], dVt[j][c][3]);
        }
      }
    }
  };
  kg_pass(0, std::true_type());
  for (int kg = 4 * KTW; kg < nkt; kg += 4 * KTW) kg_pass(kg, std::false_type());
  SKF_STAMP();   // wave done
  __syncthreads();
  SKF_STAMP();   // all waves done
  if (!CAUSAL && KTW > 1 && nkt == 4 * (KTW - 1) + 1) {
    // partial dK^T / dV^T of the shared key tile: Qs is free now, use it as [4 waves][2][DH][16] scratch
    float* sh = Qs + wave * 2 * DH * 16;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sh[(c * 16 + g * 4 + r) * 16 + i] = dK_shared[c][r];
        sh[DH * 16 + (c * 16 + g * 4 + r) * 16 + i] = dV_shared[c][r];
      }
    __syncthreads();
    const int krow0 = (nkt - 1) * 16;
    for (int e = tid; e < 2 * DH * 16; e += 256) {
      const int which = e / (DH * 16), dd = (e % (DH * 16)) / 16, kk = e % 16;
      const float v = Qs[e] + Qs[2 * DH * 16 + e] + Qs[4 * DH * 16 + e] + Qs[6 * DH * 16 + e];
      if (krow0 + kk < p.Lk) {
        float* dst = which ? p.dV : p.dK;
        const int ld = which ? p.lddv : p.lddk;
        dst[(size_t)(b * p.Lk + krow0 + kk) * ld + h * DH + dd] = v;
      }
    }
    __syncthreads();
  }
}

size_t fwd_smem(int DH, int Lk, bool split) {
  const size_t n16 = (size_t)(Lk + 15) / 16 * 16;
  const size_t krow = split ? 3 * KPP / 4 : DH + 4;   // floats per key row of the K image
  return (n16 * krow + (size_t)DH * (n16 + 8) + n16 + n16 / 16 + 4) * sizeof(float);
}
size_t bwd_smem(int DH, int Lq) {
  const size_t QR = (size_t)(Lq + 15) / 16 * 16;
  return (2 * QR * (DH + 4) + (size_t)2 * 4 * 16 * (DH + 1) + 3 * QR + (size_t)4 * (SKF_ATTN_BWD_TRP ? SKF_ATTN_BWD_TRP : 64 / DH) * 16 * 20 + 4) * sizeof(float);
}

template <typename K>
int set_smem(K kfn, size_t bytes) {
  SKF_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return SKF_OK;
}

inline bool mfma_head(int dh) { return dh == 16 || dh == 32 || dh == 64; }
int check_common(const AttnParams& p, int dh) {
  SKF_CHECK_ARG(mfma_head(dh) || skf_attention_any_supported(dh, p.Lq, p.Lk), "head dim must be 16, 32, 64 (MFMA kernels) or any size <= 128 with sequences <= 1024 (fallback)");
  SKF_CHECK_ARG(p.B > 0 && p.H > 0 && p.Lq > 0 && p.Lk > 0, "empty problem");
  SKF_CHECK_ARG((p.ldq & 3) == 0 && (p.ldk & 3) == 0 && (p.ldv & 3) == 0 && (p.ldo & 3) == 0, "row strides must be multiples of 4");
  SKF_CHECK_ARG(!p.causal || p.Lq == p.Lk, "causal attention needs Lq == Lk");
  return SKF_OK;
}

}  // namespace

extern "C" int skf_attention_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                                 const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk,
                                 int dh, float* O, int ldo, float* stats, int precision, skf_stream_t stream) {
  return skf_attention_fwd_ordered(Q, ldq, K, ldk, V, ldv, key_mask, key_mask_ld, causal, B, H, Lq, Lk, dh, O, ldo, stats, precision, nullptr, stream);
}

extern "C" int skf_attention_fwd_ordered(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                                         const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk,
                                         int dh, float* O, int ldo, float* stats, int precision, const int* sample_order, skf_stream_t stream) {
  AttnParams p{};
  p.order = (B & 7) == 0 ? sample_order : nullptr;      // (the deal needs whole rounds of the 8 XCDs; results never depend on it)
  p.Q = Q; p.K = K; p.V = V; p.O = O; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.key_mask = key_mask; p.key_mask_ld = key_mask_ld; p.causal = causal; p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.stats = stats;
  { const char* e = skf_knob("SKF_ATTN_XCD"); p.xcd_remap = !(e && e[0] == '0'); }
#if SKF_MEASURE
  { const char* db = skf_knob("SKF_ATTN_DBG"); p.dbg = db ? (long long*)strtoull(db, nullptr, 0) : nullptr; }
#endif
  int rc = check_common(p, dh);
  if (rc) return rc;
  SKF_CHECK_ARG(Q && K && V && O, "null operand");
  if (!mfma_head(dh)) return skf_attention_fwd_any(p, dh, (hipStream_t)stream);
  SKF_CHECK_ARG(Lk <= 512, "Lk > 512 not supported");
  SKF_CHECK_ARG((double)Lq * ldo * 4 < 2147483648.0, "one sample's output rows exceed 32-bit byte offsets");
  // S^T on the bf16 pipe follows the Dense arithmetic switch (SKF_ATTN_SPLIT=0 turns it off).  With padded 48-byte plane
  // rows it removed 26 % of the MFMA cycles and changed nothing (the forward is wait-bound: 45 % of the wave cycles parked,
  // and the planes cost the fourth resident workgroup per CU); with unpadded rows (four workgroups per CU again, 2-way bank
  // conflicts) it is 4-11 % faster than the fp32-MFMA tiles: 36.4 / 29.0 / 42.1 vs 39.2 / 30.2 / 47.2 us.
  static const bool split_off = skf_knob("SKF_ATTN_SPLIT") && skf_knob("SKF_ATTN_SPLIT")[0] == '0';
  const bool split = dh == 16 && !split_off && precision != SKF_PREC_F32;
  const size_t smem = fwd_smem(dh, Lk, split);
  SKF_CHECK_ARG(smem <= 160 * 1024, "K/V of one head do not fit in LDS");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(B * H), block(256);
#define SKF_ATTN_FWD(DHV, MT, SP)                               \
  {                                                             \
    auto kfn = attn_fwd_kernel<DHV, MT, SP>;                    \
    if ((rc = set_smem(kfn, smem))) return rc;                  \
    hipLaunchKernelGGL(kfn, grid, block, smem, st, p);          \
  }
  static const char* const tags[3] = {"attn_fwd<dh16>", "attn_fwd<dh32>", "attn_fwd<dh64>"};
  const double visited = skf_prof_attention_fraction(key_mask, key_mask_ld, causal, B, Lq, Lk, nullptr, 16, 16);
  SkfProfScope ps(st, tags[dh == 16 ? 0 : dh == 32 ? 1 : 2], 4.0 * B * H * (double)Lq * Lk * dh,
                  4.0 * B * H * dh * (2.0 * Lq + 2.0 * Lk));
  ps.done(4.0 * B * H * (double)Lq * Lk * dh * visited, 4.0 * B * H * dh * (2.0 * Lq + 2.0 * Lk));
  const bool small = Lk <= 208;
  if (dh == 16 && split) { if (small) SKF_ATTN_FWD(16, 13, true) else SKF_ATTN_FWD(16, 32, true) }
  else if (dh == 16) { if (small) SKF_ATTN_FWD(16, 13, false) else SKF_ATTN_FWD(16, 32, false) }
  else if (dh == 32) { if (small) SKF_ATTN_FWD(32, 13, false) else SKF_ATTN_FWD(32, 32, false) }
  else { if (small) SKF_ATTN_FWD(64, 13, false) else SKF_ATTN_FWD(64, 32, false) }
#undef SKF_ATTN_FWD
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_attention_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                                 const float* O, int ldo, const float* dO, int lddo, const float* stats,
                                 const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk,
                                 int dh, float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, int precision, skf_stream_t stream) {
  return skf_attention_bwd_ordered(Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, stats, key_mask, key_mask_ld, causal, B, H, Lq, Lk, dh, dQ, lddq,
                                   dK, lddk, dV, lddv, precision, nullptr, nullptr, stream);
}

extern "C" int skf_attention_bwd_rows(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                                      const float* O, int ldo, const float* dO, int lddo, const float* stats,
                                      const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk,
                                      int dh, float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, int precision,
                                      const int* q_live_len, skf_stream_t stream) {
  return skf_attention_bwd_ordered(Q, ldq, K, ldk, V, ldv, O, ldo, dO, lddo, stats, key_mask, key_mask_ld, causal, B, H, Lq, Lk, dh, dQ, lddq,
                                   dK, lddk, dV, lddv, precision, q_live_len, nullptr, stream);
}

extern "C" int skf_attention_bwd_ordered(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                                         const float* O, int ldo, const float* dO, int lddo, const float* stats,
                                         const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk,
                                         int dh, float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, int precision,
                                         const int* q_live_len, const int* sample_order, skf_stream_t stream) {
  const bool two_pass = (precision & SKF_ATTN_TWO_PASS) != 0;
  precision &= ~SKF_ATTN_TWO_PASS;
  AttnParams p{};
  p.order = (B & 7) == 0 ? sample_order : nullptr;
  p.q_live = q_live_len;
  p.Q = Q; p.K = K; p.V = V; p.O = const_cast<float*>(O); p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.key_mask = key_mask; p.key_mask_ld = key_mask_ld; p.causal = causal; p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk;
  p.stats = const_cast<float*>(stats);
  { const char* ab = skf_knob("SKF_ATTN_ABLATE"); p.ablate = ab ? atoi(ab) : 0; }
  { const char* e = skf_knob("SKF_ATTN_XCD"); p.xcd_remap = !(e && e[0] == '0'); }
#if SKF_MEASURE
  { const char* db = skf_knob("SKF_ATTN_DBG"); p.dbg = db ? (long long*)strtoull(db, nullptr, 0) : nullptr; }
#endif
  p.dO = dO; p.lddo = lddo; p.dQ = dQ; p.dK = dK; p.dV = dV; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  int rc = check_common(p, dh);
  if (rc) return rc;
  SKF_CHECK_ARG(Q && K && V && O && dO && stats && dQ && dK && dV, "null operand");
  SKF_CHECK_ARG((lddo & 3) == 0 && (lddq & 3) == 0 && (lddk & 3) == 0 && (lddv & 3) == 0, "row strides must be multiples of 4");
  if (!mfma_head(dh)) return skf_attention_bwd_any(p, dh, (hipStream_t)stream);
  // head size 16 / 32 in the split arithmetic modes: the two-pass kernel on the bf16 matrix cores (skf_attention_bwd2.hip);
  // SKF_PREC_F32 keeps the fp32-MFMA kernel below (SKF_ATTN_BWD2=0 forces it)
  // Head size 16: only the causal (decoder self-attention) calls take it - measured at the cfg-2 shape, the one-pass kernel
  // below is faster without a look-ahead mask (encoder self 89 vs 95 us; cross 101 vs 124 us with every dO row live, and
  // with the dead query tiles left out by q_live it also wins on padded batches); SKF_ATTN_BWD2=1 forces the two-pass kernel.
  static const char* bwd2_env = skf_knob("SKF_ATTN_BWD2");       // measurement builds only
  const bool bwd2_off = bwd2_env && bwd2_env[0] == '0', bwd2_all = (bwd2_env && bwd2_env[0] == '1') || two_pass;
  // (round 3: with its prologue loads batched the one-pass kernel also wins the causal dh = 16 calls - 4.134 vs 4.149 ms/step padded,
  //  4.777 vs 4.774 full-length at cfg 2 - so head size 16 takes the two-pass kernel only when forced; head size 32 keeps it:
  //  15.9 vs 17.2 ms/step at cfg 3)
  // round 5: head size 16, sequences up to 208, split modes: skf_attention_bwd3.hip (one workgroup per head stages every operand
  // once, the dQ items and the dK / dV items of the two-pass scheme run side by side from one queue, per-score arithmetic folded
  // into operands and accumulator seeds); SKF_ATTN_BWD3=0 (measurement builds) or SKF_ATTN_TWO_PASS keep the older kernels reachable
  const char* bwd3_env = skf_knob("SKF_ATTN_BWD3");          // (per call: tools/attn_bwd3_ablate.py flips it inside one process)
  if (skf_attention_bwd3_supported(dh, Lq, Lk) && precision != SKF_PREC_F32 && !bwd2_all && !bwd2_off && !(bwd3_env && bwd3_env[0] == '0'))
    return skf_attention_bwd3_launch(p, (hipStream_t)stream);
  const bool bwd2_shape = (dh == 16 && bwd2_all) || (dh == 32 && Lk <= 256 && Lq <= 256);
  if (bwd2_shape && precision != SKF_PREC_F32 && !bwd2_off && Lk <= 512 && Lq <= 512)
    return skf_attention_bwd2_launch(p, dh, (hipStream_t)stream);
  const size_t smem = bwd_smem(dh, Lq);
  SKF_CHECK_ARG(smem <= 160 * 1024, "Q/dO/dQ of one head do not fit in LDS");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(B * H), block(256);
#define SKF_ATTN_BWD(DHV)                                       \
  {                                                             \
    if (causal) {                                               \
      auto kfn = attn_bwd_kernel<DHV, 64 / DHV, true>;          \
      if ((rc = set_smem(kfn, smem))) return rc;                \
      hipLaunchKernelGGL(kfn, grid, block, smem, st, p);        \
    } else {                                                    \
      auto kfn = attn_bwd_kernel<DHV, 64 / DHV, false>;         \
      if ((rc = set_smem(kfn, smem))) return rc;                \
      hipLaunchKernelGGL(kfn, grid, block, smem, st, p);        \
    }                                                           \
  }
  static const char* const tags[3] = {"attn_bwd<dh16>", "attn_bwd<dh32>", "attn_bwd<dh64>"};
  const double visited = skf_prof_attention_fraction(key_mask, key_mask_ld, causal, B, Lq, Lk, q_live_len, 16, 16);
  SkfProfScope ps(st, tags[dh == 16 ? 0 : dh == 32 ? 1 : 2], 8.0 * B * H * (double)Lq * Lk * dh,
                  4.0 * B * H * dh * (4.0 * Lq + 4.0 * Lk));
  ps.done(8.0 * B * H * (double)Lq * Lk * dh * visited, 4.0 * B * H * dh * (4.0 * Lq + 4.0 * Lk));
  if (dh == 16) SKF_ATTN_BWD(16) else if (dh == 32) SKF_ATTN_BWD(32) else SKF_ATTN_BWD(64)
#undef SKF_ATTN_BWD
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
