// Attention backward, head size 16, two passes on the bf16 matrix cores with exactly split fp32 operands.
//
// tape gradient of builders/utils.py:71-105 (scaled_dot_product_attention) with the masks of builders/utils.py:35-68; same
// semantics, statistics and skipping rules as attn_bwd_kernel in skf_attention.hip, which stays the SKF_PREC_F32 path.
//
// Why two passes.  The one-pass kernel keeps dK / dV of a key tile in registers and must therefore transpose dS (through LDS)
// for dQ and sum the dQ partials of its four waves (LDS slots + a barrier per query tile): a latency chain per tile pair that
// left the MFMA pipe 22-36 % busy.  Here
//   pass A (dQ)     : a wave owns query tiles.  S^T = K.Q^T and dP^T = V.dO^T come out with a lane holding 4 keys of ONE query
//                     (C layout: col = query, rows = keys), so mx / 1/sum / delta are per-lane scalars and dS^T is ALREADY the
//                     B operand of dQ^T += K^T.dS^T; K^T comes from the row-major K planes through ds_read_b64_tr_b16.
//   pass B (dK, dV) : a wave owns key tiles.  S = Q.K^T and dP = dO.V^T with a lane holding 4 queries of ONE key; P and dS are
//                     the B operands of dV^T += dO^T.P and dK^T += Q^T.dS (dO^T / Q^T by transposing reads).
// No transposes through memory, no cross-wave reduction, no barrier inside either pass; S and dP are computed twice
// (21 bf16 MFMAs of 16 cycles per tile pair instead of 20 fp32 MFMAs of 32 cycles).
//
// Arithmetic (the bf16x6 scheme of skf_common.h): every fp32 operand = three bf16 pieces, the six piece products x_i.y_j
// (i + j <= 2) are accumulated in fp32, smallest first.  All contractions here are 16 deep (dh = 16, or the 16 rows of a
// tile), so one 32-deep v_mfma_f32_16x16x32_bf16 carries TWO piece products:
//   contraction over d   : lane group g supplies d = 8(g&1).. of the first (g < 2) or second (g >= 2) product of the pair:
//                          [x1|x0].[y1|y2] + [x1|x2].[y0|y0] + [x0|x0].[y0|y1]  (A from 32-byte plane rows, B from registers)
//   contraction over rows: slot (g, e) = row 4g + (e & 3) of product (e >> 2): the four rows a lane holds in the C layout are
//                          split in registers ([y0|y1], [y0|y0], [y1|y2]) and the A side ([x0|x0], [x1|x2], [x1|x0]) is two
//                          transposing reads of plane rows 4g.. per piece - no data moves between lanes.
// LDS: two tensors x three planes x 32-byte rows (K, V for pass A; re-staged with Q, dO for pass B) + delta = 40 KB at
// L = 200: four workgroups per CU, i.e. all B*H = 1024 workgroups of a cfg-2 launch resident at once; key / query tile
// ownership rotates with the workgroup id so that the waves with one tile more land on different SIMDs.
#include <stdlib.h>
#include "skf_attention_params.h"

namespace {

typedef __bf16 b2_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned b2_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned b2_u32x2 __attribute__((ext_vector_type(2)));
typedef short b2_s4 __attribute__((ext_vector_type(4)));

// NC = head size / 16.  Plane rows are 32 NC bytes (16 NC bf16).

__device__ __forceinline__ f32x4 mfma_x(b2_u32x4 a, b2_u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b2_bf16x8, a), __builtin_bit_cast(b2_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ b2_u32x2 tr_read(const char* a) {
  const b2_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((b2_s4 __attribute__((address_space(3)))*)a);
  return __builtin_bit_cast(b2_u32x2, v);
}

// B-side operands of a contraction over d from the lane's 8 fp32 values.  NC = 1 (16 deep: two piece products per MFMA):
// d = 8(g&1).. and b1 / b2 / b3 are the paired operands [y0|y1] / [y0|y0] / [y1|y2] selected by the lane group's half.
// NC = 2 (32 deep: one piece product per MFMA): d = 8g.. and b1 / b2 / b3 are simply the pieces y0 / y1 / y2.
struct DOps { b2_u32x4 b1, b2, b3; };
template <int NC>
__device__ __forceinline__ DOps d_ops(const float4& lo4, const float4& hi4, bool first, const SkfSplitSel& sel) {
  unsigned d0[3], d1[3], d2[3], d3[3];
  skf_split2<3>(lo4.x, lo4.y, d0, sel); skf_split2<3>(lo4.z, lo4.w, d1, sel);
  skf_split2<3>(hi4.x, hi4.y, d2, sel); skf_split2<3>(hi4.z, hi4.w, d3, sel);
  const b2_u32x4 p0 = {d0[0], d1[0], d2[0], d3[0]}, p1 = {d0[1], d1[1], d2[1], d3[1]}, p2 = {d0[2], d1[2], d2[2], d3[2]};
  DOps o;
  if constexpr (NC == 1) { o.b1 = first ? p0 : p1; o.b2 = p0; o.b3 = first ? p1 : p2; }
  else { o.b1 = p0; o.b2 = p1; o.b3 = p2; }
  return o;
}
// x.y over d: A = plane rows of tensor X, B = DOps of Y.  NC = 1: a1 / a2 / a3 are the per-lane addresses of the paired planes
// [x0|x0] / [x1|x2] / [x1|x0]; NC = 2: of planes 0 / 1 / 2.  Six piece products, smallest first.
template <int NC>
__device__ __forceinline__ f32x4 dot_d(const char* a1, const char* a2, const char* a3, int tile_off, const DOps& y) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (NC == 1) {
    acc = mfma_x(*reinterpret_cast<const b2_u32x4*>(a3 + tile_off), y.b3, acc);      // x1.y1 + x0.y2
    acc = mfma_x(*reinterpret_cast<const b2_u32x4*>(a2 + tile_off), y.b2, acc);      // x1.y0 + x2.y0
    acc = mfma_x(*reinterpret_cast<const b2_u32x4*>(a1 + tile_off), y.b1, acc);      // x0.y0 + x0.y1
  } else {
    const b2_u32x4 x0 = *reinterpret_cast<const b2_u32x4*>(a1 + tile_off), x1 = *reinterpret_cast<const b2_u32x4*>(a2 + tile_off),
                   x2 = *reinterpret_cast<const b2_u32x4*>(a3 + tile_off);
    acc = mfma_x(x1, y.b2, acc);      // x1.y1
    acc = mfma_x(x0, y.b3, acc);      // x0.y2
    acc = mfma_x(x2, y.b1, acc);      // x2.y0
    acc = mfma_x(x1, y.b1, acc);      // x1.y0
    acc = mfma_x(x0, y.b2, acc);      // x0.y1
    acc = mfma_x(x0, y.b1, acc);      // x0.y0
  }
  return acc;
}
// acc[c][d][col] += sum over the 16 rows of a tile X^T[16c + d][row] . y[row][col]; y = the lane's 4 rows (C layout), X = plane
// rows read transposed (tr = per-lane address of plane 0 at tile row 4g + (j >> 2), column 4 (j & 3); planes are plane_b apart,
// the second 16 columns 32 bytes further)
template <int NC>
__device__ __forceinline__ void dot_rows(const char* tr, int plane_b, const f32x4& y, f32x4 (&acc)[NC], const SkfSplitSel& sel) {
  unsigned lo[3], hi[3];
  skf_split2<3>(y[0], y[1], lo, sel);
  skf_split2<3>(y[2], y[3], hi, sel);
  const b2_u32x4 b3 = {lo[1], hi[1], lo[2], hi[2]}, b2 = {lo[0], hi[0], lo[0], hi[0]}, b1 = {lo[0], hi[0], lo[1], hi[1]};
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const b2_u32x2 t0 = tr_read(tr + c * 32), t1 = tr_read(tr + c * 32 + plane_b), t2 = tr_read(tr + c * 32 + 2 * plane_b);
    const b2_u32x4 a3 = {t1[0], t1[1], t0[0], t0[1]};     // x1.y1 + x0.y2
    const b2_u32x4 a2 = {t1[0], t1[1], t2[0], t2[1]};     // x1.y0 + x2.y0
    const b2_u32x4 a1 = {t0[0], t0[1], t0[0], t0[1]};     // x0.y0 + x0.y1
    acc[c] = mfma_x(a3, b3, acc[c]);
    acc[c] = mfma_x(a2, b2, acc[c]);
    acc[c] = mfma_x(a1, b1, acc[c]);
  }
}

// rows [0, R) of X1 / X2 (16 columns each) -> three bf16 planes each at dst1 / dst2 (plane_b apart); rows >= nrows are zeros.
// All global loads of a batch (4 float4 per tensor and thread) are issued before the first split / LDS store: a staging loop
// that loads, splits and stores one element at a time pays one memory round trip (2-4 us under load) per iteration.
template <int NC>
__device__ __forceinline__ void stage_planes2(char* dst1, const float* X1, int ld1, int n1, char* dst2, const float* X2, int ld2, int n2,
                                              int plane_b, int R, int tid, const SkfSplitSel& sel) {
  constexpr int RP = 32 * NC, CH = 4 * NC;           // float4 chunks per row
  for (int e0 = tid; e0 < R * CH; e0 += 1024) {
    float4 a[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + 256 * u, row = e / CH, c4 = (e % CH) * 4;
      const int r1 = row < n1 ? row : 0, r2 = row < n2 ? row : 0;
      a[u] = *reinterpret_cast<const float4*>(X1 + (size_t)r1 * ld1 + c4);
      c[u] = *reinterpret_cast<const float4*>(X2 + (size_t)r2 * ld2 + c4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + 256 * u, row = e / CH, c4 = (e % CH) * 4;
      if (e < R * CH) {
        const float z1 = row < n1 ? 1.f : 0.f, z2 = row < n2 ? 1.f : 0.f;
        unsigned lo[3], hi[3];
        skf_split2<3>(a[u].x * z1, a[u].y * z1, lo, sel);
        skf_split2<3>(a[u].z * z1, a[u].w * z1, hi, sel);
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<b2_u32x2*>(dst1 + q * plane_b + row * RP + c4 * 2) = (b2_u32x2){lo[q], hi[q]};
        skf_split2<3>(c[u].x * z2, c[u].y * z2, lo, sel);
        skf_split2<3>(c[u].z * z2, c[u].w * z2, hi, sel);
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<b2_u32x2*>(dst2 + q * plane_b + row * RP + c4 * 2) = (b2_u32x2){lo[q], hi[q]};
      }
    }
  }
}

// per-query-tile operands of pass A, loaded one tile ahead (a global load under a busy chip returns after 2-4 k cycles; every load
// is unconditional - clamped row, zeroed afterwards - so that the compiler can count it instead of waiting for vmcnt(0))
struct QTile { float4 q0, q1, g0, g1, o0, o1; float2 st; };
struct KTile { float4 k0, k1, v0, v1; float kadd; };

template <int NC, bool CAUSAL>
__global__ __launch_bounds__(256, NC == 1 ? 4 : 2) void attn_bwd2_kernel(AttnParams p) {
  constexpr int RP = 32 * NC, DH = 16 * NC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bh = p.xcd_remap ? skf_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  if (p.order) { const int k = skf_deal_rank(blockIdx.x, p.H); bh = min(max(p.order[k / p.H], 0), p.B - 1) * p.H + k % p.H; }   // (clamped: a list that is no permutation must not leave the tensors)
  const int b = bh / p.H, h = bh % p.H;
  const int nkt = (p.Lk + 15) >> 4, nqt = (p.Lq + 15) >> 4;
  const int R = (nkt > nqt ? nkt : nqt) * 16;
  const int plane_b = R * RP;
  char* TA = smem;                                    // K planes (pass A) / Q planes (pass B)
  char* TB = smem + 3 * plane_b;                      // V planes (pass A) / dO planes (pass B)
  float* Dl = reinterpret_cast<float*>(smem + 6 * plane_b);      // [R] delta = rowsum(dO o O), zeros past Lq
  unsigned* Kbits = reinterpret_cast<unsigned*>(Dl + R);         // [32] bit i of word t: key 16 t + i is padded (key mask)
  int* red = reinterpret_cast<int*>(Kbits + 32);                 // [0] last un-padded key, [1] bit t: query tile t has a non-zero dO row
  const SkfSplitSel sel = skf_split_sel();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const bool first = g < 2;
  const int d0 = NC == 1 ? 8 * (g & 1) : 8 * g;      // the lane's 8 columns of a row operand
  const unsigned char* km = p.key_mask ? p.key_mask + (size_t)b * p.key_mask_ld : nullptr;
  const float* Qb = p.Q + (size_t)b * p.Lq * p.ldq + h * DH;
  const float* Kb = p.K + (size_t)b * p.Lk * p.ldk + h * DH;
  const float* Vb = p.V + (size_t)b * p.Lk * p.ldv + h * DH;
  const float* Ob = p.O + (size_t)b * p.Lq * p.ldo + h * DH;
  const float* dOb = p.dO + (size_t)b * p.Lq * p.lddo + h * DH;
  const float2* stats = reinterpret_cast<const float2*>(p.stats) + (size_t)bh * p.Lq;

  // ---------------- phase 0: K / V planes, delta, key-mask bits
  if (tid < 32) Kbits[tid] = 0u;
  if (tid == 32) { red[0] = -1; red[1] = 0; }
  __syncthreads();
  stage_planes2<NC>(TA, Kb, p.ldk, p.Lk, TB, Vb, p.ldv, p.Lk, plane_b, R, tid, sel);
  __syncthreads();
  for (int key = tid; key < p.Lk; key += 256) {
    if (km && km[key]) atomicOr(&Kbits[key >> 4], 1u << (key & 15));
    else atomicMax(&red[0], key);
  }
  __syncthreads();
  const int lastk = red[0];
  // skipping fully look-ahead-masked tiles is exact only if key 0 is visible; trailing all-padding key tiles have P == 0 exactly
  // unless some row may see no key at all (see skf_attention.hip)
  const bool can_skip = CAUSAL && !(km && km[0]);
  const int nkt_eff = (lastk >= 0 && (!CAUSAL || can_skip)) ? (lastk >> 4) + 1 : nkt;
  const float inv_sqrt = NC == 1 ? 0.25f : 0.17677669529663687f;  // 1 / sqrt(dh)
  const float c2 = 1.44269504088896340736f * inv_sqrt;
  const int wv = (wave + bh) & 3;
  // per-lane plane addresses: contraction over d (row j of a tile, bytes 16 (g&1)..) and transposing reads (row 4g + (j>>2))
  const int lane_d = j * RP + (NC == 1 ? (g & 1) : g) * 16;
  // plane offsets of the three A operands of a contraction over d (NC = 1: paired planes by lane half, NC = 2: planes 0 / 1 / 2)
  const int pl1 = 0, pl2 = NC == 1 ? (first ? plane_b : 2 * plane_b) : plane_b, pl3 = NC == 1 ? (first ? plane_b : 0) : 2 * plane_b;
  const int lane_t = (4 * g + (j >> 2)) * RP + (j & 3) * 8;

  // ---------------- pass A: dQ
  {
    const char* ka1 = TA + pl1 + lane_d;
    const char* ka2 = TA + pl2 + lane_d;
    const char* ka3 = TA + pl3 + lane_d;
    const char* va1 = TB + pl1 + lane_d;
    const char* va2 = TB + pl2 + lane_d;
    const char* va3 = TB + pl3 + lane_d;
    const char* ktr = TA + lane_t;
    auto load_q = [&](int qt_) {
      const int qq = qt_ * 16 + j, qc = qq < p.Lq ? qq : p.Lq - 1;
      QTile t;
      t.q0 = *reinterpret_cast<const float4*>(Qb + (size_t)qc * p.ldq + d0); t.q1 = *reinterpret_cast<const float4*>(Qb + (size_t)qc * p.ldq + d0 + 4);
      t.g0 = *reinterpret_cast<const float4*>(dOb + (size_t)qc * p.lddo + d0); t.g1 = *reinterpret_cast<const float4*>(dOb + (size_t)qc * p.lddo + d0 + 4);
      t.o0 = *reinterpret_cast<const float4*>(Ob + (size_t)qc * p.ldo + d0); t.o1 = *reinterpret_cast<const float4*>(Ob + (size_t)qc * p.ldo + d0 + 4);
      t.st = stats[qc];
      return t;
    };
    QTile nx = load_q(wv < nqt ? wv : 0);
    for (int qt = wv; qt < nqt; qt += 4) {
      QTile cur = nx;
      nx = load_q(qt + 4 < nqt ? qt + 4 : qt);
      const int q = qt * 16 + j;
      const bool qok = q < p.Lq;
      const float mx = cur.st.x, ri = qok ? cur.st.y : 0.f;      // rows past Lq: P == 0
      if (!qok) { cur.q0 = cur.q1 = cur.g0 = cur.g1 = make_float4(0.f, 0.f, 0.f, 0.f); }          // (delta of those rows = 0)
      // delta = rowsum(dO o O) of the lane's query (its 8 columns + the partner lane's), kept for pass B; a tile whose dO rows
      // are all exactly zero is dead (see q_live below)
      float dl = cur.g0.x * cur.o0.x + cur.g0.y * cur.o0.y + cur.g0.z * cur.o0.z + cur.g0.w * cur.o0.w +
                 cur.g1.x * cur.o1.x + cur.g1.y * cur.o1.y + cur.g1.z * cur.o1.z + cur.g1.w * cur.o1.w;
      dl += __shfl_xor(dl, 16, 64);
      if (NC == 2) dl += __shfl_xor(dl, 32, 64);
      if (g == 0) Dl[q] = dl;
      const bool nz = cur.g0.x != 0.f || cur.g0.y != 0.f || cur.g0.z != 0.f || cur.g0.w != 0.f || cur.g1.x != 0.f || cur.g1.y != 0.f ||
                      cur.g1.z != 0.f || cur.g1.w != 0.f;
      const bool live = __ballot(nz) != 0ull;
      if (live && lane == 0) atomicOr(reinterpret_cast<unsigned*>(&red[1]), 1u << qt);
      const DOps qo = d_ops<NC>(cur.q0, cur.q1, first, sel), go = d_ops<NC>(cur.g0, cur.g1, first, sel);
      const int nt = (live && !(p.ablate & 1)) ? min(can_skip ? qt + 1 : nkt, nkt_eff) : 0;   // (ablate: diagnostics)
      f32x4 dq[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) dq[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int kt = 0; kt < nt; ++kt) {
        const int toff = kt * 16 * RP;
        const f32x4 sacc = dot_d<NC>(ka1, ka2, ka3, toff, qo);   // S^T: lane = query j, rows = keys kt*16 + 4g + r
        const f32x4 dpacc = dot_d<NC>(va1, va2, va3, toff, go);
        f32x4 ds;
        const unsigned bits = Kbits[kt];
        const bool need_mask = bits != 0u || kt * 16 + 16 > p.Lk || (CAUSAL && kt >= qt);     // wave-uniform
        if (need_mask) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g + r;
            float m = key < p.Lk ? (((bits >> (4 * g + r)) & 1u) ? -1e9f : 0.f) : -INFINITY;
            if (CAUSAL && key > q) m = fminf(m, -1e9f);
            const float v = m < 0.f ? m : sacc[r] * c2;
            ds[r] = __builtin_amdgcn_exp2f(v - mx) * ri * (dpacc[r] - dl);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[r] = __builtin_amdgcn_exp2f(sacc[r] * c2 - mx) * ri * (dpacc[r] - dl);
        }
        dot_rows<NC>(ktr + toff, plane_b, ds, dq, sel);          // dQ^T[d][q] += sum_k K[k][d] dS[q][k]
      }
      if (qok) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
          *reinterpret_cast<float4*>(p.dQ + (size_t)(b * p.Lq + q) * p.lddq + h * DH + 16 * c + 4 * g) =
              make_float4(dq[c][0] * inv_sqrt, dq[c][1] * inv_sqrt, dq[c][2] * inv_sqrt, dq[c][3] * inv_sqrt);
      }
    }
  }
  __syncthreads();
  // A query tile whose dO rows are all exactly zero contributes nothing: dS = P o (0 - 0) = 0, so its dQ is 0 and dK / dV get
  // nothing from it.  That is every padded position of the decoder (masked loss -> dlogits = 0 -> zero rows through the
  // row-wise stages and the dgrad GEMMs): about half of the tile pairs of a QuickDraw-shaped batch; exact, data-dependent.
  const unsigned q_live = (unsigned)red[1];
  // ---------------- phase 1: Q / dO planes over the K / V planes
  stage_planes2<NC>(TA, Qb, p.ldq, p.Lq, TB, dOb, p.lddo, p.Lq, plane_b, R, tid, sel);
  __syncthreads();
  // ---------------- pass B: dK, dV
  {
    const char* qa1 = TA + pl1 + lane_d;
    const char* qa2 = TA + pl2 + lane_d;
    const char* qa3 = TA + pl3 + lane_d;
    const char* da1 = TB + pl1 + lane_d;
    const char* da2 = TB + pl2 + lane_d;
    const char* da3 = TB + pl3 + lane_d;
    const char* qtr = TA + lane_t;
    const char* dtr = TB + lane_t;
    auto load_k = [&](int kt_) {
      const int kk = kt_ * 16 + j, kc = kk < p.Lk ? kk : p.Lk - 1;
      KTile t;
      t.k0 = *reinterpret_cast<const float4*>(Kb + (size_t)kc * p.ldk + d0); t.k1 = *reinterpret_cast<const float4*>(Kb + (size_t)kc * p.ldk + d0 + 4);
      t.v0 = *reinterpret_cast<const float4*>(Vb + (size_t)kc * p.ldv + d0); t.v1 = *reinterpret_cast<const float4*>(Vb + (size_t)kc * p.ldv + d0 + 4);
      // keys past Lk get -inf (never -1e9): with a fully padded sample the row max itself is -1e9
      t.kadd = kk < p.Lk ? (((Kbits[kt_ & 31] >> j) & 1u) ? -1e9f : 0.f) : -INFINITY;
      return t;
    };
    // row statistics of a query tile: lane l holds those of query 16 qt + (l & 15); the four rows of a lane's C layout are
    // fetched from lanes 4g .. 4g+3 (ds_bpermute: no LDS space, which is what keeps four workgroups on a CU)
    auto load_st = [&](int qt_) {
      const int qq = qt_ * 16 + j;
      float2 st = stats[qq < p.Lq ? qq : p.Lq - 1];
      if (qq >= p.Lq) st = make_float2(0.f, 0.f);                  // rows past Lq: 1/sum = 0 -> P == 0
      return st;
    };
    KTile kn = load_k(wv < nkt ? wv : 0);
    for (int kt = wv; kt < nkt; kt += 4) {
      KTile kc_ = kn;
      kn = load_k(kt + 4 < nkt ? kt + 4 : kt);
      const int key = kt * 16 + j;
      const bool kok = key < p.Lk;
      f32x4 dkt[NC], dvt[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) { dkt[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvt[c] = dkt[c]; }
      if (kt < nkt_eff && !(p.ablate & 2)) {
        if (!kok) { kc_.k0 = kc_.k1 = kc_.v0 = kc_.v1 = make_float4(0.f, 0.f, 0.f, 0.f); }
        const DOps ko = d_ops<NC>(kc_.k0, kc_.k1, first, sel), vo = d_ops<NC>(kc_.v0, kc_.v1, first, sel);
        const float kadd = kc_.kadd;
        const int qt0 = (CAUSAL && can_skip) ? kt : 0;
        float2 sn = load_st(qt0 < nqt ? qt0 : 0);
        for (int qt = qt0; qt < nqt; ++qt) {
          const float2 sc = sn;
          sn = load_st(qt + 1 < nqt ? qt + 1 : qt);
          if (!((q_live >> qt) & 1u)) continue;
          const int toff = qt * 16 * RP, qr = qt * 16 + 4 * g;
          const f32x4 sacc = dot_d<NC>(qa1, qa2, qa3, toff, ko); // S: lane = key j, rows = queries qr + r
          const f32x4 dpacc = dot_d<NC>(da1, da2, da3, toff, vo);
          const float4 dl4 = *reinterpret_cast<const float4*>(Dl + qr);
          const float dlr[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
          f32x4 pr, ds;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = qr + r;
            const float mxr = __shfl(sc.x, 4 * g + r, 64), rir = __shfl(sc.y, 4 * g + r, 64);
            const float m = CAUSAL ? fminf(kadd, key > q ? -1e9f : 0.f) : kadd;
            const float v = m < 0.f ? m : sacc[r] * c2;
            const float pv = __builtin_amdgcn_exp2f(v - mxr) * rir;
            pr[r] = pv;
            ds[r] = pv * (dpacc[r] - dlr[r]);
          }
          dot_rows<NC>(dtr + toff, plane_b, pr, dvt, sel);       // dV^T[d][k] += sum_q dO[q][d] P[q][k]
          dot_rows<NC>(qtr + toff, plane_b, ds, dkt, sel);       // dK^T[d][k] += sum_q Q[q][d] dS[q][k]
        }
      }
      if (kok) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          *reinterpret_cast<float4*>(p.dK + (size_t)(b * p.Lk + key) * p.lddk + h * DH + 16 * c + 4 * g) =
              make_float4(dkt[c][0] * inv_sqrt, dkt[c][1] * inv_sqrt, dkt[c][2] * inv_sqrt, dkt[c][3] * inv_sqrt);
          *reinterpret_cast<float4*>(p.dV + (size_t)(b * p.Lk + key) * p.lddv + h * DH + 16 * c + 4 * g) =
              make_float4(dvt[c][0], dvt[c][1], dvt[c][2], dvt[c][3]);
        }
      }
    }
  }
}

}  // namespace

template <int NC>
static int bwd2_launch(const AttnParams& p, hipStream_t st) {
  const int nkt = (p.Lk + 15) >> 4, nqt = (p.Lq + 15) >> 4;
  const int R = (nkt > nqt ? nkt : nqt) * 16;
  const size_t smem = (size_t)6 * R * 32 * NC + (size_t)R * sizeof(float) + 32 * sizeof(unsigned) + 16;
  SKF_CHECK_ARG(smem <= 160 * 1024, "the operand planes of one head do not fit in LDS");
  SKF_CHECK_ARG(nkt <= 32 && nqt <= 32, "more than 32 key / query tiles");
  SKF_CHECK_ARG((p.ldq & 3) == 0 && (p.ldk & 3) == 0 && (p.ldv & 3) == 0 && (p.ldo & 3) == 0 && (p.lddo & 3) == 0 && (p.lddq & 3) == 0 &&
                (p.lddk & 3) == 0 && (p.lddv & 3) == 0, "row strides must be multiples of 4");
  const void* kfn = p.causal ? (const void*)attn_bwd2_kernel<NC, true> : (const void*)attn_bwd2_kernel<NC, false>;
  SKF_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));   // per launch: the attribute is per device
  // (query tiles whose dO rows are all zero are skipped as well; with the live lengths of the train step those are the tiles
  //  behind q_live - without the lengths the figure below only knows the masks)
  const double visited = skf_prof_attention_fraction(p.key_mask, p.key_mask_ld, p.causal, p.B, p.Lq, p.Lk, p.q_live, 16, 16);
  SkfProfScope ps(st, NC == 1 ? "attn_bwd2<dh16,bf16x6>" : "attn_bwd2<dh32,bf16x6>", 8.0 * p.B * p.H * (double)p.Lq * p.Lk * 16 * NC,
                  4.0 * p.B * p.H * 16 * NC * (4.0 * p.Lq + 4.0 * p.Lk));
  ps.done(8.0 * p.B * p.H * (double)p.Lq * p.Lk * 16 * NC * visited, 4.0 * p.B * p.H * 16 * NC * (4.0 * p.Lq + 4.0 * p.Lk));
  if (p.causal) hipLaunchKernelGGL((attn_bwd2_kernel<NC, true>), dim3(p.B * p.H), dim3(256), smem, st, p);
  else hipLaunchKernelGGL((attn_bwd2_kernel<NC, false>), dim3(p.B * p.H), dim3(256), smem, st, p);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

int skf_attention_bwd2_launch(const AttnParams& p, int dh, hipStream_t st) {
  return dh == 16 ? bwd2_launch<1>(p, st) : bwd2_launch<2>(p, st);
}
