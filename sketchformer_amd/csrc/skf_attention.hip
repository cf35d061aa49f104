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
// transposed through a wave-private LDS scratch for dQ^T += K^T.dS^T, which is
// accumulated across waves with LDS float atomics and written once.
#include "skf_common.h"

namespace {

struct AttnParams {
  const float* Q; const float* K; const float* V; float* O;
  int ldq, ldk, ldv, ldo;
  const unsigned char* key_mask;  // (B, key_mask_ld) 1 = masked key, or null
  int key_mask_ld;
  int causal;
  int B, H, Lq, Lk;
  float* stats;                   // (B, H, Lq, 2): row max, 1/sum
  // backward only
  const float* dO; int lddo;
  float* dQ; float* dK; float* dV;
  int lddq, lddk, lddv;
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int DH, int MAXT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  constexpr int NC = DH / 16;
  constexpr int LD = DH + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  const int nkt = (p.Lk + 15) >> 4, nqt = (p.Lq + 15) >> 4;
  float* Ks = smem;                       // [nkt*16][LD]
  float* Vs = smem + nkt * 16 * LD;       // [nkt*16][LD]
  float* Ms = Vs + nkt * 16 * LD;         // [nkt*16] additive key mask: 0, -1e9 (padded key) or -inf (key >= Lk)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: loop bounds stay scalar
  const int i = lane & 15, g = lane >> 4;

  // ---- stage K, V (zero-filled tail rows) and the key mask
  for (int e = tid; e < nkt * 16 * (DH / 4); e += 256) {
    const int row = e / (DH / 4), c4 = (e % (DH / 4)) * 4;
    float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
    if (row < p.Lk) {
      kv = *reinterpret_cast<const float4*>(p.K + (size_t)(b * p.Lk + row) * p.ldk + h * DH + c4);
      vv = *reinterpret_cast<const float4*>(p.V + (size_t)(b * p.Lk + row) * p.ldv + h * DH + c4);
    }
    *reinterpret_cast<float4*>(&Ks[row * LD + c4]) = kv;
    *reinterpret_cast<float4*>(&Vs[row * LD + c4]) = vv;
  }
  for (int key = tid; key < nkt * 16; key += 256) {
    float mv = -INFINITY;
    if (key < p.Lk) mv = (p.key_mask && p.key_mask[(size_t)b * p.key_mask_ld + key]) ? -1e9f : 0.f;
    Ms[key] = mv;
  }
  __syncthreads();

  // Causal tile skipping is exact only when key 0 is visible to every query
  // (then every row max is a real score and masked probabilities are exactly 0).
  const bool can_skip = p.causal && Ms[0] == 0.f;
  const float inv_sqrt = 1.0f / sqrtf((float)DH);
  const bool pow4 = (DH == 16 || DH == 64);

  for (int qt = wave; qt < nqt; qt += 4) {
    const int q0 = qt * 16, qrow = q0 + i;
    const bool qok = qrow < p.Lq;
    float4 qf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      qf[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qok) qf[c] = *reinterpret_cast<const float4*>(p.Q + (size_t)(b * p.Lq + qrow) * p.ldq + h * DH + c * 16 + g * 4);
    }
    const int nt = can_skip ? min(nkt, qt + 1) : nkt;
    float s[MAXT][4];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt) {
      if (kt < nt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float4 kf = *reinterpret_cast<const float4*>(&Ks[(kt * 16 + i) * LD + c * 16 + g * 4]);
          acc = mfma16(kf.x, qf[c].x, acc);
          acc = mfma16(kf.y, qf[c].y, acc);
          acc = mfma16(kf.z, qf[c].z, acc);
          acc = mfma16(kf.w, qf[c].w, acc);
        }
        // logits = (q.k)/sqrt(dh) + max(pad, look_ahead) * -1e9  (one -1e9, never two)
        const float4 m4 = *reinterpret_cast<const float4*>(&Ms[kt * 16 + g * 4]);
        const float mr[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + g * 4 + r;
          const float cm = (p.causal && key > qrow) ? -1e9f : 0.f;
          const float v = (pow4 ? acc[r] * inv_sqrt : acc[r] / sqrtf((float)DH)) + fminf(mr[r], cm);
          s[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt)
      if (kt < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[kt][r] = __expf(s[kt][r] - mx); sum += s[kt][r]; }
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float rinv = 1.0f / sum;
    f32x4 o[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < MAXT; ++kt)
      if (kt < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = s[kt][r] * rinv;
#pragma unroll
          for (int c = 0; c < NC; ++c)
            o[c] = mfma16(Vs[(kt * 16 + g * 4 + r) * LD + c * 16 + i], pv, o[c]);
        }
      }
    if (qok) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
        *reinterpret_cast<float4*>(p.O + (size_t)(b * p.Lq + qrow) * p.ldo + h * DH + c * 16 + g * 4) =
            make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
      if (g == 0 && p.stats) {
        float2* st = reinterpret_cast<float2*>(p.stats) + ((size_t)bh * p.Lq + qrow);
        *st = make_float2(mx, rinv);
      }
    }
  }
}

// KTW = key tiles owned by one wave at a time (register budget: 20*NC*KTW accumulator/fragment registers)
template <int DH, int KTW>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnParams p) {
  constexpr int NC = DH / 16;
  constexpr int LD = DH + 4;
  constexpr int TLD = 20;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  const int nkt = (p.Lk + 15) >> 4, nqt = (p.Lq + 15) >> 4;
  const int QR = nqt * 16, LDT = QR + 4;
  float* Qs = smem;                   // [QR][LD]
  float* dOs = Qs + QR * LD;          // [QR][LD]
  float* dQt = dOs + QR * LD;         // [DH][LDT]  dQ^T accumulator (d-major: the 64 atomic lanes hit 32 banks 2-way)
  float* Mx = dQt + DH * LDT;         // [QR]
  float* Ri = Mx + QR;                // [QR]
  float* Dl = Ri + QR;                // [QR]  delta = sum_d dO*O
  float* Tr = Dl + QR;                // [4 waves][16][TLD]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;

  for (int e = tid; e < QR * (DH / 4); e += 256) {
    const int row = e / (DH / 4), c4 = (e % (DH / 4)) * 4;
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), dv = qv;
    if (row < p.Lq) {
      qv = *reinterpret_cast<const float4*>(p.Q + (size_t)(b * p.Lq + row) * p.ldq + h * DH + c4);
      dv = *reinterpret_cast<const float4*>(p.dO + (size_t)(b * p.Lq + row) * p.lddo + h * DH + c4);
    }
    *reinterpret_cast<float4*>(&Qs[row * LD + c4]) = qv;
    *reinterpret_cast<float4*>(&dOs[row * LD + c4]) = dv;
  }
  for (int e = tid; e < DH * LDT; e += 256) dQt[e] = 0.f;
  for (int row = tid; row < QR; row += 256) {
    float mx = 0.f, ri = 0.f, dl = 0.f;
    if (row < p.Lq) {
      const float2 st = reinterpret_cast<const float2*>(p.stats)[(size_t)bh * p.Lq + row];
      mx = st.x; ri = st.y;
      const float* orow = p.O + (size_t)(b * p.Lq + row) * p.ldo + h * DH;
      const float* drow = p.dO + (size_t)(b * p.Lq + row) * p.lddo + h * DH;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        const float4 a = *reinterpret_cast<const float4*>(orow + c);
        const float4 d = *reinterpret_cast<const float4*>(drow + c);
        dl += a.x * d.x + a.y * d.y + a.z * d.z + a.w * d.w;
      }
    }
    Mx[row] = mx; Ri[row] = ri; Dl[row] = dl;
  }
  __syncthreads();

  const unsigned char* km = p.key_mask ? p.key_mask + (size_t)b * p.key_mask_ld : nullptr;
  // skipping fully look-ahead-masked tiles is exact only if key 0 is visible (see forward)
  const bool can_skip = p.causal && !(km && km[0]);
  const float inv_sqrt = 1.0f / sqrtf((float)DH);
  const bool pow4 = (DH == 16 || DH == 64);
  float* tr = Tr + wave * 16 * TLD;

  for (int kg = 0; kg < nkt; kg += 4 * KTW) {
    const int kt0 = kg + wave;                 // smallest key tile of this wave in this group
    if (kt0 >= nkt) continue;
    // B-operand fragments (lane = key i, contraction d = 16c+4g+s) and
    // A-operand (transposed) fragments (lane = d 16c+i, contraction key = k0+4g+s)
    float4 kb[KTW][NC], vb[KTW][NC];
    float kT[KTW][NC][4];
    float kadd[KTW], kvalid[KTW];
    f32x4 dKt[KTW][NC], dVt[KTW][NC];
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
      const int k0 = (kt0 + 4 * j) * 16, krow = k0 + i;
      const bool kok = krow < p.Lk;
      kvalid[j] = kok ? 1.f : 0.f;
      kadd[j] = (km && kok && km[krow]) ? -1e9f : 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        kb[j][c] = make_float4(0.f, 0.f, 0.f, 0.f); vb[j][c] = kb[j][c];
        if (kok) {
          kb[j][c] = *reinterpret_cast<const float4*>(p.K + (size_t)(b * p.Lk + krow) * p.ldk + h * DH + c * 16 + g * 4);
          vb[j][c] = *reinterpret_cast<const float4*>(p.V + (size_t)(b * p.Lk + krow) * p.ldv + h * DH + c * 16 + g * 4);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int kr = k0 + g * 4 + s;
          kT[j][c][s] = kr < p.Lk ? p.K[(size_t)(b * p.Lk + kr) * p.ldk + h * DH + c * 16 + i] : 0.f;
        }
        dKt[j][c] = (f32x4){0.f, 0.f, 0.f, 0.f}; dVt[j][c] = dKt[j][c];
      }
    }

    const int qt_begin = can_skip ? kt0 : 0;   // query tiles entirely above the wave's first key tile contribute 0
    const int nq_act = nqt - qt_begin;
    for (int it = 0; it < nq_act; ++it) {
      // each wave starts at a different query tile so the waves' LDS atomics on dQ do not collide
      const int qt = qt_begin + (it + wave * 3) % nq_act;
      const int q0 = qt * 16;
      float4 qa[NC], da[NC];
      float qT[NC][4], dT[NC][4];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        qa[c] = *reinterpret_cast<const float4*>(&Qs[(q0 + i) * LD + c * 16 + g * 4]);
        da[c] = *reinterpret_cast<const float4*>(&dOs[(q0 + i) * LD + c * 16 + g * 4]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          qT[c][r] = Qs[(q0 + g * 4 + r) * LD + c * 16 + i];
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
        const int kt = kt0 + 4 * j;
        if (kt >= nkt || (can_skip && kt > qt)) continue;     // wave-uniform
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
          const float cm = (p.causal && krow > q) ? -1e9f : 0.f;
          const float v = (pow4 ? sacc[r] * inv_sqrt : sacc[r] / sqrtf((float)DH)) + fminf(kadd[j], cm);
          const float pv = __expf(v - mxr[r]) * (rir[r] * kvalid[j]);
          pr[r] = pv;
          const float d = pv * (dpacc[r] - dlr[r]);
          ds[r] = pow4 ? d * inv_sqrt : d / sqrtf((float)DH);
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
#pragma unroll
        for (int r = 0; r < 4; ++r) tr[(g * 4 + r) * TLD + i] = ds[r];
        __builtin_amdgcn_wave_barrier();
        const float4 dst = *reinterpret_cast<const float4*>(&tr[i * TLD + g * 4]);
        __builtin_amdgcn_wave_barrier();
        // dQ^T[d][q] += sum_k K[k][d] dS[q][k]   (lane: d = 16c+4g+r, q = q0+i)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          dq[c] = mfma16(kT[j][c][0], dst.x, dq[c]);
          dq1[c] = mfma16(kT[j][c][1], dst.y, dq1[c]);
          dq[c] = mfma16(kT[j][c][2], dst.z, dq[c]);
          dq1[c] = mfma16(kT[j][c][3], dst.w, dq1[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&dQt[(c * 16 + g * 4 + r) * LDT + q0 + i], dq[c][r] + dq1[c][r]);
    }
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
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
  }
  __syncthreads();
  for (int e = tid; e < p.Lq * (DH / 4); e += 256) {
    const int row = e / (DH / 4), c4 = (e % (DH / 4)) * 4;
    *reinterpret_cast<float4*>(p.dQ + (size_t)(b * p.Lq + row) * p.lddq + h * DH + c4) =
        make_float4(dQt[(c4 + 0) * LDT + row], dQt[(c4 + 1) * LDT + row], dQt[(c4 + 2) * LDT + row], dQt[(c4 + 3) * LDT + row]);
  }
}

size_t fwd_smem(int DH, int Lk) { return (size_t)((Lk + 15) / 16 * 16) * (2 * (DH + 4) + 1) * sizeof(float); }
size_t bwd_smem(int DH, int Lq) {
  const size_t QR = (size_t)(Lq + 15) / 16 * 16;
  return (2 * QR * (DH + 4) + (size_t)DH * (QR + 4) + 3 * QR + 4 * 16 * 20) * sizeof(float);
}

template <typename K>
int set_smem(K kfn, size_t bytes) {
  SKF_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return SKF_OK;
}

int check_common(const AttnParams& p, int dh) {
  SKF_CHECK_ARG(dh == 16 || dh == 32 || dh == 64, "head dim must be 16, 32 or 64");
  SKF_CHECK_ARG(p.B > 0 && p.H > 0 && p.Lq > 0 && p.Lk > 0, "empty problem");
  SKF_CHECK_ARG((p.ldq & 3) == 0 && (p.ldk & 3) == 0 && (p.ldv & 3) == 0 && (p.ldo & 3) == 0, "row strides must be multiples of 4");
  SKF_CHECK_ARG(!p.causal || p.Lq == p.Lk, "causal attention needs Lq == Lk");
  return SKF_OK;
}

}  // namespace

extern "C" int skf_attention_fwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                                 const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk,
                                 int dh, float* O, int ldo, float* stats, skf_stream_t stream) {
  AttnParams p{};
  p.Q = Q; p.K = K; p.V = V; p.O = O; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.key_mask = key_mask; p.key_mask_ld = key_mask_ld; p.causal = causal; p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.stats = stats;
  int rc = check_common(p, dh);
  if (rc) return rc;
  SKF_CHECK_ARG(Q && K && V && O, "null operand");
  SKF_CHECK_ARG(Lk <= 512, "Lk > 512 not supported");
  const size_t smem = fwd_smem(dh, Lk);
  SKF_CHECK_ARG(smem <= 160 * 1024, "K/V of one head do not fit in LDS");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(B * H), block(256);
#define SKF_ATTN_FWD(DHV, MT)                                   \
  {                                                             \
    auto kfn = attn_fwd_kernel<DHV, MT>;                        \
    if ((rc = set_smem(kfn, smem))) return rc;                  \
    hipLaunchKernelGGL(kfn, grid, block, smem, st, p);          \
  }
  static const char* const tags[3] = {"attn_fwd<dh16>", "attn_fwd<dh32>", "attn_fwd<dh64>"};
  SkfProfScope ps(st, tags[dh == 16 ? 0 : dh == 32 ? 1 : 2], 4.0 * B * H * (double)Lq * Lk * dh,
                  4.0 * B * H * dh * (2.0 * Lq + 2.0 * Lk));
  const bool small = Lk <= 208;
  if (dh == 16) { if (small) SKF_ATTN_FWD(16, 13) else SKF_ATTN_FWD(16, 32) }
  else if (dh == 32) { if (small) SKF_ATTN_FWD(32, 13) else SKF_ATTN_FWD(32, 32) }
  else { if (small) SKF_ATTN_FWD(64, 13) else SKF_ATTN_FWD(64, 32) }
#undef SKF_ATTN_FWD
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_attention_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                                 const float* O, int ldo, const float* dO, int lddo, const float* stats,
                                 const unsigned char* key_mask, int key_mask_ld, int causal, int B, int H, int Lq, int Lk,
                                 int dh, float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, skf_stream_t stream) {
  AttnParams p{};
  p.Q = Q; p.K = K; p.V = V; p.O = const_cast<float*>(O); p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.key_mask = key_mask; p.key_mask_ld = key_mask_ld; p.causal = causal; p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk;
  p.stats = const_cast<float*>(stats);
  p.dO = dO; p.lddo = lddo; p.dQ = dQ; p.dK = dK; p.dV = dV; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  int rc = check_common(p, dh);
  if (rc) return rc;
  SKF_CHECK_ARG(Q && K && V && O && dO && stats && dQ && dK && dV, "null operand");
  SKF_CHECK_ARG((lddo & 3) == 0 && (lddq & 3) == 0 && (lddk & 3) == 0 && (lddv & 3) == 0, "row strides must be multiples of 4");
  const size_t smem = bwd_smem(dh, Lq);
  SKF_CHECK_ARG(smem <= 160 * 1024, "Q/dO/dQ of one head do not fit in LDS");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(B * H), block(256);
#define SKF_ATTN_BWD(DHV)                                       \
  {                                                             \
    auto kfn = attn_bwd_kernel<DHV, 64 / DHV>;                          \
    if ((rc = set_smem(kfn, smem))) return rc;                  \
    hipLaunchKernelGGL(kfn, grid, block, smem, st, p);          \
  }
  static const char* const tags[3] = {"attn_bwd<dh16>", "attn_bwd<dh32>", "attn_bwd<dh64>"};
  SkfProfScope ps(st, tags[dh == 16 ? 0 : dh == 32 ? 1 : 2], 8.0 * B * H * (double)Lq * Lk * dh,
                  4.0 * B * H * dh * (4.0 * Lq + 4.0 * Lk));
  if (dh == 16) SKF_ATTN_BWD(16) else if (dh == 32) SKF_ATTN_BWD(32) else SKF_ATTN_BWD(64)
#undef SKF_ATTN_BWD
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}
