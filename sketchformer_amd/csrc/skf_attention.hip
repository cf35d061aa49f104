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
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, g = lane >> 4;

  // ---- stage K, V (zero-filled tail rows)
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
  __syncthreads();

  const unsigned char* km = p.key_mask ? p.key_mask + (size_t)b * p.key_mask_ld : nullptr;
  // Causal tile skipping is exact only when key 0 is visible to every query
  // (then every row max is a real score and masked probabilities are exactly 0).
  const bool can_skip = p.causal && !(km && km[0]);
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
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + g * 4 + r;
          float v = pow4 ? acc[r] * inv_sqrt : acc[r] / sqrtf((float)DH);
          const bool masked = (km && key < p.Lk && km[key]) || (p.causal && key > qrow);
          if (masked) v += -1e9f;
          if (key >= p.Lk) v = -INFINITY;
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

template <int DH>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnParams p) {
  constexpr int NC = DH / 16;
  constexpr int LD = DH + 4;
  constexpr int TLD = 20;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
  const int nkt = (p.Lk + 15) >> 4, nqt = (p.Lq + 15) >> 4;
  const int QR = nqt * 16;
  float* Qs = smem;                   // [QR][LD]
  float* dOs = Qs + QR * LD;          // [QR][LD]
  float* dQs = dOs + QR * LD;         // [QR][LD]
  float* Mx = dQs + QR * LD;          // [QR]
  float* Ri = Mx + QR;                // [QR]
  float* Dl = Ri + QR;                // [QR]  delta = sum_d dO*O
  float* Tr = Dl + QR;                // [4 waves][16][TLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    *reinterpret_cast<float4*>(&dQs[row * LD + c4]) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
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
  const float inv_sqrt = 1.0f / sqrtf((float)DH);
  const bool pow4 = (DH == 16 || DH == 64);
  float* tr = Tr + wave * 16 * TLD;

  for (int kt = wave; kt < nkt; kt += 4) {
    const int k0 = kt * 16, krow = k0 + i;
    const bool kok = krow < p.Lk;
    // B-operand fragments (lane = key i, contraction d = 16c+4g+s) and
    // A-operand (transposed) fragments (lane = d 16c+i, contraction key = k0+4g+s)
    float4 kb[NC], vb[NC];
    float kT[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      kb[c] = make_float4(0.f, 0.f, 0.f, 0.f); vb[c] = kb[c];
      if (kok) {
        kb[c] = *reinterpret_cast<const float4*>(p.K + (size_t)(b * p.Lk + krow) * p.ldk + h * DH + c * 16 + g * 4);
        vb[c] = *reinterpret_cast<const float4*>(p.V + (size_t)(b * p.Lk + krow) * p.ldv + h * DH + c * 16 + g * 4);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kr = k0 + g * 4 + s;
        kT[c][s] = kr < p.Lk ? p.K[(size_t)(b * p.Lk + kr) * p.ldk + h * DH + c * 16 + i] : 0.f;
      }
    }
    const bool kmasked = km && kok && km[krow];
    f32x4 dKt[NC], dVt[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { dKt[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; dVt[c] = dKt[c]; }

    const int qt_begin = p.causal ? kt : 0;   // tiles with every q < every k contribute exactly 0
    for (int qt = qt_begin; qt < nqt; ++qt) {
      const int q0 = qt * 16;
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = sacc;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float4 qa = *reinterpret_cast<const float4*>(&Qs[(q0 + i) * LD + c * 16 + g * 4]);
        const float4 da = *reinterpret_cast<const float4*>(&dOs[(q0 + i) * LD + c * 16 + g * 4]);
        sacc = mfma16(qa.x, kb[c].x, sacc);
        sacc = mfma16(qa.y, kb[c].y, sacc);
        sacc = mfma16(qa.z, kb[c].z, sacc);
        sacc = mfma16(qa.w, kb[c].w, sacc);
        dpacc = mfma16(da.x, vb[c].x, dpacc);
        dpacc = mfma16(da.y, vb[c].y, dpacc);
        dpacc = mfma16(da.z, vb[c].z, dpacc);
        dpacc = mfma16(da.w, vb[c].w, dpacc);
      }
      // lane holds rows q = q0+4g+r, column key = k0+i
      float pr[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + g * 4 + r;
        float v = pow4 ? sacc[r] * inv_sqrt : sacc[r] / sqrtf((float)DH);
        if (kmasked || (p.causal && krow > q)) v += -1e9f;
        float pv = __expf(v - Mx[q]) * Ri[q];
        if (!kok || q >= p.Lq) pv = 0.f;
        pr[r] = pv;
        float d = pv * (dpacc[r] - Dl[q]);
        ds[r] = pow4 ? d * inv_sqrt : d / sqrtf((float)DH);
      }
      // dV^T[d][k] += sum_q dO[q][d] P[q][k];  dK^T[d][k] += sum_q Q[q][d] dS[q][k]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qr = q0 + g * 4 + r;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          dVt[c] = mfma16(dOs[qr * LD + c * 16 + i], pr[r], dVt[c]);
          dKt[c] = mfma16(Qs[qr * LD + c * 16 + i], ds[r], dKt[c]);
        }
      }
      // transpose dS through the wave-private scratch: write [q][k], read [q=i][k=4g..4g+3]
#pragma unroll
      for (int r = 0; r < 4; ++r) tr[(g * 4 + r) * TLD + i] = ds[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const float4 dst = *reinterpret_cast<const float4*>(&tr[i * TLD + g * 4]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      // dQ^T[d][q] += sum_k K[k][d] dS[q][k]   (lane: d = 16c+4g+r, q = q0+i)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
        dq = mfma16(kT[c][0], dst.x, dq);
        dq = mfma16(kT[c][1], dst.y, dq);
        dq = mfma16(kT[c][2], dst.z, dq);
        dq = mfma16(kT[c][3], dst.w, dq);
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&dQs[(q0 + i) * LD + c * 16 + g * 4 + r], dq[r]);
      }
    }
    if (kok) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        *reinterpret_cast<float4*>(p.dK + (size_t)(b * p.Lk + krow) * p.lddk + h * DH + c * 16 + g * 4) =
            make_float4(dKt[c][0], dKt[c][1], dKt[c][2], dKt[c][3]);
        *reinterpret_cast<float4*>(p.dV + (size_t)(b * p.Lk + krow) * p.lddv + h * DH + c * 16 + g * 4) =
            make_float4(dVt[c][0], dVt[c][1], dVt[c][2], dVt[c][3]);
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < p.Lq * (DH / 4); e += 256) {
    const int row = e / (DH / 4), c4 = (e % (DH / 4)) * 4;
    *reinterpret_cast<float4*>(p.dQ + (size_t)(b * p.Lq + row) * p.lddq + h * DH + c4) =
        *reinterpret_cast<const float4*>(&dQs[row * LD + c4]);
  }
}

size_t fwd_smem(int DH, int Lk) { return (size_t)2 * ((Lk + 15) / 16 * 16) * (DH + 4) * sizeof(float); }
size_t bwd_smem(int DH, int Lq) {
  const size_t QR = (size_t)(Lq + 15) / 16 * 16;
  return (3 * QR * (DH + 4) + 3 * QR + 4 * 16 * 20) * sizeof(float);
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
    auto kfn = attn_bwd_kernel<DHV>;                            \
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
