// Weight-stationary fp32 GEMM on the bf16 matrix cores: every fp32 operand is split EXACTLY into P bf16 pieces
// (x = x0 + x1 + x2, 8 significand bits each, by truncation and exact fp32 subtraction) and the product is
// evaluated as the sum of the piece products a_i.b_j with i + j < P in fp32 accumulators:
//   P = 3 -> 6 MFMA products, dropped terms <= 2^-23 |a||b| per product (the size of one fp32 rounding),
//   P = 2 -> 3 MFMA products, dropped terms <= 2^-15 |a||b|.
// Why: gfx950 issues v_mfma_f32_16x16x4_f32 (2048 FLOP) in 32 cycles but v_mfma_f32_16x16x32_bf16 (16384 FLOP)
// in 16, so six bf16 products cost 96 cycles for the work of 256 cycles of fp32 MFMA - the Dense shapes of this
// model (K = 128..512, M = 25k rows) turn from MFMA-bound into HBM-stream-bound.
//
// Structure = skf_gemm_ws.hip (persistent workgroups, weight slice in registers, A tiles global -> registers ->
// LDS two tiles ahead, C fragments stored straight from registers through buffer descriptors), except:
//   * A is split when it is handed to LDS (P planes of bf16 per tile, 8-byte ds_writes), so each element is
//     split once per workgroup, not once per wave;
//   * the weight slice is split once in the prologue: P x (K/32) x NB operands of 8 bf16 per lane;
//   * one MFMA step covers k = 32s + 8g + e (lane group g, e < 8): A fragments are one ds_read_b128 per
//     (step, piece), conflict-free with a 16-byte row pad.
// Small products go to their own accumulator (added to the a0.b0 sum at the end).
#include "skf_common.h"
#include "skf_gemm_params.h"
#include <set>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int TR = 16;   // rows per tile
#ifndef SKF_WSX_EARLY3
#define SKF_WSX_EARLY3 0   // 1: K = 128 keeps two full fragment sets also in the three-piece mode (register ring of 2)
#endif

// gfx950 hides about one VALU / LDS / VMEM instruction per v_mfma_f32_16x16x32_bf16 of the SAME wave and almost none
// of another wave's (tools/micro/mfma_bf16_valu_overlap.hip), so the tile body is left to the scheduler as one region
// (no fences between its phases) unless SKF_WSX_FENCES is defined.
#ifdef SKF_WSX_FENCES
#define SKF_WSX_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define SKF_WSX_SCHED_BARRIER() do { } while (0)
#endif

template <int N> struct VecOfX;
template <> struct VecOfX<1> { typedef float type; typedef unsigned utype; };
template <> struct VecOfX<2> { typedef float __attribute__((ext_vector_type(2))) type; typedef u32x2 utype; };
template <> struct VecOfX<4> { typedef f32x4 type; typedef u32x4 utype; };

// Descriptor over the rows [row0, M) of a row-major matrix.  The launcher guarantees M * ld * 4 < 2^31 (checked on the
// host), so the byte counts are 32-bit SALU arithmetic: 8 scalar instructions per descriptor instead of 23 with the
// 64-bit clamps (two to four descriptors per tile).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wsx_rows_rsrc(const float* base, int ld, int M, int row0, int cut = 0) {
  const int rows_left = M - row0 > 0 ? M - row0 : 0;
  const unsigned rem = rows_left > 0 ? (unsigned)rows_left * (unsigned)ld * 4u - (unsigned)cut : 0u;
  const unsigned off = rows_left > 0 ? (unsigned)row0 * (unsigned)ld * 4u : 0u;   // empty descriptor: any valid base
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base) + off), 0, rem, 0x00020000);
}
template <int NB>
__device__ __forceinline__ typename VecOfX<NB>::type wsx_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  typedef typename VecOfX<NB>::type vecn;
  if constexpr (NB == 1) return __builtin_bit_cast(vecn, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
  else if constexpr (NB == 2) return __builtin_bit_cast(vecn, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
  else return __builtin_bit_cast(vecn, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
template <int NB>
__device__ __forceinline__ void wsx_buf_store(typename VecOfX<NB>::type v, __amdgpu_buffer_rsrc_t r, unsigned voff) {
  typedef typename VecOfX<NB>::utype uvec;
  if constexpr (NB == 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uvec, v), r, voff, 0, 0);
  else if constexpr (NB == 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec, v), r, voff, 0, 0);
  else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec, v), r, voff, 0, 0);
}

template <int P> __device__ __forceinline__ void split2(float x, float y, unsigned (&out)[P], const SkfSplitSel& sel) { skf_split2<P>(x, y, out, sel); }

template <int K, int KS = 1>
__device__ __forceinline__ void wsx_load_tile(const float* __restrict__ A, int lda, int M, int tile,
                                              const unsigned (&a_voff)[TR * K / (1024 * KS)], f32x4 (&ra)[TR * K / (1024 * KS)], int cut = 0) {
#ifdef SKF_WSX_ABLATE_LOAD   // diagnostics: every A tile load hits the same (cached) rows
  tile &= 7;
#endif
  const __amdgpu_buffer_rsrc_t r = wsx_rows_rsrc(A, lda, M, tile * TR, cut);
#pragma unroll
  for (int v = 0; v < TR * K / (1024 * KS); ++v)
    ra[v] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, a_voff[v], 0, 0));
}

// one A tile (registers, fp32) -> P bf16 planes in LDS; PITCH = bytes per row
template <int K, int P, int PITCH, int KS = 1>
__device__ __forceinline__ void wsx_store_tile(char* __restrict__ dst, const f32x4 (&ra)[TR * K / (1024 * KS)], const SkfSplitSel& sel) {
#pragma unroll
  for (int v = 0; v < TR * K / (1024 * KS); ++v) {
    const int e = threadIdx.x + v * 256 * KS, row = e / (K / 4), c4 = (e % (K / 4)) * 4;
    unsigned lo[P], hi[P];
    split2<P>(ra[v][0], ra[v][1], lo, sel);
    split2<P>(ra[v][2], ra[v][3], hi, sel);
#pragma unroll
    for (int q = 0; q < P; ++q)
      *reinterpret_cast<u32x2*>(dst + (q * TR + row) * PITCH + c4 * 2) = (u32x2){lo[q], hi[q]};
  }
}

// sum over the 16 lanes of a DPP row (the lanes that share g): every lane of the row ends up with the total
__device__ __forceinline__ float wsx_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
  return v;
}

// LNF (K = 128, NB = 2, one column group: the workgroup's four waves hold the 128 columns of a tile's 16 rows): the epilogue is
// z = x + dropout(A.B + bias), out = LayerNorm(z) - builders/layers/transformer.py:221-224 / 262-272 `layernorm(x + dropout(sublayer))` -
// in the launch that produces the sublayer output.  Per tile: the residual rows are requested in the middle of the MFMA stream;
// behind it every wave forms z for its 32 columns, their mean and centred square sum (two passes in registers, DPP row sums) and
// leaves the pair in LDS; the tile's closing barrier publishes them; the four pairs of a row are combined (Chan's update: as
// accurate as two passes over the whole row) when the tile is stored, under the next tile's MFMAs.
// KS = 2 (K >= 256, no activation): 512 threads, the contraction is split between two waves of each 16-column block - waves 0-3
// take k < K/2 and own the output, waves 4-7 take the rest and hand their partial sums over through LDS behind the tile's closing
// barrier (added when the tile is stored, under the next tile's MFMAs).  Half the weight registers per wave (96 instead of 192 at
// K = 512) = two waves per SIMD instead of one: K = 512 ran at one wave per SIMD with 1536 MFMA cycles in a ~4100-cycle tile.
// EXTRA = epilogue operands of the launch: 0 none, 2 ReLU sign bits only, 3 the general form (relu_src, old C, any combination).
// (Kind 1, "accumulate only", existed in round 3 and was removed in round 4: see the note in launch_wsx.)  The kinds exist so that a launch issues only the loads it uses:
// in the general form every tile requests the relu_src rows AND the old C rows through (possibly empty) descriptors - eight VMEM
// instructions per tile and wave for nothing in the bits-only ffn input gradient.
template <int K, int NB, int P, bool B_KC, int EXTRA, bool KMASK = false, bool LNF = false, int KS = 1>
__global__ __launch_bounds__(256 * KS, (KS == 2 || (K <= 256 && NB <= 2) ? 2 : 1)) void gemm_wsx_kernel(GemmParams p, int groups, int workers) {
  constexpr int CW = 16 * NB;            // columns per wave
  constexpr int NKS = K / (32 * KS);     // MFMA k-steps per tile (of this wave)
  constexpr int NF = NKS * P;            // A fragments (ds_read_b128) per tile
  static_assert(EXTRA == 0 || EXTRA == 2 || EXTRA == 3, "epilogue kinds: none, sign bits only, general");
  static_assert(KS == 1 || (KS == 2 && K >= 256 && !LNF), "contraction split: two halves, K >= 256");
  static_assert(!LNF || (K == 128 && NB == 2 && !EXTRA && !KMASK), "LayerNorm epilogue: K = 128, two columns per lane, plain launch");
  constexpr bool EARLY = !LNF && K == 128 && (P == 2 || NB == 4 || SKF_WSX_EARLY3);   // every fragment of a tile in registers: barrier inside the MFMA stream
  constexpr int PF = EARLY ? NF : (NF < 6 ? NF : 6);
  constexpr int NCH = NB == 1 ? 2 : 1;   // accumulator chains per column block
  constexpr int PITCH = 2 * K + 32;      // bytes per LDS row.  ds_read_b128 is served in four groups of 16 lanes that MIX the lane
                                         // groups g (lanes {0-3,12-15,20-27}, ...): with quad(i, g) = 2i + g (pitch = 32 mod 256) every
                                         // group hits 16 different bank quads; +16 (quad = i + g) had a 2-way conflict in each group
  constexpr int TILE_B = P * TR * PITCH; // bytes per LDS tile buffer
  constexpr int NV = TR * K / (1024 * KS);   // float4 per thread per A tile
  constexpr int R = (K == 128 && !EARLY) ? 4 : 2;   // A tiles in flight in registers
  constexpr unsigned OOB = 0x7ffffff0u;
  typedef typename VecOfX<NB>::type vecn;
  extern __shared__ __attribute__((aligned(16))) char smem_x[];
  char* As = smem_x;                     // [2][P][TR][PITCH]

  const SkfSplitSel sel = skf_split_sel();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave_all & 3, kh = wave_all >> 2;        // column block of the wave, its half of the contraction (0 when KS = 1)
  const int s0 = kh * NKS;                                  // first k-step of this wave
  const int i = lane & 15, g = lane >> 4;
  const int logical = p.xcd_remap ? skf_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int group = logical % groups, worker = logical / groups;
  const int n_lane = group * 4 * CW + wave * CW + NB * i;
  const bool nok = n_lane < p.N;
  const int n_ld = nok ? n_lane : p.N - NB;
  const int ntiles = (p.M + TR - 1) / TR;
  // Row-block list (16-row blocks = tiles): the loop below walks list POSITIONS [0, nlive); phys() maps a position to its
  // tile (positions past the live ones map past the matrix: empty descriptors, nothing loaded or stored).  The dead
  // tiles' output rows are zero-filled at the end.
  // (constant address space: the list is written by an earlier launch, never by this one, and every index is wave-uniform - the
  //  entries arrive through s_load.  As plain global loads each lookup was `global_load_dword; s_waitcnt vmcnt(0)`: a full memory
  //  round trip per tile that also drained the A tiles requested ahead)
  typedef const __attribute__((address_space(4))) int* const_i32p;
  const const_i32p blk = (const_i32p)p.row_blocks;
  const int nlive = blk ? blk[0] : ntiles;
  auto phys = [&](int pos) -> int { return pos < nlive ? (blk ? blk[2 + pos] : pos) : ntiles; };

  // LNF: the epilogue's operands are requested before everything else (the dropout key is two dependent loads away)
  float* lnst = reinterpret_cast<float*>(smem_x + 2 * TILE_B);   // [2 tile parities][TR rows][4 waves] (mean, centred square sum)
  float ln_g[NB], ln_b[NB], ln_inv_keep = 1.f;
  uint32_t ln_key = 0u, ln_thresh = 0u;
  if constexpr (LNF) {
    if (p.ln_rate > 0.f) {
      typedef const __attribute__((address_space(4))) uint32_t* const_u32p;   // written by the step prologue launch: a scalar load, waited for at its use
      ln_key = *(const_u32p)&reinterpret_cast<const SkfStepState*>(p.ln_state)->drop_key;
      ln_thresh = skf_drop_thresh(p.ln_rate);
      ln_inv_keep = 1.0f / (1.0f - p.ln_rate);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { ln_g[nb] = p.ln_gamma[n_ld + nb]; ln_b[nb] = p.ln_beta[n_ld + nb]; }
  }
  long long* dbg = (p.dbg && lane == 0 && wave_all == 0 && (blockIdx.x % 64) == 0 && blockIdx.x / 64 < 8) ? p.dbg + (blockIdx.x / 64) * 32 : nullptr;   // same XCD: comparable clocks
  int dbi = 0;
#if SKF_WS_STAMPS   // per-phase s_memtime stamps (tools/ws_timeline.py); off by default
#define SKF_STAMP() do { if (dbg && dbi < 30) dbg[dbi++] = clock64(); } while (0)
#else
#define SKF_STAMP() do { (void)dbg; (void)dbi; } while (0)
#endif
#if SKF_WS_STAMPS
  const long long wall0 = wall_clock64(), cyc0 = clock64();   // 100 MHz constant clock vs shader clock
#endif
  SKF_STAMP();
  unsigned a_voff[NV], c_voff[4], h_voff[4];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int e = tid + v * 256 * KS, row = e / (K / 4), c4 = (e % (K / 4)) * 4;
    a_voff[v] = (unsigned)(row * p.lda + c4) * 4u;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    c_voff[r] = nok ? (unsigned)((4 * g + r) * p.ldc + n_lane) * 4u : OOB;
    h_voff[r] = nok ? (unsigned)((4 * g + r) * p.ld_relu + n_lane) * 4u : OOB;
  }
  // Register ring of R A tiles ahead of the one in LDS: with the MFMAs 2.7x shorter than in the fp32 kernel a tile
  // takes ~1.2k cycles, and a global load under a busy chip 3-4k.
  f32x4 ra[R][NV];
  int tile = worker;
  const int a_cut = KMASK ? p.a_cut : 0;
#pragma unroll
  for (int j = 0; j < R; ++j) wsx_load_tile<K, KS>(p.A, p.lda, p.M, phys(tile + j * workers), a_voff, ra[j], a_cut);

  // ---- weight slice -> split bf16 operands (once): bq[nb][s][q] = pieces q of B[k = 32s + 8g + e][n_lane + nb], e < 8
  u32x4 bq[NB][NKS][P];
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    float f[NB][8];
    if (B_KC) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int k4 = 32 * (s0 + s) + 8 * g + 4 * h;
          // KMASK: k_valid is a multiple of 4, so a 4-vector is all in or all out; out ones re-read the last valid vector (no
          // access behind the weight row) and count as zeros
          const int kl = KMASK ? (k4 < p.k_valid ? k4 : p.k_valid - 4) : k4;
          f32x4 v = *reinterpret_cast<const f32x4*>(p.B + (size_t)(n_ld + nb) * p.ldb + kl);
          if (KMASK && k4 >= p.k_valid) v = (f32x4){0.f, 0.f, 0.f, 0.f};
          f[nb][4 * h + 0] = v[0]; f[nb][4 * h + 1] = v[1]; f[nb][4 * h + 2] = v[2]; f[nb][4 * h + 3] = v[3];
        }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const vecn v = *reinterpret_cast<const vecn*>(p.B + (size_t)(32 * (s0 + s) + 8 * g + e) * p.ldb + n_ld);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) f[nb][e] = reinterpret_cast<const float*>(&v)[nb];
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        unsigned pc[P];
        split2<P>(f[nb][2 * d], f[nb][2 * d + 1], pc, sel);
#pragma unroll
        for (int q = 0; q < P; ++q) bq[nb][s][q][d] = pc[q];
      }
  }
  float bias_r[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bias_r[nb] = (p.bias && kh == 0) ? p.bias[n_ld + nb] : 0.f;

  const uint32_t ln_sk = (LNF && p.ln_rate > 0.f) ? skf_site_key(ln_key, p.ln_site) : 0u;
  SKF_STAMP();   // weight slice loaded + split
  wsx_store_tile<K, P, PITCH, KS>(As, ra[0], sel);
  __syncthreads();
  wsx_load_tile<K, KS>(p.A, p.lda, p.M, phys(tile + R * workers), a_voff, ra[0], a_cut);
  SKF_STAMP();   // first A tile in LDS

  vecn cprev[4], hsrc[4], oacc[4];
  vecn xresA[LNF ? 4 : 1], xresB[LNF ? 4 : 1];                    // LNF: residual rows of this tile and of the next one (requested a tile ahead:
                                                                  // a load issued inside the tile that consumes it stalls ~2.5k cycles per tile)
  int prev_tile = ntiles, prev_par = 0;
  const bool has_bits = EXTRA >= 2 && p.relu_bits_in != nullptr;  // relu'(.) from the forward's sign bits instead of relu_src
  const bool has_relu = EXTRA == 3 && p.relu_src != nullptr && !has_bits;
  const int ncw = groups * 4, cwi = group * 4 + wave;             // column waves of the launch / this wave's index
  unsigned long long mbits[EXTRA >= 2 ? 4 * NB : 1];                   // sign-bit words of the tile whose C is stored next (uniform: SGPRs)
  // KS = 2: partial sums of the upper contraction half, [2 tile parities][4 column waves][64 lanes] x vecn[4]
  vecn* xch = reinterpret_cast<vecn*>(smem_x + 2 * TILE_B);
  auto store_prev = [&]() {
    if (KS == 2 && kh != 0) return;                              // the upper half's waves own no output
    const __amdgpu_buffer_rsrc_t rc = wsx_rows_rsrc(p.C, p.ldc, p.M, prev_tile * TR);
    if constexpr (KS == 2) {
      const vecn* part = xch + ((prev_par * 4 + wave) * 64 + lane) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) cprev[r] += part[r];
    }
    if constexpr (LNF) {
      const __amdgpu_buffer_rsrc_t ro = wsx_rows_rsrc(p.ln_out, p.ldc, p.M, prev_tile * TR);
      const __amdgpu_buffer_rsrc_t rs = wsx_rows_rsrc(p.ln_stats, 2, p.M, prev_tile * TR);
      const float* sp = lnst + prev_par * TR * 8;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(sp + (4 * g + r) * 8), b = *reinterpret_cast<const f32x4*>(sp + (4 * g + r) * 8 + 4);
        const float mean = 0.25f * ((a[0] + a[2]) + (b[0] + b[2]));
        const float d0 = a[0] - mean, d1 = a[2] - mean, d2 = b[0] - mean, d3 = b[2] - mean;
        const float m2 = ((a[1] + a[3]) + (b[1] + b[3])) + (float)CW * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
        const float rstd = rsqrtf(m2 * (1.0f / (4 * CW)) + 1e-6f);
        vecn o;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          reinterpret_cast<float*>(&o)[nb] = (reinterpret_cast<const float*>(&cprev[r])[nb] - mean) * rstd * ln_g[nb] + ln_b[nb];
        wsx_buf_store<NB>(cprev[r], rc, c_voff[r]);
        wsx_buf_store<NB>(o, ro, c_voff[r]);
        __builtin_amdgcn_raw_buffer_store_b64((u32x2){__builtin_bit_cast(unsigned, mean), __builtin_bit_cast(unsigned, rstd)}, rs,
                                              (wave == 0 && i == 0) ? (unsigned)(4 * g + r) * 8u : OOB, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      vecn v = cprev[r];
      if (EXTRA) {
        if (EXTRA == 2 || (EXTRA == 3 && has_bits)) {     // (general form: a wave-uniform branch, no memory operation inside)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            float sel;
            asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(sel) : "v"(reinterpret_cast<const float*>(&v)[nb]), "s"(mbits[r * NB + nb]));
            reinterpret_cast<float*>(&v)[nb] = sel;
          }
        } else if (EXTRA == 3) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          reinterpret_cast<float*>(&v)[nb] = (!has_relu || reinterpret_cast<const float*>(&hsrc[r])[nb] > 0.f) ? reinterpret_cast<const float*>(&v)[nb] : 0.f;
        }
        if (EXTRA != 2) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) reinterpret_cast<float*>(&v)[nb] += reinterpret_cast<const float*>(&oacc[r])[nb];
        }
      }
#ifdef SKF_WSX_ABLATE_STORE   // diagnostics: only the first tile's stores reach memory
      if (prev_tile < workers)
#endif
      wsx_buf_store<NB>(v, rc, c_voff[r]);
    }
  };

  const int frag_off = i * PITCH + 16 * g + 64 * s0;   // byte offset of this lane's fragment inside (plane, first step of the wave)
  u32x4 afA[PF], afB[PF];
  if (EARLY) {
#pragma unroll
    for (int f = 0; f < PF; ++f) afA[f] = *reinterpret_cast<const u32x4*>(As + (f % P) * TR * PITCH + frag_off + 64 * (f / P));
  }

  if constexpr (LNF) {
    const __amdgpu_buffer_rsrc_t rx = wsx_rows_rsrc(p.ln_x, p.ldc, p.M, phys(tile) * TR);
#pragma unroll
    for (int r = 0; r < 4; ++r) xresA[r] = wsx_buf_load<NB>(rx, c_voff[r]);
  }
  auto do_tile = [&](int cur, f32x4 (&rn)[NV], u32x4 (&af)[PF], u32x4 (&afn)[PF], vecn (&xres)[LNF ? 4 : 1], vecn (&xnext)[LNF ? 4 : 1]) {
    const char* At = As + cur * TILE_B + frag_off;
    if (!EARLY) {
#pragma unroll
      for (int f = 0; f < PF; ++f) af[f] = *reinterpret_cast<const u32x4*>(At + (f % P) * TR * PITCH + 64 * (f / P));
    }
    SKF_WSX_SCHED_BARRIER();
    store_prev();
    SKF_WSX_SCHED_BARRIER();
    SKF_STAMP();   // previous C tile stored
    // One accumulator chain per column block (two for NB = 1: a dependent MFMA straight behind its producer stalls):
    // the small products first, the a0.b0 products last - small-to-large summation, no separate add.
    f32x4 acc[NB][NCH];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float b0 = c == 0 ? bias_r[nb] : 0.f;
        acc[nb][c] = (f32x4){b0, b0, b0, b0};
      }
    auto frag = [&](int s, int q) -> u32x4 {
      const int f = s * P + q;
      if (f < PF) return af[f];
      return *reinterpret_cast<const u32x4*>(At + q * TR * PITCH + 64 * s);
    };
    // K <= 256: all small products of the tile first, its a0.b0 products last (their A fragments stay in registers);
    // longer K: the a0.b0 product closes each k-step (no room to keep K/32 fragments, and no second LDS read).
    constexpr bool TWO_PHASE = K / KS <= 256;
    u32x4 a0keep[TWO_PHASE ? NKS : 1];
    int c = 0;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      u32x4 a[P];
#pragma unroll
      for (int q = 0; q < P; ++q) a[q] = frag(s, q);
      if (TWO_PHASE) a0keep[s] = a[0];
#pragma unroll
      for (int d = 1; d < P; ++d)            // d = qa + qb: products of equal magnitude together
#pragma unroll
        for (int qa = 0; qa <= d; ++qa) {
          const int qb = d - qa;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
#ifdef SKF_WSX_ABLATE_MFMA   // diagnostics: wrong results; keeps every operand live with one VALU op per 4 MFMAs
            if ((c & 3) == 0 && nb == 0) acc[nb][c % NCH][0] += __builtin_bit_cast(float, a[qa][0] ^ bq[nb][s][qb][0]);
#else
            acc[nb][c % NCH] = mfma_bf16(a[qa], bq[nb][s][qb], acc[nb][c % NCH]);
#endif
          }
          ++c;
        }
      if (!TWO_PHASE) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#ifndef SKF_WSX_ABLATE_MFMA
          acc[nb][c % NCH] = mfma_bf16(a[0], bq[nb][s][0], acc[nb][c % NCH]);
#endif
        }
        ++c;
      }
      if (s == NKS / 2 - 1) {
        SKF_WSX_SCHED_BARRIER();
        wsx_store_tile<K, P, PITCH, KS>(As + (cur ^ 1) * TILE_B, rn, sel);
        wsx_load_tile<K, KS>(p.A, p.lda, p.M, phys(tile + (R + 1) * workers), a_voff, rn, a_cut);
        if constexpr (LNF) {
          const __amdgpu_buffer_rsrc_t rx = wsx_rows_rsrc(p.ln_x, p.ldc, p.M, phys(tile + workers) * TR);
#pragma unroll
          for (int r = 0; r < 4; ++r) xnext[r] = wsx_buf_load<NB>(rx, c_voff[r]);
        }
        if (EXTRA && (KS == 1 || kh == 0)) {
          const int ptile = phys(tile);
          if constexpr (EXTRA == 3) {
            const __amdgpu_buffer_rsrc_t rh = wsx_rows_rsrc(has_relu ? p.relu_src : p.C, p.ld_relu, has_relu ? p.M : 0, ptile * TR);
#pragma unroll
            for (int r = 0; r < 4; ++r) hsrc[r] = wsx_buf_load<NB>(rh, h_voff[r]);
          }
          if constexpr (EXTRA != 2) {
            const __amdgpu_buffer_rsrc_t ro = wsx_rows_rsrc(p.C, p.ldc, p.accumulate ? p.M : 0, ptile * TR);
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[r] = wsx_buf_load<NB>(ro, c_voff[r]);
          }
          if constexpr (EXTRA >= 2) {
          // sign-bit words of this tile: one uniform (scalar) load per wave; without bits the words of tile 0 of some valid
          // buffer are fetched and ignored (no branch around a memory operation in the loop)
          // (constant address space: the words are never written by this launch, so a uniform address becomes ONE s_load)
          typedef const __attribute__((address_space(4))) unsigned long long* const_u64p;
          const const_u64p wp = (const_u64p)(has_bits ? p.relu_bits_in + ((size_t)(ptile < ntiles ? ptile : 0) * ncw + cwi) * (4 * NB)
                                                      : reinterpret_cast<const unsigned long long*>(p.B));
#pragma unroll
          for (int j = 0; j < 4 * NB; ++j) mbits[j] = wp[j];
          }
        }
        SKF_WSX_SCHED_BARRIER();
      }
      if (EARLY && s == (3 * NKS) / 4 - 1) {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        const char* An = As + (cur ^ 1) * TILE_B + frag_off;
#pragma unroll
        for (int f = 0; f < PF; ++f) afn[f] = *reinterpret_cast<const u32x4*>(An + (f % P) * TR * PITCH + 64 * (f / P));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (TWO_PHASE) {
#pragma unroll
      for (int s = 0; s < NKS; ++s) {        // the a0.b0 products
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#ifndef SKF_WSX_ABLATE_MFMA
          acc[nb][c % NCH] = mfma_bf16(a0keep[s], bq[nb][s][0], acc[nb][c % NCH]);
#endif
        }
        ++c;
      }
    }
    SKF_STAMP();   // MFMAs issued
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float v = acc[nb][0][r];
        if (NCH == 2) v += acc[nb][1][r];
        reinterpret_cast<float*>(&cprev[r])[nb] = v;
      }
    if constexpr (KS == 2) {
      if (kh != 0) {
        vecn* part = xch + ((cur * 4 + wave) * 64 + lane) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) part[r] = cprev[r];
      }
      prev_par = cur;
    }
    if constexpr (LNF) {
      const uint32_t karg0 = ((uint32_t)(phys(tile) * TR + 4 * g) * (uint32_t)(4 * CW) + (uint32_t)n_lane) * kSkfKeepStride + ln_sk;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* zr = reinterpret_cast<float*>(&cprev[r]);
        const float* xr = reinterpret_cast<const float*>(&xres[r]);
        float sum = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          float y = zr[nb];
          // element (row0 + r, n_lane + nb) of the (rows, 128) tensor: its hash argument is karg0 plus a compile-time multiple of the stride
          if (p.ln_rate > 0.f) y *= skf_keep_arg(karg0 + (uint32_t)(r * 4 * CW + nb) * kSkfKeepStride, ln_thresh) ? ln_inv_keep : 0.f;
          zr[nb] = xr[nb] + y;
          sum += zr[nb];
        }
        const float mw = wsx_row16_sum(sum) * (1.0f / CW);
        float sq = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { const float c = zr[nb] - mw; sq += c * c; }
        sq = wsx_row16_sum(sq);
        if (i == 0) *reinterpret_cast<float2*>(lnst + ((cur * TR + 4 * g + r) * 4 + wave) * 2) = make_float2(mw, sq);
      }
      prev_par = cur;
    }
    if (KS == 1 && p.act == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) reinterpret_cast<float*>(&cprev[r])[nb] = __builtin_amdgcn_fmed3f(reinterpret_cast<float*>(&cprev[r])[nb], 0.f, __builtin_inff());   // one op (fmaxf = canonicalise + max)
      if (!EXTRA) {
        // sign bits for the input-gradient launch: word j = r * NB + nb is the ballot of "> 0" over the wave, stored by lane j
        // (every lane executes the store: lanes >= 4 NB and launches without a bit buffer fall outside the descriptor)
        unsigned long long w = 0;
        if (p.relu_bits_out) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              const unsigned long long bm = __ballot(reinterpret_cast<float*>(&cprev[r])[nb] > 0.f);
              if (lane == r * NB + nb) w = bm;
            }
        }
        const int ptile = phys(tile);
        const bool wr = p.relu_bits_out != nullptr && ptile < ntiles;
        const size_t woff = wr ? ((size_t)ptile * ncw + cwi) * (4 * NB) * 8 : 0;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(wr ? (void*)p.relu_bits_out : (void*)p.C) + woff, 0, wr ? 4 * NB * 8 : 0, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, w), rb, (unsigned)lane * 8u, 0, 0);
      }
    } else if (KS == 1 && p.act == 2) {   // the bottleneck's tanh projection (one launch per step): a wave-uniform branch nobody else takes
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) reinterpret_cast<float*>(&cprev[r])[nb] = tanhf(reinterpret_cast<float*>(&cprev[r])[nb]);
    }
    prev_tile = phys(tile);
    if (!EARLY) __syncthreads();
    SKF_STAMP();   // tile done
  };
  constexpr int U = (R & 1) ? 2 * R : R;   // unroll: LDS buffer parity x ring position (written out: a `for` with
                                           // a break is not unrolled and would index the ring dynamically)
#define SKF_WSX_STEP(u)                                              \
  {                                                                  \
    if ((u) & 1) do_tile(1, ra[((u) + 1) % R], afB, afA, xresB, xresA); \
    else do_tile(0, ra[((u) + 1) % R], afA, afB, xresA, xresB);      \
    tile += workers;                                                 \
    if (tile >= nlive) break;                                        \
  }
  while (tile < nlive) {
    SKF_WSX_STEP(0) SKF_WSX_STEP(1)
    if constexpr (U > 2) { SKF_WSX_STEP(2) SKF_WSX_STEP(3) }
    if constexpr (U > 4) { SKF_WSX_STEP(4) SKF_WSX_STEP(5) }
  }
#undef SKF_WSX_STEP
  store_prev();
  if (blk && !p.accumulate && kh == 0) {             // rows of dead tiles: zeros (an accumulating call leaves them as they are)
    vecn zero;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) reinterpret_cast<float*>(&zero)[nb] = 0.f;
    for (int pos = nlive + worker; pos < ntiles; pos += workers) {
      const __amdgpu_buffer_rsrc_t rz = wsx_rows_rsrc(p.C, p.ldc, p.M, blk[2 + pos] * TR);
#pragma unroll
      for (int r = 0; r < 4; ++r) wsx_buf_store<NB>(zero, rz, c_voff[r]);
    }
  }
  SKF_STAMP();
#if SKF_WS_STAMPS
  if (dbg) { dbg[30] = wall_clock64() - wall0; dbg[31] = clock64() - cyc0; }
#endif
#undef SKF_STAMP
}

template <int K, int NB, int P, int KS = 1>
int launch_wsx(const GemmParams& p, int b_kc, hipStream_t st) {
  constexpr int CW = 16 * NB;
  const int groups = skf_cdiv(p.N, 4 * CW);
  static const int wg_target = skf_knob("SKF_WS_WGS") ? atoi(skf_knob("SKF_WS_WGS")) : (KS == 1 && K <= 256 && NB <= 2 ? 512 : 256);
  int workers = wg_target / groups;
  if (workers < 1) workers = 1;
  const int ntiles = skf_cdiv(p.M, TR);
  if (workers > ntiles) workers = ntiles;
  const size_t smem = (size_t)2 * P * TR * (2 * K + 32) + (KS == 2 ? (size_t)2 * 4 * 64 * 4 * NB * sizeof(float) : 0);
  dim3 grid(groups * workers), block(256 * KS);
  static const std::string tag = "gemm_wsx<K" + std::to_string(K) + ",CW" + std::to_string(CW) + ",bf16x" + std::to_string(P * (P + 1) / 2) + (KS == 2 ? ",ksplit" : "") + ">";
  // epilogue kind (see the kernel): 2 sign bits only, 3 anything else that needs epilogue operands.
  // There is no "accumulate only" kind.  Round 3 had one (the general form minus the relu_src requests, +0.5 % on the step) whose
  // instantiation <K = 256, one column per lane> gave run-to-run different results.  Round 4 bisected it on the hardware
  // (profiles/r04b_kind1_bisect.txt): with only the old-C add left in the epilogue the compiler fuses the four adds of a lane into
  // v_pk_add_f32 with CROSSED operand selects (op_sel:[0,1] op_sel_hi:[1,0]) in the kernel's exit block; the wrong cells (rows 12 / 14
  // of a workgroup's last tile = lanes 48-63 of the first and third add) are exactly the low halves of those instructions.  Draining
  // VMEM and 16 wait states in front of them changed nothing (not an s_waitcnt count), zero-initialised epilogue registers changed
  // nothing (not an undefined value), the same adds forced to four v_add_f32 were bit-reproducible over 210 runs.  The general form
  // never lets the compiler build that instruction (its select sits between the product and the add): tools/isa_pk_opsel.py
  // counts crossed-select packed-fp32 instructions per kernel (0 in this file, checked by tests/test_cabi_cpu.py).
  const int extra = p.relu_src ? 3 : p.relu_bits_in ? (p.accumulate ? 3 : 2) : p.accumulate ? 3 : 0;
  GemmParams q = p;
  if (q.row_block_rows != TR) q.row_blocks = nullptr;       // the list's blocks must be this kernel's tiles
  // profiling: the dense figures, and the work of the live tiles only (A rows read / multiplied; every C row is still written)
  const double live = skf_prof_list_fraction(q.row_blocks);
  const double a_c = (double)p.M * p.K + (double)p.M * p.N * ((p.accumulate ? 1 : 0) + (p.relu_src && !p.relu_bits_in ? 1 : 0));
  const double ln_c = p.ln_out ? 2.0 * p.M * p.N : 0.0;
  // SKF_PROF_FINE=1 (analysis only): one table line per output width and epilogue
  static const bool fine = skf_knob("SKF_PROF_FINE") && skf_knob("SKF_PROF_FINE")[0] == '1';
  static std::set<std::string> fine_tags;           // the profiler keeps the pointer: interned
  const char* ftag = nullptr;
  if (fine) ftag = fine_tags.insert(tag + "[N" + std::to_string(p.N) + (b_kc ? ",dgrad" : "") + (p.relu_bits_in ? ",bits" : "") + (p.relu_src ? ",relu_src" : "") +
                   (p.accumulate ? ",acc" : "") + (q.row_blocks ? ",list" : "") + (p.act ? ",act" : "") + (p.ln_out ? ",ln" : "") + "]").first->c_str();
  // (the launches with the LayerNorm epilogue stay in their family's line - same kernel template, same product - as the rocprofv3
  //  kernel names that bench.py matches against do; their residual / LayerNorm bytes are counted)
  SkfProfScope ps(st, fine ? ftag : tag.c_str(), 2.0 * p.M * p.N * p.K, 4.0 * (a_c + ln_c + (double)p.K * p.N + (double)p.M * p.N));
  ps.done(2.0 * p.M * p.N * p.K * live, 4.0 * (a_c * live + ln_c + (double)p.K * p.N + (double)p.M * p.N));
  // K >= 384 (N = 128): the two column groups of a worker read the same A tiles - XCD-contiguous ids keep the second read
  // in the L2 (PMC: 132 -> ~80 MB per launch); with one or two groups of short tiles (K <= 256) the remap only costs
  // K = 128 with three or more column groups (N = 384 / 512 / 1004): round-robin ids put the group-mates of a worker on
  // different XCDs, i.e. every A tile is fetched into `groups` L2s (PMC, round 1: 1.55x the algorithmic bytes)
  static const char* xcd_env = skf_knob("SKF_WS_XCD");     // "0" / "1" force it (measurement)
  q.xcd_remap = xcd_env ? xcd_env[0] == '1' : (groups > 1 && (K >= 384 || groups >= 3));
#define SKF_WSX_LAUNCH(BKC, EX)                                                                                    \
  do {                                                                                                             \
    static SkfOncePerDevice attr_done;                                                                             \
    if (attr_done.needed()) {                                                                                       \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wsx_kernel<K, NB, P, BKC, EX, false, false, KS>),  \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess)                \
        attr_done.mark();                     /* (a failed call shows up as the launch error below) */              \
    }                                                                                                              \
    SKF_LAUNCH_TAIL((gemm_wsx_kernel<K, NB, P, BKC, EX, false, false, KS>), grid, block, smem, st, q, groups, workers); \
  } while (0)
  if constexpr (K == 128 && NB == 2 && KS == 1) {
    if (q.ln_out) {            // residual + dropout + LayerNorm epilogue (skf_gemm_ln_residual_f32 checked the shape)
      if (groups != 1 || extra || q.row_blocks || b_kc || q.act != 0) { skf_set_error("gemm_wsx: LayerNorm epilogue on an unsupported launch"); return SKF_EUNSUPPORTED; }
      const size_t smem_ln = smem + (size_t)2 * TR * 4 * 2 * sizeof(float);
      static SkfOncePerDevice attr_ln;
      if (attr_ln.needed()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wsx_kernel<K, NB, P, false, 0, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_ln) == hipSuccess)
          attr_ln.mark();
      }
      SKF_LAUNCH_TAIL((gemm_wsx_kernel<K, NB, P, false, 0, false, true>), grid, block, smem_ln, st, q, groups, workers);
      SKF_LAUNCH_CHECK();
      return SKF_OK;
    }
  }
  if constexpr (K == 512 && NB == 1) {
    if (q.k_valid > 0) {       // masked last slice of a long contraction (dgrad form only: skf_gemm_ws_dispatch)
      static SkfOncePerDevice attr_m[2];
      if (attr_m[extra ? 1 : 0].needed()) {
        const hipError_t ea = extra ? hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wsx_kernel<K, NB, P, true, 3, true, false, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
                                    : hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wsx_kernel<K, NB, P, true, 0, true, false, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (ea == hipSuccess) attr_m[extra ? 1 : 0].mark();
      }
      if (extra) SKF_LAUNCH_TAIL((gemm_wsx_kernel<K, NB, P, true, 3, true, false, KS>), grid, block, smem, st, q, groups, workers);
      else SKF_LAUNCH_TAIL((gemm_wsx_kernel<K, NB, P, true, 0, true, false, KS>), grid, block, smem, st, q, groups, workers);
      SKF_LAUNCH_CHECK();
      return SKF_OK;
    }
  }
  // (the forward form [K][N] only ever carries the general epilogue: relu_src / accumulate there are test-only combinations)
  if (b_kc && extra == 2) SKF_WSX_LAUNCH(true, 2);
  else if (b_kc && extra) SKF_WSX_LAUNCH(true, 3);
  else if (b_kc) SKF_WSX_LAUNCH(true, 0);
  else if (extra) SKF_WSX_LAUNCH(false, 3);
  else SKF_WSX_LAUNCH(false, 0);
#undef SKF_WSX_LAUNCH
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

template <int P>
int launch_wsx_k(const GemmParams& p, int b_kc, hipStream_t st) {
  switch (p.K) {
    // (four columns per lane = 64 per wave, one workgroup per CU, was measured for N >= 256: 30.8 vs 28.7 us at N = 512 -
    //  one wave per SIMD loses more to exposed waits than the doubled MFMA : overhead ratio wins)
    case 128: return launch_wsx<128, 2, P>(p, b_kc, st);
    case 256: {
      // K = 256: only for N <= 128 (one or two column groups, 256 workgroups either way); with more column groups the 512-thread form
      // was 4 % slower at cfg 3 (N = 256 ... 1024).  SKF_WSX_KSPLIT256=1 forces it for every N, =0 turns it off (measurement)
      static const char* ks2 = skf_knob("SKF_WSX_KSPLIT256");
      const bool ks_on2 = ks2 ? ks2[0] == '1' : (p.N <= 128 && !(skf_knob("SKF_WSX_KSPLIT") && skf_knob("SKF_WSX_KSPLIT")[0] == '0'));
      if (ks_on2 && p.act == 0) return launch_wsx<256, 1, P, 2>(p, b_kc, st);
      return launch_wsx<256, 1, P>(p, b_kc, st);
    }
    case 384: {
      static const bool ks_off3 = (skf_knob("SKF_WSX_KSPLIT") && skf_knob("SKF_WSX_KSPLIT")[0] == '0') || (skf_knob("SKF_WSX_KSPLIT384") && skf_knob("SKF_WSX_KSPLIT384")[0] == '0');
      if (!ks_off3 && p.act == 0) return launch_wsx<384, 1, P, 2>(p, b_kc, st);
      return launch_wsx<384, 1, P>(p, b_kc, st);
    }
    default: {
      // K = 512 without an activation: the contraction split between wave pairs (two waves per SIMD); SKF_WSX_KSPLIT=0: A/B knob
      static const bool ks_off = skf_knob("SKF_WSX_KSPLIT") && skf_knob("SKF_WSX_KSPLIT")[0] == '0';
      if (!ks_off && p.act == 0) return launch_wsx<512, 1, P, 2>(p, b_kc, st);
      return launch_wsx<512, 1, P>(p, b_kc, st);
    }
  }
}

}  // namespace

// bytes of the sign-bit buffer of an (M, N, K) launch: [16-row tile][column wave][4 * NB] 64-bit words (see GemmParams)
size_t skf_gemm_wsx_relu_bits_bytes(int M, int N, int K) {
  const int NB = K == 128 ? 2 : 1, CW = 16 * NB;
  return (size_t)skf_cdiv(M, TR) * (size_t)skf_cdiv(N, 4 * CW) * 4 * (size_t)(4 * NB) * sizeof(unsigned long long);
}

// pieces = 3 (six products, fp32-equivalent) or 2 (three products); same applicability rules as skf_gemm_ws_dispatch
int skf_gemm_wsx_launch(const GemmParams& p, int b_kcontig, int pieces, hipStream_t st) {
  // 32-bit byte offsets inside the kernels (callers fall back to the fp32 kernels otherwise: skf_gemm_wsx_fits)
  return pieces == 2 ? launch_wsx_k<2>(p, b_kcontig, st) : launch_wsx_k<3>(p, b_kcontig, st);
}
