// The feed-forward block of a transformer layer in ONE launch per direction (d_model = 128, dff = 512):
//   forward  (builders/layers/transformer.py:194-198 point_wise_feed_forward_network, :221-224 / :270-272 the layer's
//             `layernorm(out + dropout(ffn(out)))`):  h = relu(x.W1 + b1);  y = h.W2 + b2;  z = x + dropout(y);  out = LayerNorm(z)
//   backward (its tape gradient):  dh = (dy.W2^T) o relu'(h);  dx (+)= dh.W1^T
// with the arithmetic of skf_gemm_wsx.hip: every fp32 operand split EXACTLY into P bf16 pieces, the P(P+1)/2 largest piece products
// summed in fp32 on the bf16 matrix cores.  The three (forward) / two (backward) launches it replaces each paid a launch, a weight
// prologue, a first-tile latency and a drain for ~100 rows of work per CU, and the hidden tensor made a 52 MB round trip between them.
//
// Both directions are the same chain  Y = (f(A.B1)).B2  with A [M][128], B1 [128][512], B2 [512][128]:
//   * A persistent workgroup (512 threads = 8 waves, two per SIMD, one workgroup per CU) owns a contiguous range of 16-row tiles and
//     walks it in sub-groups of up to four tiles (64 rows).  The A rows of a sub-group are split once into P bf16 planes in LDS.
//   * The weights cannot be stationary: B1 and B2 as bf16 planes are 786 KB, a CU has 512 KB of registers and 160 KB of LDS.  They
//     stream from the L2 in PRE-SPLIT, FRAGMENT-ORDERED images (skf_ffn_weight_images: one kernel per step for all layers) - a wave's
//     twelve MFMA B operands of a stage are 12 KB of contiguous memory, each a coalesced 1 KB wave load with no arithmetic behind it
//     (the weight-stationary kernels spend 32 strided loads + ~350 VALU per wave and launch on the same job), requested one stage ahead.
//   * The hidden dimension goes by in four blocks of 128: stage 1 = hidden block b of all row tiles of the sub-group (wave w owns 16
//     hidden columns: 24 MFMAs per row tile), written to global memory as fp32 (the weight gradient reads it), its sign bits as wave
//     ballots, and as P bf16 planes into one of two LDS buffers; one barrier; stage 2 = Y += Hblock.B2[block rows] (wave w owns 16
//     of the 128 output columns; accumulators of all row tiles stay in registers across the four blocks).  The hidden tensor is
//     never read back.
//   * Epilogue: Y through LDS so that a half-wave owns a whole 128-column row: residual + dropout + LayerNorm exactly as
//     ln_fwd_v4_kernel (skf_rowops.hip) computes them (forward), or the accumulation into dx with 16-byte accesses (backward).
// LDS rows are 256 bytes (128 bf16) with the 16-byte chunks XOR-swizzled by the row index: every ds_read_b128 service group of a
// fragment read (lanes {0-3,12-15,20-27}, ...) then touches 16 different bank quads, and the 2-byte hidden-plane stores of the four
// lane groups g (rows 4g + r, same columns) land in different banks too - no padding, so that X planes + two hidden buffers fit.
#include "skf_common.h"
#include "skf_ffn_fused.h"
#include <string>
#include <vector>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int FD = 128;               // model width
constexpr int FF = 512;               // hidden width
constexpr int HB = 128;               // hidden units per block
constexpr int NBLK = FF / HB;
constexpr int TR = 16;                // rows per tile
constexpr int MAXRT = 4;              // row tiles per sub-group
constexpr int ROWS = TR * MAXRT;
constexpr int RPITCH = 256;           // bytes per plane row
constexpr int PLANE = ROWS * RPITCH;  // bytes per plane
constexpr int NKS = 4;                // 32-deep MFMA steps per 128-deep contraction
constexpr int YPITCH = FD + 4;        // floats per row of the epilogue tile
constexpr unsigned OOB = 0x7ffffff0u;

// SKF_FFN_ABLATE (diagnostics builds, wrong results): bit 0 no MFMA (one VALU op keeps the operands live), 1 fragment reads only
// once per sub-group, 2 no hidden-tensor / sign-bit stores, 3 no hidden-plane stores, 4 no stage-1 epilogue at all, 5 weight
// fragments loaded once per launch, 6 no block barriers
#ifndef SKF_FFN_ABLATE
#define SKF_FFN_ABLATE 0
#endif
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
#if SKF_FFN_ABLATE & 1
  c[0] += __builtin_bit_cast(float, (a[0] ^ b[0]) & 0x3fffffu);
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// descriptor over rows [row0, M) of a row-major fp32 matrix (byte counts are 32-bit: the launcher checks M * ld * 4 < 2^31)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ffn_rows_rsrc(const float* base, int ld, int M, int row0) {
  const int rows_left = M - row0 > 0 ? M - row0 : 0;
  const unsigned rem = rows_left > 0 ? (unsigned)rows_left * (unsigned)ld * 4u : 0u;
  const unsigned off = rows_left > 0 ? (unsigned)row0 * (unsigned)ld * 4u : 0u;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base) + off), 0, rem, 0x00020000);
}

size_t image_bytes(int pieces) { return (size_t)2 * pieces * (FD * FF * 2); }

// ---------------------------------------------------------------- weight images
// Image of an MFMA B operand B [K][N] (K % 32 == 0, N % 16 == 0): fragment (cb < N / 16, s < K / 32, q < P) = pieces q of
// B[32s + 8g + e][16cb + i], lane = 16g + i, e < 8 (16 bytes per lane, 1 KB per fragment), fragments ordered [cb][s][q].
// transpose = 0: B[k][n] = src[k * ld + n]; 1: B[k][n] = src[n * ld + k].
// The feed-forward pair: image 1 = B1 [128][512] followed by image 2 = B2 [512][128]; forward B1 = W1, B2 = W2; input gradient
// B1 = W2^T, B2 = W1^T.
struct DenseImageDesc { const float* src; char* img; int ld, transpose, K, N, block_begin, pad; };
constexpr int kImageBatch = 80;   // (80 descriptors of 40 bytes: 3.1 KB of kernel arguments; the 66 images of a cfg-2 train step are one launch)
struct DenseImageBatch { DenseImageDesc d[kImageBatch]; int n; };

template <int P>
__global__ __launch_bounds__(256) void dense_image_kernel(DenseImageBatch batch) {
  int j = 0;
  while (j + 1 < batch.n && (int)blockIdx.x >= batch.d[j + 1].block_begin) ++j;       // (wave-uniform, <= 80 steps)
  const DenseImageDesc d = batch.d[j];
  const SkfSplitSel sel = skf_split_sel();
  const int t = ((int)blockIdx.x - d.block_begin) * 256 + threadIdx.x;
  const int nks = d.K / 32, f = t >> 6, lane = t & 63, i = lane & 15, g = lane >> 4;
  if (f >= (d.N / 16) * nks) return;
  const int cb = f / nks, s = f % nks;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * s + 8 * g + e, n = 16 * cb + i;
    v[e] = d.transpose ? d.src[(size_t)n * d.ld + k] : d.src[(size_t)k * d.ld + n];
  }
  u32x4 pc[P];
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) {
    unsigned o[P];
    skf_split2<P>(v[2 * dd], v[2 * dd + 1], o, sel);
#pragma unroll
    for (int q = 0; q < P; ++q) pc[q][dd] = o[q];
  }
  char* base = d.img + (size_t)f * P * 1024 + lane * 16;
#pragma unroll
  for (int q = 0; q < P; ++q) *reinterpret_cast<u32x4*>(base + q * 1024) = pc[q];
}

int launch_images(const DenseImageDesc* descs, int n, int P, hipStream_t st) {
  for (int b0 = 0; b0 < n; b0 += kImageBatch) {
    DenseImageBatch batch{};
    batch.n = n - b0 < kImageBatch ? n - b0 : kImageBatch;
    int blocks = 0;
    for (int j = 0; j < batch.n; ++j) {
      batch.d[j] = descs[b0 + j];
      batch.d[j].block_begin = blocks;
      blocks += skf_cdiv((batch.d[j].N / 16) * (batch.d[j].K / 32) * 64, 256);
    }
    if (P == 2) hipLaunchKernelGGL(dense_image_kernel<2>, dim3(blocks), dim3(256), 0, st, batch);
    else hipLaunchKernelGGL(dense_image_kernel<3>, dim3(blocks), dim3(256), 0, st, batch);
    SKF_LAUNCH_CHECK();
  }
  return SKF_OK;
}

// ---------------------------------------------------------------- the block
// a wave's NKS * P operand fragments (12 KB of contiguous image): buffer loads - one descriptor per image, the lane's 16 bytes in the
// VGPR offset, everything else wave-uniform in the scalar offset.  (As global loads from per-lane 64-bit addresses the twelve
// addresses of a rarely used image were spilled one by one: scratch reload + vmcnt(0) + load, twelve dependent round trips.)
template <int P>
__device__ __forceinline__ void load_frags(u32x4 (&w)[NKS][P], __amdgpu_buffer_rsrc_t img, unsigned lane16, int soff, bool first = false) {
#if SKF_FFN_ABLATE & 32
  if (!first) return;
#endif
#pragma unroll
  for (int s = 0; s < NKS; ++s)
#pragma unroll
    for (int q = 0; q < P; ++q) w[s][q] = __builtin_amdgcn_raw_buffer_load_b128(img, lane16, soff + (s * P + q) * 1024, 0);
}

// A 16-row x 16-column x 128-deep product goes by in two halves of two 32-deep steps.  The fragments of a half are requested while
// the previous half multiplies (an LDS read issued straight in front of its MFMA costs its whole latency: the first version of
// this kernel ran at the pace of its ds_read_b128s), so they live in two named sets.  Within a half: the small piece products
// first, its a0.b0 products last (two accumulator chains).
template <int P> struct FfnFrags { u32x4 f[2][2][P]; };   // [half][step of the half][piece]
template <int P>
__device__ __forceinline__ void load_half(const char* rows, const unsigned (&a_off)[NKS], u32x4 (&f)[2][P], int half, bool first = false) {
#if SKF_FFN_ABLATE & 2
  if (!first) return;
#endif
#pragma unroll
  for (int sl = 0; sl < 2; ++sl)
#pragma unroll
    for (int q = 0; q < P; ++q) f[sl][q] = *reinterpret_cast<const u32x4*>(rows + q * PLANE + a_off[2 * half + sl]);
}
template <int P>
__device__ __forceinline__ void half_products(const u32x4 (&w)[NKS][P], const u32x4 (&f)[2][P], int half, f32x4& acc0, f32x4& acc1, int& c) {
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int s = 2 * half + sl;
#pragma unroll
    for (int d = 1; d < P; ++d)
#pragma unroll
      for (int qa = 0; qa <= d; ++qa) {
        if (c & 1) acc1 = mfma_bf16(w[s][d - qa], f[sl][qa], acc1);
        else acc0 = mfma_bf16(w[s][d - qa], f[sl][qa], acc0);
        ++c;
      }
  }
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    if (c & 1) acc1 = mfma_bf16(w[2 * half + sl][0], f[sl][0], acc1);
    else acc0 = mfma_bf16(w[2 * half + sl][0], f[sl][0], acc0);
    ++c;
  }
}

// SKF_FFN_STAMPS (diagnostics builds): s_memtime stamps of wave 0 / wave 7 of every 32nd workgroup (tools/ffn_timeline.py)
#ifndef SKF_FFN_STAMPS
#define SKF_FFN_STAMPS 0
#endif
#if SKF_FFN_STAMPS
__device__ long long g_ffn_stamps[16 * 64];
#define FFN_STAMP() do { if (dbg && dbi < 62) dbg[dbi++] = clock64(); } while (0)
#else
#define FFN_STAMP() do { } while (0)
#endif

// sum over the 32 lanes that share a row of the epilogue (lanes 0-31 / 32-63): a DPP all-reduce inside each 16-lane row, then
// one exchange with the neighbouring row (five __shfl_xor steps are five dependent ds_bpermute round trips)
__device__ __forceinline__ float ffn_half_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
  return v + __shfl_xor(v, 16, 64);      // the neighbouring row: one ds_bpermute
}

// MODE 0 forward, 1 backward.  LNB (backward only): the launch starts at the gradient of the LayerNorm OUTPUT - the rows it stages are
// dy = dropout'(dz), dz = LayerNorm'(dout) (skf_rowops.hip ln_bwd_v4_kernel, same arithmetic), formed on the way to LDS; dy goes
// to global memory for the weight gradient, dz stays in registers and closes the block as dx = dz + dh.W1^T, the column sums of
// dout o xhat and dout (dgamma, dbeta) leave as one partial row pair per workgroup.
// POST (forward only): the launch goes on to the Dense that consumes its LayerNorm output (the next layer's q|k|v projection,
// builders/layers/transformer.py:154-158: K = 128, N = 128 or 384) - the rows are split again on their way out, one more set of
// first-stage products per 128 output columns, stored from the transposed fragments like the hidden tensor.
// PRE (forward only): the launch STARTS one sublayer earlier, at the attention output a: x1 = LayerNorm(x + dropout(a . Wo + bo)) - the
// MultiHeadAttention output projection with its residual LayerNorm (builders/layers/transformer.py:186, 216-224) - is formed by one
// product stage (K = N = 128) and a row epilogue in front of the feed-forward block, whose input and residual it is.
// NOFFN (forward, PRE and POST only): the launch is the leading stage followed at once by the chained projection of ITS LayerNorm
// output - the decoder's self-attention tail with the cross-attention query projection behind it (builders/layers/transformer.py:258-262):
// no hidden blocks, no second LayerNorm; runs as attn_tail_proj_kernel (a name of its own in profiles).
template <int P, int MODE, bool LNB, bool POST, bool PRE, bool NOFFN>
__device__ __forceinline__ void ffn_fused_body(const FfnFusedParams& p) {
  static_assert(!LNB || MODE == 1, "LayerNorm-backward prologue: backward only");
  static_assert(!POST || MODE == 0, "chained projection: forward only");
  static_assert(!PRE || MODE == 0, "leading projection + LayerNorm: forward only");
  static_assert(!NOFFN || (PRE && POST), "tail + projection: needs both stages");
  extern __shared__ __attribute__((aligned(16))) char smem_f[];
  char* Xp = smem_f;                     // [P][ROWS][256]
  char* Hp = smem_f + P * PLANE;         // [2][P][ROWS][256]
  // epilogue tile [ROWS][YPITCH]: hidden buffer 0 (last read in the third block's second stage) where it fits, else its own region
  float* Yt = reinterpret_cast<float*>(P * PLANE >= ROWS * YPITCH * 4 ? Hp : Hp + 2 * P * PLANE);
  float* B1s = reinterpret_cast<float*>(Hp + 2 * P * PLANE + (P * PLANE >= ROWS * YPITCH * 4 ? 0 : ROWS * YPITCH * 4));   // first bias [FF]

  const SkfSplitSel sel = skf_split_sel();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
#if SKF_FFN_STAMPS
  long long* dbg = (lane == 0 && (wave == 0 || wave == 7) && (blockIdx.x % 32) == 0 && blockIdx.x / 32 < 8) ? g_ffn_stamps + ((blockIdx.x / 32) * 2 + (wave == 7)) * 64 : nullptr;
  int dbi = 0;
#endif
  FFN_STAMP();
  const int ntiles = (p.M + TR - 1) / TR;
  typedef const __attribute__((address_space(4))) int* const_i32p;
  const const_i32p blk = (const_i32p)p.row_blocks;
  const int nlive = blk ? blk[0] : ntiles;
  auto phys = [&](int pos) -> int { return pos < nlive ? (blk ? blk[2 + pos] : pos) : ntiles; };

  // this workgroup's positions [pos_begin, pos_begin + count)
  const int G = gridDim.x, wg = blockIdx.x;
  const int base = nlive / G, rem = nlive % G;
  const int pos_begin = wg * base + (wg < rem ? wg : rem), count = base + (wg < rem ? 1 : 0);
  const int nsub = (count + MAXRT - 1) / MAXRT;

  // per-lane constants
  unsigned a_off[NKS];                   // fragment read: row i, 16-byte chunk (4s + g) ^ i
#pragma unroll
  for (int s = 0; s < NKS; ++s) a_off[s] = (unsigned)(i * RPITCH + (((4 * s + g) ^ i) << 4));
  // The products are formed TRANSPOSED (weight fragment as the MFMA's first operand): lane (i, g) then holds four CONSECUTIVE
  // columns 16 wave + 4g + r of ONE row i, so a tile's hidden values leave as one 16-byte global store and one 8-byte LDS store
  // per plane (row-major C fragments took four 4-byte global stores and twelve 2-byte LDS stores per tile: 14 of the 70 us)
  const unsigned hw_off = (unsigned)(i * RPITCH + (((2 * wave + (g >> 1)) ^ i) << 4) + 8 * (g & 1));   // hidden plane store
  const unsigned hc_voff = (unsigned)(i * FF + 16 * wave + 4 * g) * 4u;                                 // hidden tensor store
  const int st_row = tid >> 5, st_c = tid & 31;            // staging: thread -> (row of the tile, float4 of the row)
  const unsigned st_off = (unsigned)(st_row * RPITCH + (((st_c >> 1) ^ st_row) << 4) + (st_c & 1) * 8);
  const unsigned st_voff = (unsigned)(st_row * p.lda + 4 * st_c) * 4u;
  if (MODE == 0 && !NOFFN && tid < FF / 4) *reinterpret_cast<f32x4*>(B1s + 4 * tid) = *reinterpret_cast<const f32x4*>(p.bias1 + 4 * tid);   // (read behind the staging barrier)
  float* B3s = B1s + FF;                 // POST: the chained projection's bias [n2 <= 384] (a global load in front of each block's
                                         // first product was a cold miss per block: the chained launch took 34 us for 24 us of projection)
  if (POST && tid >= 128 && tid - 128 < p.n2 / 4) *reinterpret_cast<f32x4*>(B3s + 4 * (tid - 128)) = *reinterpret_cast<const f32x4*>(p.bias3 + 4 * (tid - 128));
  float* PREs = B3s + 384;               // PRE: [bias 128][gamma 128][beta 128] of the leading projection / LayerNorm
  if (PRE && tid >= 256 && tid < 256 + 96) {
    const int j = tid - 256, which = j >> 5, c4 = (j & 31) * 4;
    const float* src = which == 0 ? p.pre_bias : which == 1 ? p.pre_gamma : p.pre_beta;
    *reinterpret_cast<f32x4*>(PREs + which * FD + c4) = *reinterpret_cast<const f32x4*>(src + c4);
  }
  const float* bias1_p = B1s + 16 * wave + 4 * g;
  // (s_setprio 1 for the second-dispatched half, which loses the issue arbitration on its SIMD to the older wave and makes waves
  //  0-3 wait 2-4 k cycles at every block barrier, only swaps the roles: measured with stamps, zero-sum)
  const f32x4 bias2_r = MODE == 0 && !NOFFN ? *reinterpret_cast<const f32x4*>(p.bias2 + 16 * wave + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};

  // epilogue constants: a half-wave owns a row (32 lanes x 4 columns)
  const int e_half = lane >> 5, e_sub = lane & 31;     // row 2 wave + e_half = tid >> 5 = st_row, float4 e_sub = st_c: the staging map
  float inv_keep = 1.f;
  uint32_t thresh = 0u, sk = 0u;
  f32x4 gm = (f32x4){0.f, 0.f, 0.f, 0.f}, bt = gm;
  if constexpr (MODE == 0 || LNB) {
    if (p.rate > 0.f) {
      typedef const __attribute__((address_space(4))) uint32_t* const_u32p;
      const uint32_t key = *(const_u32p)&reinterpret_cast<const SkfStepState*>(p.state)->drop_key;
      sk = skf_site_key(key, p.site);
      thresh = skf_drop_thresh(p.rate);
      inv_keep = 1.0f / (1.0f - p.rate);
    }
    if constexpr (!NOFFN) {
      gm = *reinterpret_cast<const f32x4*>(p.gamma + 4 * e_sub);
      if constexpr (MODE == 0) bt = *reinterpret_cast<const f32x4*>(p.beta + 4 * e_sub);
    }
  }
  f32x4 ln_dg = (f32x4){0.f, 0.f, 0.f, 0.f}, ln_db = ln_dg;     // LNB: this thread's columns of dgamma / dbeta over its rows

  // ONE descriptor per tensor for the whole launch, the tile in the VGPR offset (rows behind M fall outside and are dropped / read
  // as zeros): a descriptor per (tile, block) was ~20 SALU instructions per tile and kept the kernel spilling SGPRs into VGPR lanes
#if SKF_FFN_ABLATE & 4
  const __amdgpu_buffer_rsrc_t r_H = __builtin_amdgcn_make_buffer_rsrc(p.H, 0, 0, 0x00020000);
#else
  const __amdgpu_buffer_rsrc_t r_H = __builtin_amdgcn_make_buffer_rsrc(p.H, 0, p.M * (FF * 4), 0x00020000);
#endif
  const __amdgpu_buffer_rsrc_t r_A = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(LNB ? p.ln_dout : p.A), 0, p.M * p.lda * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_Z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(LNB ? p.ln_z : p.A), 0, LNB ? p.M * (FD * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_S = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(LNB ? p.ln_stats : p.A), 0, LNB ? p.M * 8 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_DY = __builtin_amdgcn_make_buffer_rsrc(LNB ? p.ln_dy : p.H, 0, LNB ? p.M * (FD * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bits = __builtin_amdgcn_make_buffer_rsrc(MODE == 0 && p.bits_out ? (void*)p.bits_out : (void*)p.H, 0,
                                                                          MODE == 0 && p.bits_out ? ntiles * (NBLK * 8 * 32) : 0, 0x00020000);
  const unsigned bits_voff = lane < 4 ? (unsigned)lane * 8u : OOB;
  const __amdgpu_buffer_rsrc_t r_O2 = __builtin_amdgcn_make_buffer_rsrc(POST ? (void*)p.out2 : (void*)p.H, 0, POST ? p.M * p.n2 * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_R = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRE ? p.pre_out : p.A), 0, p.M * (PRE ? FD : p.lda) * 4, 0x00020000);   // residual rows of the main epilogue
  const __amdgpu_buffer_rsrc_t r_PR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRE ? p.pre_res : p.A), 0, PRE ? p.M * (FD * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_imgp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(PRE ? p.pre_img : p.img1), 0, PRE ? P * FD * FD * 2 : 0, 0x00020000);
  uint32_t sk_pre = 0u;
  if constexpr (PRE) {
    if (p.rate > 0.f) {
      typedef const __attribute__((address_space(4))) uint32_t* const_u32p;
      sk_pre = skf_site_key(*(const_u32p)&reinterpret_cast<const SkfStepState*>(p.state)->drop_key, p.pre_site);
    }
  }
  const unsigned lane16 = (unsigned)lane * 16u;
  const __amdgpu_buffer_rsrc_t r_img1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.img1), 0, P * (FD * FF * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_img2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.img2), 0, P * (FD * FF * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_img3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(POST ? p.img3 : p.img1), 0, POST ? P * FD * 2 * p.n2 : 0, 0x00020000);
  const int img1_w = wave * (NKS * P * 1024);            // + b * 8 * NKS * P * 1024
  const int img2_w = wave * (16 * P * 1024);             // + b * NKS * P * 1024
  u32x4 w1[NKS][P], w2[NKS][P];
  if (!NOFFN && nsub > 0) load_frags<P>(w1, r_img1, lane16, img1_w, true);
#if SKF_FFN_ABLATE & 32
  load_frags<P>(w2, r_img2, lane16, img2_w, true);
#endif

  // A rows of a sub-group: thread -> float4 st_c of row st_row of every tile.  Requested one sub-group ahead (the first one here,
  // the next one under the current one's last hidden block).  In the forward they are also the residual of the epilogue, whose
  // (row, float4) -> thread map is the same: re-requested there (L2) rather than held in 16 registers across the blocks.
  f32x4 xn[MAXRT];
  f32x4 zn[LNB ? MAXRT : 1];       // LNB: the z rows and (mean, rstd) of the same rows
  u32x2 sn[LNB ? MAXRT : 1];
  int tn[MAXRT];
  auto sub_tiles = [&](int pos0, int sub_i, int (&t)[MAXRT]) -> int {     // even split of what is left -> tile ids, count
    const int left = count - (pos0 - pos_begin);
    const int n = sub_i < nsub ? (left + (nsub - sub_i) - 1) / (nsub - sub_i) : 0;
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) t[rt] = rt < n ? phys(pos0 + rt) : ntiles;
    return n;
  };
  auto request_rows = [&](const int (&t)[MAXRT], f32x4 (&x)[MAXRT]) {
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) {
      const unsigned voff = t[rt] < ntiles ? st_voff + (unsigned)t[rt] * (unsigned)(TR * p.lda * 4) : OOB;
      x[rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_A, voff, 0, 0));
      if constexpr (LNB) {
        zn[rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_Z, voff, 0, 0));
        sn[rt] = __builtin_amdgcn_raw_buffer_load_b64(r_S, t[rt] < ntiles ? (unsigned)(t[rt] * TR + st_row) * 8u : OOB, 0, 0);
      }
    }
  };
  int pos = pos_begin;
  int nrt_next = sub_tiles(pos, 0, tn);
  request_rows(tn, xn);
  for (int sub = 0; sub < nsub; ++sub) {
    const int nrt = nrt_next;
    int tl[MAXRT];
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) tl[rt] = tn[rt];
    pos += nrt;

    // One code path for 1..4 tiles: the tile bodies are scheduling regions of their own anyway (see the sched_barriers), absent tiles
    // are skipped by wave-uniform branches.  (Four instantiations over the tile count behind a switch made the register allocator
    // spill 231 registers around the sub-group; each alone needs 238 and spills nothing.)
    {
    constexpr int NRT = MAXRT;
    constexpr int TILE = TR * RPITCH;
    // ---- stage the A rows: registers -> P planes in LDS
    f32x4 dzr[LNB ? NRT : 1];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      if constexpr (LNB) {
        // LayerNorm backward of this thread's float4 of row (tile rt, st_row); rows outside the matrix / absent tiles read as zeros
        const f32x4 dv = xn[rt], zv = zn[rt];
        // (whole-vector cast: __builtin_bit_cast on ONE element of an ext_vector yields element 0 with this compiler)
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t ms = __builtin_bit_cast(f32x2_t, sn[rt]);
        const float mean = ms[0], rstd = ms[1];
        const f32x4 xh = (zv - mean) * rstd, gg = dv * gm;
        ln_dg += dv * xh; ln_db += dv;
        const float s1 = ffn_half_wave_sum((gg[0] + gg[1]) + (gg[2] + gg[3])) * (1.0f / FD);
        const float s2 = ffn_half_wave_sum((gg[0] * xh[0] + gg[1] * xh[1]) + (gg[2] * xh[2] + gg[3] * xh[3])) * (1.0f / FD);
        const f32x4 gz = rstd * (gg - s1 - xh * s2);
        dzr[rt] = gz;
        f32x4 gy = gz;
        const unsigned voff = tl[rt] < ntiles ? st_voff + (unsigned)tl[rt] * (unsigned)(TR * FD * 4) : OOB;
        if (p.rate > 0.f) {
#pragma unroll
          for (int e = 0; e < 4; ++e) gy[e] = gz[e] * (skf_keep(sk, (voff >> 2) + e, thresh) ? inv_keep : 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gy), r_DY, voff, 0, 0);
        xn[rt] = gy;
      }
      unsigned lo[P], hi[P];
      skf_split2<P>(xn[rt][0], xn[rt][1], lo, sel);
      skf_split2<P>(xn[rt][2], xn[rt][3], hi, sel);
#pragma unroll
      for (int q = 0; q < P; ++q) *reinterpret_cast<u32x2*>(Xp + q * PLANE + rt * TILE + st_off) = (u32x2){lo[q], hi[q]};
    }
    FFN_STAMP();   // rows staged
    __syncthreads();
    FFN_STAMP();   // behind the staging barrier
    FfnFrags<P> fr;
    // z = residual + dropout(y), out = LayerNorm(z) for this thread's float4 of row (tile rt, 2 wave + e_half), exactly as
    // ln_fwd_v4_kernel; optionally the planes of `out` for the next product stage (same thread map as the staging)
    auto ln_row = [&](int rt, const f32x4& yv, const f32x4& xv, uint32_t skey, const f32x4& gmv, const f32x4& btv, float* zp, float* op,
                      float* sp, bool planes) {
      const int grow = tl[rt] * TR + 2 * wave + e_half;
      const bool ok = grow < p.M;
      const size_t off = (size_t)(ok ? grow : 0) * FD + 4 * e_sub;
      f32x4 z;
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float yy = yv[e];
        if (p.rate > 0.f) yy *= skf_keep(skey, (uint32_t)off + e, thresh) ? inv_keep : 0.f;
        z[e] = xv[e] + yy;
        sum += z[e];
      }
      const float mean = ffn_half_wave_sum(sum) * (1.0f / FD);
      float sq = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float c = z[e] - mean; sq += c * c; }
      const float rstd = rsqrtf(ffn_half_wave_sum(sq) * (1.0f / FD) + 1e-6f);
      const f32x4 o = (z - mean) * rstd * gmv + btv;
      if (ok) {
        *reinterpret_cast<f32x4*>(zp + off) = z;
        *reinterpret_cast<f32x4*>(op + off) = o;
        if (e_sub == 0) { sp[2 * (size_t)grow] = mean; sp[2 * (size_t)grow + 1] = rstd; }
      }
      if (planes) {
        unsigned lo[P], hi[P];
        skf_split2<P>(o[0], o[1], lo, sel);
        skf_split2<P>(o[2], o[3], hi, sel);
#pragma unroll
        for (int q = 0; q < P; ++q) *reinterpret_cast<u32x2*>(Xp + q * PLANE + rt * TILE + st_off) = (u32x2){lo[q], hi[q]};
      }
    };
    if constexpr (PRE) {
      // ---- the leading projection: y = a . Wo + bo for the staged rows (wave w: 16 of the 128 columns; operands in w2, free until
      // the first hidden block requests its second-stage operands), through the Y tile, then the row epilogue -> z1, x1, statistics
      // and the planes of x1 where the staged rows were
      load_frags<P>(w2, r_imgp, lane16, wave * (NKS * P * 1024));
      f32x4 xres[NRT];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
        xres[rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_PR, tl[rt] < ntiles ? st_voff + (unsigned)tl[rt] * (unsigned)(TR * FD * 4) : OOB, 0, 0));
      const f32x4 bp = *reinterpret_cast<const f32x4*>(PREs + 16 * wave + 4 * g);
      load_half<P>(Xp, a_off, fr.f[0], 0);
#pragma unroll
      for (int t = 0; t < NRT; ++t) {
        if (t < nrt) {
          load_half<P>(Xp + t * TILE, a_off, fr.f[1], 1);
          f32x4 acc0 = bp, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
          int c = 0;
          half_products<P>(w2, fr.f[0], 0, acc0, acc1, c);
          if (t + 1 < NRT) load_half<P>(Xp + (t + 1) * TILE, a_off, fr.f[0], 0);
          half_products<P>(w2, fr.f[1], 1, acc0, acc1, c);
          *reinterpret_cast<f32x4*>(Yt + (t * TR + i) * YPITCH + 16 * wave + 4 * g) = acc0 + acc1;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
      const f32x4 gmp = *reinterpret_cast<const f32x4*>(PREs + FD + 4 * e_sub), btp = *reinterpret_cast<const f32x4*>(PREs + 2 * FD + 4 * e_sub);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(Yt + (rt * TR + 2 * wave + e_half) * YPITCH + 4 * e_sub);
        ln_row(rt, yv, xres[rt], sk_pre, gmp, btp, p.pre_z, p.pre_out, p.pre_stats, true);
      }
      __syncthreads();               // the planes of x1 are complete (and the Y tile free for the first hidden block)
    }

    f32x4 y[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) y[rt] = bias2_r;
#if SKF_FFN_ABLATE & 2
    load_half<P>(Xp, a_off, fr.f[0], 0, true); load_half<P>(Xp, a_off, fr.f[1], 1, true);
#endif

#pragma unroll 1
    for (int b = 0; b < (NOFFN ? 0 : NBLK); ++b) {
      char* Hb = Hp + (b & 1) * (P * PLANE);
      const f32x4 b1 = MODE == 0 ? *reinterpret_cast<const f32x4*>(bias1_p + HB * b) : (f32x4){0.f, 0.f, 0.f, 0.f};   // (LDS)
      if (b == NBLK - 1) {                      // the next sub-group's rows (all four descriptors: absent tiles read as zeros)
        nrt_next = sub_tiles(pos, sub + 1, tn);
        request_rows(tn, xn);
      }
      // the stage-1 epilogue of a tile: activation / mask, sign bits, the hidden rows to global memory and, split, to the LDS planes
      auto hidden_out = [&](int rt, const f32x4& acc0, const f32x4& acc1) {
#if SKF_FFN_ABLATE & 16
        if (p.M >= 0) { if (acc0[0] + acc1[0] == 1.2345f) Hb[0] = 1; return; }
#endif
        f32x4 h = acc0 + acc1;
        const int tile = tl[rt];
        if constexpr (MODE == 0) {
          // sign bits: word r = ballot(h[r] > 0) goes to lane r with v_writelane (a compare-and-select per word and half cost
          // 16 VALU per tile), lanes 0-3 store (the others sit outside the descriptor)
          unsigned wlo = 0u, whi = 0u;
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = __builtin_amdgcn_fmed3f(h[r], 0.f, __builtin_inff());
          const unsigned long long m0 = __ballot(h[0] > 0.f), m1 = __ballot(h[1] > 0.f), m2 = __ballot(h[2] > 0.f), m3 = __ballot(h[3] > 0.f);
          // (s_nop 1: a VALU-written SGPR needs two wait states before v_writelane reads it - the compiler inserts them for its own
          //  v_writelane, it cannot see into inline assembly; without them a few words per launch came out wrong)
#define FFN_WRITELANE(M, L) asm("s_nop 1\n\tv_writelane_b32 %0, %2, " #L "\n\tv_writelane_b32 %1, %3, " #L : "+v"(wlo), "+v"(whi) : "s"((unsigned)(M)), "s"((unsigned)((M) >> 32)))
          FFN_WRITELANE(m0, 0); FFN_WRITELANE(m1, 1); FFN_WRITELANE(m2, 2); FFN_WRITELANE(m3, 3);
#undef FFN_WRITELANE
          __builtin_amdgcn_raw_buffer_store_b64((u32x2){wlo, whi}, r_bits, bits_voff, ((tile * NBLK + b) * 8 + wave) * 32, 0);
        } else {
          typedef const __attribute__((address_space(4))) unsigned long long* const_u64p;
          const const_u64p wp = (const_u64p)(p.bits_in + (((size_t)tile * NBLK + b) * 8 + wave) * 4);
          f32x4 hm;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned long long m = wp[r];
            float v;
            asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(v) : "v"(h[r]), "s"(m));
            hm[r] = v;
          }
          h = hm;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), r_H, hc_voff + (unsigned)tile * (TR * FF * 4), b * (HB * 4), 0);
        unsigned p01[P], p23[P];
        skf_split2<P>(h[0], h[1], p01, sel);
        skf_split2<P>(h[2], h[3], p23, sel);
        char* hrow = Hb + rt * TILE;
#if SKF_FFN_ABLATE & 8
        if (p.M < 0)
#endif
#pragma unroll
        for (int q = 0; q < P; ++q) *reinterpret_cast<u32x2*>(hrow + q * PLANE + hw_off) = (u32x2){p01[q], p23[q]};
      };
      // ---- stage 1: hidden block b of every row tile; the epilogue of tile t - 1 sits between the halves of tile t
      load_half<P>(Xp, a_off, fr.f[0], 0);
      f32x4 pacc0 = b1, pacc1 = b1;
#pragma unroll
      for (int t = 0; t < NRT; ++t) {
        if (t < nrt) {
          load_half<P>(Xp + t * TILE, a_off, fr.f[1], 1);
          f32x4 acc0 = b1, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
          int c = 0;
          half_products<P>(w1, fr.f[0], 0, acc0, acc1, c);
          if (t == 0) {
            // the second stage's operands: behind the block's first products, so that the wait in front of those (vmcnt(0) at a loop
            // head) covers only the first-stage operands requested a whole stage ago, not these
            __builtin_amdgcn_sched_barrier(0);
            load_frags<P>(w2, r_img2, lane16, img2_w + b * (NKS * P * 1024));
            __builtin_amdgcn_sched_barrier(0);
          }
          if (t > 0) hidden_out(t - 1, pacc0, pacc1);
          if (t + 1 < NRT) load_half<P>(Xp + (t + 1) * TILE, a_off, fr.f[0], 0);
          half_products<P>(w1, fr.f[1], 1, acc0, acc1, c);
          pacc0 = acc0; pacc1 = acc1;
          __builtin_amdgcn_sched_barrier(0);      // (no fragment reads hoisted across tiles)
        }
      }
      // the last tile's epilogue (nrt - 1 is wave-uniform: a short chain of uniform branches instead of a dynamic register index)
#pragma unroll
      for (int t = 0; t < NRT; ++t)
        if (t == nrt - 1) hidden_out(t, pacc0, pacc1);
      // the next first-stage operands (block 0 again behind the last block: the next sub-group starts with them)
      load_frags<P>(w1, r_img1, lane16, img1_w + ((b + 1) & (NBLK - 1)) * (8 * NKS * P * 1024));
      FFN_STAMP();   // stage 1 issued
#if !(SKF_FFN_ABLATE & 64)
      __syncthreads();
#endif
      FFN_STAMP();   // behind the block barrier
      // ---- stage 2: Y += Hblock . B2[block rows]
      load_half<P>(Hb, a_off, fr.f[0], 0);
#pragma unroll
      for (int t = 0; t < NRT; ++t) {
        if (t < nrt) {
          load_half<P>(Hb + t * TILE, a_off, fr.f[1], 1);
          int c = 0;
          f32x4 ya = (f32x4){0.f, 0.f, 0.f, 0.f}, yb = ya;      // this block's two chains; one persistent accumulator per tile
          half_products<P>(w2, fr.f[0], 0, ya, yb, c);
          if (t + 1 < NRT) load_half<P>(Hb + (t + 1) * TILE, a_off, fr.f[0], 0);
          half_products<P>(w2, fr.f[1], 1, ya, yb, c);
          y[t] += ya + yb;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      FFN_STAMP();   // stage 2 issued
    }

    if constexpr (NOFFN) {                      // (no hidden block to hide them under: the next sub-group's rows are requested here)
      nrt_next = sub_tiles(pos, sub + 1, tn);
      request_rows(tn, xn);
    }
    // ---- epilogue: Y through LDS, a half-wave per row
    if constexpr (POST) load_frags<P>(w2, r_img3, lane16, wave * (NKS * P * 1024));   // the chained projection's first operands
    if constexpr (!NOFFN) {
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) *reinterpret_cast<f32x4*>(Yt + (rt * TR + i) * YPITCH + 16 * wave + 4 * g) = y[rt];
    f32x4 xres[NRT];
    if constexpr (MODE == 0) {
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
        xres[rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_R, tl[rt] < ntiles ? st_voff + (unsigned)tl[rt] * (unsigned)(TR * (PRE ? FD : p.lda) * 4) : OOB, 0, 0));
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      const int rloc = rt * TR + 2 * wave + e_half;
      const int grow = tl[rt] * TR + 2 * wave + e_half;
      const bool ok = grow < p.M;
      const size_t off = (size_t)(ok ? grow : 0) * FD + 4 * e_sub;
      const f32x4 yv = *reinterpret_cast<const f32x4*>(Yt + rloc * YPITCH + 4 * e_sub);
      if constexpr (MODE == 0) {
        // (the residual is the block's input row: the row this thread staged, or with PRE the x1 row it wrote above)
        ln_row(rt, yv, xres[rt], sk, gm, bt, p.C, p.out, p.stats, POST);
      } else {
        if (ok) {
          f32x4 v = yv;
          if constexpr (LNB) v += dzr[rt];
          else if (p.accumulate) v += *reinterpret_cast<const f32x4*>(p.C + off);
          *reinterpret_cast<f32x4*>(p.C + off) = v;
        }
      }
    }
    }   // !NOFFN
    if constexpr (POST) {
      // ---- the chained projection: out2[rows][nb * 128 + 16 wave + 4g ..] = out . B3 + bias3, 128 output columns at a time.
      // Operand registers: w2 for even blocks (free since the last hidden block), w1 for odd ones (it holds the next sub-group's
      // first operands: re-requested behind the last block).
      const int nb2 = p.n2 >> 7;
      const int img3_w = wave * (NKS * P * 1024);           // + nb * 8 * NKS * P * 1024
      const unsigned o_voff = (unsigned)(i * p.n2 + 16 * wave + 4 * g) * 4u;
      auto post_block = [&](const u32x4 (&w)[NKS][P], int nb) {
        const f32x4 b3 = *reinterpret_cast<const f32x4*>(B3s + nb * 128 + 16 * wave + 4 * g);
        load_half<P>(Xp, a_off, fr.f[0], 0);
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
          if (t < nrt) {
            load_half<P>(Xp + t * TILE, a_off, fr.f[1], 1);
            f32x4 acc0 = b3, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
            int c = 0;
            half_products<P>(w, fr.f[0], 0, acc0, acc1, c);
            if (t + 1 < NRT) load_half<P>(Xp + (t + 1) * TILE, a_off, fr.f[0], 0);
            half_products<P>(w, fr.f[1], 1, acc0, acc1, c);
            const f32x4 r = acc0 + acc1;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r), r_O2, o_voff + (unsigned)tl[t] * (unsigned)(TR * 4) * (unsigned)p.n2, nb * 512, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      FFN_STAMP();   // (POST) row epilogue done
      __syncthreads();                                   // the planes of `out` are complete
      FFN_STAMP();   // (POST) behind the planes barrier
      if (nb2 > 1) load_frags<P>(w1, r_img3, lane16, img3_w + 1 * (8 * NKS * P * 1024));
      post_block(w2, 0);
      FFN_STAMP();   // (POST) block 0
      if (nb2 > 1) {
        if (nb2 > 2) load_frags<P>(w2, r_img3, lane16, img3_w + 2 * (8 * NKS * P * 1024));
        post_block(w1, 1);
        FFN_STAMP();   // (POST) block 1
        if (nb2 > 2) post_block(w2, 2);
        FFN_STAMP();   // (POST) block 2
        if constexpr (!NOFFN) load_frags<P>(w1, r_img1, lane16, img1_w);        // the next sub-group's first operands again
      }
      __syncthreads();                                   // the next sub-group's staging overwrites the planes
    }
    FFN_STAMP();   // epilogue done
    }
    // (the next sub-group's staging writes Xp, which nobody reads behind the last block's barrier; its first stage writes Hp
    //  buffer 0 = Yt behind the staging barrier, which every wave reaches after its epilogue reads)
  }

  // rows of dead tiles: dh = 0 (the weight gradient's 32-row blocks may contain a dead 16-row tile); dx is left alone when
  // accumulating and zero otherwise
  if (MODE == 1 && blk) {
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int dpos = nlive + wg; dpos < ntiles; dpos += G) {
      const int tile = blk[2 + dpos];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int e = tid + v * 512, row = tile * TR + e / (FF / 4), c4 = (e % (FF / 4)) * 4;
        if (row < p.M) *reinterpret_cast<f32x4*>(p.H + (size_t)row * FF + c4) = zero;
      }
      if (LNB || !p.accumulate) {
        const int row = tile * TR + (tid >> 5);
        if (row < p.M) {
          *reinterpret_cast<f32x4*>(p.C + (size_t)row * FD + 4 * (tid & 31)) = zero;
          if constexpr (LNB) *reinterpret_cast<f32x4*>(p.ln_dy + (size_t)row * FD + 4 * (tid & 31)) = zero;
        }
      }
    }
  }
  if constexpr (LNB) {
    // dgamma / dbeta partials of this workgroup: the two rows of a wave, then the eight waves through LDS -> part[wg][2][FD]
    // (every workgroup writes its pair, also one without tiles: the batched reduction reads gridDim.x of them)
#pragma unroll
    for (int e = 0; e < 4; ++e) { ln_dg[e] += __shfl_xor(ln_dg[e], 32, 64); ln_db[e] += __shfl_xor(ln_db[e], 32, 64); }
    float* red = reinterpret_cast<float*>(smem_f);       // [8 waves][2][FD]
    __syncthreads();
    if (lane < 32) {
      *reinterpret_cast<f32x4*>(red + (wave * 2 + 0) * FD + 4 * lane) = ln_dg;
      *reinterpret_cast<f32x4*>(red + (wave * 2 + 1) * FD + 4 * lane) = ln_db;
    }
    __syncthreads();
    if (tid < 2 * FD) {
      float t = red[tid];
#pragma unroll
      for (int k = 1; k < 8; ++k) t += red[k * 2 * FD + tid];
      p.ln_part[(size_t)wg * 2 * FD + tid] = t;
    }
  }
}

template <int P, int MODE, bool LNB = false, bool POST = false, bool PRE = false>
__global__ __launch_bounds__(512, 2) void ffn_fused_kernel(FfnFusedParams p) { ffn_fused_body<P, MODE, LNB, POST, PRE, false>(p); }
template <int P>
__global__ __launch_bounds__(512, 2) void attn_tail_proj_kernel(FfnFusedParams p) { ffn_fused_body<P, 0, false, true, true, true>(p); }

// ---------------------------------------------------------------- LayerNorm backward + the input gradient of the Dense in front of it
// The attention sublayers' counterpart of the LNB prologue above: out = LayerNorm(x + dropout(Dense(a))) (the MultiHeadAttention
// output projection, builders/layers/transformer.py:186, 221-224).  One launch forms dz = LayerNorm'(dout) (written: the residual
// path), dy = dropout'(dz) (written: the weight gradient reads it) and da = dy . W^T (K = N = 128: a wave keeps its 16 columns of
// the pre-split W^T image in registers for the whole launch) - it replaces the LayerNorm-backward launch and a weight-stationary
// GEMM launch that each moved the same rows (12 + 12 launches per step at cfg 2).
struct LnDgradParams {
  const float* dout; const float* z; const float* stats; const float* gamma;
  float* dz; float* dy; float* C;
  const char* img;
  float* part;
  int M; float rate; unsigned site; const void* state;
  const int* row_blocks;
  // LEAD: the gradient of the LayerNorm output is dout + lead_a . Wl^T (lead_img = the transposed image of Wl [128][128]): the input
  // gradient of a Dense that consumes the LayerNorm output (the cross-attention query projection behind the decoder's self-attention
  // sublayer, builders/layers/transformer.py:258-262) formed in the same launch instead of accumulated into dout by one of its own
  const float* lead_a; const char* lead_img;
};

template <int P, bool LEAD>
__global__ __launch_bounds__(512, 2) void ln_bwd_dgrad_kernel(LnDgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_f[];
  char* Xp = smem_f;                     // [P][ROWS][256]
  float* Yt = reinterpret_cast<float*>(smem_f + P * PLANE);     // LEAD: the leading product's rows [ROWS][YPITCH]
  constexpr int TILE = TR * RPITCH;
  const SkfSplitSel sel = skf_split_sel();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int ntiles = (p.M + TR - 1) / TR;
  typedef const __attribute__((address_space(4))) int* const_i32p;
  const const_i32p blk = (const_i32p)p.row_blocks;
  const int nlive = blk ? blk[0] : ntiles;
  auto phys = [&](int pos) -> int { return pos < nlive ? (blk ? blk[2 + pos] : pos) : ntiles; };
  const int G = gridDim.x, wg = blockIdx.x;
  const int base = nlive / G, rem = nlive % G;
  const int pos_begin = wg * base + (wg < rem ? wg : rem), count = base + (wg < rem ? 1 : 0);
  const int nsub = (count + MAXRT - 1) / MAXRT;

  unsigned a_off[NKS];
#pragma unroll
  for (int s = 0; s < NKS; ++s) a_off[s] = (unsigned)(i * RPITCH + (((4 * s + g) ^ i) << 4));
  const unsigned c_voff = (unsigned)(i * FD + 16 * wave + 4 * g) * 4u;          // transposed products: row i, columns 16 wave + 4g ..
  const int st_row = tid >> 5, st_c = tid & 31;
  const unsigned st_off = (unsigned)(st_row * RPITCH + (((st_c >> 1) ^ st_row) << 4) + (st_c & 1) * 8);
  const unsigned st_voff = (unsigned)(st_row * FD + 4 * st_c) * 4u;
  float inv_keep = 1.f;
  uint32_t thresh = 0u, sk = 0u;
  if (p.rate > 0.f) {
    typedef const __attribute__((address_space(4))) uint32_t* const_u32p;
    const uint32_t key = *(const_u32p)&reinterpret_cast<const SkfStepState*>(p.state)->drop_key;
    sk = skf_site_key(key, p.site);
    thresh = skf_drop_thresh(p.rate);
    inv_keep = 1.0f / (1.0f - p.rate);
  }
  const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + 4 * st_c);
  f32x4 ln_dg = (f32x4){0.f, 0.f, 0.f, 0.f}, ln_db = ln_dg;
  const unsigned row_bytes = (unsigned)p.M * (FD * 4);
  const __amdgpu_buffer_rsrc_t r_D = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dout), 0, row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_Z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.z), 0, row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_S = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.stats), 0, (unsigned)p.M * 8u, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_DZ = __builtin_amdgcn_make_buffer_rsrc(p.dz, 0, row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_DY = __builtin_amdgcn_make_buffer_rsrc(p.dy, 0, row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_C = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_L = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(LEAD ? p.lead_a : p.dout), 0, LEAD ? row_bytes : 0, 0x00020000);

  u32x4 w[NKS][P];
  if (nsub > 0) load_frags<P>(w, __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.img), 0, P * FD * FD * 2, 0x00020000), (unsigned)lane * 16u, wave * (NKS * P * 1024));
  u32x4 wl[LEAD ? NKS : 1][P];
  if constexpr (LEAD) {
    if (nsub > 0) load_frags<P>(wl, __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.lead_img), 0, P * FD * FD * 2, 0x00020000), (unsigned)lane * 16u, wave * (NKS * P * 1024));
  }

  f32x4 xn[MAXRT], zn[MAXRT];
  f32x4 an[LEAD ? MAXRT : 1];
  u32x2 sn[MAXRT];
  int tn[MAXRT];
  auto sub_tiles = [&](int pos0, int sub_i, int (&t)[MAXRT]) -> int {
    const int left = count - (pos0 - pos_begin);
    const int n = sub_i < nsub ? (left + (nsub - sub_i) - 1) / (nsub - sub_i) : 0;
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) t[rt] = rt < n ? phys(pos0 + rt) : ntiles;
    return n;
  };
  auto request_rows = [&]() {
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) {
      const unsigned voff = tn[rt] < ntiles ? st_voff + (unsigned)tn[rt] * (unsigned)(TR * FD * 4) : OOB;
      if constexpr (LEAD) an[rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_L, voff, 0, 0));
      xn[rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_D, voff, 0, 0));
      zn[rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_Z, voff, 0, 0));
      sn[rt] = __builtin_amdgcn_raw_buffer_load_b64(r_S, tn[rt] < ntiles ? (unsigned)(tn[rt] * TR + st_row) * 8u : OOB, 0, 0);
    }
  };
  int pos = pos_begin;
  int nrt_next = sub_tiles(pos, 0, tn);
  request_rows();
  for (int sub = 0; sub < nsub; ++sub) {
    const int nrt = nrt_next;
    int tl[MAXRT];
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) tl[rt] = tn[rt];
    pos += nrt;
    if constexpr (LEAD) {
      // ---- the leading product: rows of lead_a -> planes -> y = lead_a . Wl^T (a wave's 16 columns) -> the Y tile
#pragma unroll
      for (int rt = 0; rt < MAXRT; ++rt) {
        unsigned lo[P], hi[P];
        skf_split2<P>(an[rt][0], an[rt][1], lo, sel);
        skf_split2<P>(an[rt][2], an[rt][3], hi, sel);
#pragma unroll
        for (int q = 0; q < P; ++q) *reinterpret_cast<u32x2*>(Xp + q * PLANE + rt * TILE + st_off) = (u32x2){lo[q], hi[q]};
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < MAXRT; ++t) {
        if (t < nrt) {
          FfnFrags<P> fr;
          load_half<P>(Xp + t * TILE, a_off, fr.f[0], 0);
          load_half<P>(Xp + t * TILE, a_off, fr.f[1], 1);
          f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
          int c = 0;
          half_products<P>(wl, fr.f[0], 0, acc0, acc1, c);
          half_products<P>(wl, fr.f[1], 1, acc0, acc1, c);
          *reinterpret_cast<f32x4*>(Yt + (t * TR + i) * YPITCH + 16 * wave + 4 * g) = acc0 + acc1;
        }
      }
      __syncthreads();                          // the Y tile is complete, the planes are free for dy
    }
    // ---- LayerNorm backward of this thread's float4 of row (tile rt, st_row) (ln_bwd_v4_kernel's arithmetic) on the way to LDS
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) {
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      const f32x2_t ms = __builtin_bit_cast(f32x2_t, sn[rt]);
      f32x4 dv = xn[rt];
      if constexpr (LEAD) { if (rt < nrt) dv += *reinterpret_cast<const f32x4*>(Yt + (rt * TR + st_row) * YPITCH + 4 * st_c); }
      const f32x4 zv = zn[rt];
      const f32x4 xh = (zv - ms[0]) * ms[1], gg = dv * gm;
      ln_dg += dv * xh; ln_db += dv;
      const float s1 = ffn_half_wave_sum((gg[0] + gg[1]) + (gg[2] + gg[3])) * (1.0f / FD);
      const float s2 = ffn_half_wave_sum((gg[0] * xh[0] + gg[1] * xh[1]) + (gg[2] * xh[2] + gg[3] * xh[3])) * (1.0f / FD);
      const f32x4 gz = ms[1] * (gg - s1 - xh * s2);
      f32x4 gy = gz;
      const unsigned voff = tl[rt] < ntiles ? st_voff + (unsigned)tl[rt] * (unsigned)(TR * FD * 4) : OOB;
      if (p.rate > 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gy[e] = gz[e] * (skf_keep(sk, (voff >> 2) + e, thresh) ? inv_keep : 0.f);
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gz), r_DZ, voff, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gy), r_DY, voff, 0, 0);
      unsigned lo[P], hi[P];
      skf_split2<P>(gy[0], gy[1], lo, sel);
      skf_split2<P>(gy[2], gy[3], hi, sel);
#pragma unroll
      for (int q = 0; q < P; ++q) *reinterpret_cast<u32x2*>(Xp + q * PLANE + rt * TILE + st_off) = (u32x2){lo[q], hi[q]};
    }
    nrt_next = sub_tiles(pos, sub + 1, tn);       // the next sub-group's rows fly under the products
    request_rows();
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MAXRT; ++t) {
      if (t < nrt) {
        FfnFrags<P> fr;
        load_half<P>(Xp + t * TILE, a_off, fr.f[0], 0);
        load_half<P>(Xp + t * TILE, a_off, fr.f[1], 1);
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        int c = 0;
        half_products<P>(w, fr.f[0], 0, acc0, acc1, c);
        half_products<P>(w, fr.f[1], 1, acc0, acc1, c);
        const f32x4 r = acc0 + acc1;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r), r_C, c_voff + (unsigned)tl[t] * (TR * FD * 4), 0, 0);
      }
    }
    __syncthreads();                            // the planes are overwritten by the next sub-group's staging
  }
  if (blk) {                                    // rows of dead tiles: dout == 0 there -> dz = dy = da = 0
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int dpos = nlive + wg; dpos < ntiles; dpos += G) {
      const int row = blk[2 + dpos] * TR + st_row;
      if (row < p.M) {
        const size_t off = (size_t)row * FD + 4 * st_c;
        *reinterpret_cast<f32x4*>(p.dz + off) = zero;
        *reinterpret_cast<f32x4*>(p.dy + off) = zero;
        *reinterpret_cast<f32x4*>(p.C + off) = zero;
      }
    }
  }
  // dgamma / dbeta partials -> part[wg][2][FD]
#pragma unroll
  for (int e = 0; e < 4; ++e) { ln_dg[e] += __shfl_xor(ln_dg[e], 32, 64); ln_db[e] += __shfl_xor(ln_db[e], 32, 64); }
  float* red = reinterpret_cast<float*>(smem_f);
  __syncthreads();
  if (lane < 32) {
    *reinterpret_cast<f32x4*>(red + (wave * 2 + 0) * FD + 4 * lane) = ln_dg;
    *reinterpret_cast<f32x4*>(red + (wave * 2 + 1) * FD + 4 * lane) = ln_db;
  }
  __syncthreads();
  if (tid < 2 * FD) {
    float t = red[tid];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k * 2 * FD + tid];
    p.part[(size_t)wg * 2 * FD + tid] = t;
  }
}

// CU count of the CURRENT device (one process may drive several GPUs; cached per device id, read-mostly: a benign race writes the same value)
int n_cus() {
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!cache[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cache[dev] = n;
  }
  return cache[dev];
}

// One workgroup per CU - minus one: these launches need a whole CU per workgroup (LDS), so a single-workgroup kernel of the side
// stream that holds one CU (the embedding-gradient sort: 56 us) made the 256th workgroup, and with it the launch, wait for it
// (first forward block 111 instead of 82 us).  1600 tiles over 255 workgroups are still at most 7 per workgroup.
int ffn_grid(int M) {
  const int ntiles = skf_cdiv(M, TR);
  int g = n_cus();
  if (g > 64) g -= 1;
  return g > ntiles ? ntiles : g;
}

template <int P, int MODE, bool LNB = false, bool POST = false, bool PRE = false>
int launch_ffn(const FfnFusedParams& p, hipStream_t st) {
  const int grid = ffn_grid(p.M);
  const size_t smem = (size_t)3 * P * PLANE + (P * PLANE >= ROWS * YPITCH * 4 ? 0 : ROWS * YPITCH * 4) + (FF + 384 + 3 * FD) * sizeof(float);
  // per launch like skf_attention.hip's set_smem: the attribute belongs to the (function, device) pair and its failure must surface here
  SKF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_fused_kernel<P, MODE, LNB, POST, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  // one profiler line for the family (the kernel template's forward / backward / LayerNorm-prologue / chained-projection variants,
  // like the epilogue kinds of gemm_wsx); SKF_PROF_FINE=1 (measurement builds): one line per variant
  static const bool fine = skf_knob("SKF_PROF_FINE") && skf_knob("SKF_PROF_FINE")[0] == '1';
  static const std::string tag = std::string("ffn_fused") + (!fine ? "" : MODE == 0 ? (PRE ? (POST ? "_fwd_pre_proj" : "_fwd_pre") : POST ? "_fwd_proj" : "_fwd") : LNB ? "_bwd_ln" : "_bwd") +
                                 "<d128,dff512,bf16x" + std::to_string(P * (P + 1) / 2) + ">";
  const double live = skf_prof_list_fraction(p.row_blocks);
  const double flops = 2.0 * 2.0 * p.M * FD * FF + (POST ? 2.0 * p.M * FD * p.n2 : 0.0) + (PRE ? 2.0 * p.M * FD * FD : 0.0);
  const double bytes = 4.0 * ((double)p.M * FD * (MODE == 0 ? 4 : 3) + (double)p.M * FF + (POST ? (double)p.M * p.n2 : 0.0) + (PRE ? 4.0 * p.M * FD : 0.0)) + 2.0 * image_bytes(P) / 2;
  SkfProfScope ps(st, tag.c_str(), flops, bytes);
  ps.done(flops * live, bytes * live);
  SKF_LAUNCH_TAIL((ffn_fused_kernel<P, MODE, LNB, POST, PRE>), dim3(grid), dim3(512), smem, st, p);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

// the self-attention tail with the next projection (NOFFN): a = p.A, x1 = LayerNorm(pre_res + dropout(a . Bp + pre_bias)), out2 = x1 . B3 + bias3
template <int P>
int launch_tail_proj(const FfnFusedParams& p, hipStream_t st) {
  const int grid = ffn_grid(p.M);
  const size_t smem = (size_t)3 * P * PLANE + (P * PLANE >= ROWS * YPITCH * 4 ? 0 : ROWS * YPITCH * 4) + (FF + 384 + 3 * FD) * sizeof(float);
  SKF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_tail_proj_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  static const std::string tag = std::string("attn_tail_proj<d128,bf16x") + std::to_string(P * (P + 1) / 2) + ">";
  const double flops = 2.0 * p.M * FD * FD + 2.0 * p.M * FD * p.n2;
  const double bytes = 4.0 * ((double)p.M * FD * 4 + (double)p.M * p.n2) + 2.0 * P * FD * (FD + p.n2);
  SkfProfScope ps(st, tag.c_str(), flops, bytes);
  ps.done(flops, bytes);
  hipLaunchKernelGGL((attn_tail_proj_kernel<P>), dim3(grid), dim3(512), smem, st, p);
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

}  // namespace

int skf_ffn_fused_launch(const FfnFusedParams& p, int pieces, int direction, hipStream_t st) {
  if (direction == 0 && p.pre_img && p.img3 && !p.img1) return pieces == 2 ? launch_tail_proj<2>(p, st) : launch_tail_proj<3>(p, st);
  if (direction == 1 && p.ln_dout) return pieces == 2 ? launch_ffn<2, 1, true>(p, st) : launch_ffn<3, 1, true>(p, st);
  if (direction == 0 && p.pre_img && p.img3) return pieces == 2 ? launch_ffn<2, 0, false, true, true>(p, st) : launch_ffn<3, 0, false, true, true>(p, st);
  if (direction == 0 && p.pre_img) return pieces == 2 ? launch_ffn<2, 0, false, false, true>(p, st) : launch_ffn<3, 0, false, false, true>(p, st);
  if (direction == 0 && p.img3) return pieces == 2 ? launch_ffn<2, 0, false, true>(p, st) : launch_ffn<3, 0, false, true>(p, st);
  if (pieces == 2) return direction == 0 ? launch_ffn<2, 0>(p, st) : launch_ffn<2, 1>(p, st);
  return direction == 0 ? launch_ffn<3, 0>(p, st) : launch_ffn<3, 1>(p, st);
}

#if SKF_FFN_STAMPS
extern "C" int skf_ffn_debug_stamps(long long* out_host) {   // diagnostics builds only
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_ffn_stamps), sizeof(long long) * 16 * 64) == hipSuccess ? 0 : 1;
}
#endif
// ---------------------------------------------------------------- C ABI
extern "C" int skf_ffn_fused_supported(int M, int d, int dff, int precision) {
  return (precision == SKF_PREC_BF16X6 || precision == SKF_PREC_BF16X3) && d == FD && dff == FF && M >= 1 && (double)M * FF * 4 < 2147483648.0;
}
extern "C" size_t skf_ffn_image_bytes(int d, int dff, int precision) {
  if (!skf_ffn_fused_supported(1, d, dff, precision)) return 0;
  return image_bytes(precision == SKF_PREC_BF16X3 ? 2 : 3);
}
extern "C" size_t skf_ffn_relu_bits_bytes(int M, int d, int dff, int precision) {
  if (!skf_ffn_fused_supported(M, d, dff, precision)) return 0;
  return (size_t)skf_cdiv(M, TR) * NBLK * 8 * 4 * sizeof(unsigned long long);
}

extern "C" size_t skf_dense_image_bytes(int K, int N, int precision) {
  if ((precision != SKF_PREC_BF16X6 && precision != SKF_PREC_BF16X3) || K <= 0 || N <= 0 || (K & 31) || (N & 15)) return 0;
  return (size_t)(precision == SKF_PREC_BF16X3 ? 2 : 3) * K * N * 2;
}

extern "C" int skf_dense_weight_images(int n, const float* const* src, const int* ld, const int* transpose, const int* K, const int* N,
                                       void* const* images, int precision, skf_stream_t stream) {
  SKF_CHECK_ARG(precision == SKF_PREC_BF16X6 || precision == SKF_PREC_BF16X3, "pre-split images exist in the split-arithmetic modes only");
  SKF_CHECK_ARG(n >= 0 && (n == 0 || (src && ld && transpose && K && N && images)), "null argument");
  std::vector<DenseImageDesc> descs((size_t)n);
  for (int j = 0; j < n; ++j) {
    SKF_CHECK_ARG(src[j] && images[j] && ((uintptr_t)images[j] & 15) == 0, "null / unaligned operand");
    SKF_CHECK_ARG(K[j] > 0 && N[j] > 0 && (K[j] & 31) == 0 && (N[j] & 15) == 0 && ld[j] >= (transpose[j] ? K[j] : N[j]), "K % 32, N % 16, pitch");
    descs[j] = DenseImageDesc{src[j], (char*)images[j], ld[j], transpose[j] ? 1 : 0, K[j], N[j], 0, 0};
  }
  return launch_images(descs.data(), n, precision == SKF_PREC_BF16X3 ? 2 : 3, (hipStream_t)stream);
}

extern "C" int skf_ffn_weight_images(int n, const float* const* W1, const int* ld1, const float* const* W2, const int* ld2,
                                     const int* transpose, void* const* images, int d, int dff, int precision, skf_stream_t stream) {
  SKF_CHECK_ARG(skf_ffn_fused_supported(1, d, dff, precision), "skf_ffn_weight_images: d = 128, dff = 512 in a split-arithmetic mode only");
  SKF_CHECK_ARG(n >= 0 && (n == 0 || (W1 && ld1 && W2 && ld2 && transpose && images)), "null argument");
  const int P = precision == SKF_PREC_BF16X3 ? 2 : 3;
  std::vector<DenseImageDesc> descs;
  for (int j = 0; j < n; ++j) {
    SKF_CHECK_ARG(W1[j] && W2[j] && images[j] && ld1[j] >= dff && ld2[j] >= d, "bad weight operand");
    SKF_CHECK_ARG(((uintptr_t)images[j] & 15) == 0, "images must be 16-byte aligned");
    char* img = (char*)images[j];
    // forward: B1 = W1 [d][dff], B2 = W2 [dff][d]; input gradient: B1 = W2^T, B2 = W1^T
    if (!transpose[j]) {
      descs.push_back(DenseImageDesc{W1[j], img, ld1[j], 0, d, dff, 0, 0});
      descs.push_back(DenseImageDesc{W2[j], img + image_bytes(P) / 2, ld2[j], 0, dff, d, 0, 0});
    } else {
      descs.push_back(DenseImageDesc{W2[j], img, ld2[j], 1, d, dff, 0, 0});
      descs.push_back(DenseImageDesc{W1[j], img + image_bytes(P) / 2, ld1[j], 1, dff, d, 0, 0});
    }
  }
  return launch_images(descs.data(), (int)descs.size(), P, (hipStream_t)stream);
}

static int ffn_common_checks(int M, int d, int dff, int precision, const void* a, const void* img, const void* hid, const void* c) {
  SKF_CHECK_ARG(skf_ffn_fused_supported(M, d, dff, precision), "fused feed-forward block: d = 128, dff = 512 in a split-arithmetic mode only (skf_ffn_fused_supported)");
  SKF_CHECK_ARG(a && img && hid && c, "null operand");
  SKF_CHECK_ARG((((uintptr_t)a | (uintptr_t)img | (uintptr_t)hid | (uintptr_t)c) & 15) == 0, "operands must be 16-byte aligned");
  return SKF_OK;
}

extern "C" int skf_ffn_fused_fwd_f32(int M, int d, int dff, const float* x, const void* image, const float* b1, const float* b2,
                                     float* h, void* relu_bits_out, const float* gamma, const float* beta, float* z, float* out,
                                     float* stats, float rate, unsigned site, const void* step_state, int precision, skf_stream_t stream) {
  const int rc = ffn_common_checks(M, d, dff, precision, x, image, h, z);
  if (rc != SKF_OK) return rc;
  SKF_CHECK_ARG(gamma && beta && out && stats && b1 && b2, "null bias / LayerNorm operand");
  SKF_CHECK_ARG(rate >= 0.f && rate < 1.f && (rate == 0.f || step_state), "dropout needs 0 <= rate < 1 and the step state");
  SKF_CHECK_ARG((((uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && (((uintptr_t)stats | (uintptr_t)relu_bits_out) & 7) == 0, "operands must be 16-byte aligned");
  const int P = precision == SKF_PREC_BF16X3 ? 2 : 3;
  FfnFusedParams p{};
  p.A = x; p.lda = d; p.M = M;
  p.img1 = (const char*)image; p.img2 = (const char*)image + image_bytes(P) / 2;
  p.bias1 = b1; p.bias2 = b2; p.H = h; p.bits_out = (unsigned long long*)relu_bits_out;
  p.C = z; p.res = x; p.gamma = gamma; p.beta = beta; p.out = out; p.stats = stats;
  p.rate = rate; p.site = site; p.state = step_state;
  return skf_ffn_fused_launch(p, P, 0, (hipStream_t)stream);
}

extern "C" int skf_ffn_fused_bwd_f32(int M, int d, int dff, const float* dy, const void* image_t, const void* relu_bits_in,
                                     float* dh, float* dx, int accumulate, const int* row_blocks, int row_block_rows,
                                     int precision, skf_stream_t stream) {
  const int rc = ffn_common_checks(M, d, dff, precision, dy, image_t, dh, dx);
  if (rc != SKF_OK) return rc;
  SKF_CHECK_ARG(relu_bits_in && ((uintptr_t)relu_bits_in & 7) == 0, "the backward needs the sign bits the forward wrote");
  SKF_CHECK_ARG(!row_blocks || row_block_rows == TR, "row-block lists of this kernel have 16-row blocks");
  const int P = precision == SKF_PREC_BF16X3 ? 2 : 3;
  FfnFusedParams p{};
  p.A = dy; p.lda = d; p.M = M;
  p.img1 = (const char*)image_t; p.img2 = (const char*)image_t + image_bytes(P) / 2;
  p.H = dh; p.bits_in = (const unsigned long long*)relu_bits_in;
  p.C = dx; p.accumulate = accumulate; p.row_blocks = row_blocks;
  return skf_ffn_fused_launch(p, P, 1, (hipStream_t)stream);
}

extern "C" int skf_ffn_fused_ln_partials(int M) { return ffn_grid(M); }

extern "C" int skf_ffn_fused_bwd_ln_f32(int M, int d, int dff, const float* dout, const float* z, const float* stats, const float* gamma,
                                        float rate, unsigned site, const void* step_state, const void* image_t, const void* relu_bits_in,
                                        float* dy, float* dh, float* dx, float* ln_partials, size_t ln_partials_bytes,
                                        const int* row_blocks, int row_block_rows, int precision, skf_stream_t stream) {
  const int rc = ffn_common_checks(M, d, dff, precision, dout, image_t, dh, dx);
  if (rc != SKF_OK) return rc;
  SKF_CHECK_ARG(z && stats && gamma && dy && ln_partials, "null LayerNorm operand");
  SKF_CHECK_ARG((((uintptr_t)z | (uintptr_t)gamma | (uintptr_t)dy | (uintptr_t)ln_partials) & 15) == 0 && ((uintptr_t)stats & 7) == 0, "operands must be 16-byte aligned");
  SKF_CHECK_ARG(relu_bits_in && ((uintptr_t)relu_bits_in & 7) == 0, "the backward needs the sign bits the forward wrote");
  SKF_CHECK_ARG(rate >= 0.f && rate < 1.f && (rate == 0.f || step_state), "dropout needs 0 <= rate < 1 and the step state");
  SKF_CHECK_ARG(!row_blocks || row_block_rows == TR, "row-block lists of this kernel have 16-row blocks");
  SKF_CHECK_ARG(ln_partials_bytes >= (size_t)ffn_grid(M) * 2 * FD * sizeof(float), "partial buffer too small (skf_ffn_fused_ln_partials(M) x 2 x d floats)");
  const int P = precision == SKF_PREC_BF16X3 ? 2 : 3;
  FfnFusedParams p{};
  p.A = dout; p.lda = d; p.M = M;
  p.img1 = (const char*)image_t; p.img2 = (const char*)image_t + image_bytes(P) / 2;
  p.H = dh; p.bits_in = (const unsigned long long*)relu_bits_in;
  p.C = dx; p.accumulate = 0; p.row_blocks = row_blocks;
  p.gamma = gamma; p.rate = rate; p.site = site; p.state = step_state;
  p.ln_dout = dout; p.ln_z = z; p.ln_stats = stats; p.ln_dy = dy; p.ln_part = ln_partials;
  return skf_ffn_fused_launch(p, P, 1, (hipStream_t)stream);
}

extern "C" int skf_layernorm_bwd_dgrad_supported(int M, int d, int precision) {
  return (precision == SKF_PREC_BF16X6 || precision == SKF_PREC_BF16X3) && d == FD && M >= 1 && (double)M * FD * 4 < 2147483648.0;
}
extern "C" int skf_layernorm_bwd_dgrad_partials(int M) { return ffn_grid(M); }

extern "C" int skf_layernorm_bwd_dgrad_f32(int M, int d, const float* dout, const float* z, const float* stats, const float* gamma,
                                           float rate, unsigned site, const void* step_state, const void* image_t, float* dz, float* dy,
                                           float* da, float* ln_partials, size_t ln_partials_bytes, const int* row_blocks,
                                           int row_block_rows, int precision, skf_stream_t stream) {
  return skf_layernorm_bwd_dgrad_lead_f32(M, d, dout, nullptr, nullptr, z, stats, gamma, rate, site, step_state, image_t, dz, dy, da, ln_partials,
                                          ln_partials_bytes, row_blocks, row_block_rows, precision, stream);
}

extern "C" int skf_layernorm_bwd_dgrad_lead_f32(int M, int d, const float* dout, const float* lead_a, const void* lead_image_t, const float* z,
                                                const float* stats, const float* gamma, float rate, unsigned site, const void* step_state,
                                                const void* image_t, float* dz, float* dy, float* da, float* ln_partials, size_t ln_partials_bytes,
                                                const int* row_blocks, int row_block_rows, int precision, skf_stream_t stream) {
  SKF_CHECK_ARG(skf_layernorm_bwd_dgrad_supported(M, d, precision), "LayerNorm backward + input gradient: d = 128 in a split-arithmetic mode only");
  SKF_CHECK_ARG(!lead_a == !lead_image_t && (((uintptr_t)lead_a | (uintptr_t)lead_image_t) & 15) == 0, "leading product: rows and the transposed image together, 16-byte aligned");
  SKF_CHECK_ARG(dout && z && stats && gamma && image_t && dz && dy && da && ln_partials, "null operand");
  SKF_CHECK_ARG((((uintptr_t)dout | (uintptr_t)z | (uintptr_t)gamma | (uintptr_t)image_t | (uintptr_t)dz | (uintptr_t)dy | (uintptr_t)da |
                  (uintptr_t)ln_partials) & 15) == 0 && ((uintptr_t)stats & 7) == 0, "operands must be 16-byte aligned");
  SKF_CHECK_ARG(rate >= 0.f && rate < 1.f && (rate == 0.f || step_state), "dropout needs 0 <= rate < 1 and the step state");
  SKF_CHECK_ARG(!row_blocks || row_block_rows == TR, "row-block lists of this kernel have 16-row blocks");
  SKF_CHECK_ARG(ln_partials_bytes >= (size_t)ffn_grid(M) * 2 * FD * sizeof(float), "partial buffer too small (skf_layernorm_bwd_dgrad_partials(M) x 2 x d floats)");
  const int P = precision == SKF_PREC_BF16X3 ? 2 : 3;
  LnDgradParams p{};
  p.dout = dout; p.z = z; p.stats = stats; p.gamma = gamma; p.dz = dz; p.dy = dy; p.C = da; p.img = (const char*)image_t;
  p.part = ln_partials; p.M = M; p.rate = rate; p.site = site; p.state = step_state; p.row_blocks = row_blocks;
  p.lead_a = lead_a; p.lead_img = (const char*)lead_image_t;
  const int grid = ffn_grid(M);
  const bool lead = lead_a != nullptr;
  const size_t smem = (size_t)P * PLANE + (lead ? (size_t)ROWS * YPITCH * sizeof(float) : 0);
  hipStream_t st = (hipStream_t)stream;
  static const std::string tag2 = "ln_bwd_dgrad<d128,bf16x3>", tag3 = "ln_bwd_dgrad<d128,bf16x6>";
  const double live = skf_prof_list_fraction(row_blocks);
  const double nprod = lead ? 2.0 : 1.0, nrows = lead ? 6.0 : 5.0;
  SkfProfScope ps(st, (P == 2 ? tag2 : tag3).c_str(), nprod * 2.0 * M * FD * FD, 4.0 * nrows * M * FD);
  ps.done(nprod * 2.0 * M * FD * FD * live, 4.0 * nrows * M * FD * live);
#define SKF_LN_DGRAD_GO(PV, LV)                                                                                                                  \
  {                                                                                                                                              \
    SKF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_dgrad_kernel<PV, LV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    SKF_LAUNCH_TAIL((ln_bwd_dgrad_kernel<PV, LV>), dim3(grid), dim3(512), smem, st, p);                                                          \
  }
  if (P == 2) { if (lead) SKF_LN_DGRAD_GO(2, true) else SKF_LN_DGRAD_GO(2, false) }
  else { if (lead) SKF_LN_DGRAD_GO(3, true) else SKF_LN_DGRAD_GO(3, false) }
#undef SKF_LN_DGRAD_GO
  SKF_LAUNCH_CHECK();
  return SKF_OK;
}

extern "C" int skf_ffn_fused_fwd_proj_f32(int M, int d, int dff, const float* x, const void* image, const float* b1, const float* b2,
                                          float* h, void* relu_bits_out, const float* gamma, const float* beta, float* z, float* out,
                                          float* stats, float rate, unsigned site, const void* step_state, const void* proj_image,
                                          const float* proj_bias, int proj_n, float* proj_out, int precision, skf_stream_t stream) {
  const int rc = ffn_common_checks(M, d, dff, precision, x, image, h, z);
  if (rc != SKF_OK) return rc;
  SKF_CHECK_ARG(gamma && beta && out && stats && b1 && b2, "null bias / LayerNorm operand");
  SKF_CHECK_ARG(rate >= 0.f && rate < 1.f && (rate == 0.f || step_state), "dropout needs 0 <= rate < 1 and the step state");
  SKF_CHECK_ARG((((uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && (((uintptr_t)stats | (uintptr_t)relu_bits_out) & 7) == 0, "operands must be 16-byte aligned");
  SKF_CHECK_ARG(proj_image && proj_bias && proj_out && (proj_n == 128 || proj_n == 256 || proj_n == 384), "chained projection: N in {128, 256, 384} with its image, bias and output");
  SKF_CHECK_ARG((((uintptr_t)proj_image | (uintptr_t)proj_bias | (uintptr_t)proj_out) & 15) == 0 && (double)M * proj_n * 4 < 2147483648.0, "chained projection operands");
  const int P = precision == SKF_PREC_BF16X3 ? 2 : 3;
  FfnFusedParams p{};
  p.A = x; p.lda = d; p.M = M;
  p.img1 = (const char*)image; p.img2 = (const char*)image + image_bytes(P) / 2;
  p.bias1 = b1; p.bias2 = b2; p.H = h; p.bits_out = (unsigned long long*)relu_bits_out;
  p.C = z; p.res = x; p.gamma = gamma; p.beta = beta; p.out = out; p.stats = stats;
  p.rate = rate; p.site = site; p.state = step_state;
  p.img3 = (const char*)proj_image; p.bias3 = proj_bias; p.out2 = proj_out; p.n2 = proj_n;
  return skf_ffn_fused_launch(p, P, 0, (hipStream_t)stream);
}

extern "C" int skf_ffn_block_fwd_f32(const SkfFfnBlockFwd* b, skf_stream_t stream) {
  SKF_CHECK_ARG(b && b->struct_size == sizeof(SkfFfnBlockFwd), "SkfFfnBlockFwd.struct_size does not match this library's include/skf.h");
  const bool tail_only = !b->image;           // no feed-forward block: the leading stage and the projection of ITS LayerNorm output
  if (tail_only) {
    SKF_CHECK_ARG(skf_ffn_fused_supported(b->M, b->d, b->dff > 0 ? b->dff : FF, b->precision), "row-owner launches: d = 128 in a split-arithmetic mode only");
    SKF_CHECK_ARG(b->x && b->pre_image && b->proj_image && ((uintptr_t)b->x & 15) == 0, "without a feed-forward image the launch needs the leading stage and the chained projection");
  } else {
    const int rc = ffn_common_checks(b->M, b->d, b->dff, b->precision, b->x, b->image, b->h, b->z);
    if (rc != SKF_OK) return rc;
    SKF_CHECK_ARG(b->gamma && b->beta && b->out && b->stats && b->b1 && b->b2, "null bias / LayerNorm operand");
    SKF_CHECK_ARG((((uintptr_t)b->out | (uintptr_t)b->gamma | (uintptr_t)b->beta | (uintptr_t)b->b1 | (uintptr_t)b->b2) & 15) == 0 &&
                  (((uintptr_t)b->stats | (uintptr_t)b->relu_bits_out) & 7) == 0, "operands must be 16-byte aligned");
  }
  SKF_CHECK_ARG(b->rate >= 0.f && b->rate < 1.f && (b->rate == 0.f || b->step_state), "dropout needs 0 <= rate < 1 and the step state");
  const int P = b->precision == SKF_PREC_BF16X3 ? 2 : 3;
  FfnFusedParams p{};
  p.A = b->x; p.lda = b->d; p.M = b->M;
  p.img1 = (const char*)b->image; p.img2 = tail_only ? nullptr : (const char*)b->image + image_bytes(P) / 2;
  p.bias1 = b->b1; p.bias2 = b->b2; p.H = b->h; p.bits_out = (unsigned long long*)b->relu_bits_out;
  p.C = b->z; p.res = b->x; p.gamma = b->gamma; p.beta = b->beta; p.out = b->out; p.stats = b->stats;
  p.rate = b->rate; p.site = b->site; p.state = b->step_state;
  if (b->pre_image) {
    SKF_CHECK_ARG(b->pre_bias && b->pre_residual && b->pre_gamma && b->pre_beta && b->pre_z && b->pre_out && b->pre_stats, "leading projection: null operand");
    SKF_CHECK_ARG((((uintptr_t)b->pre_image | (uintptr_t)b->pre_bias | (uintptr_t)b->pre_residual | (uintptr_t)b->pre_gamma | (uintptr_t)b->pre_beta |
                    (uintptr_t)b->pre_z | (uintptr_t)b->pre_out) & 15) == 0 && ((uintptr_t)b->pre_stats & 7) == 0, "leading projection: alignment");
    p.pre_img = (const char*)b->pre_image; p.pre_bias = b->pre_bias; p.pre_res = b->pre_residual; p.pre_gamma = b->pre_gamma;
    p.pre_beta = b->pre_beta; p.pre_site = b->pre_site; p.pre_z = b->pre_z; p.pre_out = b->pre_out; p.pre_stats = b->pre_stats;
  }
  if (b->proj_image) {
    SKF_CHECK_ARG(b->proj_bias && b->proj_out && (b->proj_n == 128 || b->proj_n == 256 || b->proj_n == 384), "chained projection: N in {128, 256, 384} with its image, bias and output");
    SKF_CHECK_ARG((((uintptr_t)b->proj_image | (uintptr_t)b->proj_bias | (uintptr_t)b->proj_out) & 15) == 0 && (double)b->M * b->proj_n * 4 < 2147483648.0, "chained projection operands");
    p.img3 = (const char*)b->proj_image; p.bias3 = b->proj_bias; p.out2 = b->proj_out; p.n2 = b->proj_n;
  }
  return skf_ffn_fused_launch(p, P, 0, (hipStream_t)stream);
}
