// Shared device/host helpers for the sketchformer_amd HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/skf.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SKF_WAVE 64

// The shipped library reads NO environment variable and keeps no configuration state (include/skf.h): the A/B, ablation and
// timeline knobs of tools/ exist only in measurement builds (SKF_EXTRA_HIPCC_FLAGS=-DSKF_MEASURE=1 python -m
// sketchformer_amd.build --force); in the default build skf_knob() is a constant null and every branch on it folds away.
#ifndef SKF_MEASURE
#define SKF_MEASURE 0
#endif
#if SKF_MEASURE
#include <stdlib.h>
static inline const char* skf_knob(const char* name) { return getenv(name); }
#else
static inline const char* skf_knob(const char*) { return nullptr; }
#endif

void skf_set_error(const char* fmt, ...);

#define SKF_CHECK_ARG(cond, msg)                                   \
  do {                                                             \
    if (!(cond)) {                                                 \
      skf_set_error("%s: %s (%s)", __func__, msg, #cond);          \
      return SKF_EINVAL;                                           \
    }                                                              \
  } while (0)

#define SKF_LAUNCH_CHECK()                                          \
  do {                                                              \
    hipError_t e__ = hipGetLastError();                             \
    if (e__ != hipSuccess) {                                        \
      skf_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
      return SKF_EHIP;                                              \
    }                                                               \
  } while (0)

#define SKF_HIP(call)                                               \
  do {                                                              \
    hipError_t e__ = (call);                                        \
    if (e__ != hipSuccess) {                                        \
      skf_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e__)); \
      return SKF_EHIP;                                              \
    }                                                               \
  } while (0)

// Optional per-launch timing (HIP events on the launch stream), used by bench.py for the
// roofline figures.  Inactive (one branch) unless skf_profiler_enable(1) was called.
struct SkfProfScope {
  SkfProfScope(hipStream_t st, const char* tag, double flops, double bytes);
  ~SkfProfScope();
  // Work the launch actually performs when it walks a live-row / live-tile list or skips masked tiles (flops, bytes are the
  // dense figures).  Call only when active(): the fractions below synchronise the device and read the lists back.
  void done(double flops_done, double bytes_done);
  bool active() const { return idx_ >= 0; }
  hipStream_t st_;
  int idx_;
};
// Profiling only (both synchronise the device): live share of a {n_live, n_total, ...} list (skf_row_blocks.hip), and the
// share of (query tile, key tile) pairs an attention launch visits: per sample the query tiles below q_live[b] (all when
// null), the key tiles up to the last un-padded key (all when the mask is null) and, with a look-ahead mask that may be
// skipped (key 0 visible), only key tiles kt <= qt.
double skf_prof_list_fraction(const int* list);
double skf_prof_attention_fraction(const unsigned char* key_mask, int mask_ld, int causal, int B, int Lq, int Lk,
                                   const int* q_live, int qtile, int ktile);

static inline int skf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// An event that means "everything this stream was given so far is complete" costs the stream ~5 us as a packet of its own
// (hipEventRecord) and ~1.3 us attached to the kernel launch in front of it as that command's completion signal (hipExtLaunchKernelGGL's
// stop event; tools/micro/event_cost.hip, profiles/r06c_event_cost.txt: 25.4 vs 21.7 us per iteration against 20.4 without any event).
// A caller that is about to record such an event behind a launcher's LAST kernel parks it in skf_tls_stop_event first; a launcher built
// with SKF_LAUNCH_TAIL attaches it and clears the slot (a launcher that is not leaves it: the caller then records the event as before).
extern thread_local hipEvent_t skf_tls_stop_event;
#define SKF_LAUNCH_TAIL(kernel, grid, block, smem, stream, ...)                                        \
  do {                                                                                                 \
    hipEvent_t ev__ = skf_tls_stop_event;                                                              \
    if (ev__) {                                                                                        \
      skf_tls_stop_event = nullptr;                                                                    \
      hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, nullptr, ev__, 0, __VA_ARGS__);         \
    } else {                                                                                           \
      hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                              \
    }                                                                                                  \
  } while (0)

// "Once per DEVICE" guard for hipFuncSetAttribute (the attribute belongs to the (function, device) pair; one process may drive
// several GPUs, one host thread each):
//     static SkfOncePerDevice once;  if (once.needed()) { SKF_HIP(hipFuncSetAttribute(...)); once.mark(); }
// The bit is set AFTER the attribute call succeeded (a second host thread on the same device either sees the bit and finds the
// attribute applied, or sets the attribute again itself - idempotent) and with an atomic OR (threads of different devices share the word).
struct SkfOncePerDevice {
  unsigned long long done = 0;
  static int device() {
    int dev = 0;
    return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) ? dev : -1;
  }
  bool needed() const {
    const int dev = device();
    return dev < 0 || !((__atomic_load_n(&done, __ATOMIC_ACQUIRE) >> dev) & 1ull);      // unknown device: always set
  }
  void mark() {
    const int dev = device();
    if (dev >= 0) __atomic_fetch_or(&done, 1ull << dev, __ATOMIC_RELEASE);
  }
};

// ---------------------------------------------------------------- device
// Bijective XCD-aware remap of a 1-D grid: block b runs on XCD b % 8 (observed dispatch order), so giving
// every XCD a CONTIGUOUS range of logical ids keeps workgroups that share operand panels on one L2.
__device__ __forceinline__ int skf_xcd_remap(int orig, int nwg) {
  const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
// Numbering of work items (group g, part w) whose cost depends on the PART with a period that divides the dispatcher's own.
// Consecutive workgroups of an XCD are handed to its 4 shader engines in turn and a workgroup only ever runs on the 8 CUs of the
// engine it was handed to, so with 4 parts per group and the part as the fastest index every heavy part (the one live 128-key block
// of a padded sample, the last query block under the look-ahead mask) queues on the same quarter of the chip and the launch takes
// as long as if all parts were heavy (measured, bf16 dK/dV pass, B 128 / H 8 / L 512: 8 live keys of 512 223 us against 236 us
// full-length, 111 us with this numbering; profiles/r05n_bf16_attn_dispatch.txt).  Here: chunks of 32 groups, part-major inside a
// chunk, so 32 consecutive workgroups carry the same part (part 0 first) and the parts of a group stay close enough in time to
// share the L2.
__device__ __forceinline__ void skf_part_major(int lid, int ngroups, int nparts, int* g, int* w) {
  constexpr int G = 32;
  const int chunk = lid / (G * nparts), j = lid - chunk * G * nparts;
  const int gc = min(G, ngroups - chunk * G);
  *w = j / gc;
  *g = chunk * G + j - *w * gc;
}
// The other way round: workgroups whose cost is known per SAMPLE (padded batches: one workgroup per (sample, head)).  Given the
// samples sorted by cost (heaviest first), workgroup `bid` (-> XCD bid % 8, arrival i = bid / 8 in that XCD -> engine i % 4) takes head
// i % H of the sample with rank (bid % 8) + 8 * (i / H): the sorted list is DEALT over the XCDs, heaviest first, and the heads of a
// sample stay on one XCD, spread over its engines - every XCD and every engine gets the same mix of lengths, and an XCD reads whole
// activation rows (its heads' slices side by side: with the heads of a row on eight XCDs each L2 saw one 128-byte slice per KB and the
// bf16 dQ pass ran 27 % slower).  Returns rank * H + head; needs B % 8 == 0 (the caller checks).
__device__ __forceinline__ int skf_deal_rank(int bid, int H) {
  const int x = bid & 7, i = bid >> 3, m = i / H;
  return (x + 8 * m) * H + (i - m * H);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// (x, y) -> P dwords; dword q = bf16 piece q of x in the low half, of y in the high half.  Pieces are the top 16 bits
// of the running remainder (truncation), remainders are exact: x = x0 + x1 + x2 for every finite fp32 (24 significand
// bits = 3 x 8).  The remainder x - x_q is one v_dot2c_f32_bf16 (x += piece . (-1, 0)) instead of a mask and a subtract.
typedef __bf16 skf_bf16x2 __attribute__((ext_vector_type(2)));
// the (-1, 0) / (0, -1) selectors of the remainder dot products, kept opaque in SGPRs (folded into the inline constant
// "-1.0" the instruction subtracted the wrong half on gfx950).  Create ONCE per kernel: the asm is not hoisted.
struct SkfSplitSel { unsigned lo, hi; };
__device__ __forceinline__ SkfSplitSel skf_split_sel() {
  SkfSplitSel r{0x0000bf80u, 0xbf800000u};
  asm volatile("" : "+s"(r.lo), "+s"(r.hi));
  return r;
}
template <int P>
__device__ __forceinline__ void skf_split2(float x, float y, unsigned (&out)[P], const SkfSplitSel& sel) {
#pragma unroll
  for (int q = 0; q < P; ++q) {
    out[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, y), __builtin_bit_cast(unsigned, x), 0x07060302u);   // (y & 0xffff0000) | (x >> 16)
    if (q + 1 < P) {
#ifdef SKF_SPLIT_NO_DOT2
      x -= __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
      y -= __builtin_bit_cast(float, __builtin_bit_cast(unsigned, y) & 0xffff0000u);
#else
      const skf_bf16x2 pc = __builtin_bit_cast(skf_bf16x2, out[q]);
      x = __builtin_amdgcn_fdot2_f32_bf16(pc, __builtin_bit_cast(skf_bf16x2, sel.lo), x, false);
      y = __builtin_amdgcn_fdot2_f32_bf16(pc, __builtin_bit_cast(skf_bf16x2, sel.hi), y, false);
#endif
    }
  }
}

// Counter-based dropout RNG: one 32-bit hash per element, keyed on
// (key = f(seed, step), site, flat element index).  Deterministic, stateless,
// identical on host (skf_dropout_keep_mask) and device.
__host__ __device__ __forceinline__ uint32_t skf_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t skf_site_key(uint32_t key, uint32_t site) {
  return skf_hash32(key ^ (0x9e3779b9U * (site + 1)));
}
constexpr uint32_t kSkfKeepStride = 0x9e3779b1U;   // hash argument of element idx = idx * kSkfKeepStride + site_key
// the same decision from the hash argument itself: kernels that walk idx in fixed steps add multiples of kSkfKeepStride
// instead of multiplying per element (v_mul_lo_u32 is a quarter-rate instruction)
__host__ __device__ __forceinline__ bool skf_keep_arg(uint32_t arg, uint32_t thresh) { return skf_hash32(arg) >= thresh; }
__host__ __device__ __forceinline__ bool skf_keep(uint32_t site_key, uint32_t idx, uint32_t thresh) {
  // keep iff u >= rate  (tf.nn.dropout: random_uniform >= rate)
  return skf_keep_arg(idx * kSkfKeepStride + site_key, thresh);
}
__host__ __device__ __forceinline__ uint32_t skf_drop_thresh(float rate) {
  double t = (double)rate * 4294967296.0;
  return t >= 4294967295.0 ? 0xffffffffU : (uint32_t)t;
}

// Per-step scalars that live in device memory so a captured hipGraph can be
// replayed: written by the step-prologue kernel, read by dropout sites / Adam.
struct SkfStepState {
  long long iterations;   // optimizer.iterations (pre-increment value used this step)
  float lr;               // schedule(iterations)
  float alpha;            // lr * sqrt(1-b2^t)/(1-b1^t)
  uint32_t drop_key;      // hash(seed, iterations)
  uint32_t pad;
};
