// Train-step orchestrator: the fixed launch sequence of Transformer.call +
// model_trainer (models/sketchformer.py:131-181, 325-349) over caller-owned flat
// device buffers, optionally captured into hipGraphs (one for forward+backward,
// one for the optimizer, so a data-parallel caller can all-reduce the flat
// gradient buffer in between).  No autograd: the backward sequence is explicit.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <map>
#include <algorithm>
#include "skf_common.h"
#include "skf_decode_fused.h"

// ------------------------------------------------------------------ error plumbing
#if SKF_MEASURE
// measurement builds: SKF_EVLOG=1 lists every event record / stream wait of the THIRD train step with its source line and stream
// (tools: which packets still sit on the main stream between two kernels)
static hipStream_t g_evlog_main = nullptr;
static int g_evlog_step = 0;
static inline bool evlog_on() { static const bool on = skf_knob("SKF_EVLOG") != nullptr; return on && g_evlog_step == 3; }
static inline hipError_t skf_logged_record(hipEvent_t e, hipStream_t st, int line) {
  if (evlog_on()) fprintf(stderr, "EVLOG record line %d on %s\n", line, st == g_evlog_main ? "MAIN" : "side");
  return hipEventRecord(e, st);
}
static inline hipError_t skf_logged_wait(hipStream_t st, hipEvent_t e, unsigned f, int line) {
  if (evlog_on()) fprintf(stderr, "EVLOG wait   line %d on %s\n", line, st == g_evlog_main ? "MAIN" : "side");
  return hipStreamWaitEvent(st, e, f);
}
#define hipEventRecord(e, st) skf_logged_record(e, st, __LINE__)
#define hipStreamWaitEvent(st, e, f) skf_logged_wait(st, e, f, __LINE__)
#endif
static thread_local char g_err[512] = "";
thread_local hipEvent_t skf_tls_stop_event = nullptr;      // skf_common.h: an event for the next SKF_LAUNCH_TAIL launch of this thread
void skf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* skf_last_error(void) { return g_err; }
extern "C" int skf_version(void) { return 100; }
extern "C" int skf_device_info(char* name_host, size_t name_len, int* n_devices_host) {
  int n = 0;
  SKF_HIP(hipGetDeviceCount(&n));
  if (n_devices_host) *n_devices_host = n;
  if (name_host && name_len) {
    int dev = 0;
    SKF_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    SKF_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(name_host, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  return SKF_OK;
}

// ------------------------------------------------------------------ launch profiler
int skf_adam_step_launch(float* w, const float* g, float* m, float* v, size_t n, void* step_state, float grad_scale, float beta1, float beta2,
                         float eps, int advance, hipStream_t stream);
int skf_pool_bwd_partials(float* u_inout_dpre, const float* Vw, const float* x, const float* a, const float* demb, int B, int L, int U, int d,
                          float* dx, float* dV_part, hipStream_t s);
int skf_expander_bwd_partials(const float* dpre, const float* emb, const float* w, int B, int L, int d, float* demb, int demb_accumulate,
                              float* p1, float* p2, hipStream_t s);
int skf_stage_inputs_launch(const void* inp, void* dinp, const void* tar, void* dtar, size_t row, size_t src_row, size_t copy, int batch,
                            const void* labels, void* dlabels, hipStream_t st, unsigned char* emask = nullptr, unsigned char* dmask = nullptr,
                            int mask_L = 0);   // skf_rowops.hip
namespace {
struct ProfRec { const char* tag; double flops, bytes, flops_done, bytes_done; hipEvent_t e0, e1; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
thread_local int g_capturing = 0;      // > 0 while this thread records a step into a hipGraph (capture_or_run)
}  // namespace

SkfProfScope::SkfProfScope(hipStream_t st, const char* tag, double flops, double bytes) : st_(st), idx_(-1) {
  if (!g_prof_on) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
  ProfRec r{tag, flops, bytes, flops, bytes, nullptr, nullptr};
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
  (void)hipEventRecord(r.e0, st);
  g_prof.push_back(r);
  idx_ = (int)g_prof.size() - 1;
}
SkfProfScope::~SkfProfScope() {
  if (idx_ >= 0) (void)hipEventRecord(g_prof[idx_].e1, st_);
}
void SkfProfScope::done(double flops_done, double bytes_done) {
  if (idx_ < 0) return;
  g_prof[idx_].flops_done = flops_done;
  g_prof[idx_].bytes_done = bytes_done;
}
double skf_prof_list_fraction(const int* list) {
  // (never inside a stream capture: a device synchronisation there invalidates the capture - hipErrorStreamCaptureUnsupported)
  if (!list || !g_prof_on || g_capturing) return 1.0;
  int h[2] = {0, 0};
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, list, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess || h[1] <= 0) return 1.0;
  return (double)h[0] / h[1];
}
double skf_prof_attention_fraction(const unsigned char* key_mask, int mask_ld, int causal, int B, int Lq, int Lk,
                                   const int* q_live, int qtile, int ktile) {
  if (!g_prof_on || g_capturing || (!key_mask && !q_live && !causal)) return 1.0;
  if (hipDeviceSynchronize() != hipSuccess) return 1.0;
  std::vector<unsigned char> km;
  std::vector<int> ql;
  if (key_mask) {
    km.resize((size_t)B * mask_ld);
    if (hipMemcpy(km.data(), key_mask, km.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1.0;
  }
  if (q_live) {
    ql.resize(B);
    if (hipMemcpy(ql.data(), q_live, (size_t)B * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 1.0;
  }
  const int nqt_all = (Lq + qtile - 1) / qtile, nkt_all = (Lk + ktile - 1) / ktile;
  double visited = 0.0;
  for (int b = 0; b < B; ++b) {
    int nqt = nqt_all, nkt = nkt_all;
    bool can_skip = causal != 0;
    if (q_live) nqt = std::min(nqt_all, (std::max(ql[b], 0) + qtile - 1) / qtile);
    if (key_mask) {
      const unsigned char* m = km.data() + (size_t)b * mask_ld;
      int lastk = -1;
      for (int k = 0; k < Lk; ++k) if (!m[k]) lastk = k;
      can_skip = can_skip && !m[0];
      if (lastk >= 0 && (!causal || can_skip)) nkt = lastk / ktile + 1;
    }
    for (int qt = 0; qt < nqt; ++qt) {
      // keys this query tile can see under the look-ahead rule, in key tiles
      const int lim = can_skip ? std::min(nkt, ((qt + 1) * qtile - 1) / ktile + 1) : nkt;
      visited += lim;
    }
  }
  return visited / ((double)B * nqt_all * nkt_all);
}

extern "C" int skf_profiler_enable(int on) {
  for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_prof.clear();
  g_prof_on = on != 0;
  return SKF_OK;
}

extern "C" int skf_profiler_report(char* buf_host, size_t len) {
  SKF_CHECK_ARG(buf_host && len > 2, "bad buffer");
  struct Agg { int count = 0; double ms = 0, flops = 0, bytes = 0, flops_done = 0, bytes_done = 0; };
  std::vector<std::pair<std::string, Agg>> order;
  std::map<std::string, size_t> index;
  for (auto& r : g_prof) {
    SKF_HIP(hipEventSynchronize(r.e1));
    float ms = 0.f;
    SKF_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
    auto it = index.find(r.tag);
    if (it == index.end()) { index[r.tag] = order.size(); order.push_back({r.tag, Agg()}); it = index.find(r.tag); }
    Agg& a = order[it->second].second;
    a.count += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes; a.flops_done += r.flops_done; a.bytes_done += r.bytes_done;
  }
  std::string out = "[";
  char line[384];
  for (size_t i = 0; i < order.size(); ++i) {
    const Agg& a = order[i].second;
    snprintf(line, sizeof(line), "%s{\"tag\":\"%s\",\"count\":%d,\"ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e,\"flops_done\":%.6e,\"bytes_done\":%.6e}",
             i ? "," : "", order[i].first.c_str(), a.count, a.ms, a.flops, a.bytes, a.flops_done, a.bytes_done);
    out += line;
  }
  out += "]";
  SKF_CHECK_ARG(out.size() + 1 <= len, "report buffer too small");
  memcpy(buf_host, out.c_str(), out.size() + 1);
  return SKF_OK;
}

namespace {

inline size_t pad4(size_t n) { return (n + 3) & ~(size_t)3; }

struct DenseP { size_t w, b; int in, out, ld; };     // offsets into the flat buffer
struct LnP { size_t g, b; };
struct SelfMhaP { DenseP qkv, o; };
struct CrossMhaP { DenseP q, kv, o; };
struct EncLayerP { SelfMhaP mha; DenseP f1, f2; LnP ln1, ln2; };
struct DecLayerP { SelfMhaP mha1; CrossMhaP mha2; DenseP f1, f2; LnP ln1, ln2, ln3; };

// models/sketchformer.py:76-108: the bottleneck (+ expander) exists when lowerdim > 0, the class head only inside that
// block and only with do_classification, the decoder / output layer only with do_reconstruction
inline bool has_bott(const SkfConfig& c) { return c.lowerdim > 0; }
inline bool has_cls(const SkfConfig& c) { return c.lowerdim > 0 && c.do_classification != 0; }
inline bool do_recon(const SkfConfig& c) { return c.do_reconstruction != 0; }

struct Layout {
  size_t total = 0;
  size_t enc_emb = 0, dec_emb = 0;      // token mode: (V,d) tables
  DenseP enc_embd{}, dec_embd{};         // continuous mode: Dense(5 -> d)
  std::vector<EncLayerP> enc;
  std::vector<DecLayerP> dec;
  DenseP bott_w{};      // W_attn + b_attn
  size_t bott_v = 0;    // V_attn
  DenseP bott_e{};      // SelfAttnV2 only: Dense(lowerdim) after the pooling (builders/layers/transformer.py:92,128)
  std::vector<DenseP> cbuf;   // class_buffer Dense(lowerdim, relu) layers (models/sketchformer.py:101-104)
  size_t dec_off = 0;   // offset of the decoder embedding (== total when there is no decoder)
  int E = 0, Ua = 0;    // embedding width (d for V1, lowerdim for V2); units of the attention scorer (lowerdim / d)
  DenseP cls{}, out{};
  size_t exp_w = 0, exp_b = 0;
  std::vector<SkfParamEntry> entries;
};

void add_entry(Layout& L, const std::string& name, size_t off, int rows, int cols, int stride) {
  SkfParamEntry e;
  memset(&e, 0, sizeof(e));
  snprintf(e.name, sizeof(e.name), "%s", name.c_str());
  e.offset = (int64_t)off; e.rows = rows; e.cols = cols; e.row_stride = stride;
  L.entries.push_back(e);
}

size_t alloc(Layout& L, size_t n) { size_t o = L.total; L.total += pad4(n); return o; }

DenseP dense(Layout& L, const std::string& name, int in, int out) {
  DenseP d; d.in = in; d.out = out; d.ld = out;
  d.w = alloc(L, (size_t)in * out); d.b = alloc(L, out);
  add_entry(L, name + "/kernel", d.w, in, out, out);
  add_entry(L, name + "/bias", d.b, 1, out, out);
  return d;
}

// fused [in][nparts*out] block exposed as nparts strided (in,out) kernels
DenseP fused_dense(Layout& L, const std::string& prefix, const char* const* names, int nparts, int in, int out) {
  DenseP d; d.in = in; d.out = nparts * out; d.ld = nparts * out;
  d.w = alloc(L, (size_t)in * d.out); d.b = alloc(L, d.out);
  for (int i = 0; i < nparts; ++i) {
    add_entry(L, prefix + "/" + names[i] + "/kernel", d.w + (size_t)i * out, in, out, d.ld);
    add_entry(L, prefix + "/" + names[i] + "/bias", d.b + (size_t)i * out, 1, out, out);
  }
  return d;
}

LnP lnp(Layout& L, const std::string& name, int d) {
  LnP p; p.g = alloc(L, d); p.b = alloc(L, d);
  add_entry(L, name + "/gamma", p.g, 1, d, d);
  add_entry(L, name + "/beta", p.b, 1, d, d);
  return p;
}

Layout build_layout(const SkfConfig& c) {
  Layout L;
  const int d = c.d_model;
  // SelfAttnV1 returns (B,d), V2 projects to (B,lowerdim); without a bottleneck the "embedding" is the encoder output
  const int E = (has_bott(c) && c.attn_version == 2) ? c.lowerdim : d;
  const int Ua = c.attn_version == 2 ? d : c.lowerdim;    // W_attn is (d,units) in V1, (d,d) in V2
  L.E = E; L.Ua = Ua;
  static const char* const qkv_names[3] = {"wq", "wk", "wv"};
  static const char* const kv_names[2] = {"wk", "wv"};
  if (c.continuous) {
    L.enc_embd = dense(L, "encoder/embedding", 5, d);
  } else {
    L.enc_emb = alloc(L, (size_t)c.vocab_size * d);
    add_entry(L, "encoder/embedding", L.enc_emb, c.vocab_size, d, d);
  }
  for (int i = 0; i < c.num_layers; ++i) {
    const std::string p = "encoder/layer" + std::to_string(i);
    EncLayerP e;
    e.mha.qkv = fused_dense(L, p + "/mha", qkv_names, 3, d, d);
    e.mha.o = dense(L, p + "/mha/dense", d, d);
    e.f1 = dense(L, p + "/ffn/dense1", d, c.dff);
    e.f2 = dense(L, p + "/ffn/dense2", c.dff, d);
    e.ln1 = lnp(L, p + "/layernorm1", d);
    e.ln2 = lnp(L, p + "/layernorm2", d);
    L.enc.push_back(e);
  }
  if (has_bott(c)) {
    L.bott_w.in = d; L.bott_w.out = Ua; L.bott_w.ld = Ua;
    L.bott_w.w = alloc(L, (size_t)d * Ua); L.bott_w.b = alloc(L, Ua);
    L.bott_v = alloc(L, Ua);
    add_entry(L, "bottleneck/W_attn", L.bott_w.w, d, Ua, Ua);
    add_entry(L, "bottleneck/b_attn", L.bott_w.b, 1, Ua, Ua);
    add_entry(L, "bottleneck/V_attn", L.bott_v, Ua, 1, 1);
    if (c.attn_version == 2) L.bott_e = dense(L, "bottleneck/embeding_layer", d, c.lowerdim);
  }
  if (has_cls(c)) {
    for (int i = 0; i < c.class_buffer_layers; ++i)
      L.cbuf.push_back(dense(L, "class_buffer/" + std::to_string(i), i == 0 ? E : c.lowerdim, c.lowerdim));
    L.cls = dense(L, "classify", c.class_buffer_layers ? c.lowerdim : E, c.n_classes);
  }
  L.dec_off = L.total;                   // first float of the decoder-side variables (gradient bucket boundary)
  if (!do_recon(c)) return L;
  if (has_bott(c)) {
    L.exp_w = alloc(L, c.seq_len); L.exp_b = alloc(L, c.seq_len);
    add_entry(L, "expand/kernel", L.exp_w, 1, c.seq_len, c.seq_len);
    add_entry(L, "expand/bias", L.exp_b, 1, c.seq_len, c.seq_len);
  }
  L.dec_off = L.total;
  if (c.continuous) {
    L.dec_embd = dense(L, "decoder/embedding", 5, d);
  } else {
    L.dec_emb = alloc(L, (size_t)c.vocab_size * d);
    add_entry(L, "decoder/embedding", L.dec_emb, c.vocab_size, d, d);
  }
  for (int i = 0; i < c.num_layers; ++i) {
    const std::string p = "decoder/layer" + std::to_string(i);
    DecLayerP e;
    e.mha1.qkv = fused_dense(L, p + "/mha1", qkv_names, 3, d, d);
    e.mha1.o = dense(L, p + "/mha1/dense", d, d);
    e.mha2.q = dense(L, p + "/mha2/wq", d, d);
    e.mha2.kv = fused_dense(L, p + "/mha2", kv_names, 2, E, d);
    e.mha2.o = dense(L, p + "/mha2/dense", d, d);
    e.f1 = dense(L, p + "/ffn/dense1", d, c.dff);
    e.f2 = dense(L, p + "/ffn/dense2", c.dff, d);
    e.ln1 = lnp(L, p + "/layernorm1", d);
    e.ln2 = lnp(L, p + "/layernorm2", d);
    e.ln3 = lnp(L, p + "/layernorm3", d);
    L.dec.push_back(e);
  }
  L.out = dense(L, "output", d, c.continuous ? 5 : c.vocab_size);
  return L;
}

// ------------------------------------------------------------------ workspace plan
struct Bump {
  size_t off = 0;
  size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
};

struct EncAct { size_t x_in, qkv, o, z1, st1, astats, x1, h, z2, st2, x2, hbits, img[2], img_o, img_qkv, img_of; };   // img: pre-split ffn weight images (forward, backward); img_o: Wo^T; img_qkv: this layer's Wqkv (read by the PREVIOUS layer's feed-forward launch)
struct DecAct { size_t x_in, qkv, o1, z1, st1, astats1, out1, q2, kv2, o2, astats2, z2, st2, out2, h, z3, st3, out3, hbits, img[2], img_o1, img_o2, img_qkv, img_o2f, img_o1f, img_q2, img_q2t; };

struct Plan {
  size_t bytes = 0;
  size_t inp, tar, labels, enc_mask, dec_mask;
  size_t order;            // (B) samples sorted by length, longest first (skf_sample_order): the attention launches deal their workgroups from it
  std::vector<EncAct> enc;
  std::vector<DecAct> dec;
  size_t u, pool_a, emb, pooled, dpooled, cls_logits, cls_probs, pre, logits;
  std::vector<size_t> cb_h, cb_f;       // class buffers: relu output, post-dropout output  (B, lowerdim) each
  size_t dcb[2];
  size_t recon_loss, recon_hit, cls_loss, cls_hit, row_mask, cont_scal;
  size_t gA, gB, gC, dqkv, dh, do_, dpre, dkv2, dq2, demb;
  // Buffers that weight-gradient GEMMs read (dY operands).  Two sets, alternating by layer: the wgrads of a layer are
  // issued together on the side stream at the end of that layer, so their operands must stay untouched until the
  // layer after next starts (every cross-stream event costs ~5 us of dead time on the main stream).
  struct GradSet { size_t dy[3], dh, dq2, dkv2, dqkv; };
  std::vector<GradSet> gs;        // gradient buffers of the backward, n_gs sets in rotation (one per layer where memory allows: see plan())
  int n_gs = 2;
  size_t gemm_ws, gemm_ws_bytes, small_ws, small_ws_bytes;
  size_t emb_sort[2] = {0, 0}, emb_sort_bytes = 0;         // token positions sorted by id (encoder, decoder): skf_embed_sort
  size_t slab_arena, slab_arena_bytes, descs, n_wgrads;   // deferred split-K reduction (eager path)
  size_t ln_part, ln_part_stride;                          // per-LayerNorm dgamma|dbeta partials [5N][g][2d], reduced in the same batch
  size_t bott_part = 0;                                    // expander / pooling gradient partials (see build_plan)
  // KV-cached greedy decode (inference): per-layer self-attention K|V cache (B, L, 2d) + one-row-per-sample step buffers
  std::vector<size_t> dc_cache;
  size_t dc_x[2], dc_q, dc_o, dc_z, dc_out1, dc_out2, dc_h, dc_logits, dc_stats, dc_mask, dc_flags, dc_limit;
  size_t dc_tok, dc_cont, dc_kvnew, dc_dyn;
  size_t live_len = 0, live16 = 0, live32 = 0;   // decoder-side live rows of the step (token mode): per-sample count, block lists   // internal (B, L+1) output image, newest K|V rows, per-call scalars + step index
};

size_t wgrad_ws(int in, int out, int rows) {
  return skf_gemm_workspace_bytes(in, out, rows, skf_gemm_default_splits(in, out, rows), 1);
}

Plan build_plan(const SkfConfig& c) {
  Plan P;
  Bump b;
  const size_t B = c.batch, L = c.seq_len, Ld = c.seq_len - 1, d = c.d_model, F = c.dff, U = c.lowerdim;
  const size_t Me = B * L, Md = B * Ld, H = c.num_heads, f = sizeof(float);
  const size_t E = (has_bott(c) && c.attn_version == 2) ? U : d, Ua = c.attn_version == 2 ? d : U;
  const size_t in_bytes = c.continuous ? B * L * 5 * 4 : B * L * 8;   // (B,L,5) f32 or (B,L) i64
  const size_t Vout = c.continuous ? 5 : (size_t)c.vocab_size;
  P.inp = b.take(in_bytes); P.tar = b.take(in_bytes); P.labels = b.take(B * 8);
  P.enc_mask = b.take(B * L); P.dec_mask = b.take(B * L);
  for (int i = 0; i < c.num_layers; ++i) {
    EncAct a;
    a.x_in = b.take(Me * d * f); a.qkv = b.take(Me * 3 * d * f); a.o = b.take(Me * d * f); a.z1 = b.take(Me * d * f);
    a.st1 = b.take(Me * 2 * f); a.astats = b.take(B * H * L * 2 * f); a.x1 = b.take(Me * d * f);
    a.h = b.take(Me * F * f); a.z2 = b.take(Me * d * f); a.st2 = b.take(Me * 2 * f);
    a.hbits = b.take(std::max(skf_gemm_relu_bits_bytes((int)Me, (int)F, (int)d, c.gemm_precision),        // 0 bytes: no sign-bit path for this shape
                              skf_ffn_relu_bits_bytes((int)Me, (int)d, (int)F, c.gemm_precision)));
    a.img[0] = b.take(skf_ffn_image_bytes((int)d, (int)F, c.gemm_precision)); a.img[1] = b.take(skf_ffn_image_bytes((int)d, (int)F, c.gemm_precision));
    a.img_o = b.take(skf_dense_image_bytes((int)d, (int)d, c.gemm_precision));
    a.img_qkv = b.take(skf_dense_image_bytes((int)d, 3 * (int)d, c.gemm_precision));
    a.img_of = b.take(skf_dense_image_bytes((int)d, (int)d, c.gemm_precision));
    a.x2 = 0;
    P.enc.push_back(a);
  }
  const size_t enc_out = b.take(Me * d * f);
  for (int i = 0; i < c.num_layers; ++i) P.enc[i].x2 = (i + 1 < c.num_layers) ? P.enc[i + 1].x_in : enc_out;
  P.u = b.take(Me * Ua * f); P.pool_a = b.take(B * L * f); P.emb = b.take(B * E * f);
  P.pooled = b.take(B * d * f); P.dpooled = b.take(B * d * f);
  for (int i = 0; i < c.class_buffer_layers; ++i) { P.cb_h.push_back(b.take(B * U * f)); P.cb_f.push_back(b.take(B * U * f)); }
  P.dcb[0] = b.take(B * U * f); P.dcb[1] = b.take(B * U * f);
  P.cls_logits = b.take(B * c.n_classes * f); P.cls_probs = b.take(B * c.n_classes * f);
  P.pre = b.take(Me * E * f);
  for (int i = 0; i < c.num_layers; ++i) {
    DecAct a;
    a.x_in = b.take(Md * d * f); a.qkv = b.take(Md * 3 * d * f); a.o1 = b.take(Md * d * f); a.z1 = b.take(Md * d * f);
    a.st1 = b.take(Md * 2 * f); a.astats1 = b.take(B * H * Ld * 2 * f); a.out1 = b.take(Md * d * f);
    a.q2 = b.take(Md * d * f); a.kv2 = b.take(Me * 2 * d * f); a.o2 = b.take(Md * d * f);   // kv2 = pre (Me,E) . Wkv (E,2d)
    a.astats2 = b.take(B * H * Ld * 2 * f); a.z2 = b.take(Md * d * f); a.st2 = b.take(Md * 2 * f);
    a.out2 = b.take(Md * d * f); a.h = b.take(Md * F * f); a.z3 = b.take(Md * d * f); a.st3 = b.take(Md * 2 * f);
    a.hbits = b.take(std::max(skf_gemm_relu_bits_bytes((int)Md, (int)F, (int)d, c.gemm_precision),
                              skf_ffn_relu_bits_bytes((int)Md, (int)d, (int)F, c.gemm_precision)));
    a.img[0] = b.take(skf_ffn_image_bytes((int)d, (int)F, c.gemm_precision)); a.img[1] = b.take(skf_ffn_image_bytes((int)d, (int)F, c.gemm_precision));
    a.img_o1 = b.take(skf_dense_image_bytes((int)d, (int)d, c.gemm_precision)); a.img_o2 = b.take(skf_dense_image_bytes((int)d, (int)d, c.gemm_precision));
    a.img_qkv = b.take(skf_dense_image_bytes((int)d, 3 * (int)d, c.gemm_precision));
    a.img_o2f = b.take(skf_dense_image_bytes((int)d, (int)d, c.gemm_precision));
    a.img_o1f = b.take(skf_dense_image_bytes((int)d, (int)d, c.gemm_precision)); a.img_q2 = b.take(skf_dense_image_bytes((int)d, (int)d, c.gemm_precision));
    a.img_q2t = b.take(skf_dense_image_bytes((int)d, (int)d, c.gemm_precision));
    a.out3 = 0;
    P.dec.push_back(a);
  }
  const size_t dec_out = b.take(Md * d * f);
  for (int i = 0; i < c.num_layers; ++i) P.dec[i].out3 = (i + 1 < c.num_layers) ? P.dec[i + 1].x_in : dec_out;
  P.logits = b.take(Md * Vout * f);
  P.recon_loss = b.take(Md * f); P.recon_hit = b.take(Md * f); P.cls_loss = b.take(B * f); P.cls_hit = b.take(B * f);
  P.row_mask = b.take(Md * f); P.cont_scal = b.take(64);
  P.gA = b.take(Me * d * f); P.gB = b.take(Me * d * f); P.gC = b.take(Me * d * f);
  P.dqkv = b.take(Me * 3 * d * f); P.dh = b.take(Me * F * f); P.do_ = b.take(Me * d * f);
  P.dpre = b.take(Me * E * f); P.dkv2 = b.take(Me * 2 * d * f); P.dq2 = b.take(Md * d * f); P.demb = b.take(B * E * f);
  // A layer's weight gradients (side stream) read its dy / dh / dq|k|v buffers long after the main stream has moved on, so the buffers
  // rotate.  With two sets the main stream waits for the group of two layers ago in front of every layer - finished long since, but a
  // wait in the queue costs the waiting stream ~6 us whether or not it has to wait (tools/wait_cost.py).  One set per layer: no buffer
  // is written twice in a step and those waits are gone (cfg 2: 8 x 170 MB); above 8 GB the sets fall back to two.
  {
    const size_t per_set = (3 * Me * d + Me * F + Md * d + Me * 2 * d + Me * 3 * d) * f;
    static const bool two_sets = skf_knob("SKF_TWO_GRAD_SETS") && skf_knob("SKF_TWO_GRAD_SETS")[0] == '1';      // (measurement builds)
    const size_t layers = (size_t)c.num_layers * (do_recon(c) ? 2 : 1);
    P.n_gs = (two_sets || per_set * layers > ((size_t)8 << 30) || layers < 2) ? 2 : (int)layers;
    P.gs.resize(P.n_gs);
  }
  for (int k = 0; k < P.n_gs; ++k) {
    for (int j = 0; j < 3; ++j) P.gs[k].dy[j] = b.take(Me * d * f);
    P.gs[k].dh = k == 0 ? P.dh : b.take(Me * F * f);
    P.gs[k].dq2 = k == 0 ? P.dq2 : b.take(Md * d * f);
    P.gs[k].dkv2 = k == 0 ? P.dkv2 : b.take(Me * 2 * d * f);
    P.gs[k].dqkv = k == 0 ? P.dqkv : b.take(Me * 3 * d * f);
  }
  size_t g = 0;
  auto mx = [&](size_t v) { if (v > g) g = v; };
  mx(wgrad_ws(d, 3 * d, Me)); mx(wgrad_ws(d, d, Me)); mx(wgrad_ws(d, F, Me)); mx(wgrad_ws(F, d, Me));
  mx(wgrad_ws((int)E, 2 * d, Me)); mx(wgrad_ws(d, (int)Vout, Md));
  if (has_bott(c)) {
    mx(wgrad_ws(d, (int)Ua, Me));
    mx(wgrad_ws((int)E, c.n_classes, B)); mx(wgrad_ws(U, c.n_classes, B)); mx(wgrad_ws(d, U, B)); mx(wgrad_ws((int)E, U, B)); mx(wgrad_ws(U, U, B));
  }
  P.gemm_ws_bytes = g; P.gemm_ws = b.take(g);
  P.n_wgrads = 4 + 11 * (size_t)c.num_layers + (size_t)c.class_buffer_layers + 5 * (size_t)c.num_layers + 3;   // + one entry per LayerNorm + expander (2) / pooling (1) partials
  // per-sample partials of the expander's kernel / bias gradients [2][B][L] and of the pooling scorer's V gradient [B][Ua]: column
  // sums in the batched reduction instead of three one-workgroup launches on the main stream between the decoder and encoder backward
  P.bott_part = b.take((2 * B * L + B * 4096) * f);
  P.ln_part_stride = (skf_layernorm_bwd_workspace_bytes((int)Me, (int)d) + 255) & ~(size_t)255;
  P.ln_part = b.take(5 * (size_t)c.num_layers * P.ln_part_stride);
  P.slab_arena_bytes = P.n_wgrads * ((g + 255) & ~(size_t)255);
  P.slab_arena = b.take(P.slab_arena_bytes);
  P.descs = b.take(P.n_wgrads * sizeof(SkfReduceDesc));
  size_t s = skf_layernorm_bwd_workspace_bytes((int)Me, (int)d);
  if (B * Ua * f > s) s = B * Ua * f;
  if (2 * B * L * f > s) s = 2 * B * L * f;
  if (c.continuous && skf_embed_continuous_bwd_workspace_bytes((int)Me, (int)d) > s) s = skf_embed_continuous_bwd_workspace_bytes((int)Me, (int)d);
  P.small_ws_bytes = s; P.small_ws = b.take(s);
  if (!c.continuous && c.vocab_size <= 12288 && c.d_model <= 512 && !skf_knob("SKF_NO_EMBED_SORT")) {   // (the sorted kernel's partial slab is sized for rows of <= 512 floats)
    P.emb_sort_bytes = (skf_embed_sort_workspace_bytes((int)B, (int)L, c.vocab_size) + 255) & ~(size_t)255;
    P.emb_sort[0] = b.take(P.emb_sort_bytes); P.emb_sort[1] = b.take(P.emb_sort_bytes);
  }
  for (int i = 0; i < c.num_layers; ++i) P.dc_cache.push_back(b.take(B * L * 2 * d * f));
  P.dc_x[0] = b.take(B * d * f); P.dc_x[1] = b.take(B * d * f); P.dc_q = b.take(B * d * f); P.dc_o = b.take(B * d * f);
  P.dc_z = b.take(B * d * f); P.dc_out1 = b.take(B * d * f); P.dc_out2 = b.take(B * d * f); P.dc_h = b.take(B * F * f);
  P.dc_logits = b.take(B * Vout * f); P.dc_stats = b.take(B * 2 * f); P.dc_mask = b.take(B * (L + 1));
  P.dc_flags = b.take((B + 16) * sizeof(int)); P.dc_limit = b.take(B * sizeof(int));
  P.dc_tok = b.take(B * (L + 1) * 8); P.dc_cont = b.take(B * (L + 1) * 5 * f); P.dc_kvnew = b.take(B * 2 * d * f);
  P.dc_dyn = b.take(64);
  P.live_len = b.take(B * sizeof(int));
  P.order = b.take(B * sizeof(int));
  P.live16 = b.take(skf_row_blocks_bytes(B * (L - 1), 16)); P.live32 = b.take(skf_row_blocks_bytes(B * (L - 1), 32));
  P.bytes = b.off;
  return P;
}

}  // namespace

#define SKF_BF16_PART 1
#include "skf_model_bf16.inc"
#undef SKF_BF16_PART

struct SkfModel {
  SkfConfig cfg;
  uint32_t flags = 0;                // skf_model_set_flags
  bool no_ln_fuse = false, no_relu_bits = false;   // a fused entry answered SKF_EUNSUPPORTED once: this model takes the general pair
  bool ffn_fused = false;            // the feed-forward blocks run as one launch per direction (skf_ffn_fused.hip); set per forward
  bool masks_staged = false;         // the padding masks of this call were written by its staging launch (stage_inputs)
  hipEvent_t last_ready = nullptr;   // ffn_ln_bwd: the event attached to its fused launch (valid until the caller's next main-stream launch)
  Layout lay;
  Plan plan;
  Plan16 p16;                        // bf16 path (cfg.act_dtype == SKF_ACT_BF16): its own workspace plan
  bool bf16 = false;
  float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr, *metrics = nullptr;
  const float* pos = nullptr;
  char* ws = nullptr;
  void* state = nullptr;
  hipGraphExec_t g_fb = nullptr, g_opt = nullptr, g_dec = nullptr;   // g_dec: one greedy-decode step
  long long dec_dyn_host[2] = {0, 0};
  float g_opt_scale = 0.f;
  // weight-gradient GEMMs run on a side stream, off the dgrad critical path
  hipStream_t side = nullptr;
  hipEvent_t fork_event = nullptr, join_event = nullptr;      // use_graph = 2: the side stream's entry into / exit from the capture
  hipEvent_t inputs_staged = nullptr;                         // recorded behind the staging copies of every call (skf_model_wait_inputs_staged)
  bool inputs_staged_valid = false;
  std::vector<hipEvent_t> events;
  size_t next_event = 0;
  // Side-stream events carry a sequence number (their record order on the in-order side stream): once the main stream has
  // waited for event k, every event <= k is complete too - later waits for those are dropped (a decoder layer's seven dY
  // buffers share one `done` event: one barrier packet on the main stream instead of seven, ~5 us each)
  struct SideEvent { hipEvent_t e; long seq; };
  long side_seq = 0, side_waited = 0;
  bool no_wait_dedupe = skf_knob("SKF_NO_WAIT_DEDUPE") && skf_knob("SKF_NO_WAIT_DEDUPE")[0] == '1';     // A/B knob
  std::map<const void*, SideEvent> pending_readers;    // buffer -> completion event of its last side-stream reader
  // kind 0: dW = X^T dY (+ bias grad); kind 1: an input gradient nobody on the main stream needs soon (dx (+)= dY W^T)
  struct QueuedWgrad { DenseP w; const float* x; int ldx; const float* dy; int lddy; int rows; int kind = 0; float* dx = nullptr; int lddx = 0; int accumulate = 0; const int* blocks32 = nullptr; };
  // Live row blocks of the decoder-side backward (skf_row_blocks.hip): set while the decoder layers' gradients are issued,
  // consulted by dense_dgrad / dense_wgrad for problems with exactly `live_rows` rows; null = every row is visited
  const int* live16 = nullptr; const int* live32 = nullptr; int live_rows = 0;
  const int* order = nullptr;        // this step's samples sorted by length (run_forward), or null
  hipEvent_t pre_ready = nullptr, masks_ready = nullptr;    // train step: forward_preamble ran on the side stream; the forward waits for the masks before its first attention, for the images behind it
  bool lists_built = false;             // this step's lists are in P.live16 / P.live32 (issued, not necessarily complete)
  std::map<const void*, SideEvent> pending_writers;    // buffers a side-stream dgrad still writes
  std::vector<QueuedWgrad> wq;                         // wgrads of the current layer, not yet issued
  std::vector<QueuedWgrad> wq_held;                    // the PREVIOUS layer's group, held back until the next layer's first kernel is queued (hold_wgrads)
  bool side_used = false;
  std::vector<SkfReduceDesc> descs;     // one per wgrad of the step, in launch order
  bool descs_uploaded = false;
  size_t slab_cursor = 0, desc_cursor = 0, ln_cursor = 0;
  int reduce_blocks = 0;
  // gradient buckets (data parallelism): the flat gradient buffer becomes final in two pieces, in production order -
  // [dec_off, total) after the decoder backward, [0, dec_off) at the end; an event marks each piece complete so that
  // its all-reduce can start while the encoder backward / the previous piece's optimizer sweep still runs
  size_t phase_desc_begin = 0;
  hipEvent_t bucket_ready[2] = {nullptr, nullptr};
  int n_buckets = 1;

  hipEvent_t new_event() {
    if (next_event == events.size()) {
      hipEvent_t e = nullptr;
      // Events that only order the library's own two streams on ONE device: a device-scope release is all the waiter needs
      // (the default system-scope fence of hipEventRecord writes caches back for host / peer visibility).  SKF_EVENT_SCOPE=system
      // restores the default for A/B measurements.  (The gradient-bucket events handed to the caller keep the default.)
      static const bool sys_scope = skf_knob("SKF_EVENT_SCOPE") && skf_knob("SKF_EVENT_SCOPE")[0] == 's';
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming | (sys_scope ? 0u : hipEventReleaseToDevice)) != hipSuccess) return nullptr;
      events.push_back(e);
    }
    return events[next_event++];
  }
  std::map<std::string, std::pair<size_t, std::pair<int, int>>> named;

  template <typename T> T* at(size_t off) const { return reinterpret_cast<T*>(ws + off); }
  float* P(size_t off) const { return params + off; }
  float* G(size_t off) const { return grads + off; }
};

#define SKF_TRY(call)            \
  do {                           \
    int rc__ = (call);           \
    if (rc__ != SKF_OK) return rc__; \
  } while (0)

namespace {

int dense_fwd(SkfModel* M, const DenseP& w, const float* x, int rows, float* y, int act, hipStream_t s) {
  return skf_gemm_f32(1, 0, rows, w.out, w.in, x, w.in, M->P(w.w), w.ld, y, w.out, M->P(w.b), act, nullptr, 0, 0, 1,
                      nullptr, 0, nullptr, 0, M->cfg.gemm_precision, s);
}
// y = Dense(a); z = x + dropout(y); out = LayerNorm(z): one launch where the fused kernel exists (the attention output projection
// at d_model = 128 in the split-arithmetic modes), else the Dense launch followed by the LayerNorm launch.  SKF_NO_LN_FUSE=1: A/B knob.
int dense_ln_fwd(SkfModel* M, const DenseP& w, const float* a, int rows, const float* x, float* z, const LnP& ln, float* out,
                 float* stats, float rate, unsigned site, hipStream_t s) {
  static const bool fuse_off = skf_knob("SKF_NO_LN_FUSE") && skf_knob("SKF_NO_LN_FUSE")[0] == '1';
  if (!fuse_off && !M->no_ln_fuse && skf_gemm_ln_residual_supported(rows, w.out, w.in, M->cfg.gemm_precision)) {
    const int rc = skf_gemm_ln_residual_f32(rows, w.out, w.in, a, w.in, M->P(w.w), w.ld, M->P(w.b), x, M->P(ln.g), M->P(ln.b), z, out, stats,
                                            rate, site, M->state, M->cfg.gemm_precision, s);
    // the shape test above does not see pitches / alignment: a launch the fused entry declines takes the general pair (from now on)
    if (rc != SKF_EUNSUPPORTED) return rc;
    M->no_ln_fuse = true;
  }
  SKF_TRY(dense_fwd(M, w, a, rows, z, 0, s));
  return skf_layernorm_residual_fwd(x, z, M->P(ln.g), M->P(ln.b), out, stats, rows, w.out, rate, site, M->state, s);
}
// sign-bit buffer of an ffn hidden tensor (rows x dff from d inputs), or null when the shape has no such path / SKF_NO_RELU_BITS=1
void* hbits_of(SkfModel* M, size_t off, int rows) {
  static const bool bits_off = skf_knob("SKF_NO_RELU_BITS") && skf_knob("SKF_NO_RELU_BITS")[0] == '1';
  if (M->ffn_fused) return M->at<char>(off);      // (the fused block always writes / reads its own sign-bit words)
  if (bits_off || M->no_relu_bits || !skf_gemm_relu_bits_bytes(rows, M->cfg.dff, M->cfg.d_model, M->cfg.gemm_precision)) return nullptr;
  return M->at<char>(off);
}
// ffn dense1 (relu): also leaves the sign bits of the hidden tensor for the backward when the shape has that path (bits != null)
int dense_fwd_relu_bits(SkfModel* M, const DenseP& w, const float* x, int rows, float* y, void* bits, hipStream_t s) {
  if (!bits) return dense_fwd(M, w, x, rows, y, 1, s);
  const int rc = skf_gemm_f32_bits(1, 0, rows, w.out, w.in, x, w.in, M->P(w.w), w.ld, y, w.out, M->P(w.b), 1, nullptr, 0, 0, 1,
                                   nullptr, 0, nullptr, 0, M->cfg.gemm_precision, nullptr, 0, bits, nullptr, s);
  if (rc != SKF_EUNSUPPORTED) return rc;
  // the weight-stationary dispatch declined (pitch / alignment): the general kernels, and the backward of this and every later
  // step reads the hidden tensor (relu_src) instead of sign bits nobody wrote - hbits_of() answers null from here on
  M->no_relu_bits = true;
  return dense_fwd(M, w, x, rows, y, 1, s);
}
// strided-input variant (x has row stride ldx)
int dense_fwd_ld(SkfModel* M, const DenseP& w, const float* x, int ldx, int rows, float* y, int ldy, int act, hipStream_t s) {
  return skf_gemm_f32(1, 0, rows, w.out, w.in, x, ldx, M->P(w.w), w.ld, y, ldy, M->P(w.b), act, nullptr, 0, 0, 1,
                      nullptr, 0, nullptr, 0, M->cfg.gemm_precision, s);
}
int dense_wgrad_on(SkfModel* M, const DenseP& w, const float* x, int ldx, const float* dy, int lddy, int rows, hipStream_t s) {
  const int splits = skf_gemm_default_splits(w.in, w.out, rows);
  return skf_gemm_f32(0, 0, w.in, w.out, rows, x, ldx, dy, lddy, M->G(w.w), w.ld, nullptr, 0, nullptr, 0, 0, splits,
                      M->G(w.b), 0, M->at<char>(M->plan.gemm_ws), M->plan.gemm_ws_bytes, M->cfg.gemm_precision, s);
}
int issue_wgrads(SkfModel* M, hipStream_t s, hipEvent_t ready_recorded = nullptr, bool on_main = false);
int issue_held_wgrads(SkfModel* M, hipStream_t s, hipEvent_t ready_recorded = nullptr);
// The "main stream has reached this point" event of a held weight-gradient group as the COMPLETION SIGNAL of the launch in front of it
// (skf_common.h: SKF_LAUNCH_TAIL) instead of a packet of its own: park_ready() before the launcher, take_ready() behind it - the
// event when the launcher attached it, null when it did not (or nothing is held / the step is being captured).
hipEvent_t park_ready(SkfModel* M);
hipEvent_t take_ready(hipEvent_t parked) {
  const bool attached = parked && skf_tls_stop_event == nullptr;
  skf_tls_stop_event = nullptr;
  return attached ? parked : nullptr;
}
// Main-stream kernels that overwrite `buf` must first wait for the side-stream wgrad that still reads it
// (a wgrad that is still queued is issued first; with the alternating gradient-buffer sets this is the rare case).
int before_write(SkfModel* M, const void* buf, hipStream_t s) {
  for (const auto* qs : {&M->wq_held, &M->wq})
    for (const auto& q : *qs)
      if (q.dy == buf || q.x == buf) { SKF_TRY(issue_wgrads(M, s)); break; }
  auto it = M->pending_readers.find(buf);
  if (it == M->pending_readers.end()) return SKF_OK;
  if (it->second.seq > M->side_waited || M->no_wait_dedupe) {
    SKF_HIP(hipStreamWaitEvent(s, it->second.e, 0));
    M->side_waited = std::max(M->side_waited, it->second.seq);
  }
  M->pending_readers.erase(it);
  return SKF_OK;
}
// dW = X^T dY (+ bias grad).  Eager path: queued, and issued per layer on the side stream by issue_wgrads().
int dense_wgrad(SkfModel* M, const DenseP& w, const float* x, int ldx, const float* dy, int lddy, int rows, hipStream_t s) {
  if (!M->side) return dense_wgrad_on(M, w, x, ldx, dy, lddy, rows, s);
  SkfModel::QueuedWgrad q{w, x, ldx, dy, lddy, rows};
  if (M->live32 && rows == M->live_rows) q.blocks32 = M->live32;
  M->wq.push_back(q);
  return SKF_OK;
}
// Main-stream kernels that read `buf` first wait for the side-stream dgrad that writes it.
int before_read(SkfModel* M, const void* buf, hipStream_t s) {
  for (const auto* qs : {&M->wq_held, &M->wq})
    for (const auto& q : *qs)
      if (q.kind == 1 && q.dx == buf) { SKF_TRY(issue_wgrads(M, s)); break; }
  auto it = M->pending_writers.find(buf);
  if (it == M->pending_writers.end()) return SKF_OK;
  if (it->second.seq > M->side_waited || M->no_wait_dedupe) {
    SKF_HIP(hipStreamWaitEvent(s, it->second.e, 0));
    M->side_waited = std::max(M->side_waited, it->second.seq);
  }
  M->pending_writers.erase(it);
  return SKF_OK;
}
// Issue the queued wgrads on the side stream: ONE ready event (everything queued on `s` so far is complete before they
// start) and ONE done event for the whole group; they are serialized among themselves and joined before the optimizer.
// The fused feed-forward backward is the FIRST kernel of a layer's backward, and its workgroups (147 KB of LDS, two waves per SIMD)
// cannot share a CU with a weight-gradient workgroup (66 KB, 272 registers): issued together - the previous layer's group on the side
// stream, the block on the main stream - they ran one after the other (134 + 136 us where 45 + 100 were expected, per layer).  So a
// layer's group is HELD at the end of the layer and goes out right behind the next layer's first launch: it then runs beside the
// LayerNorm / projection / attention kernels of that layer, which share CUs with it well.
hipEvent_t park_ready(SkfModel* M) {
  static const bool off = skf_knob("SKF_NO_STOP_EVENTS") && skf_knob("SKF_NO_STOP_EVENTS")[0] == '1';   // (measurement builds only)
  if (off || M->wq_held.empty() || !M->side || g_capturing) return nullptr;
  hipEvent_t e = M->new_event();
  skf_tls_stop_event = e;
  return e;
}
// the same for a group that is issued right behind ONE call's last launch (no held group needed): a fresh event, or null
hipEvent_t park_fresh(SkfModel* M) {
  static const bool off = skf_knob("SKF_NO_STOP_EVENTS") && skf_knob("SKF_NO_STOP_EVENTS")[0] == '1';   // (measurement builds only)
  if (off || !M->side || g_capturing) return nullptr;
  hipEvent_t e = M->new_event();
  skf_tls_stop_event = e;
  return e;
}
// side-stream launches must never pick up an event that is parked for the main stream's next launch
struct ParkedEventGuard {
  hipEvent_t saved;
  ParkedEventGuard() : saved(skf_tls_stop_event) { skf_tls_stop_event = nullptr; }
  ~ParkedEventGuard() { skf_tls_stop_event = saved; }
};
int issue_held_wgrads(SkfModel* M, hipStream_t s, hipEvent_t ready_recorded) {
  if (M->wq_held.empty()) return SKF_OK;
  std::vector<SkfModel::QueuedWgrad> cur;
  cur.swap(M->wq);
  M->wq.swap(M->wq_held);
  const int rc = issue_wgrads(M, s, ready_recorded);
  M->wq.swap(cur);
  return rc;
}
int hold_wgrads(SkfModel* M, hipStream_t s) {
  if (!M->wq_held.empty()) SKF_TRY(issue_wgrads(M, s));       // (never two groups held)
  M->wq_held.swap(M->wq);
  return SKF_OK;
}
// on_main: the queued group runs on the MAIN stream, in place (no events, no hop) - for the one weight gradient at the very end of
// the backward that the side stream would finish last (see run_backward)
int issue_wgrads(SkfModel* M, hipStream_t s, hipEvent_t ready_recorded, bool on_main) {
  ParkedEventGuard guard;                                      // (this function may run INSIDE a parked call: before_write)
  if (!M->wq_held.empty()) {                                   // the held group first, as a group of its own
    std::vector<SkfModel::QueuedWgrad> cur;
    cur.swap(M->wq);
    M->wq.swap(M->wq_held);
    const int rc = issue_wgrads(M, s);
    M->wq.swap(cur);
    if (rc != SKF_OK) return rc;
  }
  if (M->wq.empty()) return SKF_OK;
  std::vector<SkfModel::QueuedWgrad> group;
  group.swap(M->wq);
  hipStream_t ws = on_main ? s : M->side;                      // the stream the group runs on
  hipEvent_t ready = nullptr, done = nullptr;
  if (!on_main) {
    ready = ready_recorded ? ready_recorded : M->new_event(); done = M->new_event();
    SKF_CHECK_ARG(ready && done, "event allocation failed");
    if (!ready_recorded) SKF_HIP(hipEventRecord(ready, s));   // (else: already the completion signal of the launch in front of this call)
    SKF_HIP(hipStreamWaitEvent(M->side, ready, 0));
  }
  // deferred input gradients first, with their own completion event: their reader must not wait for the weight gradients
  hipEvent_t dgrad_done = nullptr;
  for (const auto& q : group) {
    if (q.kind != 1) continue;
    const DenseP& w = q.w;
    SKF_TRY(skf_gemm_f32(1, 1, q.rows, w.in, w.out, q.dy, q.lddy, M->P(w.w), w.ld, q.dx, q.lddx, nullptr, 0, nullptr, 0,
                         q.accumulate, 1, nullptr, 0, nullptr, 0, M->cfg.gemm_precision, ws));
    if (!dgrad_done && !on_main) { dgrad_done = M->new_event(); SKF_CHECK_ARG(dgrad_done, "event allocation failed"); }
  }
  long dgrad_seq = 0;
  if (dgrad_done) { SKF_HIP(hipEventRecord(dgrad_done, M->side)); dgrad_seq = ++M->side_seq; }
  // the large problems of the group: partial tiles by ONE grouped launch (up to 8 problems each); every slab of the phase is
  // reduced by one launch in flush_wgrads()
  std::vector<SkfWgradProblem> probs;
  std::vector<const SkfModel::QueuedWgrad*> prob_q;
  for (const auto& q : group) {
    const DenseP& w = q.w;
    if (q.kind == 1) continue;
    if ((double)w.in * w.out * q.rows <= 33554432.0) {
      // batch-sized problems (classifier, class buffers, SelfAttnV2 projection): one small-GEMM launch, no split-K slab
      SKF_TRY(dense_wgrad_on(M, w, q.x, q.ldx, q.dy, q.lddy, q.rows, ws));
      continue;
    }
    const int splits = skf_gemm_default_splits(w.in, w.out, q.rows);
    const size_t bytes = (skf_gemm_workspace_bytes(w.in, w.out, q.rows, splits, 1) + 255) & ~(size_t)255;
    SKF_CHECK_ARG(M->slab_cursor + bytes <= M->plan.slab_arena_bytes && M->desc_cursor + probs.size() < M->plan.n_wgrads, "slab arena exhausted");
    SkfWgradProblem pr{};
    pr.A = q.x; pr.B = q.dy; pr.slab = M->at<float>(M->plan.slab_arena + M->slab_cursor); pr.slab_bytes = bytes;
    pr.row_blocks = q.blocks32; pr.row_block_rows = 32;
    pr.M = w.in; pr.N = w.out; pr.K = q.rows; pr.lda = q.ldx; pr.ldb = q.lddy; pr.splits = splits; pr.with_bias_grad = 1;
    M->slab_cursor += bytes;
    probs.push_back(pr); prob_q.push_back(&q);
  }
  // (measured: grouping the 1-3 GFLOP problems of cfg 2 shortens the step by 0.6 %, grouping the 3-13 GFLOP ones of cfg 3 lengthens
  //  it by 1.3 % - those fill the chip for ~90 us each and gain nothing from sharing a grid)
  bool small = true;
  for (const auto& pr : probs) small = small && 2.0 * pr.M * pr.N * pr.K < 4e9;
  const size_t gmax = small ? 8 : 1;
  for (size_t b0 = 0; b0 < probs.size(); b0 += gmax) {
    const int nb = (int)std::min<size_t>(gmax, probs.size() - b0);
    SKF_TRY(skf_gemm_wgrad_partial_group(probs.data() + b0, nb, M->cfg.gemm_precision, ws));
  }
  for (size_t i = 0; i < probs.size(); ++i) {
    const DenseP& w = prob_q[i]->w;
    SkfReduceDesc d;
    d.slab = probs[i].slab; d.C = M->G(w.w); d.bias_grad = M->G(w.b); d.splits = probs[i].splits_used; d.M = w.in; d.N = w.out; d.ldc = w.ld;
    d.block_begin = M->reduce_blocks; d.pad = 0;
    if (!M->descs_uploaded) M->descs.push_back(d);
    else {
      const SkfReduceDesc& o = M->descs[M->desc_cursor];
      SKF_CHECK_ARG(o.slab == d.slab && o.C == d.C && o.splits == d.splits && o.block_begin == d.block_begin, "wgrad sequence changed between steps");
    }
    M->reduce_blocks += skf_splitk_reduce_blocks(w.in, w.out);
    M->desc_cursor += 1;
  }
  M->side_used = true;                                         // (the slabs are reduced by the batched launch either way)
  if (on_main) return SKF_OK;                                  // same stream as every later reader / writer of the operands: nothing to track
  SKF_HIP(hipEventRecord(done, M->side));
  const long done_seq = ++M->side_seq;
  for (const auto& q : group) {
    M->pending_readers[q.dy] = SkfModel::SideEvent{done, done_seq};
    if (q.x) M->pending_readers[q.x] = SkfModel::SideEvent{done, done_seq};
    if (q.kind == 1) M->pending_writers[q.dx] = SkfModel::SideEvent{dgrad_done, dgrad_seq};
  }
  return SKF_OK;
}
// Reduce the split-K partials of the wgrads issued since the last flush (one batched launch on the side stream) and
// mark gradient bucket `bucket` complete.  final = the main stream waits for the side stream (before the optimizer).
int flush_wgrads(SkfModel* M, hipStream_t s, int bucket, bool final, bool issue_queued = true, hipEvent_t main_here = nullptr) {
  if (issue_queued) SKF_TRY(issue_wgrads(M, s));      // (false: reduce what has been issued; queued / held groups stay where they are)
  const size_t begin = M->phase_desc_begin, end = M->desc_cursor;
  hipStream_t ready_on = s;
  bool bucket_recorded = false;
  if (M->side && M->side_used && end > begin) {
    if (!M->descs_uploaded) {      // the launch sequence is fixed: descriptors are built and uploaded once (first step)
      SKF_HIP(hipMemcpy(M->at<SkfReduceDesc>(M->plan.descs) + begin, M->descs.data() + begin,
                        (end - begin) * sizeof(SkfReduceDesc), hipMemcpyHostToDevice));
      if (final) M->descs_uploaded = true;
    }
    // LayerNorm partials and the embedding gradients of this bucket were written by the main stream: the batched
    // reduction (wgrad slabs + LayerNorm partials) and the bucket-ready event are ordered after both streams
    static const bool tail_on_main = !(skf_knob("SKF_TAIL_REDUCE_SIDE") && skf_knob("SKF_TAIL_REDUCE_SIDE")[0] == '1');
    if (final && tail_on_main) {
      // end of the backward: the optimizer waits for this reduction anyway, so it runs on the MAIN stream behind ONE hop
      // (side -> main after the last weight gradient) instead of two (main -> side for the partials, side -> main for the result)
      hipEvent_t e = M->new_event();
      SKF_CHECK_ARG(e, "event allocation failed");
      SKF_HIP(hipEventRecord(e, M->side));
      SKF_HIP(hipStreamWaitEvent(s, e, 0));
      // the bucket-ready event rides on the reduction launch as its completion signal (skf_common.h: SKF_LAUNCH_TAIL)
      hipEvent_t br = (bucket >= 0 && !g_capturing) ? M->bucket_ready[bucket] : nullptr;
      skf_tls_stop_event = br;
      const int rc = skf_splitk_reduce_batch(M->at<SkfReduceDesc>(M->plan.descs) + begin, (int)(end - begin), M->reduce_blocks, s);
      bucket_recorded = br && skf_tls_stop_event == nullptr;
      skf_tls_stop_event = nullptr;
      SKF_TRY(rc);
      ready_on = s;
    } else {
    hipEvent_t em = main_here ? main_here : M->new_event();      // (main_here: already the completion signal of the main stream's last launch)
    SKF_CHECK_ARG(em, "event allocation failed");
    if (!main_here) SKF_HIP(hipEventRecord(em, s));
    SKF_HIP(hipStreamWaitEvent(M->side, em, 0));
    SKF_TRY(skf_splitk_reduce_batch(M->at<SkfReduceDesc>(M->plan.descs) + begin, (int)(end - begin), M->reduce_blocks, M->side));
    ready_on = M->side;
    if (final) {
      hipEvent_t e = M->new_event();
      SKF_CHECK_ARG(e, "event allocation failed");
      SKF_HIP(hipEventRecord(e, M->side));
      SKF_HIP(hipStreamWaitEvent(s, e, 0));
      ready_on = s;
    }
    }
  }
  if (bucket >= 0 && M->bucket_ready[bucket] && !bucket_recorded) SKF_HIP(hipEventRecord(M->bucket_ready[bucket], ready_on));   // bucket < 0: an intermediate reduction
  M->phase_desc_begin = end;
  M->reduce_blocks = 0;                 // block numbering of the next batch starts again at 0
  if (final) {
    M->pending_readers.clear();
    M->pending_writers.clear();
    M->side_used = false;
    M->side_waited = M->side_seq;        // the main stream has joined the side stream: every event recorded so far is behind it
  }
  return SKF_OK;
}
int dense_dgrad(SkfModel* M, const DenseP& w, const float* dy, int lddy, int rows, float* dx, int lddx, int accumulate,
                const float* relu_src, int ld_relu, hipStream_t s, const void* relu_bits = nullptr) {
  SKF_TRY(before_write(M, dx, s));
  const int* blocks = (M->live16 && rows == M->live_rows) ? M->live16 : nullptr;
  if (relu_bits)      // relu'(hidden) from the sign bits the forward left (skf_gemm_f32_bits): the hidden tensor is not re-read
    return skf_gemm_f32_bits(1, 1, rows, w.in, w.out, dy, lddy, M->P(w.w), w.ld, dx, lddx, nullptr, 0, nullptr, 0,
                             accumulate, 1, nullptr, 0, nullptr, 0, M->cfg.gemm_precision, blocks, 16, nullptr, relu_bits, s);
  return skf_gemm_f32_rows(1, 1, rows, w.in, w.out, dy, lddy, M->P(w.w), w.ld, dx, lddx, nullptr, 0, relu_src, ld_relu,
                           accumulate, 1, nullptr, 0, nullptr, 0, M->cfg.gemm_precision, blocks, 16, s);
}

// dx (+)= dY W^T for a dx that the main stream reads much later (the encoder-output gradient sent back by the decoder's
// cross-attention K/V projections): queued behind this layer's weight gradients on the side stream; the reader calls
// before_read(dx).  Successive deferred writers of one dx stay in order (one side stream).
int dense_dgrad_deferred(SkfModel* M, const DenseP& w, const float* dy, int lddy, int rows, float* dx, int lddx, int accumulate,
                         hipStream_t s) {
  static const bool off = skf_knob("SKF_NO_DEFERRED_DGRAD") != nullptr;
  if (!M->side || off) return dense_dgrad(M, w, dy, lddy, rows, dx, lddx, accumulate, nullptr, 0, s);
  SkfModel::QueuedWgrad q{w, nullptr, 0, dy, lddy, rows};
  q.kind = 1; q.dx = dx; q.lddx = lddx; q.accumulate = accumulate;
  M->wq.push_back(q);
  return SKF_OK;
}

// site ids follow oracle.dropout_sites()
inline unsigned site_enc_embed() { return 0; }
inline unsigned site_enc(int layer, int j) { return 1 + 2 * layer + j; }
inline unsigned site_dec_embed(int N) { return 1 + 2 * N; }
inline unsigned site_class(int N, int i) { return 2 + 5 * N + i; }   // after the 1 + 2N encoder and 1 + 3N decoder sites
inline unsigned site_dec(int N, int layer, int j) { return 2 + 2 * N + 3 * layer + j; }

// classify_from_embedding (models/sketchformer.py:183-199): optional Dense(lowerdim, relu) + Dropout(class_dropout)
// buffers, then the classify layer -> logits in P.cls_logits (its softmax is fused into the CE kernel).
int classify_fwd(SkfModel* M, bool training, hipStream_t s) {
  const SkfConfig& c = M->cfg;
  const Layout& L = M->lay;
  const Plan& P = M->plan;
  const float* fc = M->at<float>(P.emb);
  for (int i = 0; i < c.class_buffer_layers; ++i) {
    float* h = M->at<float>(P.cb_h[i]);
    float* fdrop = M->at<float>(P.cb_f[i]);
    SKF_TRY(dense_fwd(M, L.cbuf[i], fc, c.batch, h, 1, s));
    const float r = training ? c.class_dropout : 0.f;
    SKF_TRY(skf_dropout(h, fdrop, (size_t)c.batch * c.lowerdim, r, site_class(c.num_layers, i), M->state, s));
    fc = fdrop;
  }
  return dense_fwd(M, L.cls, fc, c.batch, M->at<float>(P.cls_logits), 0, s);
}

// The feed-forward block as one launch per direction (skf_ffn_fused.hip) where that kernel exists (d_model 128, dff 512, split
// arithmetic) unless SKF_MODEL_FFN_LAUNCHES asks for the separate launches.  Its pre-split weight images are rebuilt from the fp32
// masters at the start of every forward (one or two launches for all layers: whoever changed the weights - the optimizer, a
// checkpoint restore, a test - did not have to tell the library).
bool ffn_fused_on(const SkfModel* M) {
  const SkfConfig& c = M->cfg;
  static const bool off = skf_knob("SKF_NO_FFN_FUSE") && skf_knob("SKF_NO_FFN_FUSE")[0] == '1';   // (measurement builds only)
  return !off && !(M->flags & SKF_MODEL_FFN_LAUNCHES) && skf_ffn_fused_supported(c.batch * c.seq_len, c.d_model, c.dff, c.gemm_precision) &&
         skf_ffn_image_bytes(c.d_model, c.dff, c.gemm_precision) > 0;
}
int build_ffn_images(SkfModel* M, bool with_backward, bool encoder_only, hipStream_t s) {
  const SkfConfig& c = M->cfg;
  const Layout& L = M->lay;
  const Plan& P = M->plan;
  const int d = c.d_model, F = c.dff;
  const size_t half = skf_ffn_image_bytes(d, F, c.gemm_precision) / 2;
  std::vector<const float*> src; std::vector<int> ld, tr, K, N; std::vector<void*> img;
  auto one = [&](const DenseP& w, int t, int k, int n, char* im) {
    src.push_back(M->P(w.w)); ld.push_back(w.ld); tr.push_back(t); K.push_back(k); N.push_back(n); img.push_back(im);
  };
  auto ffn = [&](const DenseP& f1, const DenseP& f2, const size_t (&im)[2]) {
    one(f1, 0, d, F, M->at<char>(im[0])); one(f2, 0, F, d, M->at<char>(im[0]) + half);                  // forward: B1 = W1, B2 = W2
    if (with_backward) { one(f2, 1, d, F, M->at<char>(im[1])); one(f1, 1, F, d, M->at<char>(im[1]) + half); }   // backward: B1 = W2^T, B2 = W1^T
  };
  const bool dec = !encoder_only && do_recon(c);
  for (int i = 0; i < c.num_layers; ++i) {
    ffn(L.enc[i].f1, L.enc[i].f2, P.enc[i].img);
    if (i > 0) one(L.enc[i].mha.qkv, 0, d, 3 * d, M->at<char>(P.enc[i].img_qkv));
    one(L.enc[i].mha.o, 0, d, d, M->at<char>(P.enc[i].img_of));
    if (with_backward) one(L.enc[i].mha.o, 1, d, d, M->at<char>(P.enc[i].img_o));
  }
  if (dec)
    for (int i = 0; i < c.num_layers; ++i) {
      ffn(L.dec[i].f1, L.dec[i].f2, P.dec[i].img);
      if (i > 0) one(L.dec[i].mha1.qkv, 0, d, 3 * d, M->at<char>(P.dec[i].img_qkv));
      one(L.dec[i].mha2.o, 0, d, d, M->at<char>(P.dec[i].img_o2f));
      one(L.dec[i].mha1.o, 0, d, d, M->at<char>(P.dec[i].img_o1f)); one(L.dec[i].mha2.q, 0, d, d, M->at<char>(P.dec[i].img_q2));   // self-attention tail + query projection
      if (with_backward) { one(L.dec[i].mha1.o, 1, d, d, M->at<char>(P.dec[i].img_o1)); one(L.dec[i].mha2.o, 1, d, d, M->at<char>(P.dec[i].img_o2)); }
      if (with_backward) one(L.dec[i].mha2.q, 1, d, d, M->at<char>(P.dec[i].img_q2t));      // its input gradient rides in the self-attention sublayer's LayerNorm launch
    }
  return skf_dense_weight_images((int)src.size(), src.data(), ld.data(), tr.data(), K.data(), N.data(), img.data(), c.gemm_precision, s);
}
// out = LayerNorm(x + dropout(ffn(x))): one launch, or Dense(relu) + Dense + residual-LayerNorm
// next / next_image / next_out: the Dense that consumes `out` (the next layer's q|k|v projection), taken into the same launch when
// the fused kernel runs (*next_done = true), else left to the caller
int ffn_ln_fwd(SkfModel* M, const DenseP& f1, const DenseP& f2, const LnP& ln, const float* x, int rows, float* h, void* bits,
               const void* image, float* z, float* out, float* stats, float rate, unsigned site, hipStream_t s,
               const DenseP* next = nullptr, const void* next_image = nullptr, float* next_out = nullptr, bool* next_done = nullptr) {
  const int d = M->cfg.d_model;
  static const bool chain_off = skf_knob("SKF_NO_FFN_CHAIN") && skf_knob("SKF_NO_FFN_CHAIN")[0] == '1';   // (measurement builds only)
  if (next_done) *next_done = false;
  if (M->ffn_fused && next && !chain_off && next->in == d && (next->out == 128 || next->out == 256 || next->out == 384) && next->ld == next->out) {
    if (next_done) *next_done = true;
    return skf_ffn_fused_fwd_proj_f32(rows, d, M->cfg.dff, x, image, M->P(f1.b), M->P(f2.b), h, bits, M->P(ln.g), M->P(ln.b), z, out, stats,
                                      rate, site, M->state, next_image, M->P(next->b), next->out, next_out, M->cfg.gemm_precision, s);
  }
  if (M->ffn_fused)
    return skf_ffn_fused_fwd_f32(rows, d, M->cfg.dff, x, image, M->P(f1.b), M->P(f2.b), h, bits, M->P(ln.g), M->P(ln.b), z, out, stats,
                                 rate, site, M->state, M->cfg.gemm_precision, s);
  SKF_TRY(dense_fwd_relu_bits(M, f1, x, rows, h, bits, s));
  SKF_TRY(dense_fwd(M, f2, h, rows, z, 0, s));
  return skf_layernorm_residual_fwd(x, z, M->P(ln.g), M->P(ln.b), out, stats, rows, d, rate, site, M->state, s);
}

// The tail of a layer behind its last attention: x1 = LayerNorm(x + dropout(o_proj(a))), out = LayerNorm(x1 + dropout(ffn(x1))) and,
// when there is one, the next layer's q|k|v projection - ONE launch (skf_ffn_block_fwd_f32) where the fused kernel runs, else the
// output-projection launch followed by ffn_ln_fwd.
int attn_tail_ffn_fwd(SkfModel* M, const DenseP& o, const LnP& ln_a, const float* a, const float* x, float* z1, float* x1, float* st1,
                      unsigned site_a, const void* o_image, const DenseP& f1, const DenseP& f2, const LnP& ln, float* h, void* bits,
                      const void* image, float* z, float* out, float* stats, unsigned site, int rows, float rate, hipStream_t s,
                      const DenseP* next, const void* next_image, float* next_out, bool* next_done) {
  const int d = M->cfg.d_model;
  static const bool pre_off = skf_knob("SKF_NO_FFN_PRE") && skf_knob("SKF_NO_FFN_PRE")[0] == '1';   // (measurement builds only)
  static const bool chain_off = skf_knob("SKF_NO_FFN_CHAIN") && skf_knob("SKF_NO_FFN_CHAIN")[0] == '1';
  if (!M->ffn_fused || pre_off || o.in != d || o.out != d || ln_a.b != ln_a.g + (size_t)d) {
    SKF_TRY(dense_ln_fwd(M, o, a, rows, x, z1, ln_a, x1, st1, rate, site_a, s));
    return ffn_ln_fwd(M, f1, f2, ln, x1, rows, h, bits, image, z, out, stats, rate, site, s, next, next_image, next_out, next_done);
  }
  SkfFfnBlockFwd b{};
  b.struct_size = sizeof(SkfFfnBlockFwd); b.M = rows; b.d = d; b.dff = M->cfg.dff; b.precision = M->cfg.gemm_precision;
  b.x = a; b.image = image; b.b1 = M->P(f1.b); b.b2 = M->P(f2.b); b.h = h; b.relu_bits_out = bits;
  b.gamma = M->P(ln.g); b.beta = M->P(ln.b); b.z = z; b.out = out; b.stats = stats; b.rate = rate; b.site = site; b.step_state = M->state;
  b.pre_image = o_image; b.pre_bias = M->P(o.b); b.pre_residual = x; b.pre_gamma = M->P(ln_a.g); b.pre_beta = M->P(ln_a.b);
  b.pre_z = z1; b.pre_out = x1; b.pre_stats = st1; b.pre_site = site_a;
  const bool chain = next && !chain_off && next->in == d && (next->out == 128 || next->out == 256 || next->out == 384) && next->ld == next->out;
  if (chain) { b.proj_image = next_image; b.proj_bias = M->P(next->b); b.proj_out = next_out; b.proj_n = next->out; }
  if (next_done) *next_done = chain;
  return skf_ffn_block_fwd_f32(&b, s);
}

// What the forward needs besides its inputs: the pre-split weight images of the row-owner launches (the weights changed in the last
// optimizer step), the two padding masks, and the samples sorted by length (both masks), longest first - every (sample, head)
// attention launch of the step deals its workgroups from that list.  None of it is read before the first attention.
int forward_preamble(SkfModel* M, bool with_backward, bool encoder_only, hipStream_t s, hipEvent_t masks_ready = nullptr) {
  const SkfConfig& c = M->cfg;
  const Plan& P = M->plan;
  const int B = c.batch, Le = c.seq_len, Ld = c.seq_len - 1;
  unsigned char* emask = M->at<unsigned char>(P.enc_mask);
  unsigned char* dmask = M->at<unsigned char>(P.dec_mask);
  if (M->masks_staged) {
    // (written by the staging launch of this call)
  } else if (c.continuous) {
    SKF_TRY(skf_padding_mask_continuous(M->at<float>(P.inp), Le, B, Le, emask, s));
    SKF_TRY(skf_padding_mask_continuous(M->at<float>(P.tar), Le, B, Ld, dmask, s));
  } else {
    SKF_TRY(skf_padding_mask(M->at<long long>(P.inp), Le, B, Le, emask, s));
    SKF_TRY(skf_padding_mask(M->at<long long>(P.tar), Le, B, Ld, dmask, s));
  }
  static const bool order_off = skf_knob("SKF_ATTN_ORDER") && skf_knob("SKF_ATTN_ORDER")[0] == '0';      // (measurement builds)
  M->order = nullptr;
  if (!order_off && B <= 4096) {
    SKF_TRY(skf_sample_order(emask, Le, Le, encoder_only ? nullptr : dmask, Ld, Ld, B, M->at<int>(P.order), s));
    M->order = M->at<int>(P.order);
  }
  if (masks_ready) SKF_HIP(hipEventRecord(masks_ready, s));      // (the images are only read by the launch BEHIND the first attention)
  M->ffn_fused = ffn_fused_on(M);
  if (M->ffn_fused) SKF_TRY(build_ffn_images(M, with_backward, encoder_only, s));
  return SKF_OK;
}

int run_forward(SkfModel* M, bool training, bool with_loss, hipStream_t s, bool encoder_only = false) {
  const SkfConfig& c = M->cfg;
  const Layout& L = M->lay;
  const Plan& P = M->plan;
  const int B = c.batch, Le = c.seq_len, Ld = c.seq_len - 1, d = c.d_model, H = c.num_heads, dh = d / H;
  const int Me = B * Le, Md = B * Ld, N = c.num_layers;
  const float rate = training ? c.dropout_rate : 0.f;
  const long long* inp = M->at<long long>(P.inp);
  const long long* tar = M->at<long long>(P.tar);
  unsigned char* emask = M->at<unsigned char>(P.enc_mask);
  unsigned char* dmask = M->at<unsigned char>(P.dec_mask);

  const float* inpf = M->at<float>(P.inp);      // continuous mode: (B, L, 5) stroke-5 rows
  const float* tarf = M->at<float>(P.tar);
  // weight images, padding masks, sample order: here, unless the train step already put them on the side stream (issue_embed_sorts)
  hipEvent_t images_ready = M->pre_ready, masks_ready = M->masks_ready;
  M->pre_ready = M->masks_ready = nullptr;
  if (!images_ready) SKF_TRY(forward_preamble(M, training && with_loss, encoder_only, s));
  const int* order = M->order;

  // ---------------- encoder (builders/layers/transformer.py:288-301)
  if (c.continuous)
    SKF_TRY(skf_embed_continuous_fwd(inpf, Le, B, Le, M->P(L.enc_embd.w), M->P(L.enc_embd.b), d, M->pos,
                                     M->at<float>(P.enc[0].x_in), rate, site_enc_embed(), M->state, s));
  else
    SKF_TRY(skf_embed_fwd(inp, Le, B, Le, M->P(L.enc_emb), c.vocab_size, d, M->pos, M->at<float>(P.enc[0].x_in), rate,
                          site_enc_embed(), M->state, s));
  bool enc_qkv_done = false;
  for (int i = 0; i < N; ++i) {
    const EncLayerP& w = L.enc[i];
    const EncAct& a = P.enc[i];
    float* x = M->at<float>(a.x_in);
    float* qkv = M->at<float>(a.qkv);
    if (!enc_qkv_done) SKF_TRY(dense_fwd(M, w.mha.qkv, x, Me, qkv, 0, s));     // (else: the previous layer's feed-forward launch wrote it)
    if (i == 0 && images_ready) SKF_HIP(hipStreamWaitEvent(s, masks_ready, 0));    // masks and order were built beside the embedding and this projection
    SKF_TRY(skf_attention_fwd_ordered(qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, emask, Le, 0, B, H, Le, Le, dh,
                                      M->at<float>(a.o), d, M->at<float>(a.astats), M->cfg.gemm_precision, order, s));
    const bool has_next = i + 1 < N;
    if (i == 0 && images_ready) SKF_HIP(hipStreamWaitEvent(s, images_ready, 0));      // ... and the weight images beside the first attention
    SKF_TRY(attn_tail_ffn_fwd(M, w.mha.o, w.ln1, M->at<float>(a.o), x, M->at<float>(a.z1), M->at<float>(a.x1), M->at<float>(a.st1),
                              site_enc(i, 0), M->at<char>(a.img_of), w.f1, w.f2, w.ln2, M->at<float>(a.h), hbits_of(M, a.hbits, Me),
                              M->at<char>(a.img[0]), M->at<float>(a.z2), M->at<float>(a.x2), M->at<float>(a.st2), site_enc(i, 1), Me, rate, s,
                              has_next ? &L.enc[i + 1].mha.qkv : nullptr, has_next ? M->at<char>(P.enc[i + 1].img_qkv) : nullptr,
                              has_next ? M->at<float>(P.enc[i + 1].qkv) : nullptr, &enc_qkv_done));
  }
  float* enc_out = M->at<float>(P.enc[N - 1].x2);
  // ---------------- bottleneck + classifier + expander (models/sketchformer.py:149-160,183-199,170-176)
  const int E = L.E, Ua = L.Ua;
  const bool bott = has_bott(c), cls = has_cls(c), recon = do_recon(c);
  if (bott) {
    SKF_TRY(dense_fwd(M, L.bott_w, enc_out, Me, M->at<float>(P.u), 2, s));
    SKF_TRY(skf_pool_fwd(M->at<float>(P.u), M->P(L.bott_v), enc_out, B, Le, Ua, d, M->at<float>(P.pool_a),
                         M->at<float>(c.attn_version == 2 ? P.pooled : P.emb), s));
    if (c.attn_version == 2)   // SelfAttnV2: o = embeding_layer(o) (builders/layers/transformer.py:128-129)
      SKF_TRY(dense_fwd(M, L.bott_e, M->at<float>(P.pooled), B, M->at<float>(P.emb), 0, s));
  }
  if (cls) SKF_TRY(classify_fwd(M, training, s));
  if (encoder_only || !recon) {
    if (encoder_only || !with_loss) {   // encode_from_seq / predict_class (models/sketchformer.py:162-168,223-228): class probabilities only
      if (!cls) return SKF_OK;
      return skf_softmax_ce(M->at<float>(P.cls_logits), c.n_classes, B, c.n_classes, M->at<long long>(P.labels), 1, 1, 0, 0, 0.f,
                            M->at<float>(P.cls_loss), M->at<float>(P.cls_hit), M->at<float>(P.cls_probs), 0, s);
    }
  }
  // pre_decoder: the expanded embedding, or the encoder output itself when there is no bottleneck (:172-176)
  float* pre = bott ? M->at<float>(P.pre) : enc_out;
  // pre_decoder is all the side stream's cross-attention K|V projections wait for: their event rides on the expander launch
  // (the event pool restarts here: the step's earlier events - masks, images - were waited for in front of the first encoder layer)
  hipEvent_t dec_in_ready = nullptr;
  bool dec_in_recorded = false;
  if (recon && M->side) {
    M->next_event = 0;
    dec_in_ready = M->new_event();
    SKF_CHECK_ARG(dec_in_ready, "event allocation failed");
  }
  if (recon && bott) {
    if (dec_in_ready && !g_capturing) skf_tls_stop_event = dec_in_ready;
    const int rc_ex = skf_expander_fwd(M->at<float>(P.emb), M->P(L.exp_w), M->P(L.exp_b), B, Le, E, pre, s);
    dec_in_recorded = dec_in_ready && !g_capturing && skf_tls_stop_event == nullptr;
    skf_tls_stop_event = nullptr;
    SKF_TRY(rc_ex);
  }

  // ---------------- decoder (builders/layers/transformer.py:325-344)
  if (recon) {
  if (c.continuous)
    SKF_TRY(skf_embed_continuous_fwd(tarf, Le, B, Ld, M->P(L.dec_embd.w), M->P(L.dec_embd.b), d, M->pos,
                                     M->at<float>(P.dec[0].x_in), rate, site_dec_embed(N), M->state, s));
  else
    SKF_TRY(skf_embed_fwd(tar, Le, B, Ld, M->P(L.dec_emb), c.vocab_size, d, M->pos, M->at<float>(P.dec[0].x_in), rate,
                          site_dec_embed(N), M->state, s));
  const unsigned char* cross_mask = c.blind_decoder_mask ? nullptr : emask;
  // The cross-attention K|V projections of ALL decoder layers only depend on pre_decoder: on the eager path they run
  // on the side stream under the first layer's self-attention block (one event pair) instead of on the critical path.
  // (round 5: the first layer's cross-attention waits for ITS projection only - it used to wait for all of them, 45 us with the main
  //  stream idle at cfg 2 - the second layer's for the rest)
  hipEvent_t kv_done = nullptr, kv_first = nullptr;
  if (M->side) {
    kv_done = M->new_event();
    static const bool wait_all = skf_knob("SKF_KV_WAIT_ALL") && skf_knob("SKF_KV_WAIT_ALL")[0] == '1';      // (measurement builds)
    kv_first = N > 1 && !wait_all ? M->new_event() : kv_done;
    SKF_CHECK_ARG(dec_in_ready && kv_done && kv_first, "event allocation failed");
    if (!dec_in_recorded) SKF_HIP(hipEventRecord(dec_in_ready, s));
    SKF_HIP(hipStreamWaitEvent(M->side, dec_in_ready, 0));
    for (int i = 0; i < N; ++i) {
      SKF_TRY(dense_fwd(M, L.dec[i].mha2.kv, pre, Me, M->at<float>(P.dec[i].kv2), 0, M->side));
      if (i == 0 && kv_first != kv_done) SKF_HIP(hipEventRecord(kv_first, M->side));
    }
    SKF_HIP(hipEventRecord(kv_done, M->side));
  }
  bool dec_qkv_done = false;
  for (int i = 0; i < N; ++i) {
    const DecLayerP& w = L.dec[i];
    const DecAct& a = P.dec[i];
    float* x = M->at<float>(a.x_in);
    float* qkv = M->at<float>(a.qkv);
    if (!dec_qkv_done) SKF_TRY(dense_fwd(M, w.mha1.qkv, x, Md, qkv, 0, s));
    SKF_TRY(skf_attention_fwd_ordered(qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, dmask, Ld, 1, B, H, Ld, Ld, dh,
                                      M->at<float>(a.o1), d, M->at<float>(a.astats1), M->cfg.gemm_precision, order, s));
    // out1 = LayerNorm(x + dropout(o1 . Wo + bo)) and q2 = out1 . Wq + bq: one row-owner launch where that kernel runs
    // (skf_ffn_block_fwd_f32 without a feed-forward image), else the fused Dense + LayerNorm launch and the projection launch
    static const bool tail_off = skf_knob("SKF_NO_TAIL_PROJ") && skf_knob("SKF_NO_TAIL_PROJ")[0] == '1';   // (measurement builds only)
    const bool tail_proj = M->ffn_fused && !tail_off && w.mha1.o.in == d && w.mha1.o.out == d && w.mha2.q.in == d && w.mha2.q.out == d &&
                           w.mha2.q.ld == d && w.ln1.b == w.ln1.g + (size_t)d;
    if (tail_proj) {
      SkfFfnBlockFwd tb{};
      tb.struct_size = sizeof(SkfFfnBlockFwd); tb.M = Md; tb.d = d; tb.dff = c.dff; tb.precision = c.gemm_precision;
      tb.x = M->at<float>(a.o1); tb.rate = rate; tb.step_state = M->state;
      tb.pre_image = M->at<char>(a.img_o1f); tb.pre_bias = M->P(w.mha1.o.b); tb.pre_residual = x; tb.pre_gamma = M->P(w.ln1.g); tb.pre_beta = M->P(w.ln1.b);
      tb.pre_z = M->at<float>(a.z1); tb.pre_out = M->at<float>(a.out1); tb.pre_stats = M->at<float>(a.st1); tb.pre_site = site_dec(N, i, 0);
      tb.proj_image = M->at<char>(a.img_q2); tb.proj_bias = M->P(w.mha2.q.b); tb.proj_out = M->at<float>(a.q2); tb.proj_n = d;
      SKF_TRY(skf_ffn_block_fwd_f32(&tb, s));
    } else {
      SKF_TRY(dense_ln_fwd(M, w.mha1.o, M->at<float>(a.o1), Md, x, M->at<float>(a.z1), w.ln1, M->at<float>(a.out1), M->at<float>(a.st1),
                           rate, site_dec(N, i, 0), s));
      SKF_TRY(dense_fwd(M, w.mha2.q, M->at<float>(a.out1), Md, M->at<float>(a.q2), 0, s));
    }
    float* kv2 = M->at<float>(a.kv2);
    if (!kv_done) SKF_TRY(dense_fwd(M, w.mha2.kv, pre, Me, kv2, 0, s));
    else if (i == 0) SKF_HIP(hipStreamWaitEvent(s, kv_first, 0));
    else if (i == 1) SKF_HIP(hipStreamWaitEvent(s, kv_done, 0));
    SKF_TRY(skf_attention_fwd_ordered(M->at<float>(a.q2), d, kv2, 2 * d, kv2 + d, 2 * d, cross_mask, Le, 0, B, H, Ld, Le, dh,
                                      M->at<float>(a.o2), d, M->at<float>(a.astats2), M->cfg.gemm_precision, order, s));
    const bool has_next = i + 1 < N;
    SKF_TRY(attn_tail_ffn_fwd(M, w.mha2.o, w.ln2, M->at<float>(a.o2), M->at<float>(a.out1), M->at<float>(a.z2), M->at<float>(a.out2),
                              M->at<float>(a.st2), site_dec(N, i, 1), M->at<char>(a.img_o2f), w.f1, w.f2, w.ln3, M->at<float>(a.h),
                              hbits_of(M, a.hbits, Md), M->at<char>(a.img[0]), M->at<float>(a.z3), M->at<float>(a.out3), M->at<float>(a.st3),
                              site_dec(N, i, 2), Md, rate, s, has_next ? &L.dec[i + 1].mha1.qkv : nullptr,
                              has_next ? M->at<char>(P.dec[i + 1].img_qkv) : nullptr, has_next ? M->at<float>(P.dec[i + 1].qkv) : nullptr,
                              &dec_qkv_done));
  }
  SKF_TRY(dense_fwd(M, L.out, M->at<float>(P.dec[N - 1].out3), Md, M->at<float>(P.logits), 0, s));
  }

  // ---------------- losses + metrics (models/sketchformer.py:334-346)
  const long long* labels = M->at<long long>(P.labels);
  if (with_loss) {
    // tar_real = tar[:, 1:]  -> target offset 1 within rows of stride L
    const float* recon_scalar = nullptr;
    if (recon) {
      if (c.continuous) {
        SKF_TRY(skf_continuous_loss(M->at<float>(P.logits), tarf, Le, Ld, 1, Md, c.recon_weight, M->at<float>(P.recon_loss),
                                    M->at<float>(P.recon_hit), M->at<float>(P.row_mask), M->at<float>(P.cont_scal), 1, s));
        recon_scalar = M->at<float>(P.cont_scal) + 3;
      } else {
        SKF_TRY(skf_softmax_ce(M->at<float>(P.logits), c.vocab_size, Md, c.vocab_size, tar, Le, Ld, 1, 1,
                               c.recon_weight / (float)Md, M->at<float>(P.recon_loss), M->at<float>(P.recon_hit), nullptr, 1, s));
      }
    }
    if (cls)
      SKF_TRY(skf_softmax_ce(M->at<float>(P.cls_logits), c.n_classes, B, c.n_classes, labels, 1, 1, 0, 0,
                             c.class_weight / (float)B, M->at<float>(P.cls_loss), M->at<float>(P.cls_hit),
                             M->at<float>(P.cls_probs), 1, s));
    // absent heads contribute 0 rows: their loss is 0 in total_loss (sum(all_losses), models/sketchformer.py:345)
    SKF_TRY(skf_metrics_update(M->at<float>(P.recon_loss), M->at<float>(P.recon_hit), recon ? Md : 0, c.recon_weight,
                               M->at<float>(P.cls_loss), M->at<float>(P.cls_hit), cls ? B : 0, c.class_weight, recon_scalar,
                               M->metrics, s));
  } else if (cls) {
    SKF_TRY(skf_softmax_ce(M->at<float>(P.cls_logits), c.n_classes, B, c.n_classes, labels, 1, 1, 0, 0, 0.f,
                           M->at<float>(P.cls_loss), M->at<float>(P.cls_hit), M->at<float>(P.cls_probs), 0, s));
  }
  return SKF_OK;
}

int ffn_bwd(SkfModel* M, const DenseP& f1, const DenseP& f2, const float* x_in, const float* h, const float* dy,
            float* dh, float* dx_acc, int rows, hipStream_t s, const void* hbits, const void* image_t) {
  if (M->ffn_fused) {       // both input gradients in one launch; the weight gradients read dy / h and x / dh as before
    SKF_TRY(dense_wgrad(M, f2, h, f2.in, dy, f2.out, rows, s));
    SKF_TRY(before_write(M, dh, s));
    SKF_TRY(before_write(M, dx_acc, s));
    const int* blocks = (M->live16 && rows == M->live_rows) ? M->live16 : nullptr;
    hipEvent_t parked = park_ready(M);
    const int rc = skf_ffn_fused_bwd_f32(rows, M->cfg.d_model, M->cfg.dff, dy, image_t, hbits, dh, dx_acc, 1, blocks, blocks ? 16 : 0,
                                         M->cfg.gemm_precision, s);
    const hipEvent_t ready = take_ready(parked);
    SKF_TRY(rc);
    SKF_TRY(issue_held_wgrads(M, s, ready));
    return dense_wgrad(M, f1, x_in, f1.in, dh, f1.out, rows, s);
  }
  SKF_TRY(dense_wgrad(M, f2, h, f2.in, dy, f2.out, rows, s));
  SKF_TRY(dense_dgrad(M, f2, dy, f2.out, rows, dh, f2.in, 0, h, f2.in, s, hbits));
  SKF_TRY(dense_wgrad(M, f1, x_in, f1.in, dh, f1.out, rows, s));
  SKF_TRY(dense_dgrad(M, f1, dh, f1.out, rows, dx_acc, f1.in, 1, nullptr, 0, s));
  return SKF_OK;
}

// `splits` partial row pairs [splits][2][d] of a LayerNorm's (dgamma, dbeta) -> one descriptor of the batched split-K reduction
int ln_partials_desc(SkfModel* M, const LnP& ln, const float* part, int splits) {
  const int d = M->cfg.d_model;
  SkfReduceDesc r;
  r.slab = part; r.C = M->G(ln.g); r.bias_grad = nullptr; r.splits = splits; r.M = 1; r.N = 2 * d;
  r.ldc = 2 * d; r.block_begin = M->reduce_blocks; r.pad = 0;
  if (!M->descs_uploaded) M->descs.push_back(r);
  else {
    const SkfReduceDesc& o = M->descs[M->desc_cursor];
    SKF_CHECK_ARG(o.slab == r.slab && o.C == r.C && o.splits == r.splits && o.block_begin == r.block_begin, "reduction sequence changed between steps");
  }
  M->reduce_blocks += skf_splitk_reduce_blocks(1, 2 * d);
  M->desc_cursor += 1;
  M->ln_cursor += 1;
  M->side_used = true;
  return SKF_OK;
}

// column sums of per-sample partials part[splits][n] -> C[n] in the batched reduction (a "slab" of `splits` splits of a 1 x n matrix)
int colsum_desc(SkfModel* M, const float* part, int splits, int n, float* C) {
  SKF_CHECK_ARG(M->desc_cursor < M->plan.n_wgrads, "reduction descriptor table exhausted");
  SkfReduceDesc r;
  r.slab = part; r.C = C; r.bias_grad = nullptr; r.splits = splits; r.M = 1; r.N = n;
  r.ldc = n; r.block_begin = M->reduce_blocks; r.pad = 0;
  if (!M->descs_uploaded) M->descs.push_back(r);
  else {
    const SkfReduceDesc& o = M->descs[M->desc_cursor];
    SKF_CHECK_ARG(o.slab == r.slab && o.C == r.C && o.splits == r.splits && o.block_begin == r.block_begin, "reduction sequence changed between steps");
  }
  M->reduce_blocks += skf_splitk_reduce_blocks(1, n);
  M->desc_cursor += 1;
  M->side_used = true;
  return SKF_OK;
}

int ln_bwd(SkfModel* M, const LnP& ln, const float* dout, const float* z, const float* st, float* dz, float* dy,
           int rows, float rate, unsigned site, hipStream_t s) {
  const Plan& P = M->plan;
  const int d = M->cfg.d_model;
  SKF_TRY(before_write(M, dz, s));
  if (dy != dz) SKF_TRY(before_write(M, dy, s));
  // decoder side of a padded batch: rows behind a sample's live length have dout == 0 and are not read
  const int* ll = (M->live16 && rows == M->live_rows) ? M->at<int>(P.live_len) : nullptr;
  const int rps = M->cfg.seq_len - 1;
  if (!M->side || ln.b != ln.g + (size_t)d)
    return skf_layernorm_residual_bwd_rows(dout, z, st, M->P(ln.g), dz, dy, M->G(ln.g), M->G(ln.b), rows, d, rate,
                                           site, M->state, M->at<char>(P.small_ws), P.small_ws_bytes, ll, rps, s);
  // eager path: leave the [g][2d] partials in this LayerNorm's own slice; their column sums ride in the batched
  // split-K reduction of the wgrads (a "slab" of g splits of a 1 x 2d matrix) instead of one tiny launch per LayerNorm
  SKF_CHECK_ARG(M->ln_cursor < 5 * (size_t)M->cfg.num_layers && M->desc_cursor < P.n_wgrads, "LayerNorm partial arena exhausted");
  float* part = M->at<float>(P.ln_part + M->ln_cursor * P.ln_part_stride);
  const size_t bytes = skf_layernorm_bwd_workspace_bytes(rows, d);
  SKF_TRY(skf_layernorm_residual_bwd_rows(dout, z, st, M->P(ln.g), dz, dy, nullptr, nullptr, rows, d, rate, site, M->state, part,
                                          bytes, ll, rps, s));
  return ln_partials_desc(M, ln, part, (int)(bytes / (8 * (size_t)d)));
}

// Both input gradients of a feed-forward block AND the backward of the LayerNorm that closes it in one launch
// (skf_ffn_fused_bwd_ln_f32), where that kernel exists and its dgamma / dbeta partials can ride in the batched reduction;
// otherwise the LayerNorm launch followed by ffn_bwd.  dout: gradient of the LayerNorm output; dx = dz + d(ffn input) is WRITTEN.
int ffn_ln_bwd(SkfModel* M, const LnP& ln, const DenseP& f1, const DenseP& f2, const float* dout, const float* z, const float* st,
               const float* x_in, const float* h, float* dy, float* dh, float* dx, int rows, float rate, unsigned site, hipStream_t s,
               const void* hbits, const void* image_t) {
  const Plan& P = M->plan;
  const int d = M->cfg.d_model;
  M->last_ready = nullptr;
  static const bool ln_off = skf_knob("SKF_NO_FFN_LN_BWD") && skf_knob("SKF_NO_FFN_LN_BWD")[0] == '1';   // (measurement builds only)
  const size_t pbytes = (size_t)skf_ffn_fused_ln_partials(rows) * 2 * d * sizeof(float);
  if (ln_off || !M->ffn_fused || !M->side || ln.b != ln.g + (size_t)d || pbytes > P.ln_part_stride) {
    SKF_TRY(ln_bwd(M, ln, dout, z, st, dx, dy, rows, rate, site, s));
    return ffn_bwd(M, f1, f2, x_in, h, dy, dh, dx, rows, s, hbits, image_t);
  }
  SKF_CHECK_ARG(M->ln_cursor < 5 * (size_t)M->cfg.num_layers && M->desc_cursor < P.n_wgrads, "LayerNorm partial arena exhausted");
  float* part = M->at<float>(P.ln_part + M->ln_cursor * P.ln_part_stride);
  SKF_TRY(before_write(M, dy, s));
  SKF_TRY(before_write(M, dh, s));
  SKF_TRY(before_write(M, dx, s));
  const int* blocks = (M->live16 && rows == M->live_rows) ? M->live16 : nullptr;
  hipEvent_t parked = park_ready(M);
  const int rc = skf_ffn_fused_bwd_ln_f32(rows, d, M->cfg.dff, dout, z, st, M->P(ln.g), rate, site, M->state, image_t, hbits, dy, dh, dx, part,
                                          pbytes, blocks, blocks ? 16 : 0, M->cfg.gemm_precision, s);
  const hipEvent_t ready = take_ready(parked);
  SKF_TRY(rc);
  SKF_TRY(ln_partials_desc(M, ln, part, skf_ffn_fused_ln_partials(rows)));
  M->last_ready = ready;                        // (nothing else reaches the main stream before this function returns: the caller may reuse it)
  SKF_TRY(issue_held_wgrads(M, s, ready));      // the previous layer's weight gradients: behind this launch (see hold_wgrads)
  SKF_TRY(dense_wgrad(M, f2, h, f2.in, dy, f2.out, rows, s));
  return dense_wgrad(M, f1, x_in, f1.in, dh, f1.out, rows, s);
}

// The backward of an attention sublayer's tail, out = LayerNorm(x + dropout(o_proj(a))): LayerNorm backward + the projection's
// input gradient in one launch (skf_layernorm_bwd_dgrad_f32) where it exists, else the two launches.  The weight gradient is queued.
// lead_a / lead_w / lead_image_t (optional): the gradient of the LayerNorm output is dout + lead_a . lead_w^T - one launch where the fused
// kernel takes it (*lead_done = true), else the caller's accumulating GEMM has to run first (*lead_done = false, nothing done yet)
bool ln_oproj_bwd_takes_lead(SkfModel* M, const LnP& ln, const DenseP& o, const DenseP& lead_w, int rows) {
  const Plan& P = M->plan;
  const int d = M->cfg.d_model;
  static const bool off = (skf_knob("SKF_NO_LN_DGRAD") && skf_knob("SKF_NO_LN_DGRAD")[0] == '1') || (skf_knob("SKF_NO_LN_LEAD") && skf_knob("SKF_NO_LN_LEAD")[0] == '1');
  const size_t pbytes = (size_t)skf_layernorm_bwd_dgrad_partials(rows) * 2 * d * sizeof(float);
  return !off && M->ffn_fused && M->side && ln.b == ln.g + (size_t)d && pbytes <= P.ln_part_stride && o.in == d && o.out == d &&
         lead_w.in == d && lead_w.out == d && skf_layernorm_bwd_dgrad_supported(rows, d, M->cfg.gemm_precision);
}
int ln_oproj_bwd(SkfModel* M, const LnP& ln, const DenseP& o, const float* dout, const float* z, const float* st, const float* a_in,
                 float* dz, float* dy, float* da, int rows, float rate, unsigned site, hipStream_t s, const void* image_t,
                 const float* lead_a = nullptr, const void* lead_image_t = nullptr) {
  const Plan& P = M->plan;
  const int d = M->cfg.d_model;
  static const bool off = skf_knob("SKF_NO_LN_DGRAD") && skf_knob("SKF_NO_LN_DGRAD")[0] == '1';   // (measurement builds only)
  const size_t pbytes = (size_t)skf_layernorm_bwd_dgrad_partials(rows) * 2 * d * sizeof(float);
  if (off || !M->ffn_fused || !M->side || ln.b != ln.g + (size_t)d || pbytes > P.ln_part_stride || o.in != d || o.out != d ||
      !skf_layernorm_bwd_dgrad_supported(rows, d, M->cfg.gemm_precision)) {
    SKF_TRY(ln_bwd(M, ln, dout, z, st, dz, dy, rows, rate, site, s));
    SKF_TRY(dense_wgrad(M, o, a_in, d, dy, d, rows, s));
    return dense_dgrad(M, o, dy, d, rows, da, d, 0, nullptr, 0, s);
  }
  SKF_CHECK_ARG(M->ln_cursor < 5 * (size_t)M->cfg.num_layers && M->desc_cursor < P.n_wgrads, "LayerNorm partial arena exhausted");
  float* part = M->at<float>(P.ln_part + M->ln_cursor * P.ln_part_stride);
  SKF_TRY(before_write(M, dz, s));
  SKF_TRY(before_write(M, dy, s));
  SKF_TRY(before_write(M, da, s));
  const int* blocks = (M->live16 && rows == M->live_rows) ? M->live16 : nullptr;
  SKF_TRY(skf_layernorm_bwd_dgrad_lead_f32(rows, d, dout, lead_a, lead_image_t, z, st, M->P(ln.g), rate, site, M->state, image_t, dz, dy, da, part,
                                           pbytes, blocks, blocks ? 16 : 0, M->cfg.gemm_precision, s));
  SKF_TRY(ln_partials_desc(M, ln, part, skf_layernorm_bwd_dgrad_partials(rows)));
  return dense_wgrad(M, o, a_in, d, dy, d, rows, s);
}

// Row-block lists of the decoder-side backward (token mode, split arithmetic only: the fp32-MFMA kernels ignore them)
int build_row_lists(SkfModel* M, hipStream_t s) {
  const SkfConfig& c = M->cfg;
  const Plan& P = M->plan;
  static const bool rows_off = skf_knob("SKF_NO_ROW_BLOCKS") && skf_knob("SKF_NO_ROW_BLOCKS")[0] == '1';
  M->lists_built = false;
  if (c.continuous || rows_off || !do_recon(c) || c.gemm_precision == SKF_PREC_F32) return SKF_OK;
  const int B = c.batch, Le = c.seq_len, Ld = c.seq_len - 1;
  SKF_TRY(skf_target_live_len(M->at<long long>(P.tar), Le, B, Ld, M->at<int>(P.live_len), s));
  SKF_TRY(skf_row_blocks_build(M->at<int>(P.live_len), B, Ld, 16, M->at<int>(P.live16), s));
  SKF_TRY(skf_row_blocks_build(M->at<int>(P.live_len), B, Ld, 32, M->at<int>(P.live32), s));
  M->lists_built = true;
  return SKF_OK;
}

int run_backward(SkfModel* M, hipStream_t s) {
  M->live16 = M->live32 = nullptr; M->live_rows = 0;      // (set below for the decoder layers only; see reset_live_rows)
  M->next_event = 0;
  M->pending_readers.clear();
  M->side_used = false;
  M->slab_cursor = 0; M->desc_cursor = 0; M->reduce_blocks = 0; M->phase_desc_begin = 0; M->ln_cursor = 0;
  const SkfConfig& c = M->cfg;
  const Layout& L = M->lay;
  const Plan& P = M->plan;
  const int B = c.batch, Le = c.seq_len, Ld = c.seq_len - 1, d = c.d_model, H = c.num_heads, dh = d / H;
  const int Me = B * Le, Md = B * Ld, N = c.num_layers;
  const float rate = c.dropout_rate;
  const long long* inp = M->at<long long>(P.inp);
  const long long* tar = M->at<long long>(P.tar);
  const unsigned char* emask = M->at<unsigned char>(P.enc_mask);
  const unsigned char* dmask = M->at<unsigned char>(P.dec_mask);
  float* G = M->at<float>(P.gA);
  float* G2 = M->at<float>(P.gB);
  float* dO = M->at<float>(P.do_);
  float* dpre = M->at<float>(P.dpre);
  float* demb = M->at<float>(P.demb);
  M->wq.clear(); M->wq_held.clear();
  int layer_no = 0;     // running layer counter: picks the gradient-buffer set

  const bool bott = has_bott(c), cls = has_cls(c), recon = do_recon(c);
  float* enc_out = M->at<float>(P.enc[N - 1].x2);
  const float* pre = bott ? M->at<float>(P.pre) : enc_out;      // pre_decoder (see run_forward)
  if (recon) {
  // From the output layer to the decoder embedding every (B * Ld)-row gradient is exactly zero behind a sample's last trained position
  // (skf_row_blocks.hip): the split-arithmetic GEMMs walk the live row blocks only.  Not in continuous mode (its pen-state
  // loss has a gradient at every position), not for the fp32-MFMA kernels (they ignore the lists).
  // (the lists only depend on the staged targets: issue_embed_sorts builds them on the side stream under the forward)
  if (!M->lists_built) SKF_TRY(build_row_lists(M, s));
  if (M->lists_built) { M->live16 = M->at<int>(P.live16); M->live32 = M->at<int>(P.live32); M->live_rows = Md; }
  M->lists_built = false;
  // output layer: logits buffer now holds dlogits
  const float* dlog = M->at<float>(P.logits);
  SKF_TRY(dense_wgrad(M, L.out, M->at<float>(P.dec[N - 1].out3), d, dlog, L.out.out, Md, s));
  {
    hipEvent_t parked = park_fresh(M);       // the group's "main stream is here" event rides on the input-gradient launch (the last of its chain)
    const int rc = dense_dgrad(M, L.out, dlog, L.out.out, Md, G, d, 0, nullptr, 0, s);
    const hipEvent_t ready = take_ready(parked);
    SKF_TRY(rc);
    SKF_TRY(issue_wgrads(M, s, ready));
  }
  const unsigned char* cross_mask = c.blind_decoder_mask ? nullptr : emask;
  for (int i = N - 1; i >= 0; --i, ++layer_no) {
    const DecLayerP& w = L.dec[i];
    const DecAct& a = P.dec[i];
    const Plan::GradSet& gs = P.gs[layer_no % P.n_gs];
    float* dy3 = M->at<float>(gs.dy[0]); float* dy2 = M->at<float>(gs.dy[1]); float* dy1 = M->at<float>(gs.dy[2]);
    float* dqkv = M->at<float>(gs.dqkv); float* dkv2 = M->at<float>(gs.dkv2); float* dq2 = M->at<float>(gs.dq2);
    // out3 = LN3(out2 + drop(ffn(out2)))
    SKF_TRY(ffn_ln_bwd(M, w.ln3, w.f1, w.f2, G, M->at<float>(a.z3), M->at<float>(a.st3), M->at<float>(a.out2), M->at<float>(a.h), dy3,
                       M->at<float>(gs.dh), G2, Md, rate, site_dec(N, i, 2), s, hbits_of(M, a.hbits, Md), M->at<char>(a.img[1])));
    // out2 = LN2(out1 + drop(mha2(pre, pre, out1)))
    SKF_TRY(ln_oproj_bwd(M, w.ln2, w.mha2.o, G2, M->at<float>(a.z2), M->at<float>(a.st2), M->at<float>(a.o2), G, dy2, dO, Md, rate,
                         site_dec(N, i, 1), s, M->at<char>(a.img_o2)));
    const float* kv2 = M->at<float>(a.kv2);
    SKF_TRY(before_write(M, dq2, s));
    SKF_TRY(before_write(M, dkv2, s));
    const int* qlive = M->live16 ? M->at<int>(P.live_len) : nullptr;      // decoder query rows behind it have dO == 0
    SKF_TRY(skf_attention_bwd_ordered(M->at<float>(a.q2), d, kv2, 2 * d, kv2 + d, 2 * d, M->at<float>(a.o2), d, dO, d,
                                      M->at<float>(a.astats2), cross_mask, Le, 0, B, H, Ld, Le, dh, dq2, d, dkv2, 2 * d,
                                      dkv2 + d, 2 * d, M->cfg.gemm_precision, qlive, M->order, s));
    SKF_TRY(dense_wgrad(M, w.mha2.q, M->at<float>(a.out1), d, dq2, d, Md, s));
    // d(out1) += dq2 . Wq^T: inside the LayerNorm-backward launch of the self-attention sublayer below where that kernel takes it
    const bool lead = ln_oproj_bwd_takes_lead(M, w.ln1, w.mha1.o, w.mha2.q, Md);
    if (!lead) SKF_TRY(dense_dgrad(M, w.mha2.q, dq2, d, Md, G, d, 1, nullptr, 0, s));
    SKF_TRY(dense_wgrad(M, w.mha2.kv, pre, L.E, dkv2, 2 * d, Me, s));
    // (the last layer of the loop runs it on the main stream: its reader follows too soon to gain anything)
    if (i == 0) {
      SKF_TRY(before_read(M, dpre, s));   // the side-stream writers of the layers above have finished accumulating
      SKF_TRY(dense_dgrad(M, w.mha2.kv, dkv2, 2 * d, Me, dpre, L.E, i != N - 1, nullptr, 0, s));
    } else SKF_TRY(dense_dgrad_deferred(M, w.mha2.kv, dkv2, 2 * d, Me, dpre, L.E, i != N - 1, s));
    // out1 = LN1(x + drop(mha1(x,x,x)))
    SKF_TRY(ln_oproj_bwd(M, w.ln1, w.mha1.o, G, M->at<float>(a.z1), M->at<float>(a.st1), M->at<float>(a.o1), G2, dy1, dO, Md, rate,
                         site_dec(N, i, 0), s, M->at<char>(a.img_o1), lead ? dq2 : nullptr, lead ? M->at<char>(a.img_q2t) : nullptr));
    const float* qkv = M->at<float>(a.qkv);
    SKF_TRY(before_write(M, dqkv, s));
    SKF_TRY(skf_attention_bwd_ordered(qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, M->at<float>(a.o1), d, dO, d,
                                      M->at<float>(a.astats1), dmask, Ld, 1, B, H, Ld, Ld, dh, dqkv, 3 * d, dqkv + d, 3 * d,
                                      dqkv + 2 * d, 3 * d, M->cfg.gemm_precision, qlive, M->order, s));
    SKF_TRY(dense_wgrad(M, w.mha1.qkv, M->at<float>(a.x_in), d, dqkv, 3 * d, Md, s));
    // the 8 weight gradients of this layer: one event pair - held until the next layer's fused feed-forward launch is queued
    static const bool hold_off = skf_knob("SKF_NO_WGRAD_HOLD") && skf_knob("SKF_NO_WGRAD_HOLD")[0] == '1';   // (measurement builds only)
    const bool hold = M->ffn_fused && i > 0 && !hold_off;
    hipEvent_t parked = hold ? nullptr : park_fresh(M);         // not held: the group's event rides on this layer's last launch
    const int rc_dg = dense_dgrad(M, w.mha1.qkv, dqkv, 3 * d, Md, G2, d, 1, nullptr, 0, s);
    const hipEvent_t ready = take_ready(parked);
    SKF_TRY(rc_dg);
    float* t = G; G = G2; G2 = t;
    if (hold) SKF_TRY(hold_wgrads(M, s));
    else SKF_TRY(issue_wgrads(M, s, ready));
  }
  M->live16 = M->live32 = nullptr; M->live_rows = 0;
  hipEvent_t dec_emb_done = nullptr;
  // decoder embedding
  if (c.continuous) {
    SKF_TRY(skf_embed_continuous_bwd(M->at<float>(P.tar), Le, B, Ld, G, d, M->G(L.dec_embd.w), M->G(L.dec_embd.b), rate,
                                     site_dec_embed(N), M->state, M->at<char>(P.small_ws), P.small_ws_bytes, s));
  } else {
    if (P.emb_sort_bytes) {
      hipEvent_t parked = M->n_buckets == 2 ? park_fresh(M) : nullptr;      // the bucket's reduction (side stream) waits for this launch's own signal
      const int rc_e = skf_embed_bwd_sorted(M->at<char>(P.emb_sort[1]), B, Ld, G, c.vocab_size, d, M->G(L.dec_emb), rate, site_dec_embed(N),
                                            M->state, s);
      dec_emb_done = take_ready(parked);
      SKF_TRY(rc_e);
    } else {
      SKF_HIP(hipMemsetAsync(M->G(L.dec_emb), 0, (size_t)c.vocab_size * d * sizeof(float), s));
      SKF_TRY(skf_embed_bwd(tar, Le, B, Ld, G, c.vocab_size, d, M->G(L.dec_emb), rate, site_dec_embed(N), M->state, s));
    }
  }
  // every gradient of [decoder embedding .. output layer] is issued: first bucket of the flat buffer
  if (M->n_buckets == 2) SKF_TRY(flush_wgrads(M, s, 0, false, true, dec_emb_done));
  SKF_TRY(before_read(M, dpre, s));     // the deferred K/V-projection input gradients (side stream) are complete
  }   // recon
  const int E = L.E, Ua = L.Ua, U = c.lowerdim, NB = c.class_buffer_layers;
  hipEvent_t bott_ready = nullptr;
  if (bott) {
    // expander, classifier
    static const bool part_off = skf_knob("SKF_NO_BOTT_PARTIALS") && skf_knob("SKF_NO_BOTT_PARTIALS")[0] == '1';   // (measurement builds only)
    const bool defer_sums = !part_off && M->side && Ua <= 4096;      // (wherever the batched reduction runs: the eager step and its two-stream capture)
    float* xp1 = M->at<float>(P.bott_part);
    float* xp2 = xp1 + (size_t)B * Le;
    float* pvp = xp2 + (size_t)B * Le;
    if (recon && defer_sums) {
      SKF_TRY(skf_expander_bwd_partials(dpre, M->at<float>(P.emb), M->P(L.exp_w), B, Le, E, demb, 0, xp1, xp2, s));
      SKF_TRY(colsum_desc(M, xp1, B, Le, M->G(L.exp_w)));
      SKF_TRY(colsum_desc(M, xp2, B, Le, M->G(L.exp_b)));
    } else if (recon)
      SKF_TRY(skf_expander_bwd(dpre, M->at<float>(P.emb), M->P(L.exp_w), B, Le, E, demb, 0, M->G(L.exp_w), M->G(L.exp_b),
                               M->at<char>(P.small_ws), P.small_ws_bytes, s));
    const int acc_emb = recon ? 1 : 0;        // without a decoder the class head is the only source of d(embedding)
    // classifier (+ class buffers): d fc_i = dropout'(.) then relu'(.) - both are element-wise masks and commute
    const float* dcls = M->at<float>(P.cls_logits);
    if (cls && NB == 0) {
      SKF_TRY(dense_wgrad(M, L.cls, M->at<float>(P.emb), E, dcls, c.n_classes, B, s));
      SKF_TRY(dense_dgrad(M, L.cls, dcls, c.n_classes, B, demb, E, acc_emb, nullptr, 0, s));
    } else if (cls) {
      float* dz = M->at<float>(P.dcb[0]);
      float* dz2 = M->at<float>(P.dcb[1]);
      SKF_TRY(before_write(M, dz, s));
      SKF_TRY(dense_wgrad(M, L.cls, M->at<float>(P.cb_f[NB - 1]), U, dcls, c.n_classes, B, s));
      SKF_TRY(dense_dgrad(M, L.cls, dcls, c.n_classes, B, dz, U, 0, M->at<float>(P.cb_h[NB - 1]), U, s));
      for (int i = NB - 1; i >= 0; --i) {
        SKF_TRY(skf_dropout(dz, dz, (size_t)B * U, c.class_dropout, site_class(N, i), M->state, s));
        const float* in = i == 0 ? M->at<float>(P.emb) : M->at<float>(P.cb_f[i - 1]);
        const int in_w = i == 0 ? E : U;
        SKF_TRY(dense_wgrad(M, L.cbuf[i], in, in_w, dz, U, B, s));
        if (i == 0) {
          SKF_TRY(dense_dgrad(M, L.cbuf[0], dz, U, B, demb, E, acc_emb, nullptr, 0, s));
        } else {
          SKF_TRY(before_write(M, dz2, s));
          SKF_TRY(dense_dgrad(M, L.cbuf[i], dz, U, B, dz2, U, 0, M->at<float>(P.cb_h[i - 1]), U, s));
          float* t = dz; dz = dz2; dz2 = t;
        }
      }
    }
    // bottleneck
    const float* dpool = demb;
    if (c.attn_version == 2) {
      SKF_TRY(before_write(M, M->at<float>(P.dpooled), s));
      SKF_TRY(dense_wgrad(M, L.bott_e, M->at<float>(P.pooled), d, demb, U, B, s));
      SKF_TRY(dense_dgrad(M, L.bott_e, demb, U, B, M->at<float>(P.dpooled), d, 0, nullptr, 0, s));
      dpool = M->at<float>(P.dpooled);
    }
    SKF_TRY(before_write(M, G, s));
    if (defer_sums) {
      SKF_TRY(skf_pool_bwd_partials(M->at<float>(P.u), M->P(L.bott_v), enc_out, M->at<float>(P.pool_a), dpool, B, Le, Ua, d, G, pvp, s));
      SKF_TRY(colsum_desc(M, pvp, B, Ua, M->G(L.bott_v)));
    } else
    SKF_TRY(skf_pool_bwd(M->at<float>(P.u), M->P(L.bott_v), enc_out, M->at<float>(P.pool_a), dpool, B, Le, Ua, d,
                         G, M->G(L.bott_v), M->at<char>(P.small_ws), P.small_ws_bytes, s));
    SKF_TRY(dense_wgrad(M, L.bott_w, enc_out, d, M->at<float>(P.u), Ua, Me, s));
    hipEvent_t parked = park_fresh(M);
    const int rc_dg = dense_dgrad(M, L.bott_w, M->at<float>(P.u), Ua, Me, G, d, 1, nullptr, 0, s);
    bott_ready = take_ready(parked);
    SKF_TRY(rc_dg);
  } else {
    // no bottleneck: d(enc_output) is what the cross-attention K/V projections of all decoder layers sent back
    float* spare = (G == M->at<float>(P.gA)) ? M->at<float>(P.gB) : M->at<float>(P.gA);
    G = dpre; G2 = spare;
  }
  SKF_TRY(issue_wgrads(M, s, bott_ready));            // expander / classifier / bottleneck group
  for (int i = N - 1; i >= 0; --i, ++layer_no) {
    const EncLayerP& w = L.enc[i];
    const EncAct& a = P.enc[i];
    const Plan::GradSet& gs = P.gs[layer_no % P.n_gs];
    float* dy2 = M->at<float>(gs.dy[0]); float* dy1 = M->at<float>(gs.dy[1]);
    float* dqkv = M->at<float>(gs.dqkv);
    SKF_TRY(ffn_ln_bwd(M, w.ln2, w.f1, w.f2, G, M->at<float>(a.z2), M->at<float>(a.st2), M->at<float>(a.x1), M->at<float>(a.h), dy2,
                       M->at<float>(gs.dh), G2, Me, rate, site_enc(i, 1), s, hbits_of(M, a.hbits, Me), M->at<char>(a.img[1])));
    // last layer of the backward: nothing is left on the main stream to hide a whole layer's weight gradients behind
    // (only the embedding gradient follows), so they go out per sublayer - the step's tail before Adam is one wgrad, not four
    static const bool early_tail = !skf_knob("SKF_NO_EARLY_TAIL");
    // (the fused launch's completion signal already served the held group as its "main stream is here" event: this group shares it)
    if (i == 0 && early_tail) SKF_TRY(issue_wgrads(M, s, M->last_ready));
    M->last_ready = nullptr;
    {
      hipEvent_t parked = (i == 0 && early_tail) ? park_fresh(M) : nullptr;
      const int rc_ln = ln_oproj_bwd(M, w.ln1, w.mha.o, G2, M->at<float>(a.z1), M->at<float>(a.st1), M->at<float>(a.o), G, dy1, dO, Me, rate,
                                     site_enc(i, 0), s, M->at<char>(a.img_o));
      const hipEvent_t ready = take_ready(parked);
      SKF_TRY(rc_ln);
      if (i == 0 && early_tail) SKF_TRY(issue_wgrads(M, s, ready));
    }
    const float* qkv = M->at<float>(a.qkv);
    SKF_TRY(before_write(M, dqkv, s));
    SKF_TRY(skf_attention_bwd_ordered(qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, M->at<float>(a.o), d, dO, d,
                                      M->at<float>(a.astats), emask, Le, 0, B, H, Le, Le, dh, dqkv, 3 * d, dqkv + d, 3 * d,
                                      dqkv + 2 * d, 3 * d, M->cfg.gemm_precision, nullptr, M->order, s));
    SKF_TRY(dense_wgrad(M, w.mha.qkv, M->at<float>(a.x_in), d, dqkv, 3 * d, Me, s));
    // last layer of the backward: the weight gradient only needs dqkv, so it goes out BEFORE the input-gradient GEMM - the hop
    // to the side stream and the kernel itself then run under that GEMM and the embedding gradient instead of behind them.
    // Round 6: it runs on the MAIN stream.  The side stream still has this layer's feed-forward and output-projection gradients
    // queued behind the held group of the layer above and finished ~20 us AFTER the main stream's last kernel
    // (profiles/r06h_timeline.txt: 34 us of idle main stream in front of the final reduction); with the 18-us q|k|v gradient in line
    // here both streams end together and the final reduction starts without waiting for a hop.
    static const bool tail_main = !(skf_knob("SKF_TAIL_WGRAD_SIDE") && skf_knob("SKF_TAIL_WGRAD_SIDE")[0] == '1');   // (measurement builds only)
    if (i == 0 && early_tail) SKF_TRY(issue_wgrads(M, s, nullptr, tail_main && M->wq_held.empty()));
    SKF_TRY(dense_dgrad(M, w.mha.qkv, dqkv, 3 * d, Me, G, d, 1, nullptr, 0, s));
    static const bool hold_off_e = skf_knob("SKF_NO_WGRAD_HOLD") && skf_knob("SKF_NO_WGRAD_HOLD")[0] == '1';
    if (M->ffn_fused && i > 0 && !hold_off_e) SKF_TRY(hold_wgrads(M, s));
    else SKF_TRY(issue_wgrads(M, s));
    // half-way through the encoder: the slabs and LayerNorm partials finished so far are reduced on the side stream now, under the
    // remaining layers - the final reduction, which the optimizer waits for on the main stream, shrinks to the last layers' share
    // (with a held group: only what is already on the side stream - issuing the held group here would put it beside the next layer's
    //  fused feed-forward launch again)
    static const bool mid_flush = !(skf_knob("SKF_MID_FLUSH") && skf_knob("SKF_MID_FLUSH")[0] == '0');
    // (not with the fused feed-forward blocks: the reduction launch lands beside a fused launch it cannot share CUs with - A/B 3.95 vs 3.99 ms)
    if (mid_flush && !M->ffn_fused && M->side && N >= 2 && i == N / 2) SKF_TRY(flush_wgrads(M, s, -1, false, M->wq_held.empty()));
  }
  if (c.continuous) {
    SKF_TRY(skf_embed_continuous_bwd(M->at<float>(P.inp), Le, B, Le, G, d, M->G(L.enc_embd.w), M->G(L.enc_embd.b), rate,
                                     site_enc_embed(), M->state, M->at<char>(P.small_ws), P.small_ws_bytes, s));
  } else {
    if (P.emb_sort_bytes) {
      SKF_TRY(skf_embed_bwd_sorted(M->at<char>(P.emb_sort[0]), B, Le, G, c.vocab_size, d, M->G(L.enc_emb), rate, site_enc_embed(),
                                   M->state, s));
    } else {
      SKF_HIP(hipMemsetAsync(M->G(L.enc_emb), 0, (size_t)c.vocab_size * d * sizeof(float), s));
      SKF_TRY(skf_embed_bwd(inp, Le, B, Le, G, c.vocab_size, d, M->G(L.enc_emb), rate, site_enc_embed(), M->state, s));
    }
  }
  return flush_wgrads(M, s, M->n_buckets - 1, true);
}

// KV-cached greedy reconstruction (models/sketchformer.py:255-311).  The reference re-runs the decoder on the whole
// prefix for every token; one step here touches only the newest position (rows = batch):
//   x = embed(token_i) * sqrt(d) + pos[i]                                   (transformer.py:325-334, dropout off)
//   per layer: q = x Wq ; [k|v] = x [Wk|Wv] written straight into cache row i ; attention over keys 0..i with the
//   target padding mask (look-ahead is implicit: later keys do not exist yet) ; LN ; cross attention over the
//   cached K/V of pre_decoder ; LN ; FFN ; LN                               (transformer.py:245-262)
//   logits of position i -> argmax / stroke-5 row -> appended                (sketchformer.py:285-301)
int run_greedy_decode(SkfModel* M, const float* embedding, const int* expected_len_host, int n_valid, long long sos,
                      long long eos, int max_steps, void* out, int* out_len_host, hipStream_t s) {
  const SkfConfig& c = M->cfg;
  const Layout& L = M->lay;
  const Plan& P = M->plan;
  const int B = c.batch, Le = c.seq_len, d = c.d_model, H = c.num_heads, dh = d / H, N = c.num_layers, F = c.dff;
  const int T = max_steps + 1;                         // columns of the output buffer
  const int Vout = c.continuous ? 5 : c.vocab_size;
  (void)F;
  const bool bott = has_bott(c);
  float* enc_out = M->at<float>(P.enc[N - 1].x2);
  // the embedding is (B, E) with a bottleneck, else the whole encoder output (B, L, d) = pre_decoder itself
  const float* pre = bott ? M->at<float>(P.pre) : enc_out;
  if (bott) {
    if (embedding && embedding != M->at<float>(P.emb))
      SKF_HIP(hipMemcpyAsync(M->at<float>(P.emb), embedding, (size_t)B * L.E * sizeof(float), hipMemcpyDeviceToDevice, s));
  } else if (embedding && embedding != enc_out) {
    SKF_HIP(hipMemcpyAsync(M->at<float>(P.pre), embedding, (size_t)B * Le * d * sizeof(float), hipMemcpyDeviceToDevice, s));
    pre = M->at<float>(P.pre);
  }
  int* eos_seen = M->at<int>(P.dc_flags);
  int* done_step = eos_seen + B;
  long long* dyn = M->at<long long>(P.dc_dyn);          // [0] n_valid, [1] eos  (read by the selection kernel)
  int* step_dev = reinterpret_cast<int*>(dyn + 4);      // index of the position being decoded
  unsigned char* selfmask = M->at<unsigned char>(P.dc_mask);
  // the running output lives in an internal (B, Le+1) image so that the captured step has constant arguments
  const int Ti = Le + 1;
  long long* tokens = c.continuous ? nullptr : M->at<long long>(P.dc_tok);
  float* cont = c.continuous ? M->at<float>(P.dc_cont) : nullptr;
  SKF_TRY(skf_decode_init(tokens, Ti, cont, Ti, selfmask, Le + 1, eos_seen, done_step, B, sos, step_dev, s));
  M->dec_dyn_host[0] = n_valid; M->dec_dyn_host[1] = eos;
  SKF_HIP(hipMemcpyAsync(dyn, M->dec_dyn_host, 2 * sizeof(long long), hipMemcpyHostToDevice, s));
  int* limit = nullptr;                                  // per-sample key limit of the cross attention (non-blind only)
  if (!c.blind_decoder_mask) {
    limit = M->at<int>(P.dc_limit);
    if (expected_len_host) SKF_HIP(hipMemcpyAsync(limit, expected_len_host, (size_t)B * sizeof(int), hipMemcpyHostToDevice, s));
    else SKF_HIP(hipMemsetAsync(limit, 0xff, (size_t)B * sizeof(int), s));        // -1: nattn = step + 1
  }
  // pre_decoder and the cross-attention K/V of every layer: once
  if (bott)
    SKF_TRY(skf_expander_fwd(M->at<float>(P.emb), M->P(L.exp_w), M->P(L.exp_b), B, Le, L.E, M->at<float>(P.pre), s));
  for (int l = 0; l < N; ++l)
    SKF_TRY(dense_fwd(M, L.dec[l].mha2.kv, pre, B * Le, M->at<float>(P.dec[l].kv2), 0, s));
  if (has_cls(c)) {
    SKF_TRY(classify_fwd(M, false, s));                                                       // classify_from_embedding
    SKF_TRY(skf_softmax_ce(M->at<float>(P.cls_logits), c.n_classes, B, c.n_classes, M->at<long long>(P.labels), 1, 1, 0, 0, 0.f,
                           M->at<float>(P.cls_loss), M->at<float>(P.cls_hit), M->at<float>(P.cls_probs), 0, s));
  }

  float* q = M->at<float>(P.dc_q); float* o = M->at<float>(P.dc_o); float* z = M->at<float>(P.dc_z);
  float* out1 = M->at<float>(P.dc_out1); float* out2 = M->at<float>(P.dc_out2); float* hbuf = M->at<float>(P.dc_h);
  float* stats = M->at<float>(P.dc_stats); float* logits = M->at<float>(P.dc_logits);
  float* kvnew = M->at<float>(P.dc_kvnew);
  // One decode step.  Every argument is the same for every step and every call (the step index, n_valid and eos are
  // read from device memory), so the ~50 small launches are captured once into a hipGraph and replayed.
  auto issue_step = [&]() -> int {
    float* x = M->at<float>(P.dc_x[0]);
    float* xn = M->at<float>(P.dc_x[1]);
    SKF_TRY(skf_decode_embed(tokens, cont, Ti, B, c.continuous ? nullptr : M->P(L.dec_emb), c.vocab_size,
                             c.continuous ? M->P(L.dec_embd.w) : nullptr, c.continuous ? M->P(L.dec_embd.b) : nullptr, d,
                             M->pos, step_dev, x, s));
    for (int l = 0; l < N; ++l) {
      const DecLayerP& w = L.dec[l];
      float* cache = M->at<float>(P.dc_cache[l]);                       // (B, Le, 2d): K | V of the positions so far
      const DenseP wq{w.mha1.qkv.w, w.mha1.qkv.b, d, d, w.mha1.qkv.ld};
      const DenseP wkv{w.mha1.qkv.w + d, w.mha1.qkv.b + d, d, 2 * d, w.mha1.qkv.ld};
      SKF_TRY(dense_fwd_ld(M, wq, x, d, B, q, d, 0, s));
      SKF_TRY(dense_fwd_ld(M, wkv, x, d, B, kvnew, 2 * d, 0, s));
      // keys 0..step: the cache plus the row just projected (which the kernel also appends to the cache)
      SKF_TRY(skf_attention_decode(q, d, cache, cache + d, 2 * d, (long long)Le * 2 * d, selfmask, Le + 1, nullptr, 0, B, H,
                                   Le, dh, o, d, step_dev, kvnew, kvnew + d, 2 * d, 0, s));
      SKF_TRY(dense_fwd(M, w.mha1.o, o, B, z, 0, s));
      SKF_TRY(skf_layernorm_residual_fwd(x, z, M->P(w.ln1.g), M->P(w.ln1.b), out1, stats, B, d, 0.f, 0, M->state, s));
      const float* kv2 = M->at<float>(P.dec[l].kv2);
      SKF_TRY(dense_fwd(M, w.mha2.q, out1, B, q, 0, s));
      // cross mask (models/sketchformer.py:172,279-283): none when blind, else keys >= nattn (expected length or step+1)
      SKF_TRY(skf_attention_decode(q, d, kv2, kv2 + d, 2 * d, (long long)Le * 2 * d, nullptr, 0, limit, 0, B, H, Le, dh, o, d,
                                   step_dev, nullptr, nullptr, 0, c.blind_decoder_mask ? 0 : 1, s));
      SKF_TRY(dense_fwd(M, w.mha2.o, o, B, z, 0, s));
      SKF_TRY(skf_layernorm_residual_fwd(out1, z, M->P(w.ln2.g), M->P(w.ln2.b), out2, stats, B, d, 0.f, 0, M->state, s));
      SKF_TRY(dense_fwd(M, w.f1, out2, B, hbuf, 1, s));
      SKF_TRY(dense_fwd(M, w.f2, hbuf, B, z, 0, s));
      SKF_TRY(skf_layernorm_residual_fwd(out2, z, M->P(w.ln3.g), M->P(w.ln3.b), xn, stats, B, d, 0.f, 0, M->state, s));
      float* t = x; x = xn; xn = t;
    }
    SKF_TRY(dense_fwd(M, L.out, x, B, logits, 0, s));
    if (c.continuous)
      return skf_decode_select_continuous(logits, Vout, B, 0, 0, cont, Ti, selfmask, Le + 1, done_step, step_dev, dyn, s);
    return skf_decode_select_tokens(logits, Vout, B, Vout, 0, 0, 0, tokens, Ti, selfmask, Le + 1, eos_seen, done_step,
                                    step_dev, dyn, s);
  };
  // One launch per position (skf_decode_fused.hip) unless SKF_MODEL_DECODE_LAYERWISE (skf_model_set_flags) asks for the layer-by-layer path above
  const bool fused_off = (M->flags & SKF_MODEL_DECODE_LAYERWISE) != 0;
  const bool fused = !fused_off && skf_decode_fused_supported(d, H, F, Le, N, Vout);
  SkfDecodeFused fp{};
  if (fused) {
    auto dn = [&](const DenseP& w) {
      SkfDecDense r{M->P(w.w), M->P(w.b), w.in, w.out, w.ld, 0};
      r.vec4 = (w.ld & 3) == 0 && (w.out & 3) == 0 && ((uintptr_t)r.w & 15) == 0;
      return r;
    };
    fp.B = B; fp.Le = Le; fp.d = d; fp.H = H; fp.F = F; fp.N = N; fp.Vout = Vout; fp.vocab = c.vocab_size;
    fp.blind = c.blind_decoder_mask ? 1 : 0; fp.hs_len = F > Vout ? F : Vout;
    for (int l = 0; l < N; ++l) {
      const DecLayerP& w = L.dec[l];
      SkfDecLayer& o = fp.layer[l];
      o.qkv = dn(w.mha1.qkv); o.o = dn(w.mha1.o); o.q2 = dn(w.mha2.q); o.o2 = dn(w.mha2.o); o.f1 = dn(w.f1); o.f2 = dn(w.f2);
      o.ln1_g = M->P(w.ln1.g); o.ln1_b = M->P(w.ln1.b); o.ln2_g = M->P(w.ln2.g); o.ln2_b = M->P(w.ln2.b);
      o.ln3_g = M->P(w.ln3.g); o.ln3_b = M->P(w.ln3.b);
      o.cache = M->at<float>(P.dc_cache[l]); o.kv2 = M->at<float>(P.dec[l].kv2);
    }
    fp.out = dn(L.out);
    fp.emb_table = c.continuous ? nullptr : M->P(L.dec_emb);
    fp.embd_w = c.continuous ? M->P(L.dec_embd.w) : nullptr; fp.embd_b = c.continuous ? M->P(L.dec_embd.b) : nullptr;
    fp.pos = M->pos; fp.tokens = tokens; fp.cont = cont; fp.Ti = Ti; fp.selfmask = selfmask; fp.mask_ld = Le + 1;
    fp.eos_seen = eos_seen; fp.done_step = done_step; fp.step_dev = step_dev; fp.ticket = done_step + 1;
    fp.dyn = dyn; fp.limit = limit;
    SKF_HIP(hipMemsetAsync(fp.ticket, 0, sizeof(int), s));
  }
  static const bool use_graph = !(skf_knob("SKF_DECODE_GRAPH") && skf_knob("SKF_DECODE_GRAPH")[0] == '0');
  if (fused) {
    for (int i = 0; i < max_steps; ++i) {
      SKF_TRY(skf_decode_fused_launch(fp, s));
      if ((i & 7) == 7 || i + 1 == max_steps) {
        int done = -1;
        SKF_HIP(hipMemcpyAsync(&done, done_step, sizeof(int), hipMemcpyDeviceToHost, s));
        SKF_HIP(hipStreamSynchronize(s));
        if (done >= 0 || i + 1 == max_steps) {
          const int ncols = (done >= 0 ? done + 1 : i + 1) + 1;
          if (out_len_host) *out_len_host = ncols;
          const size_t esz = c.continuous ? 5 * sizeof(float) : sizeof(long long);
          SKF_HIP(hipMemcpy2DAsync(out, (size_t)T * esz, c.continuous ? (const void*)cont : (const void*)tokens, (size_t)Ti * esz,
                                   (size_t)ncols * esz, B, hipMemcpyDeviceToDevice, s));
          return SKF_OK;
        }
      }
    }
  }
  if (use_graph && !M->g_dec) {
    hipGraph_t graph = nullptr;
    SKF_HIP(hipStreamSynchronize(s));        // nothing of the setup above may end up inside the captured step
    SKF_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = issue_step();
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc != SKF_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) { skf_set_error("hipStreamEndCapture (decode step): %s", hipGetErrorString(e)); return SKF_EHIP; }
    e = hipGraphInstantiate(&M->g_dec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { skf_set_error("hipGraphInstantiate (decode step): %s", hipGetErrorString(e)); M->g_dec = nullptr; return SKF_EHIP; }
  }
  int done = -1, steps_run = 0;
  for (int i = 0; i < max_steps; ++i) {
    if (use_graph) SKF_HIP(hipGraphLaunch(M->g_dec, s));
    else SKF_TRY(issue_step());
    steps_run = i + 1;
    if ((i & 7) == 7 || i + 1 == max_steps) {            // the reference syncs every token; every 8th is enough here
      SKF_HIP(hipMemcpyAsync(&done, done_step, sizeof(int), hipMemcpyDeviceToHost, s));
      SKF_HIP(hipStreamSynchronize(s));
      if (done >= 0) break;
    }
  }
  const int ncols = (done >= 0 ? done + 1 : steps_run) + 1;     // start symbol + emitted positions
  if (out_len_host) *out_len_host = ncols;
  // hand the valid columns to the caller's (B, max_steps + 1[, 5]) buffer
  const size_t esz = c.continuous ? 5 * sizeof(float) : sizeof(long long);
  SKF_HIP(hipMemcpy2DAsync(out, (size_t)T * esz, c.continuous ? (const void*)cont : (const void*)tokens, (size_t)Ti * esz,
                           (size_t)ncols * esz, B, hipMemcpyDeviceToDevice, s));
  return SKF_OK;
}

int prologue(SkfModel* M, hipStream_t s) {
  const SkfConfig& c = M->cfg;
  return skf_step_prologue(M->state, c.schedule, c.sched_p0, c.sched_p1, c.sched_p2, c.sched_p3, c.beta1, c.beta2,
                           c.seed, s);
}

// The live-row state is host-side and only valid while ONE backward is being issued: an early (error) return from a backward
// must not leave it set for the next call on the model, whose batch has other live rows.
inline void reset_live_rows(SkfModel* M) { M->live16 = M->live32 = nullptr; M->live_rows = 0; }

int stage_inputs(SkfModel* M, const void* inp, const void* tar, int tar_ld, const long long* labels, hipStream_t s) {
  const SkfConfig& c = M->cfg;
  const Plan& P = M->plan;
  reset_live_rows(M);
  M->lists_built = false;
  M->pre_ready = M->masks_ready = nullptr;
  M->masks_staged = false;
  SKF_CHECK_ARG(inp && tar, "null input");
  const size_t row = c.continuous ? (size_t)c.seq_len * 5 * sizeof(float) : (size_t)c.seq_len * 8;     // bytes per sample
  const size_t src_row = c.continuous ? (size_t)tar_ld * 5 * sizeof(float) : (size_t)tar_ld * 8;
  // one launch for the three copies (skf_rowops.hip); operands that are not 4-byte aligned take the copy engine below
  static const bool stage_off = skf_knob("SKF_NO_STAGE_KERNEL") && skf_knob("SKF_NO_STAGE_KERNEL")[0] == '1';   // (measurement builds only)
  if (!stage_off) {
    // token mode: the two padding masks are written by the same launch (forward_preamble then skips its mask launches)
    static const bool mask_off = skf_knob("SKF_NO_STAGED_MASKS") && skf_knob("SKF_NO_STAGED_MASKS")[0] == '1';   // (measurement builds only)
    const bool masks = !c.continuous && !mask_off && tar_ld >= c.seq_len;
    const int rc = skf_stage_inputs_launch(inp, M->at<char>(P.inp), tar, M->at<char>(P.tar), row, src_row, row < src_row ? row : src_row, c.batch,
                                           labels, M->at<char>(P.labels), s, masks ? M->at<unsigned char>(P.enc_mask) : nullptr,
                                           masks ? M->at<unsigned char>(P.dec_mask) : nullptr, masks ? c.seq_len : 0);
    if (rc == SKF_OK) M->masks_staged = masks;
    if (rc != SKF_EUNSUPPORTED) return rc;
  }
  SKF_HIP(hipMemcpyAsync(M->at<char>(P.inp), inp, row * c.batch, hipMemcpyDeviceToDevice, s));
  if (tar_ld == c.seq_len) {
    SKF_HIP(hipMemcpyAsync(M->at<char>(P.tar), tar, row * c.batch, hipMemcpyDeviceToDevice, s));
  } else {
    SKF_HIP(hipMemcpy2DAsync(M->at<char>(P.tar), row, tar, src_row, row < src_row ? row : src_row, c.batch,
                             hipMemcpyDeviceToDevice, s));
  }
  if (labels) SKF_HIP(hipMemcpyAsync(M->at<char>(P.labels), labels, (size_t)c.batch * 8, hipMemcpyDeviceToDevice, s));
  else SKF_HIP(hipMemsetAsync(M->at<char>(P.labels), 0, (size_t)c.batch * 8, s));
  return SKF_OK;
}

// The staging copies with the model's `inputs_staged` event behind them: a caller that hands over DEVICE tensors it will refill in place
// makes its own stream wait for that event (skf_model_wait_inputs_staged) instead of cloning the tensors in front of every call - the
// clones were two copy kernels on the caller's stream that the step then waited for, ~19 us of idle GPU at the head of every step.
// The event rides on the staging launch as its completion signal where there is one (SKF_LAUNCH_TAIL), else it is recorded.
template <typename F>
int stage_with_event(SkfModel* M, hipStream_t s, F stage) {
  if (!M->inputs_staged) SKF_HIP(hipEventCreateWithFlags(&M->inputs_staged, hipEventDisableTiming));
  skf_tls_stop_event = M->inputs_staged;
  const int rc = stage();
  const bool attached = skf_tls_stop_event == nullptr;
  skf_tls_stop_event = nullptr;
  SKF_TRY(rc);
  if (!attached) SKF_HIP(hipEventRecord(M->inputs_staged, s));
  M->inputs_staged_valid = true;
  return SKF_OK;
}

// use_graph = 1: the step is captured on ONE stream (no side stream exists).  use_graph = 2 (round 5): the two-stream step is captured -
// the side stream joins the capture through an event recorded on the capturing stream (fork) and is joined back before the capture
// ends, so the weight-gradient groups / embedding sorts / K|V projections become parallel branches of the graph.  The first call runs
// eagerly: it builds and uploads the reduction descriptors (a synchronous copy, not capturable) that the captured sequence reuses.
template <typename F>
int capture_or_run(SkfModel* M, hipGraphExec_t* exec, hipStream_t s, F body) {
  if (!M->cfg.use_graph) return body();
  // use_graph = 2 replays a MULTI-BRANCH graph, and hipGraphLaunch of such a graph reads past the end of the exec's stream vector in the
  // HIP runtime whenever one of the exec's internal streams compares equal to the launch stream (hip::Graph::UpdateStreams: analysis in
  // include/skf.h at SKF_MODEL_TWO_STREAM_GRAPH, DESIGN.md section 6 "Round 6").  Whether that happens is decided by the runtime's state
  // in the PROCESS, so the mode needs the caller's explicit SKF_MODEL_TWO_STREAM_GRAPH; without it the same launches go out eagerly on
  // the two streams (bit-equal results, and the faster form anyway).
  if (M->cfg.use_graph == 2 && !(M->flags & SKF_MODEL_TWO_STREAM_GRAPH)) return body();
  const bool two_stream = M->side != nullptr && exec == &M->g_fb;
  if (two_stream && !M->descs_uploaded) return body();
  if (!*exec) {
    hipGraph_t graph = nullptr;
    SKF_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    ++g_capturing;
    int rc = SKF_OK;
    if (two_stream) {
      hipEvent_t fork = M->fork_event;
      if (hipEventRecord(fork, s) != hipSuccess || hipStreamWaitEvent(M->side, fork, 0) != hipSuccess) rc = SKF_EHIP;
    }
    if (rc == SKF_OK) rc = body();
    if (two_stream && rc == SKF_OK) {
      hipEvent_t join = M->join_event;
      if (hipEventRecord(join, M->side) != hipSuccess || hipStreamWaitEvent(s, join, 0) != hipSuccess) rc = SKF_EHIP;
    }
    --g_capturing;
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc != SKF_OK) { if (graph) (void)hipGraphDestroy(graph); if (rc == SKF_EHIP) skf_set_error("two-stream capture: fork / join failed"); return rc; }
    if (e != hipSuccess) { skf_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return SKF_EHIP; }
    e = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { skf_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); *exec = nullptr; return SKF_EHIP; }
  }
  SKF_HIP(hipGraphLaunch(*exec, s));
  return SKF_OK;
}

}  // namespace

#define SKF_BF16_PART 2
#include "skf_model_bf16.inc"
#undef SKF_BF16_PART

// =========================================================================== C ABI
extern "C" size_t skf_config_size(void) { return sizeof(SkfConfig); }

extern "C" int skf_model_set_flags(SkfModel* M, uint32_t flags) {
  SKF_CHECK_ARG(M, "null model");
  SKF_CHECK_ARG((flags & ~(SKF_MODEL_DECODE_LAYERWISE | SKF_MODEL_FFN_LAUNCHES | SKF_MODEL_TWO_STREAM_GRAPH)) == 0, "unknown flag bits");
  // SKF_MODEL_FFN_LAUNCHES changes the launch sequence of the step, hence the split counts of the LayerNorm partials and the
  // reduction descriptors that were built (and uploaded once) by the first step, and the captured step graphs: drop them like
  // skf_model_bind does, so that the next step rebuilds its descriptors / re-captures with the new sequence.
  if ((M->flags ^ flags) & SKF_MODEL_FFN_LAUNCHES) {
    SKF_HIP(hipDeviceSynchronize());     // steps in flight still read the descriptor table the next step uploads again
    if (M->g_fb) { (void)hipGraphExecDestroy(M->g_fb); M->g_fb = nullptr; }
    if (M->g_opt) { (void)hipGraphExecDestroy(M->g_opt); M->g_opt = nullptr; }
    M->descs.clear(); M->descs_uploaded = false;
    M->slab_cursor = 0; M->desc_cursor = 0; M->ln_cursor = 0; M->reduce_blocks = 0; M->phase_desc_begin = 0;
  }
  M->flags = flags;
  return SKF_OK;
}

extern "C" int skf_config_validate(const SkfConfig* c) {
  SKF_CHECK_ARG(c, "null config");
  if (c->struct_size != sizeof(SkfConfig)) {
    skf_set_error("SkfConfig.struct_size is %u, this library's SkfConfig has %zu bytes: the caller's declaration of the struct does not "
                  "match include/skf.h (set struct_size = sizeof(SkfConfig))", c->struct_size, sizeof(SkfConfig));
    return SKF_EINVAL;
  }
  SKF_CHECK_ARG(c->batch > 0 && c->seq_len > 1 && c->num_layers > 0, "bad sizes");
  SKF_CHECK_ARG(c->d_model % c->num_heads == 0, "d_model must be divisible by num_heads");
  const int dh = c->d_model / c->num_heads;
  // Any d_model % num_heads == 0 like the reference (builders/layers/transformer.py:150-152).  The MFMA kernels serve d_model in
  // {64,128,256,512} with head sizes {16,32,64} (every BASELINE config); other shapes run on the plain fp32 kernels of
  // skf_generic.hip (LayerNorm, expander, attention) and the generic GEMM.  Limits of those: d_model % 4 == 0 and
  // head size % 4 == 0 (16-byte rows and head slices), d_model <= 1024, head size <= 128.
  if ((c->d_model & 3) || c->d_model > 1024 || dh > 128 || (dh & 3)) {
    skf_set_error("d_model %d / head size %d: need d_model %% 4 == 0, d_model <= 1024, head size %% 4 == 0, head size <= 128", c->d_model, dh);
    return SKF_EUNSUPPORTED; }
  SKF_CHECK_ARG(c->attn_version == 1 || c->attn_version == 2, "attn_version must be 1 (SelfAttnV1) or 2 (SelfAttnV2)");
  SKF_CHECK_ARG(c->lowerdim >= 0, "lowerdim must be >= 0");
  // models/sketchformer.py:96-108,338: the class head only exists with a bottleneck; asking for it without one fails
  // in the reference too (the 'class' loss is never registered)
  SKF_CHECK_ARG(c->lowerdim > 0 || !c->do_classification, "do_classification needs lowerdim > 0");
  SKF_CHECK_ARG(c->do_reconstruction || (c->lowerdim > 0 && c->do_classification), "nothing to train: no decoder and no class head");
  if (c->lowerdim > 0 && c->attn_version == 2 && ((c->lowerdim & 3) || c->lowerdim > 1024)) {
    skf_set_error("attn_version=2: lowerdim %d (the embedding width) must be a multiple of 4, at most 1024", c->lowerdim); return SKF_EUNSUPPORTED; }
  SKF_CHECK_ARG(c->class_buffer_layers >= 0 && c->class_buffer_layers <= 8, "class_buffer_layers must be in [0, 8]");
  SKF_CHECK_ARG(c->class_dropout >= 0.f && c->class_dropout < 1.f, "class_dropout out of range");
  SKF_CHECK_ARG(c->optimizer == 0 || c->optimizer == 1, "optimizer must be 0 (Adam) or 1 (SGD with momentum)");
  SKF_CHECK_ARG(c->gemm_precision == SKF_PREC_F32 || c->gemm_precision == SKF_PREC_BF16X3 || c->gemm_precision == SKF_PREC_BF16X6,
                "gemm_precision must be 0 (fp32 MFMA), 6 (bf16x6) or 3 (bf16x3)");
  SKF_CHECK_ARG(c->n_classes > 0, "bad number of classes");
  if (!c->continuous) {
    SKF_CHECK_ARG(c->vocab_size > 0, "bad vocab size");
    SKF_CHECK_ARG(c->vocab_size % 4 == 0, "vocab_size must be a multiple of 4");
  }
  SKF_CHECK_ARG(c->dff % 4 == 0 && c->lowerdim % 4 == 0, "dff and lowerdim must be multiples of 4");
  SKF_CHECK_ARG(c->max_pos >= c->seq_len, "max_pos < seq_len");
  SKF_CHECK_ARG(c->dropout_rate >= 0.f && c->dropout_rate < 1.f, "dropout_rate out of range");
  SKF_CHECK_ARG(c->seq_len <= 512, "seq_len > 512 not supported");
  SKF_CHECK_ARG(c->act_dtype == SKF_ACT_F32 || c->act_dtype == SKF_ACT_BF16, "act_dtype must be 0 (fp32) or 1 (bf16)");
  if (c->act_dtype == SKF_ACT_BF16) {
    // the bf16 path is built for the default structure of the model (what BASELINE cfg 5 trains)
    if (c->continuous || c->attn_version != 1 || c->lowerdim <= 0 || !c->do_classification || !c->do_reconstruction ||
        c->class_buffer_layers != 0) {
      skf_set_error("act_dtype=bf16 supports token mode with attn_version=1, bottleneck + classifier + decoder, no class buffers");
      return SKF_EUNSUPPORTED;
    }
    if (dh != 64) { skf_set_error("act_dtype=bf16: head size %d, the streaming attention kernels are built for 64", dh); return SKF_EUNSUPPORTED; }
    if (!(c->d_model == 128 || c->d_model == 256 || c->d_model == 512)) { skf_set_error("act_dtype=bf16: d_model %d not in {128,256,512}", c->d_model); return SKF_EUNSUPPORTED; }
    SKF_CHECK_ARG(c->dff % 8 == 0 && c->lowerdim % 8 == 0, "act_dtype=bf16: dff and lowerdim must be multiples of 8");
    SKF_CHECK_ARG(c->vocab_size <= 2040, "act_dtype=bf16: vocab_size > 2040 (the cross-entropy kernel keeps a row in registers)");
  }
  return SKF_OK;
}

extern "C" size_t skf_model_param_floats(const SkfConfig* cfg) {
  if (skf_config_validate(cfg) != SKF_OK) return 0;
  return build_layout(*cfg).total;
}

extern "C" int skf_model_param_entries(const SkfConfig* cfg, SkfParamEntry* out_host, int max_entries) {
  int rc = skf_config_validate(cfg);
  if (rc != SKF_OK) return rc;
  Layout L = build_layout(*cfg);
  const int n = (int)L.entries.size();
  if (out_host) for (int i = 0; i < n && i < max_entries; ++i) out_host[i] = L.entries[i];
  return n;
}

extern "C" size_t skf_model_workspace_bytes(const SkfConfig* cfg) {
  if (skf_config_validate(cfg) != SKF_OK) return 0;
  if (cfg->act_dtype == SKF_ACT_BF16) return build_plan16(*cfg, build_layout(*cfg)).bytes;
  return build_plan(*cfg).bytes;
}

extern "C" int skf_model_create(const SkfConfig* cfg, SkfModel** out) {
  SKF_CHECK_ARG(out, "null out");
  int rc = skf_config_validate(cfg);
  if (rc != SKF_OK) return rc;
  SkfModel* M = new SkfModel();
  M->cfg = *cfg;
  M->lay = build_layout(*cfg);
  if (cfg->act_dtype == SKF_ACT_BF16) {
    M->bf16 = true;
    M->p16 = build_plan16(*cfg, M->lay);
    // one stream; gradient buckets like the fp32 path (events cannot be recorded for outside waiters in a captured graph)
    if (!cfg->use_graph) {
      M->n_buckets = 2;
      for (int i = 0; i < 2; ++i) SKF_HIP(hipEventCreateWithFlags(&M->bucket_ready[i], hipEventDisableTiming));
    }
    *out = M;
    return SKF_OK;
  }
  M->plan = build_plan(*cfg);
  const Plan& P = M->plan;
  const int B = cfg->batch, L = cfg->seq_len, Ld = L - 1, d = cfg->d_model, N = cfg->num_layers;
  auto reg = [&](const char* n, size_t off, int r, int c) { M->named[n] = {off, {r, c}}; };
  reg("logits", P.logits, B * Ld, cfg->continuous ? 5 : cfg->vocab_size);
  reg("class_probs", P.cls_probs, B, cfg->n_classes);
  reg("class_logits", P.cls_logits, B, cfg->n_classes);
  if (has_bott(*cfg)) reg("embedding", P.emb, B, M->lay.E);
  else reg("embedding", P.enc[N - 1].x2, B * L, d);              // no bottleneck: the encoder output (models/sketchformer.py:158-159)
  reg("enc_output", P.enc[N - 1].x2, B * L, d);
  reg("dec_output", P.dec[N - 1].out3, B * Ld, d);
  reg("pre_decoder", has_bott(*cfg) ? P.pre : P.enc[N - 1].x2, B * L, M->lay.E);
  reg("bottleneck_attn", P.pool_a, B, L);
  // the FFN hidden activations relu(x W1 + b1): parity tests read the ReLU branch taken on the device from them
  static char hnames[2][64][24];
  for (int i = 0; i < N && i < 64; ++i) {
    snprintf(hnames[0][i], sizeof(hnames[0][i]), "encoder/layer%d/ffn_h", i);
    reg(hnames[0][i], P.enc[i].h, B * L, cfg->dff);
    if (do_recon(*cfg)) {
      snprintf(hnames[1][i], sizeof(hnames[1][i]), "decoder/layer%d/ffn_h", i);
      reg(hnames[1][i], P.dec[i].h, B * Ld, cfg->dff);
    }
  }
  reg("enc_embed_out", P.enc[0].x_in, B * L, d);
  reg("dec_embed_out", P.dec[0].x_in, B * Ld, d);
  // measured on MI355X: eager launches + a wgrad side stream beat hipGraph replay (graph nodes of different
  // streams do not overlap, 7.50 vs 7.72 ms/step), so the side stream is only used on the eager path
  // (round 5: use_graph = 2 captures the two-stream step - capture_or_run)
  if (cfg->use_graph != 1 && !(skf_knob("SKF_NO_SIDE_STREAM") && skf_knob("SKF_NO_SIDE_STREAM")[0] == '1'))
    SKF_HIP(hipStreamCreateWithFlags(&M->side, hipStreamNonBlocking));
  if (cfg->use_graph == 2 && M->side) {
    SKF_HIP(hipEventCreateWithFlags(&M->fork_event, hipEventDisableTiming));
    SKF_HIP(hipEventCreateWithFlags(&M->join_event, hipEventDisableTiming));
  }
  if (!cfg->use_graph) {     // events cannot be recorded for outside waiters inside a captured graph: one bucket there
    M->n_buckets = do_recon(*cfg) ? 2 : 1;
    for (int i = 0; i < 2; ++i) SKF_HIP(hipEventCreateWithFlags(&M->bucket_ready[i], hipEventDisableTiming));
  }
  *out = M;
  return SKF_OK;
}

extern "C" void skf_model_destroy(SkfModel* m) {
  if (!m) return;
  if (m->g_fb) (void)hipGraphExecDestroy(m->g_fb);
  if (m->g_opt) (void)hipGraphExecDestroy(m->g_opt);
  if (m->g_dec) (void)hipGraphExecDestroy(m->g_dec);
  for (hipEvent_t e : m->events) (void)hipEventDestroy(e);
  for (int i = 0; i < 2; ++i) if (m->bucket_ready[i]) (void)hipEventDestroy(m->bucket_ready[i]);
  if (m->inputs_staged) (void)hipEventDestroy(m->inputs_staged);
  if (m->fork_event) (void)hipEventDestroy(m->fork_event);
  if (m->join_event) (void)hipEventDestroy(m->join_event);
  if (m->side) (void)hipStreamDestroy(m->side);
  delete m;
}

extern "C" int skf_model_bind(SkfModel* m, float* params, float* grads, float* adam_m, float* adam_v, const float* pos,
                              void* workspace, size_t workspace_bytes, float* metrics, void* step_state) {
  SKF_CHECK_ARG(m && params && grads && adam_m && adam_v && pos && workspace && metrics && step_state, "null buffer");
  SKF_CHECK_ARG(workspace_bytes >= (m->bf16 ? m->p16.bytes : m->plan.bytes), "workspace too small");
  SKF_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "workspace must be 256-byte aligned");
  SKF_CHECK_ARG((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)adam_m | (uintptr_t)adam_v) & 15) == 0, "flat buffers must be 16-byte aligned");
  m->params = params; m->grads = grads; m->m = adam_m; m->v = adam_v; m->pos = pos;
  m->ws = (char*)workspace; m->metrics = metrics; m->state = step_state;
  if (m->g_fb) { (void)hipGraphExecDestroy(m->g_fb); m->g_fb = nullptr; }
  if (m->g_opt) { (void)hipGraphExecDestroy(m->g_opt); m->g_opt = nullptr; }
  if (m->g_dec) { (void)hipGraphExecDestroy(m->g_dec); m->g_dec = nullptr; }
  m->descs.clear(); m->descs_uploaded = false;
  return SKF_OK;
}

extern "C" int skf_model_forward(SkfModel* m, const void* inp, const void* tar, int tar_ld, int training,
                                 skf_stream_t stream) {
  SKF_CHECK_ARG(m && m->ws, "model not bound");
  hipStream_t s = (hipStream_t)stream;
  if (m->bf16) {
    SKF_TRY(stage_with_event(m, s, [&]() { return stage_inputs16(m, inp, tar, tar_ld, nullptr, s); }));
    if (training) SKF_TRY(prologue(m, s));
    return run_forward16(m, training != 0, false, s);
  }
  SKF_TRY(stage_with_event(m, s, [&]() { return stage_inputs(m, inp, tar, tar_ld, nullptr, s); }));
  if (training) SKF_TRY(prologue(m, s));
  return run_forward(m, training != 0, false, s);
}

extern "C" int skf_model_encode(SkfModel* m, const void* inp, skf_stream_t stream) {
  SKF_CHECK_ARG(m && m->ws, "model not bound");
  hipStream_t s = (hipStream_t)stream;
  if (m->bf16) {
    SKF_TRY(stage_with_event(m, s, [&]() { return stage_inputs16(m, inp, inp, m->cfg.seq_len, nullptr, s); }));
    return run_forward16(m, false, false, s, true);
  }
  SKF_TRY(stage_with_event(m, s, [&]() { return stage_inputs(m, inp, inp, m->cfg.seq_len, nullptr, s); }));
  return run_forward(m, false, false, s, true);
}

extern "C" int skf_model_greedy_decode(SkfModel* m, const float* embedding, const int* expected_len_host, int n_valid,
                                       long long sos, long long eos, int max_steps, void* out, int* out_len_host,
                                       skf_stream_t stream) {
  SKF_CHECK_ARG(m && m->ws, "model not bound");
  SKF_CHECK_ARG(out, "null output");
  SKF_CHECK_ARG(n_valid > 0 && n_valid <= m->cfg.batch, "n_valid must be in [1, batch]");
  SKF_CHECK_ARG(max_steps > 0 && max_steps <= m->cfg.seq_len, "max_steps must be in [1, seq_len]");
  SKF_CHECK_ARG(m->cfg.do_reconstruction, "the model was built without a decoder (do_reconstruction = 0)");
  if (m->bf16) return run_greedy_decode16(m, embedding, expected_len_host, n_valid, sos, eos, max_steps, out, out_len_host, (hipStream_t)stream);
  return run_greedy_decode(m, embedding, expected_len_host, n_valid, sos, eos, max_steps, out, out_len_host,
                           (hipStream_t)stream);
}

// The embedding gradients' counting sorts depend on the staged tokens only.  Eager path with a decoder: side stream, under
// the forward (the main stream joins the side stream at the first cross-attention, long before the backward reads the
// sorted positions); otherwise (hipGraph capture, encoder-only configurations) in place on the main stream.
int issue_embed_sorts(SkfModel* M, hipStream_t s) {
  const SkfConfig& c = M->cfg;
  const Layout& L = M->lay;
  const Plan& P = M->plan;
  if (!P.emb_sort_bytes) return SKF_OK;
  const int B = c.batch, Le = c.seq_len, Ld = c.seq_len - 1, d = c.d_model;
  hipStream_t ss = s;
  if (M->side && do_recon(c)) {
    // (the staged inputs are all the side stream's first launches read: it waits for the staging launch's own completion signal,
    //  no packet of its own on the main stream)
    hipEvent_t staged = (M->inputs_staged && M->inputs_staged_valid && !g_capturing) ? M->inputs_staged : nullptr;
    if (!staged) {
      staged = M->new_event();
      SKF_CHECK_ARG(staged, "event allocation failed");
      SKF_HIP(hipEventRecord(staged, s));
    }
    SKF_HIP(hipStreamWaitEvent(M->side, staged, 0));
    ss = M->side;
    // first on the side stream: what the forward does not need before its first attention (forward_preamble) - the main stream goes
    // straight to the embedding and the first q|k|v projection (30 us of one-workgroup and short launches off the critical path)
    static const bool pre_off = skf_knob("SKF_NO_SIDE_PREAMBLE") && skf_knob("SKF_NO_SIDE_PREAMBLE")[0] == '1';   // (measurement builds)
    if (!pre_off) {
      hipEvent_t masks = M->new_event(), ready = M->new_event();
      SKF_CHECK_ARG(masks && ready, "event allocation failed");
      SKF_TRY(forward_preamble(M, true, false, ss, masks));
      SKF_HIP(hipEventRecord(ready, ss));
      M->masks_ready = masks; M->pre_ready = ready;
    }
  }
  SKF_TRY(skf_embed_sort(M->at<long long>(P.inp), Le, B, Le, c.vocab_size, M->G(L.enc_emb), d, M->at<char>(P.emb_sort[0]),
                         P.emb_sort_bytes, ss));
  if (do_recon(c))
    SKF_TRY(skf_embed_sort(M->at<long long>(P.tar), Le, B, Ld, c.vocab_size, M->G(L.dec_emb), d, M->at<char>(P.emb_sort[1]),
                           P.emb_sort_bytes, ss));
  return build_row_lists(M, ss);
}

extern "C" int skf_model_forward_backward(SkfModel* m, const void* inp, const void* tar, int tar_ld,
                                          const long long* labels, skf_stream_t stream) {
  SKF_CHECK_ARG(m && m->ws, "model not bound");
  SKF_CHECK_ARG(labels, "null labels");
  hipStream_t s = (hipStream_t)stream;
#if SKF_MEASURE
  g_evlog_main = s; ++g_evlog_step;
#endif
  if (m->bf16) {
    SKF_TRY(stage_with_event(m, s, [&]() { return stage_inputs16(m, inp, tar, tar_ld, labels, s); }));
    return capture_or_run(m, &m->g_fb, s, [&]() -> int {
      SKF_TRY(prologue(m, s));
      SKF_TRY(issue_embed_sorts16(m, s));
      SKF_TRY(run_forward16(m, true, true, s));
      return run_backward16(m, s);
    });
  }
  SKF_TRY(stage_with_event(m, s, [&]() { return stage_inputs(m, inp, tar, tar_ld, labels, s); }));
  return capture_or_run(m, &m->g_fb, s, [&]() -> int {
    SKF_TRY(prologue(m, s));
    SKF_TRY(issue_embed_sorts(m, s));
    SKF_TRY(run_forward(m, true, true, s));
    return run_backward(m, s);
  });
}

extern "C" int skf_model_wait_inputs_staged(SkfModel* m, skf_stream_t stream) {
  SKF_CHECK_ARG(m, "null model");
  SKF_CHECK_ARG(m->inputs_staged && m->inputs_staged_valid, "no call has staged inputs yet");
  SKF_HIP(hipStreamWaitEvent((hipStream_t)stream, m->inputs_staged, 0));
  return SKF_OK;
}

extern "C" int skf_model_apply_gradients(SkfModel* m, float grad_scale, skf_stream_t stream) {
  SKF_CHECK_ARG(m && m->ws, "model not bound");
  hipStream_t s = (hipStream_t)stream;
  if (m->g_opt && m->g_opt_scale != grad_scale) { (void)hipGraphExecDestroy(m->g_opt); m->g_opt = nullptr; }
  m->g_opt_scale = grad_scale;
  return capture_or_run(m, &m->g_opt, s, [&]() -> int {
    if (m->cfg.optimizer == 1)   // tf.keras.optimizers.SGD(lr_schedule, momentum) - the Adam m buffer is the velocity slot
      SKF_TRY(skf_sgd_momentum_step(m->params, m->grads, m->m, m->lay.total, m->state, grad_scale, m->cfg.momentum, s));
    else      // (the Adam sweep of a whole step also advances optimizer.iterations: one launch instead of two)
      return skf_adam_step_launch(m->params, m->grads, m->m, m->v, m->lay.total, m->state, grad_scale, m->cfg.beta1, m->cfg.beta2, m->cfg.eps, 1, s);
    return skf_step_epilogue(m->state, s);
  });
}

extern "C" int skf_model_grad_buckets(SkfModel* m, int max_buckets, size_t* offsets_host, size_t* counts_host) {
  SKF_CHECK_ARG(m && offsets_host && counts_host && max_buckets >= 2, "bad argument");
  const size_t dec_off = m->lay.dec_off;
  if (m->n_buckets == 2) {
    offsets_host[0] = dec_off; counts_host[0] = m->lay.total - dec_off;      // decoder embedding .. output layer
    offsets_host[1] = 0; counts_host[1] = dec_off;                           // encoder .. expander
  } else {
    offsets_host[0] = 0; counts_host[0] = m->lay.total;
  }
  return m->n_buckets;
}

extern "C" int skf_model_wait_grad_bucket(SkfModel* m, int bucket, skf_stream_t stream) {
  SKF_CHECK_ARG(m && bucket >= 0 && bucket < m->n_buckets, "bad bucket");
  if (m->bucket_ready[bucket]) SKF_HIP(hipStreamWaitEvent((hipStream_t)stream, m->bucket_ready[bucket], 0));
  return SKF_OK;
}

extern "C" int skf_model_apply_gradients_range(SkfModel* m, size_t offset, size_t count, float grad_scale, int last,
                                               skf_stream_t stream) {
  SKF_CHECK_ARG(m && m->ws, "model not bound");
  SKF_CHECK_ARG((offset & 3) == 0 && offset + count <= m->lay.total, "range must start on a multiple of 4 floats inside the buffer");
  hipStream_t s = (hipStream_t)stream;
  if (count) {
    if (m->cfg.optimizer == 1)
      SKF_TRY(skf_sgd_momentum_step(m->params + offset, m->grads + offset, m->m + offset, count, m->state, grad_scale,
                                    m->cfg.momentum, s));
    else
      SKF_TRY(skf_adam_step(m->params + offset, m->grads + offset, m->m + offset, m->v + offset, count, m->state,
                            grad_scale, m->cfg.beta1, m->cfg.beta2, m->cfg.eps, s));
  }
  return last ? skf_step_epilogue(m->state, s) : SKF_OK;
}

extern "C" int skf_model_buffer_info(SkfModel* m, const char* name, void** ptr, int* rows, int* cols, int* ld, int* is_bf16) {
  SKF_CHECK_ARG(m && m->ws && name && ptr && rows && cols && ld && is_bf16, "bad argument");
  if (m->bf16) {
    auto it = m->p16.named.find(name);
    if (it == m->p16.named.end()) { skf_set_error("skf_model_buffer_info: unknown buffer '%s'", name); return SKF_EINVAL; }
    *ptr = m->ws + it->second.off; *rows = it->second.rows; *cols = it->second.cols; *ld = it->second.ld; *is_bf16 = it->second.bf16;
    return SKF_OK;
  }
  float* p = nullptr;
  int rc = skf_model_buffer(m, name, &p, rows, cols);
  if (rc) return rc;
  *ptr = p; *ld = *cols; *is_bf16 = 0;
  return SKF_OK;
}

extern "C" int skf_model_buffer(SkfModel* m, const char* name, float** ptr, int* rows, int* cols) {
  SKF_CHECK_ARG(m && m->ws && name && ptr, "bad argument");
  if (m->bf16) {
    auto it16 = m->p16.named.find(name);
    if (it16 == m->p16.named.end() || it16->second.bf16) {
      skf_set_error("skf_model_buffer: '%s' is not an fp32 buffer of this bf16 model (use skf_model_buffer_info)", name);
      return SKF_EINVAL;
    }
    *ptr = reinterpret_cast<float*>(m->ws + it16->second.off);
    if (rows) *rows = it16->second.rows;
    if (cols) *cols = it16->second.cols;
    return SKF_OK;
  }
  auto it = m->named.find(name);
  if (it == m->named.end()) { skf_set_error("skf_model_buffer: unknown buffer '%s'", name); return SKF_EINVAL; }
  *ptr = m->at<float>(it->second.first);
  if (rows) *rows = it->second.second.first;
  if (cols) *cols = it->second.second.second;
  return SKF_OK;
}
