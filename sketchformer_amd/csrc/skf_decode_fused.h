// Argument block of the one-launch-per-position greedy decode (skf_decode_fused.hip); internal to libskf.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

constexpr int SKF_DEC_MAX_LAYERS = 8;

struct SkfDecDense { const float* w; const float* b; int in, out, ld, vec4; };   // W[in][out], row stride ld; vec4: 16-byte column groups allowed

struct SkfDecLayer {
  SkfDecDense qkv, o, q2, o2, f1, f2;
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *ln3_g, *ln3_b;
  float* cache;           // (B, Le, 2d): self-attention K | V rows of the positions decoded so far
  const float* kv2;       // (B, Le, 2d): cross-attention K | V of pre_decoder
};

struct SkfDecodeFused {
  int B, Le, d, H, F, N, Vout, vocab, blind, hs_len;
  SkfDecLayer layer[SKF_DEC_MAX_LAYERS];
  SkfDecDense out;
  const float* emb_table; const float* embd_w; const float* embd_b; const float* pos;
  long long* tokens; float* cont; int Ti;         // running output image (B, Ti[, 5]); exactly one of tokens / cont
  unsigned char* selfmask; int mask_ld;
  int* eos_seen; int* done_step; int* step_dev; int* ticket;
  const long long* dyn;                           // [0] n_valid, [1] eos
  const int* limit;                               // per-sample cross-attention key limit (non-blind) or null
};

bool skf_decode_fused_supported(int d, int H, int F, int Le, int N, int Vout);
int skf_decode_fused_launch(const SkfDecodeFused& p, hipStream_t st);
