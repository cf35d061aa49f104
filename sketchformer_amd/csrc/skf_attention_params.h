// Launch parameters shared by the fp32-path attention kernels (skf_attention.hip, skf_attention_bwd2.hip).
#pragma once
#include "skf_common.h"

struct AttnParams {
  const float* Q; const float* K; const float* V; float* O;
  int ldq, ldk, ldv, ldo;
  const unsigned char* key_mask;  // (B, key_mask_ld) 1 = masked key, or null
  int key_mask_ld;
  int causal;
  int B, H, Lq, Lk;
  float* stats;                   // (B, H, Lq, 2): row max, 1/sum
  // backward only
  int xcd_remap;                  // XCD-contiguous (sample, head) ids (default on; env SKF_ATTN_XCD=0 turns it off)
  int ablate;                     // diagnostics (env SKF_ATTN_ABLATE): 1 = no dQ atomics
  long long* dbg;                 // diagnostics: s_memtime stamps of a few workgroups (-DSKF_MEASURE=1 builds only, env SKF_ATTN_DBG; always null in the shipped library)
  const float* dO; int lddo;
  float* dQ; float* dK; float* dV;
  int lddq, lddk, lddv;
  const int* order;               // optional (B): the samples sorted by cost, heaviest first (skf_sample_order): workgroups are dealt over the shader engines (skf_deal_rank)
  const int* q_live;              // optional (B): query rows >= q_live[b] have dO == 0 exactly (skf_target_live_len); one-pass
                                  // backward: their tiles are neither staged nor visited, dQ is stored as zeros
};

// two-pass backward on the bf16 matrix cores with exactly split fp32 operands (head size 16 or 32); returns SKF_OK after launching
int skf_attention_bwd2_launch(const AttnParams& p, int dh, hipStream_t st);

// round 5, head size 16: the two passes as independent workgroups of one launch (skf_attention_bwd3.hip); returns SKF_OK after launching
int skf_attention_bwd3_supported(int dh, int Lq, int Lk);
int skf_attention_bwd3_launch(const AttnParams& p, hipStream_t st);

// any head size <= 128 / sequence <= 1024 (skf_generic.hip): plain fp32 FMA kernels, the same semantics and statistics
int skf_attention_any_supported(int dh, int Lq, int Lk);
int skf_attention_fwd_any(const AttnParams& p, int dh, hipStream_t st);
int skf_attention_bwd_any(const AttnParams& p, int dh, hipStream_t st);
