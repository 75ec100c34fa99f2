// decode_mega.h - host / device structures of the persistent decode step (decode_mega.hip), shared with decoder.hip.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "common.h"

namespace chatts {

struct MegaLayer {          // device table, one entry per decoder layer (pointers of ChattsLayerWeights + this layer's KV cache)
  const float* input_norm;
  const uint16_t* qkv;
  const float* qkv_bias;
  const float* q_norm;
  const float* k_norm;
  const uint16_t* o;
  const float* post_norm;
  const uint16_t* gate_up;
  const uint16_t* down;
  float* kc;                // [n_kv, max_ctx, 128] of the sequence, or the layer's block pool
  float* vc;
};

struct MegaGeom {           // how one projection's row-pair tasks are spread over workgroups and compute waves
  int n, k;                 // weight rows, reduction length (= leading dimension)
  int tasks;                // row pairs (SwiGLU: gate / up pairs = output units)
  int tpw;                  // tasks per workgroup (workgroup b: tasks [b * tpw, (b + 1) * tpw))
  int nact;                 // compute waves that take tasks (wave w: local tasks w, w + nact, ...)
  int vthreads;             // threads of the stand-alone GEMV's workgroup for this shape: its RMSNorm summation order
  int swiglu;
  int pad_;
};

struct MegaSync {           // device memory; everything before `status` is zeroed before every launch
  unsigned grp_count[8 * 32];     // arrival counter per workgroup group (b & 7), 128 bytes apart
  unsigned grp_gen[8 * 32];       // generation word the members of a group poll
  unsigned top_count[32];         // arrivals of the groups' last members
  unsigned status[32];            // sticky: != 0 = a barrier timed out (results since then are invalid)
};

struct MegaHost {           // host-side plan
  MegaGeom geom[5];         // qkv, o, gate_up, down, lm_head
  int nwg;
};

struct MegaParams {
  const MegaLayer* layers;
  MegaGeom geom[5];          // qkv, o, gate_up, down, lm_head (by value: read with scalar loads from the kernarg segment)
  MegaSync* sync;
  unsigned long long* argmax_pairs;     // [nwg] (float bits << 32 | row): each workgroup's best lm_head row
  float* x;                 // [H] residual stream (input embedding at entry)
  float* qkv;               // [(n_q + 2 n_kv) * 128]
  float* attn;              // [n_q * 128]
  float* act;               // [inter]
  float* logits;            // [vocab_local]
  float* part_o;            // attention partials [n_q, n_splits, 128]
  float* part_ml;           // [n_q, n_splits, 2]
  const float* final_norm;
  const uint16_t* lm_head;
  const uint16_t* embed;
  const float* cos_tab;
  const float* sin_tab;
  const int32_t* kv_table;  // block table row of the sequence (paged cache) or NULL
  int32_t* pos_dev;
  int32_t* step_dev;
  int64_t* token_dev;
  float* token_logit_dev;
  int64_t* out_tokens;
  int64_t vocab_offset, embed_rows, embed_offset;
  float eps;
  int n_layers, hidden, n_q, n_kv, max_ctx, n_splits, kv_log_block;
  int greedy_tail;          // 1: argmax + decode-loop state + next input embedding inside the launch
  int nwg, xs_bytes;
  // optional profiling (chatts_decoder_mega_profile): s_memtime stamps of two workgroups, [2][n_phases][16] uint64:
  //   0 phase start (master), 1 after A, 2 published + drained, 3 grid barrier passed, 4 after B, 5 after C (staged),
  //   6 / 7 compute wave 0: first / last cycle of its block loop, 8..15: start of its first eight blocks
  unsigned long long* prof;
};

size_t mega_state_bytes(int n_layers, int nwg);
int mega_lds_bytes(const MegaHost& h);
bool mega_plan(MegaHost* h, int hidden, int n_q, int n_kv, int inter, int64_t vocab_local, int cus);
int mega_launch(const MegaParams& p, const MegaHost& h, hipStream_t s);

}  // namespace chatts
