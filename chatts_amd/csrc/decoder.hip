// decoder.hip - host-side composition: chatts_linear dispatch and the Qwen2/Qwen3 decoder schedule.
// Replaces Qwen2TSForCausalLM.forward / compute_logits (NetManAIOps/ChatTS chatts/vllm/chatts_vllm.py:576-610),
// whose decoder is vLLM's Qwen2Model (NOT IN REFERENCE); layer math per oracle/qwen_decoder.py.
// Nothing here allocates device memory or synchronises: every call only enqueues kernels on the
// caller's stream, so a whole decode step can be captured into one hipGraph and replayed per token.
#include <new>
#include <vector>

#include "common.h"
#include "tp_common.h"

namespace chatts {
int launch_gemv(const ChattsLinearArgs* a, hipStream_t s);
int launch_gemm(const ChattsLinearArgs* a, hipStream_t s, const RopeFuse* rope = nullptr, bool* rope_done = nullptr);
int attention_decode_batched_impl(const float* qkv_raw, int batch, int n_q, int n_kv, const float* q_norm_w, const float* k_norm_w,
                                  float norm_eps, const float* cos_tab, const float* sin_tab, int pos, const int32_t* pos_dev,
                                  const ChattsKvCache* cache, size_t seq_stride, float* out, uint16_t* out_hi, uint16_t* out_lo,
                                  int n_splits, void* workspace, size_t workspace_bytes, chatts_stream_t stream);
int attention_impl(const float* qkv, int t, int n_q, int n_kv, int pos0, const int32_t* pos0_dev, const ChattsKvCache* cache, float* out,
                   uint16_t* out_hi, uint16_t* out_lo, int n_splits, void* workspace, size_t workspace_bytes, chatts_stream_t stream);
int launch_split_bf16x2(const float* x, int m, int k, int ldx, uint16_t* hi, uint16_t* lo, int ldp, hipStream_t s);
size_t gemm_workspace(int m, int n, int k);
}  // namespace chatts

using namespace chatts;

struct ChattsDecoder {
  ChattsDecoderConfig cfg;
  ChattsDecoderWeights w;
  std::vector<ChattsLayerWeights> layers;
  ChattsDecoderBuffers b;
  int cur_seq = 0;          // sequence (KV-cache slot) the single-sequence entry points operate on
  bool chain = false;       // set by the entry points that run the layers back to back themselves (chatts_decoder_prefill,
                            // decode_step_batched): only then may a projection write the NEXT projection's normed operand
  bool x_in_xn = false;     // batched step: o_proj left the residual stream in b.xn (ping-pong, see layer_part_batched); down_proj brings it back
  bool normed = false;      // planes 0 already hold RMSNorm(x) for the projection that comes next (written by the previous
                            // projection's fused epilogue): norm_into only binds them
  bool sampling = false;    // token selection of the decode steps: greedy argmax, or the sampler with `sa`
  ChattsSamplingArgs sa{};
  ChattsTpComm* tp = nullptr;   // tensor-parallel exchange (borrowed); required by the whole-step entry points when tp_world > 1
  bool prefill_f16q = false; // opt-in parity-grade mode (chatts_decoder_set_prefill_f16q): prefill chunks of >= 96 rows on the f16 + e4m3 operand split
  bool normed_q = false;    // the f16q planes A already hold RMSNorm(x) for the projection that comes next (layer_part_f16q only)
  bool prefill_fp8 = false; // SPEED MODE (chatts_decoder_set_prefill_fp8): prefill chunks of >= 16 rows multiply fp8 x fp8 (gemm_fp8.hip)
  bool fuse_tp = false;     // set by chatts_decoder_decode_step: the M == 1 o_proj / down_proj GEMVs carry the exchange in their own launch
  bool tp_fused = false;    // ... and whether the last layer part's projection did (otherwise the caller launches chatts_allreduce)
};

static int64_t embed_rows(const ChattsDecoder* d) { return d->cfg.embed_rows > 0 ? d->cfg.embed_rows : d->cfg.vocab_local; }
static int64_t embed_offset(const ChattsDecoder* d) { return d->cfg.embed_rows > 0 ? d->cfg.embed_offset : d->cfg.vocab_offset; }

extern "C" size_t chatts_linear_workspace(int m, int n, int k) { return gemm_workspace(m, n, k); }

static int linear_impl(const ChattsLinearArgs* a, chatts_stream_t stream, const RopeFuse* rope, bool* rope_done);
extern "C" int chatts_linear(const ChattsLinearArgs* a, chatts_stream_t stream) { return linear_impl(a, stream, nullptr, nullptr); }

// rope / rope_done: the qkv projection of a prefill chunk may carry rope_kv_kernel's work in its split-K epilogue (*rope_done says
// whether it did; otherwise the caller launches chatts_rope_kv_write as before)
static int linear_impl(const ChattsLinearArgs* a, chatts_stream_t stream, const RopeFuse* rope, bool* rope_done) {
  CHATTS_REQUIRE(a != nullptr, CHATTS_E_BADARG, "linear: null args");
  CHATTS_REQUIRE(a->m >= 0 && a->n > 0 && a->k > 0, CHATTS_E_BADARG, "linear: bad sizes m=%d n=%d k=%d", a->m, a->n, a->k);
  if (a->m == 0) return CHATTS_OK;
  const bool planes = a->a_hi || a->a_lo;
  const bool cplanes = a->c_hi || a->c_lo;
  CHATTS_REQUIRE((a->a || planes) && a->w && (a->c || cplanes), CHATTS_E_BADARG, "linear: null pointer");
  if (cplanes)
    CHATTS_REQUIRE(a->c_hi && a->c_lo && a->m > 1 && a->ld_cplanes >= (a->epilogue == CHATTS_EPI_SWIGLU ? a->n / 2 : a->n),
                   CHATTS_E_SHAPE, "linear: plane output needs both planes, M > 1 and ld_cplanes >= the output width");
  if (planes)
    CHATTS_REQUIRE(a->a_hi && a->a_lo && a->ld_planes >= a->k && a->ld_planes % 8 == 0 && ((uintptr_t)a->a_hi % 16) == 0 &&
                       ((uintptr_t)a->a_lo % 16) == 0 && a->m > 1 && !a->norm_w,
                   CHATTS_E_SHAPE, "linear: pre-split A needs both planes, ld_planes >= K and %% 8, 16-byte alignment, M > 1");
  CHATTS_REQUIRE(a->epilogue >= CHATTS_EPI_NONE && a->epilogue <= CHATTS_EPI_SWIGLU, CHATTS_E_BADARG,
                 "linear: epilogue %d", a->epilogue);
  CHATTS_REQUIRE(a->epilogue != CHATTS_EPI_RESID || a->resid, CHATTS_E_BADARG, "linear: EPI_RESID without resid");
  CHATTS_REQUIRE(a->k % 32 == 0, CHATTS_E_SHAPE, "linear: K=%d must be a multiple of 32 (pad the weight)", a->k);
  CHATTS_REQUIRE(a->n % 16 == 0 && (a->epilogue != CHATTS_EPI_SWIGLU || a->n % 32 == 0), CHATTS_E_SHAPE,
                 "linear: N=%d must be a multiple of 16 (32 for SwiGLU)", a->n);
  CHATTS_REQUIRE((!a->a || (a->lda >= a->k && a->lda % 4 == 0)) && a->ldw >= a->k && a->ldw % 8 == 0, CHATTS_E_SHAPE,
                 "linear: leading dimensions lda=%d ldw=%d", a->lda, a->ldw);
  const int ncols = a->epilogue == CHATTS_EPI_SWIGLU ? a->n / 2 : a->n;
  CHATTS_REQUIRE(a->ldc >= ncols || (cplanes && a->epilogue != CHATTS_EPI_RESID), CHATTS_E_SHAPE, "linear: ldc=%d < %d", a->ldc, ncols);
  CHATTS_REQUIRE(((uintptr_t)a->a % 16) == 0 && ((uintptr_t)a->w % 16) == 0 && ((uintptr_t)a->c % 16) == 0,
                 CHATTS_E_SHAPE, "linear: pointers must be 16-byte aligned");
  if (a->w8)
    CHATTS_REQUIRE(a->w8_scale && a->ldw8 >= a->k && a->ldw8 % 16 == 0 && ((uintptr_t)a->w8 % 16) == 0 && a->k % 16 == 0,
                   CHATTS_E_SHAPE, "linear: fp8 weights need a scale, ldw8 >= K, 16-byte alignment");
  if (a->w4)
    CHATTS_REQUIRE(a->w4_sz && a->w4_group >= 16 && (a->w4_group & (a->w4_group - 1)) == 0 && a->k % a->w4_group == 0 && a->ldw4 >= a->k / 2 &&
                       a->ldw4 % 8 == 0 && ((uintptr_t)a->w4 % 8) == 0 && ((uintptr_t)a->w4_sz % 8) == 0,
                   CHATTS_E_SHAPE, "linear: 4-bit weights need w4_sz, a power-of-two group size >= 16 that divides K, ldw4 >= K/2 and %% 8");
  if (a->post_norm_w)
    CHATTS_REQUIRE(a->m > 1 && a->c && a->post_hi && a->post_lo && a->ld_post >= a->n && a->ld_post % 4 == 0 && a->n % 4 == 0 &&
                       (a->epilogue == CHATTS_EPI_NONE || a->epilogue == CHATTS_EPI_RESID) && !cplanes,
                   CHATTS_E_BADARG, "linear: post-norm planes need M > 1, float32 c, EPI_NONE / EPI_RESID, both planes, ld_post >= N");
  if (a->tp_reduce) CHATTS_REQUIRE(a->m == 1, CHATTS_E_BADARG, "linear: tp_reduce is available for M == 1 (the decode GEMV) only");
  if (a->m == 1 && a->epilogue != CHATTS_EPI_GELU) return launch_gemv(a, as_stream(stream));
  CHATTS_REQUIRE(a->norm_w == nullptr, CHATTS_E_BADARG, "linear: fused RMSNorm is only available for M == 1");
  return launch_gemm(a, as_stream(stream), rope, rope_done);
}

extern "C" int chatts_split_bf16x2(const float* x, int m, int k, int ldx, chatts_bf16* hi, chatts_bf16* lo, int ld_planes,
                                   chatts_stream_t stream) {
  CHATTS_REQUIRE(m >= 0 && k >= 0 && k % 8 == 0, CHATTS_E_SHAPE, "split_bf16x2: m=%d k=%d (K must be a multiple of 8)", m, k);
  if (m == 0 || k == 0) return CHATTS_OK;
  CHATTS_REQUIRE(x && hi && lo, CHATTS_E_BADARG, "split_bf16x2: null pointer");
  CHATTS_REQUIRE(ldx >= k && ldx % 4 == 0 && ld_planes >= k && ld_planes % 8 == 0, CHATTS_E_SHAPE,
                 "split_bf16x2: leading dimensions ldx=%d ld_planes=%d", ldx, ld_planes);
  CHATTS_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)hi % 16) == 0 && ((uintptr_t)lo % 16) == 0, CHATTS_E_SHAPE,
                 "split_bf16x2: pointers must be 16-byte aligned");
  return launch_split_bf16x2(x, m, k, ldx, hi, lo, ld_planes, as_stream(stream));
}

constexpr int kF16qMinRows = 96;      // f16q mode: chunks below this take the default kernels (also parity grade)

extern "C" size_t chatts_decoder_workspace(const ChattsDecoderConfig* c, int t_max, int n_splits_max) {
  if (!c) return 0;
  size_t ws = 0;
  const int qkv_n = (c->n_q + 2 * c->n_kv) * c->head_dim;
  const int shapes[4][2] = {{qkv_n, c->hidden}, {c->hidden, c->n_q * c->head_dim}, {2 * c->inter, c->hidden},
                            {c->hidden, c->inter}};
  for (int t = 2; t <= t_max; ++t) {   // geometry depends on M; take the max over all M (cheap: <= t_max iterations)
    for (auto& s : shapes) {
      size_t w = gemm_workspace(t, s[0], s[1]);
      if (t >= kF16qMinRows && s[1] % 128 == 0) { const size_t wq = chatts_linear_f16q_workspace(t, s[0], s[1]); if (wq > w) w = wq; }
      if (w > ws) ws = w;
    }
  }
  size_t aw = chatts_attn_workspace(1, c->n_q, n_splits_max);
  const size_t pw = chatts_attn_workspace(t_max, c->n_q, 2);      // prefill attention with the keys split in two
  if (pw > aw) aw = pw;
  return (ws > aw ? ws : aw) + 256;
}

extern "C" ChattsDecoder* chatts_decoder_create(const ChattsDecoderConfig* c, const ChattsDecoderWeights* w,
                                                const ChattsDecoderBuffers* b) {
  if (!c || !w || !b || !w->layers) { set_error("decoder_create: null argument"); return nullptr; }
  if (c->head_dim != kHeadDim) { set_error("decoder_create: head_dim %d != 128", c->head_dim); return nullptr; }
  if (c->n_q % c->n_kv != 0 || c->n_q / c->n_kv > 8) { set_error("decoder_create: unsupported GQA group"); return nullptr; }
  if (c->hidden % 32 || c->inter % 32) { set_error("decoder_create: hidden/inter must be multiples of 32"); return nullptr; }
  if (b->kv_block_table) {
    const int B = b->kv_block_size;
    if (kv_log_block(B) < 0 || c->max_ctx % B != 0 || b->kv_table_stride < c->max_ctx / B || b->kv_pool_blocks < 1) {
      set_error("decoder_create: paged KV needs a power-of-two block size in 64..32768 dividing max_ctx (%d), a table row of >= "
                "max_ctx / block_size entries and a non-empty pool (block %d, row %d, pool %d)", c->max_ctx, B, b->kv_table_stride,
                b->kv_pool_blocks);
      return nullptr;
    }
  }
  ChattsDecoder* d = new (std::nothrow) ChattsDecoder();
  if (!d) { set_error("decoder_create: out of host memory"); return nullptr; }
  d->cfg = *c;
  d->w = *w;
  d->layers.assign(w->layers, w->layers + c->n_layers);
  d->w.layers = d->layers.data();
  d->b = *b;
  return d;
}

extern "C" void chatts_decoder_destroy(ChattsDecoder* d) { delete d; }

static size_t seq_stride(const ChattsDecoder* d) {      // floats between the caches of consecutive sequences
  return (size_t)d->cfg.n_layers * d->cfg.n_kv * d->cfg.max_ctx * kHeadDim;
}

static ChattsKvCache layer_cache(const ChattsDecoder* d, int layer, int seq) {
  ChattsKvCache c{};
  c.max_ctx = d->cfg.max_ctx;
  if (d->b.kv_block_table) {      // block-paged: the layer's pool + this sequence's row of the block table
    const size_t pool = (size_t)d->b.kv_pool_blocks * d->cfg.n_kv * d->b.kv_block_size * kHeadDim;
    c.k = d->b.kv_k + pool * layer;
    c.v = d->b.kv_v + pool * layer;
    c.block_table = d->b.kv_block_table + (size_t)seq * d->b.kv_table_stride;
    c.block_size = d->b.kv_block_size;
    c.table_stride = d->b.kv_table_stride;
    return c;
  }
  const size_t per = (size_t)d->cfg.n_kv * d->cfg.max_ctx * kHeadDim;
  c.k = d->b.kv_k + seq_stride(d) * seq + per * layer;
  c.v = d->b.kv_v + seq_stride(d) * seq + per * layer;
  return c;
}

// Projections whose A operand is a pair of bf16 hi / lo planes: prefill chunks of >= 96 rows (LDS-DMA GEMM) and batched
// decode with 2..16 sequences on bf16 weights (weight-streaming kernel).
static bool planes_path(const ChattsDecoder* d, int m, int k, bool fp8 = false) {
  if (!(d->b.planes_hi && d->b.planes_lo && d->b.planes2_hi && d->b.planes2_lo) || k % 64 != 0) return false;
  if (m >= 2 && m <= 16) return !fp8 || k % 128 == 0;      // (the fp8 stream's K-step is one 128-byte line = 128 values)
  return m >= 96 && !fp8;
}

// x -> RMSNorm -> la's input: planes written by the norm kernel itself (plane path) or float32 xn.
static int norm_into(ChattsDecoder* d, const float* norm_w, ChattsLinearArgs* la, chatts_stream_t stream) {
  if (planes_path(d, la->m, la->k, la->w8 != nullptr)) {
    la->a = nullptr; la->a_hi = d->b.planes_hi; la->a_lo = d->b.planes_lo; la->ld_planes = la->k;
    if (d->normed) { d->normed = false; return CHATTS_OK; }
    return chatts_rmsnorm_planes(d->b.x, norm_w, d->b.planes_hi, d->b.planes_lo, la->k, la->m, la->k, d->cfg.rms_eps, stream);
  }
  d->normed = false;
  la->a = d->b.xn;
  return chatts_rmsnorm(d->b.x, norm_w, d->b.xn, la->m, la->k, d->cfg.rms_eps, stream);
}

// The residual-updating projection (o_proj / down_proj, tp_world == 1) also produces the NEXT projection's operand:
// RMSNorm(x) with `next_norm_w` as planes 0, when that projection will take the plane path for the same M.
static bool gemm_post_norm_small_m() { return opt_get(OPT_POST_NORM_SMALL_M, 1) != 0; }
static void request_post_norm(ChattsDecoder* d, ChattsLinearArgs* la, const float* next_norm_w, bool next_fp8) {
  if (!d->chain || d->cfg.tp_world > 1 || !next_norm_w || la->epilogue != CHATTS_EPI_RESID) return;
  // the fused epilogue runs one workgroup per row - fine for a prefill chunk; at batched-decode M it needs several workgroups per row,
  // which re-read resid while others write c: only when the caller gave it a c that does not alias resid (layer_part_batched)
  if (la->m < 64 && !(la->m <= 16 && la->c != la->resid && gemm_post_norm_small_m())) return;
  if (!planes_path(d, la->m, la->n, next_fp8)) return;
  la->post_norm_w = next_norm_w; la->post_norm_eps = d->cfg.rms_eps;
  la->post_hi = d->b.planes_hi; la->post_lo = d->b.planes_lo; la->ld_post = la->n;
  d->normed = true;
}

extern "C" int chatts_decoder_set_sampling(ChattsDecoder* d, const ChattsSamplingArgs* sa) {
  CHATTS_REQUIRE(d, CHATTS_E_BADARG, "decoder_set_sampling: null decoder");
  if (sa) CHATTS_REQUIRE(sa->temperature_rows || (sa->temperature > 0.f && sa->top_p > 0.f), CHATTS_E_BADARG,
                         "decoder_set_sampling: temperature %g / top_p %g must be positive (pass NULL for greedy)",
                         (double)sa->temperature, (double)sa->top_p);
  d->sampling = sa != nullptr;
  if (sa) d->sa = *sa;
  return CHATTS_OK;
}

extern "C" int chatts_decoder_set_tp(ChattsDecoder* d, ChattsTpComm* comm) {
  CHATTS_REQUIRE(d, CHATTS_E_BADARG, "decoder_set_tp: null decoder");
  if (comm) {
    CHATTS_REQUIRE(chatts_tp_world(comm) == d->cfg.tp_world, CHATTS_E_BADARG, "decoder_set_tp: comm of %d ranks, decoder built for %d",
                   chatts_tp_world(comm), d->cfg.tp_world);
    const int mb = d->b.max_batch > 0 ? d->b.max_batch : 1;
    CHATTS_REQUIRE(chatts_tp_max_elems(comm) >= (int64_t)mb * d->cfg.hidden, CHATTS_E_SHAPE,
                   "decoder_set_tp: exchange buffer holds %lld elements, a step needs %lld", (long long)chatts_tp_max_elems(comm),
                   (long long)mb * d->cfg.hidden);
    CHATTS_REQUIRE(d->b.tp_pair_logit && d->b.tp_pair_token && d->b.delta, CHATTS_E_BADARG,
                   "decoder_set_tp: buffers.delta / tp_pair_logit / tp_pair_token are required");
  }
  d->tp = comm;
  return CHATTS_OK;
}

// Greedy / sampled token for `batch` logits rows of this rank, agreed across the TP ranks (see the header).
extern "C" int chatts_decoder_select_tokens(ChattsDecoder* d, const float* logits, int batch, int64_t logits_stride, int64_t* token,
                                            float* token_logit, int64_t* out_tokens, int64_t out_stride, int32_t* step_dev,
                                            int32_t* pos_dev, int pos_limit, const ChattsSamplingArgs* override_sa,
                                            chatts_stream_t stream) {
  CHATTS_REQUIRE(d && logits && token && batch >= 1, CHATTS_E_BADARG, "decoder_select_tokens: bad arguments");
  StageRange stage("chatts.select_tokens");
  const ChattsDecoderConfig& c = d->cfg;
  const ChattsSamplingArgs* sa = override_sa ? override_sa : (d->sampling ? &d->sa : nullptr);
  if (c.tp_world <= 1) {
    if (sa)
      return chatts_sample_batched(logits, batch, logits_stride, c.vocab_local, c.vocab_offset, sa, token, token_logit, out_tokens,
                                   out_stride, step_dev, pos_dev, pos_limit, stream);
    // (the split-K / attention workspace is idle here: every kernel that used it is ahead of this call on the stream)
    return chatts::argmax_batched_scratch(logits, batch, logits_stride, c.vocab_local, c.vocab_offset, token, token_logit, out_tokens,
                                          out_stride, step_dev, pos_dev, pos_limit, d->b.workspace, d->b.workspace_bytes,
                                          reinterpret_cast<hipStream_t>(stream));
  }
  CHATTS_REQUIRE(d->tp, CHATTS_E_BADARG, "decoder_select_tokens: tp_world = %d but no exchange attached (chatts_decoder_set_tp)", c.tp_world);
  const int mb = d->b.max_batch > 0 ? d->b.max_batch : 1;
  CHATTS_REQUIRE(batch <= mb, CHATTS_E_SHAPE, "decoder_select_tokens: batch %d exceeds max_batch %d", batch, mb);
  int rc;
  if (!sa) {      // one (max logit, global id) pair per rank and sequence instead of the [V / W] logits
    if ((rc = chatts_argmax_batched(logits, batch, logits_stride, c.vocab_local, c.vocab_offset, d->b.tp_pair_token, d->b.tp_pair_logit,
                                    nullptr, 0, nullptr, nullptr, 0, stream)) != 0) return rc;
    return chatts_tp_argmax(d->tp, batch, d->b.tp_pair_logit, d->b.tp_pair_token, token, token_logit, out_tokens, out_stride, step_dev,
                            pos_dev, pos_limit, stream);
  }
  // sampling: every rank draws from the gathered full-vocabulary logits with the same counter-hash variate -> the same token
  CHATTS_REQUIRE(d->b.logits_full, CHATTS_E_BADARG, "decoder_select_tokens: sampling under tensor parallelism needs buffers.logits_full");
  CHATTS_REQUIRE(logits_stride == c.vocab_local || batch == 1, CHATTS_E_SHAPE, "decoder_select_tokens: logits rows must be contiguous");
  if ((rc = chatts_allgather(d->tp, logits, d->b.logits_full, batch, c.vocab_local, stream)) != 0) return rc;
  const int64_t vfull = c.vocab_local * c.tp_world;
  return chatts_sample_batched(d->b.logits_full, batch, vfull, vfull, 0, sa, token, token_logit, out_tokens, out_stride, step_dev,
                               pos_dev, pos_limit, stream);
}

extern "C" int chatts_decoder_select_sequence(ChattsDecoder* d, int seq) {
  CHATTS_REQUIRE(d && seq >= 0 && seq < (d->b.max_batch > 0 ? d->b.max_batch : 1), CHATTS_E_BADARG,
                 "decoder_select_sequence: slot %d out of range", seq);
  d->cur_seq = seq;
  return CHATTS_OK;
}

// ---- SPEED MODE: a prefill chunk's layer half on the fp8 matrix pipe (gemm_fp8.hip) ----------------------------------------------
// Activation rows are quantised per token (RMSNorm fused), the four projections are fp8 x fp8 GEMMs on the e4m3 weight copies;
// RoPE / cache write / attention / residual stream stay float32.  Scratch: the bf16 plane buffers as byte arrays, xn for the row scales.
extern "C" int chatts_decoder_set_prefill_fp8(ChattsDecoder* d, int on) {
  CHATTS_REQUIRE(d, CHATTS_E_BADARG, "decoder_set_prefill_fp8: null decoder");
  if (on) {
    const ChattsDecoderConfig& c = d->cfg;
    CHATTS_REQUIRE(c.w8_format == CHATTS_W8_FP8, CHATTS_E_BADARG, "decoder_set_prefill_fp8: the 8-bit weight copies are not e4m3");
    for (const ChattsLayerWeights& lw : d->layers)
      CHATTS_REQUIRE(lw.qkv8 && lw.o8 && lw.gate_up8 && lw.down8, CHATTS_E_BADARG, "decoder_set_prefill_fp8: a layer has no fp8 weight copies "
                     "(weight_format=\"fp8\")");
    CHATTS_REQUIRE(c.hidden % 128 == 0 && c.inter % 128 == 0, CHATTS_E_SHAPE, "decoder_set_prefill_fp8: hidden / inter must be multiples of 128");
    CHATTS_REQUIRE(d->b.planes_hi && d->b.planes_lo && d->b.planes2_hi && d->b.xn && d->b.t_max >= 1, CHATTS_E_BADARG,
                   "decoder_set_prefill_fp8: the plane buffers and xn are needed as scratch");
  }
  d->prefill_fp8 = on != 0;
  return CHATTS_OK;
}

// ---- OPT-IN, PARITY GRADE: a prefill chunk's layer half on the f16q operand split (gemm_f16q.hip) -------------------------------------
// The four projections multiply f16 high parts on the f16 MFMA and e4m3 residuals x e4m3 weights on the block-scaled fp8 MFMA; every
// activation matrix between them travels as f16q planes written by its producer (RMSNorm, attention output split, SwiGLU epilogue, the
// residual projections' post-norm).  Scratch: the bf16 plane buffers re-typed - pair 0 / pair 1 = (f16 hi | e4m3 lo, then the e8m0 scales).
extern "C" int chatts_decoder_set_prefill_f16q(ChattsDecoder* d, int on) {
  CHATTS_REQUIRE(d, CHATTS_E_BADARG, "decoder_set_prefill_f16q: null decoder");
  if (on) {
    const ChattsDecoderConfig& c = d->cfg;
    for (const ChattsLayerWeights& lw : d->layers)
      CHATTS_REQUIRE(lw.qkv16 && lw.qkv_q8 && lw.qkv_q8e && lw.o16 && lw.o_q8 && lw.o_q8e && lw.gate_up16 && lw.gate_up_q8 && lw.gate_up_q8e &&
                         lw.down16 && lw.down_q8 && lw.down_q8e, CHATTS_E_BADARG, "decoder_set_prefill_f16q: a layer has no (tiled) f16q weight copies");
    CHATTS_REQUIRE(c.hidden % 128 == 0 && c.inter % 128 == 0, CHATTS_E_SHAPE, "decoder_set_prefill_f16q: hidden / inter must be multiples of 128");
    CHATTS_REQUIRE(d->b.planes_hi && d->b.planes_lo && d->b.planes2_hi && d->b.planes2_lo && d->b.t_max >= 1, CHATTS_E_BADARG,
                   "decoder_set_prefill_f16q: the plane buffers are needed as scratch");
    CHATTS_REQUIRE(!d->prefill_fp8, CHATTS_E_BADARG, "decoder_set_prefill_f16q: the fp8 speed mode is on");
  }
  d->prefill_f16q = on != 0;
  d->normed_q = false;
  return CHATTS_OK;
}

struct F16qBuf { chatts_f16* hi; uint8_t* lo8; uint8_t* sc; };
static F16qBuf f16q_buf(const ChattsDecoder* d, int pair) {
  const ChattsDecoderConfig& c = d->cfg;
  int maxdim = c.hidden > c.inter ? c.hidden : c.inter;
  if (c.n_q * kHeadDim > maxdim) maxdim = c.n_q * kHeadDim;
  const size_t plane = (size_t)((d->b.t_max + 31) / 32 * 32) * maxdim;      // elements of one bf16 plane (rows rounded up to 32: the tiled planes
                                                                           // hold whole 16-row blocks) = bytes of the e4m3 plane
  F16qBuf b;
  b.hi = reinterpret_cast<chatts_f16*>(pair ? d->b.planes2_hi : d->b.planes_hi);
  b.lo8 = reinterpret_cast<uint8_t*>(pair ? d->b.planes2_lo : d->b.planes_lo);
  b.sc = b.lo8 + plane;
  return b;
}

static int layer_part_f16q(ChattsDecoder* d, int layer, int part, int t, int pos0, const int32_t* pos0_dev, chatts_stream_t stream) {
  StageRange stage(part == 0 ? "chatts.layer.attn.f16q" : "chatts.layer.mlp.f16q");
  const ChattsDecoderConfig& c = d->cfg;
  const ChattsLayerWeights& lw = d->layers[layer];
  const bool tp = c.tp_world > 1;
  const int H = c.hidden, na = c.n_q * kHeadDim, qkv_n = (c.n_q + 2 * c.n_kv) * kHeadDim;
  const F16qBuf A = f16q_buf(d, 0), B = f16q_buf(d, 1);
  d->normed = false;
  d->tp_fused = false;
  int rc;
  auto base = [&](ChattsLinearF16qArgs& q, const F16qBuf& in, int k, const chatts_f16* w16, const uint8_t* w8, const uint8_t* w8e, int n) {
    q = ChattsLinearF16qArgs{};
    q.a_hi = in.hi; q.a_lo8 = in.lo8; q.a_scale = in.sc; q.ld_a = k; q.ld_scale = k / 128;
    q.w16 = w16; q.w8 = w8; q.w8_exp = w8e; q.ldw = k; q.m = t; q.n = n; q.k = k;
    q.workspace = d->b.workspace; q.workspace_bytes = d->b.workspace_bytes;
    q.planes_tiled = 1; q.w_tiled = 1;      // every plane set and both weight copies in the kernel's own 1 KB / 512 B block order
  };
  // a residual projection also writes RMSNorm(x) for the projection after it - only inside the entry points that run the layers back to back
  auto post = [&](ChattsLinearF16qArgs& q, const float* next_norm_w) {
    if (!d->chain || tp || !next_norm_w) return;
    q.post_norm_w = next_norm_w; q.post_norm_eps = c.rms_eps;
    q.post_hi = A.hi; q.post_lo8 = A.lo8; q.post_scale = A.sc; q.ld_post = H; q.ld_pscale = H / 128;
    d->normed_q = true;
  };
  ChattsLinearF16qArgs q;
  if (part == 0) {
    if (!d->normed_q && (rc = chatts_rmsnorm_f16q(d->b.x, lw.input_norm, A.hi, A.lo8, A.sc, H, H / 128, t, H, c.rms_eps, 1, stream)) != 0) return rc;
    d->normed_q = false;
    base(q, A, H, lw.qkv16, lw.qkv_q8, lw.qkv_q8e, qkv_n);
    q.bias = lw.qkv_bias; q.c = d->b.qkv; q.ldc = qkv_n; q.epilogue = CHATTS_EPI_NONE;
    if ((rc = chatts_linear_f16q(&q, stream)) != 0) return rc;
    ChattsKvCache kc = layer_cache(d, layer, d->cur_seq);
    if ((rc = chatts_rope_kv_write(d->b.qkv, t, c.n_q, c.n_kv, lw.q_norm, lw.k_norm, c.rms_eps, d->w.cos_tab, d->w.sin_tab, pos0, pos0_dev,
                                   &kc, stream)) != 0) return rc;
    if ((rc = attention_impl(d->b.qkv, t, c.n_q, c.n_kv, pos0, pos0_dev, &kc, d->b.attn, nullptr, nullptr, 1, d->b.workspace,
                             d->b.workspace_bytes, stream)) != 0) return rc;
    if ((rc = chatts_split_f16q(d->b.attn, t, na, na, A.hi, A.lo8, A.sc, na, na / 128, 1, stream)) != 0) return rc;
    base(q, A, na, lw.o16, lw.o_q8, lw.o_q8e, H);
    q.ldc = H;
    if (tp) { q.c = d->b.delta; q.epilogue = CHATTS_EPI_NONE; }
    else { q.c = d->b.x; q.resid = d->b.x; q.epilogue = CHATTS_EPI_RESID; post(q, lw.post_norm); }
    return chatts_linear_f16q(&q, stream);
  }
  if (!d->normed_q && (rc = chatts_rmsnorm_f16q(d->b.x, lw.post_norm, A.hi, A.lo8, A.sc, H, H / 128, t, H, c.rms_eps, 1, stream)) != 0) return rc;
  d->normed_q = false;
  base(q, A, H, lw.gate_up16, lw.gate_up_q8, lw.gate_up_q8e, 2 * c.inter);
  q.epilogue = CHATTS_EPI_SWIGLU; q.ldc = c.inter;
  q.c_hi = B.hi; q.c_lo8 = B.lo8; q.c_scale = B.sc; q.ld_cplanes = c.inter; q.ld_cscale = c.inter / 128;
  if ((rc = chatts_linear_f16q(&q, stream)) != 0) return rc;
  base(q, B, c.inter, lw.down16, lw.down_q8, lw.down_q8e, H);
  q.ldc = H;
  if (tp) { q.c = d->b.delta; q.epilogue = CHATTS_EPI_NONE; }
  else { q.c = d->b.x; q.resid = d->b.x; q.epilogue = CHATTS_EPI_RESID; post(q, layer + 1 < c.n_layers ? d->layers[layer + 1].input_norm : nullptr); }
  return chatts_linear_f16q(&q, stream);
}

static int layer_part_fp8(ChattsDecoder* d, int layer, int part, int t, int pos0, const int32_t* pos0_dev, chatts_stream_t stream) {
  const ChattsDecoderConfig& c = d->cfg;
  const ChattsLayerWeights& lw = d->layers[layer];
  const bool tp = c.tp_world > 1;
  const int H = c.hidden, na = c.n_q * kHeadDim, qkv_n = (c.n_q + 2 * c.n_kv) * kHeadDim;
  uint8_t* a8 = reinterpret_cast<uint8_t*>(d->b.planes_hi);
  uint8_t* b8 = reinterpret_cast<uint8_t*>(d->b.planes_lo);
  float* sa = d->b.xn;
  float* sb = d->b.xn + d->b.t_max;
  d->normed = false;
  d->tp_fused = false;
  int rc;
  ChattsLinearFp8Args f{};
  if (part == 0) {
    if ((rc = chatts_quantize_rows_fp8(d->b.x, t, H, H, lw.input_norm, c.rms_eps, a8, H, sa, stream)) != 0) return rc;
    f.a8 = a8; f.a_scale = sa; f.w8 = lw.qkv8; f.w_scale = lw.qkv8_scale; f.bias = lw.qkv_bias; f.c = d->b.qkv;
    f.m = t; f.n = qkv_n; f.k = H; f.lda8 = H; f.ldw8 = H; f.ldc = qkv_n; f.epilogue = CHATTS_EPI_NONE;
    if ((rc = chatts_linear_fp8(&f, stream)) != 0) return rc;
    ChattsKvCache kc = layer_cache(d, layer, d->cur_seq);
    if ((rc = chatts_rope_kv_write(d->b.qkv, t, c.n_q, c.n_kv, lw.q_norm, lw.k_norm, c.rms_eps, d->w.cos_tab, d->w.sin_tab, pos0, pos0_dev,
                                   &kc, stream)) != 0) return rc;
    if ((rc = attention_impl(d->b.qkv, t, c.n_q, c.n_kv, pos0, pos0_dev, &kc, d->b.attn, nullptr, nullptr, 1, d->b.workspace,
                             d->b.workspace_bytes, stream)) != 0) return rc;
    if ((rc = chatts_quantize_rows_fp8(d->b.attn, t, na, na, nullptr, 0.f, b8, na, sb, stream)) != 0) return rc;
    f = ChattsLinearFp8Args{};
    f.a8 = b8; f.a_scale = sb; f.w8 = lw.o8; f.w_scale = lw.o8_scale; f.m = t; f.n = H; f.k = na; f.lda8 = na; f.ldw8 = na; f.ldc = H;
    if (tp) { f.c = d->b.delta; f.epilogue = CHATTS_EPI_NONE; }
    else { f.c = d->b.x; f.resid = d->b.x; f.epilogue = CHATTS_EPI_RESID; }
    return chatts_linear_fp8(&f, stream);
  }
  if ((rc = chatts_quantize_rows_fp8(d->b.x, t, H, H, lw.post_norm, c.rms_eps, a8, H, sa, stream)) != 0) return rc;
  f.a8 = a8; f.a_scale = sa; f.w8 = lw.gate_up8; f.w_scale = lw.gate_up8_scale; f.c = d->b.act;
  f.m = t; f.n = 2 * c.inter; f.k = H; f.lda8 = H; f.ldw8 = H; f.ldc = c.inter; f.epilogue = CHATTS_EPI_SWIGLU;
  if ((rc = chatts_linear_fp8(&f, stream)) != 0) return rc;
  uint8_t* c8 = reinterpret_cast<uint8_t*>(d->b.planes2_hi);
  if ((rc = chatts_quantize_rows_fp8(d->b.act, t, c.inter, c.inter, nullptr, 0.f, c8, c.inter, sb, stream)) != 0) return rc;
  f = ChattsLinearFp8Args{};
  f.a8 = c8; f.a_scale = sb; f.w8 = lw.down8; f.w_scale = lw.down8_scale; f.m = t; f.n = H; f.k = c.inter; f.lda8 = c.inter; f.ldw8 = c.inter;
  f.ldc = H;
  if (tp) { f.c = d->b.delta; f.epilogue = CHATTS_EPI_NONE; }
  else { f.c = d->b.x; f.resid = d->b.x; f.epilogue = CHATTS_EPI_RESID; }
  return chatts_linear_fp8(&f, stream);
}

extern "C" int chatts_decoder_layer_part(ChattsDecoder* d, int layer, int part, int t, int pos0,
                                         const int32_t* pos0_dev, int n_splits, chatts_stream_t stream) {
  CHATTS_REQUIRE(d && layer >= 0 && layer < d->cfg.n_layers && (part == 0 || part == 1), CHATTS_E_BADARG,
                 "decoder_layer_part: bad arguments");
  CHATTS_REQUIRE(t >= 1 && t <= d->b.t_max, CHATTS_E_SHAPE, "decoder_layer_part: t=%d exceeds buffers (%d)", t, d->b.t_max);
  if (d->prefill_fp8 && t >= 16) return layer_part_fp8(d, layer, part, t, pos0, pos0_dev, stream);      // speed mode, prefill chunks only
  if (d->prefill_f16q && t >= kF16qMinRows) return layer_part_f16q(d, layer, part, t, pos0, pos0_dev, stream);
  d->normed_q = false;        // (a following f16q half recomputes its norm: planes 0 are about to hold bf16 data)
  StageRange stage(part == 0 ? "chatts.layer.attn" : "chatts.layer.mlp");
  const ChattsDecoderConfig& c = d->cfg;
  const ChattsLayerWeights& lw = d->layers[layer];
  const bool tp = c.tp_world > 1;
  const int H = c.hidden;
  int rc;
  ChattsLinearArgs la;
  if (part == 0) {
    const int qkv_n = (c.n_q + 2 * c.n_kv) * kHeadDim;
    // RMSNorm -> QKV (+bias).  Decode: norm fused in the GEMV prologue; prefill: separate kernel.
    la = ChattsLinearArgs{};
    la.w = lw.qkv; la.w_tiled = lw.qkv_t; la.bias = lw.qkv_bias; la.c = d->b.qkv; la.m = t; la.n = qkv_n; la.k = H;
    la.lda = H; la.ldw = H; la.ldc = qkv_n; la.epilogue = CHATTS_EPI_NONE;
    la.workspace = d->b.workspace; la.workspace_bytes = d->b.workspace_bytes;
    if (t == 1) {
      la.a = d->b.x; la.norm_w = lw.input_norm; la.norm_eps = c.rms_eps;
      la.w8 = lw.qkv8; la.w8_scale = lw.qkv8_scale; la.ldw8 = H; la.w8_format = d->cfg.w8_format;
      la.w4 = lw.qkv4; la.w4_sz = lw.qkv4_sz; la.ldw4 = H / 2; la.w4_group = lw.w4_group;
    } else {
      if ((rc = norm_into(d, lw.input_norm, &la, stream)) != 0) return rc;
    }
    ChattsKvCache kc = layer_cache(d, layer, d->cur_seq);
    bool rope_done = false;
    const bool rope_fuse = opt_get(OPT_ROPE_FUSE, 1) != 0;
    if (t > 1 && rope_fuse) {   // prefill: the projection's split-K epilogue (when it has one) also rotates q / k and fills the cache
      RopeFuse rf;
      if ((rc = rope_fuse_prepare(t, c.n_q, c.n_kv, lw.q_norm, lw.k_norm, c.rms_eps, d->w.cos_tab, d->w.sin_tab, pos0, pos0_dev, &kc,
                                  &rf)) != 0) return rc;
      if ((rc = linear_impl(&la, stream, &rf, &rope_done)) != 0) return rc;
    } else if ((rc = chatts_linear(&la, stream)) != 0) {
      return rc;
    }
    bool attn_out_planes = false;
    if (t == 1) {   // decode: RoPE + cache write fused into the attention kernel
      if ((rc = chatts_attention_decode_fused(d->b.qkv, c.n_q, c.n_kv, lw.q_norm, lw.k_norm, c.rms_eps, d->w.cos_tab,
                                                     d->w.sin_tab, pos0, pos0_dev, &kc, d->b.attn, n_splits, d->b.workspace,
                                                     d->b.workspace_bytes, stream)) != 0) {
        return rc;
      }
    } else {
      if (!rope_done && (rc = chatts_rope_kv_write(d->b.qkv, t, c.n_q, c.n_kv, lw.q_norm, lw.k_norm, c.rms_eps, d->w.cos_tab,
                                                   d->w.sin_tab, pos0, pos0_dev, &kc, stream)) != 0) return rc;
      // long chunks: split the keys over 2 workgroups per (query tile, head) - one sequence has too few waves otherwise
      const int env_ks = opt_get(OPT_ATTN_KSPLIT, 0);
      // (the bf16x3 kernel is fast enough per tile that the second key split + its combine launch no longer pay: 73.3 us
      // against 74.0 + 12 at T = 798; the float32-MFMA kernel, CHATTS_ATTN_BF16X3=0, still wants the split)
      const bool f32_attn = opt_get(OPT_ATTN_BF16X3, 1) == 0;
      int ks = env_ks >= 1 && env_ks <= 4 ? env_ks : (t >= 256 && f32_attn ? 2 : 1);
      if (chatts_attn_workspace(t, c.n_q, ks) > d->b.workspace_bytes) ks = 1;
      // ... and, on the plane path, let the attention kernel (or its combine) write o_proj's operand format itself
      const bool out_planes = t >= 16 && planes_path(d, t, c.n_q * kHeadDim) && opt_get(OPT_ATTN_ROWS, 0) == 0;
      if ((rc = attention_impl(d->b.qkv, t, c.n_q, c.n_kv, pos0, pos0_dev, &kc, d->b.attn, out_planes ? d->b.planes_hi : nullptr,
                               out_planes ? d->b.planes_lo : nullptr, ks, d->b.workspace, d->b.workspace_bytes, stream)) != 0)
        return rc;
      attn_out_planes = out_planes;
    }
    // o_proj (+ residual, or partial sum for the TP all-reduce)
    la = ChattsLinearArgs{};
    la.a = d->b.attn; la.w = lw.o; la.w_tiled = lw.o_t; la.m = t; la.n = H; la.k = c.n_q * kHeadDim;
    la.lda = la.k; la.ldw = la.k; la.ldc = H;
    la.workspace = d->b.workspace; la.workspace_bytes = d->b.workspace_bytes;
    if (t == 1) {
      la.w8 = lw.o8; la.w8_scale = lw.o8_scale; la.ldw8 = la.k; la.w8_format = d->cfg.w8_format;
      la.w4 = lw.o4; la.w4_sz = lw.o4_sz; la.ldw4 = la.k / 2; la.w4_group = lw.w4_group;
    } else if (attn_out_planes) {              // written by the attention kernel
      la.a = nullptr; la.a_hi = d->b.planes_hi; la.a_lo = d->b.planes_lo; la.ld_planes = la.k;
    } else if (planes_path(d, t, la.k)) {      // the VALU attention kernel writes float32: split it
      if ((rc = chatts_split_bf16x2(d->b.attn, t, la.k, la.k, d->b.planes_hi, d->b.planes_lo, la.k, stream)) != 0) return rc;
      la.a_hi = d->b.planes_hi; la.a_lo = d->b.planes_lo; la.ld_planes = la.k;
    }
    d->tp_fused = false;
    if (tp && t == 1 && d->fuse_tp && d->tp && !la.w4) {      // x += sum over the ranks, inside the GEMV launch (ChattsLinearArgs.tp_reduce)
      la.c = d->b.x; la.resid = d->b.x; la.epilogue = CHATTS_EPI_RESID; la.tp_reduce = d->tp;
      d->tp_fused = true;
    } else if (tp) { la.c = d->b.delta; la.epilogue = CHATTS_EPI_NONE; }
    else { la.c = d->b.x; la.resid = d->b.x; la.epilogue = CHATTS_EPI_RESID; }
    request_post_norm(d, &la, lw.post_norm, false);
    return chatts_linear(&la, stream);
  }
  // part 1: RMSNorm -> gate_up + SwiGLU -> down (+ residual / partial)
  la = ChattsLinearArgs{};
  la.w = lw.gate_up; la.w_tiled = lw.gate_up_t; la.c = d->b.act; la.m = t; la.n = 2 * c.inter; la.k = H;
  la.lda = H; la.ldw = H; la.ldc = c.inter; la.epilogue = CHATTS_EPI_SWIGLU;
  la.workspace = d->b.workspace; la.workspace_bytes = d->b.workspace_bytes;
  if (t == 1) {
    la.a = d->b.x; la.norm_w = lw.post_norm; la.norm_eps = c.rms_eps;
    la.w8 = lw.gate_up8; la.w8_scale = lw.gate_up8_scale; la.ldw8 = H; la.w8_format = d->cfg.w8_format;
    la.w4 = lw.gate_up4; la.w4_sz = lw.gate_up4_sz; la.ldw4 = H / 2; la.w4_group = lw.w4_group;
  } else {
    if ((rc = norm_into(d, lw.post_norm, &la, stream)) != 0) return rc;
  }
  const bool act_planes = t > 1 && planes_path(d, t, c.inter);   // SwiGLU output goes straight into down_proj's operand format
  if (act_planes) { la.c = nullptr; la.c_hi = d->b.planes2_hi; la.c_lo = d->b.planes2_lo; la.ld_cplanes = c.inter; }
  if ((rc = chatts_linear(&la, stream)) != 0) return rc;
  la = ChattsLinearArgs{};
  la.a = d->b.act; la.w = lw.down; la.w_tiled = lw.down_t; la.m = t; la.n = H; la.k = c.inter;
  la.lda = c.inter; la.ldw = c.inter; la.ldc = H;
  la.workspace = d->b.workspace; la.workspace_bytes = d->b.workspace_bytes;
  if (t == 1) {
    la.w8 = lw.down8; la.w8_scale = lw.down8_scale; la.ldw8 = c.inter; la.w8_format = d->cfg.w8_format;
    la.w4 = lw.down4; la.w4_sz = lw.down4_sz; la.ldw4 = c.inter / 2; la.w4_group = lw.w4_group;
  }
  if (act_planes) { la.a = nullptr; la.a_hi = d->b.planes2_hi; la.a_lo = d->b.planes2_lo; la.ld_planes = c.inter; }
  d->tp_fused = false;
  if (tp && t == 1 && d->fuse_tp && d->tp && !la.w4) {
    la.c = d->b.x; la.resid = d->b.x; la.epilogue = CHATTS_EPI_RESID; la.tp_reduce = d->tp;
    d->tp_fused = true;
  } else if (tp) { la.c = d->b.delta; la.epilogue = CHATTS_EPI_NONE; }
  else { la.c = d->b.x; la.resid = d->b.x; la.epilogue = CHATTS_EPI_RESID; }
  request_post_norm(d, &la, layer + 1 < c.n_layers ? d->layers[layer + 1].input_norm : nullptr, false);
  return chatts_linear(&la, stream);
}

extern "C" int chatts_decoder_layer_part_add(ChattsDecoder* d, int add_delta, int layer, int part, int t, int pos0,
                                             const int32_t* pos0_dev, int n_splits, chatts_stream_t stream) {
  CHATTS_REQUIRE(d && t >= 1 && t <= d->b.t_max, CHATTS_E_BADARG, "decoder_layer_part_add: bad arguments");
  if (add_delta) {
    CHATTS_REQUIRE(d->b.delta, CHATTS_E_BADARG, "decoder_layer_part_add: no delta buffer");
    const int rc = chatts_residual_add(d->b.x, d->b.delta, (int64_t)t * d->cfg.hidden, stream);
    if (rc) return rc;
  }
  return chatts_decoder_layer_part(d, layer, part, t, pos0, pos0_dev, n_splits, stream);
}

// Batched decode: `batch` sequences advance one token each.  Row b of x / qkv / attn /
// act belongs to sequence b (cache slot b, position pos_dev[b]).  The projections run as M = batch GEMMs (bf16x2
// MFMA, weights streamed once for the whole batch); attention runs per sequence on its own cache.
extern "C" int chatts_decoder_layer_part_batched(ChattsDecoder* d, int layer, int part, int batch,
                                                 const int32_t* pos_dev, int n_splits, chatts_stream_t stream) {
  CHATTS_REQUIRE(d && layer >= 0 && layer < d->cfg.n_layers && (part == 0 || part == 1) && pos_dev, CHATTS_E_BADARG,
                 "decoder_layer_part_batched: bad arguments");
  const int maxb = d->b.max_batch > 0 ? d->b.max_batch : 1;
  CHATTS_REQUIRE(batch >= 1 && batch <= maxb && batch <= d->b.t_max, CHATTS_E_SHAPE,
                 "decoder_layer_part_batched: batch %d exceeds max_batch %d", batch, maxb);
  StageRange stage(part == 0 ? "chatts.layer.attn" : "chatts.layer.mlp");
  const ChattsDecoderConfig& c = d->cfg;
  const ChattsLayerWeights& lw = d->layers[layer];
  const bool tp = c.tp_world > 1;
  const int H = c.hidden;
  int rc;
  ChattsLinearArgs la;
  if (part == 0) {
    const int qkv_n = (c.n_q + 2 * c.n_kv) * kHeadDim;
    la = ChattsLinearArgs{};
    la.w = lw.qkv; la.w_tiled = lw.qkv_t; la.bias = lw.qkv_bias; la.c = d->b.qkv; la.m = batch; la.n = qkv_n; la.k = H;
    la.lda = H; la.ldw = H; la.ldc = qkv_n; la.epilogue = CHATTS_EPI_NONE;
    la.w8 = lw.qkv8; la.w8_scale = lw.qkv8_scale; la.ldw8 = H; la.w8_format = d->cfg.w8_format;
    la.workspace = d->b.workspace; la.workspace_bytes = d->b.workspace_bytes;
    if ((rc = norm_into(d, lw.input_norm, &la, stream)) != 0) return rc;
    if ((rc = chatts_linear(&la, stream)) != 0) return rc;
    ChattsKvCache kc = layer_cache(d, layer, 0);
    const bool attn_planes = planes_path(d, batch, c.n_q * kHeadDim, lw.o8 != nullptr);   // the combine writes o_proj's operand format
    if ((rc = attention_decode_batched_impl(d->b.qkv, batch, c.n_q, c.n_kv, lw.q_norm, lw.k_norm, c.rms_eps, d->w.cos_tab, d->w.sin_tab, 0,
                                            pos_dev, &kc, seq_stride(d), d->b.attn, attn_planes ? d->b.planes_hi : nullptr,
                                            attn_planes ? d->b.planes_lo : nullptr, n_splits, d->b.workspace, d->b.workspace_bytes,
                                            stream)) != 0) return rc;
    la = ChattsLinearArgs{};
    la.a = d->b.attn; la.w = lw.o; la.w_tiled = lw.o_t; la.m = batch; la.n = H; la.k = c.n_q * kHeadDim;
    la.lda = la.k; la.ldw = la.k; la.ldc = H;
    la.w8 = lw.o8; la.w8_scale = lw.o8_scale; la.ldw8 = la.k; la.w8_format = d->cfg.w8_format;
    la.workspace = d->b.workspace; la.workspace_bytes = d->b.workspace_bytes;
    if (attn_planes) { la.a = nullptr; la.a_hi = d->b.planes_hi; la.a_lo = d->b.planes_lo; la.ld_planes = la.k; }
    d->tp_fused = false;
    if (tp) { la.c = d->b.delta; la.epilogue = CHATTS_EPI_NONE; }
    else { la.c = d->b.x; la.resid = d->b.x; la.epilogue = CHATTS_EPI_RESID; }
    // The residual stream ping-pongs x -> xn (here) -> x (down_proj) inside a chained step, so that the post-norm epilogue may use
    // several workgroups per row (they read resid columns other workgroups update: c must not alias resid).
    d->x_in_xn = false;
    if (!tp && d->chain && d->b.xn && batch <= 16 && planes_path(d, batch, H, lw.gate_up8 != nullptr)) {
      la.c = d->b.xn;
      d->x_in_xn = true;
    }
    request_post_norm(d, &la, lw.post_norm, lw.gate_up8 != nullptr);
    if (d->x_in_xn && !la.post_norm_w) { la.c = d->b.x; d->x_in_xn = false; }      // not fused after all: the norm launch reads x
    return chatts_linear(&la, stream);
  }
  la = ChattsLinearArgs{};
  la.w = lw.gate_up; la.w_tiled = lw.gate_up_t; la.c = d->b.act; la.m = batch; la.n = 2 * c.inter; la.k = H;
  la.lda = H; la.ldw = H; la.ldc = c.inter; la.epilogue = CHATTS_EPI_SWIGLU;
  la.w8 = lw.gate_up8; la.w8_scale = lw.gate_up8_scale; la.ldw8 = H; la.w8_format = d->cfg.w8_format;
  la.workspace = d->b.workspace; la.workspace_bytes = d->b.workspace_bytes;
  if ((rc = norm_into(d, lw.post_norm, &la, stream)) != 0) return rc;
  const bool act_planes = planes_path(d, batch, c.inter, lw.down8 != nullptr) && planes_path(d, batch, H, lw.gate_up8 != nullptr);
  if (act_planes) { la.c = nullptr; la.c_hi = d->b.planes2_hi; la.c_lo = d->b.planes2_lo; la.ld_cplanes = c.inter; }
  if ((rc = chatts_linear(&la, stream)) != 0) return rc;
  la = ChattsLinearArgs{};
  la.a = d->b.act; la.w = lw.down; la.w_tiled = lw.down_t; la.m = batch; la.n = H; la.k = c.inter;
  la.lda = c.inter; la.ldw = c.inter; la.ldc = H;
  la.w8 = lw.down8; la.w8_scale = lw.down8_scale; la.ldw8 = c.inter; la.w8_format = d->cfg.w8_format;
  la.workspace = d->b.workspace; la.workspace_bytes = d->b.workspace_bytes;
  if (act_planes) { la.a = nullptr; la.a_hi = d->b.planes2_hi; la.a_lo = d->b.planes2_lo; la.ld_planes = c.inter; }
  d->tp_fused = false;
  if (tp) { la.c = d->b.delta; la.epilogue = CHATTS_EPI_NONE; }
  else { la.c = d->b.x; la.resid = d->x_in_xn ? d->b.xn : d->b.x; la.epilogue = CHATTS_EPI_RESID; }
  d->x_in_xn = false;
  request_post_norm(d, &la, layer + 1 < c.n_layers ? d->layers[layer + 1].input_norm : d->w.final_norm,
                    (layer + 1 < c.n_layers ? d->layers[layer + 1].qkv8 : d->w.lm_head8) != nullptr);
  return chatts_linear(&la, stream);
}

// final norm + lm_head for rows 0 .. batch-1 of x (cache slots 0 .. batch-1) -> logits_all [batch, vocab_local]: the tail of a
// batched step (compute_logits for a batch, chatts_vllm.py:603-610).  M = batch weight-streaming GEMM on this rank's vocabulary slice.
extern "C" int chatts_decoder_logits_batched(ChattsDecoder* d, int batch, float* logits_all, chatts_stream_t stream) {
  CHATTS_REQUIRE(d && logits_all, CHATTS_E_BADARG, "decoder_logits_batched: null argument");
  const int maxb = d->b.max_batch > 0 ? d->b.max_batch : 1;
  CHATTS_REQUIRE(batch >= 1 && batch <= maxb && batch <= d->b.t_max, CHATTS_E_SHAPE, "decoder_logits_batched: batch %d exceeds max_batch %d",
                 batch, maxb);
  const ChattsDecoderConfig& c = d->cfg;
  ChattsLinearArgs la{};
  la.w = d->w.lm_head; la.c = logits_all; la.m = batch; la.n = (int)c.vocab_local; la.k = c.hidden;
  la.lda = c.hidden; la.ldw = c.hidden; la.ldc = (int)c.vocab_local; la.epilogue = CHATTS_EPI_NONE;
  la.workspace = d->b.workspace; la.workspace_bytes = d->b.workspace_bytes;
  la.w8 = d->w.lm_head8; la.w8_scale = d->w.lm_head8_scale; la.ldw8 = c.hidden; la.w8_format = d->cfg.w8_format;
  int rc = norm_into(d, d->w.final_norm, &la, stream);      // binds planes a fused epilogue already wrote (d->normed), or launches the norm
  d->normed = false; d->normed_q = false;
  if (rc) return rc;
  return chatts_linear(&la, stream);
}

// One greedy step for `batch` sequences (TP = 1): all layers, final norm + lm_head for every row, per-sequence
// argmax (token / step / pos / out_tokens rows advance independently; pos saturates at max_ctx - 1).
// logits_all: [batch, vocab_local] float32 scratch owned by the caller.
extern "C" int chatts_decoder_decode_step_batched(ChattsDecoder* d, int batch, int32_t* pos_dev, int32_t* step_dev,
                                                  int64_t* token_dev, float* token_logit_dev, int64_t* out_tokens,
                                                  int64_t out_stride, float* logits_all, int n_splits,
                                                  chatts_stream_t stream) {
  CHATTS_REQUIRE(d && pos_dev && step_dev && token_dev && logits_all, CHATTS_E_BADARG, "decode_step_batched: null argument");
  StageRange stage("chatts.decode_step_batched");
  const bool tp = d->cfg.tp_world > 1;
  CHATTS_REQUIRE(!tp || d->tp, CHATTS_E_BADARG, "decode_step_batched: tp_world = %d but no exchange attached (chatts_decoder_set_tp)",
                 d->cfg.tp_world);
  CHATTS_REQUIRE(!tp || chatts_tp_pending(d->tp) == 0, CHATTS_E_BADARG, "decode_step_batched: the exchange has %d collectives whose epochs were "
                 "never settled (chatts_tp_flush_epochs)", tp ? chatts_tp_pending(d->tp) : 0);
  int rc;
  const ChattsDecoderConfig& c = d->cfg;
  // the step STARTS by loading the input embeddings from the current tokens (so a prefill of another request may
  // use x between two steps) and ENDS with the per-sequence argmax
  if ((rc = chatts_embed_token_batched(token_dev, batch, d->w.embed, embed_offset(d), embed_rows(d), c.hidden, d->b.x,
                                       stream)) != 0) return rc;
  d->chain = true;               // layers run back to back: a projection may write the next one's normed operand
  d->normed = false; d->normed_q = false;
  const int64_t nx = (int64_t)batch * c.hidden;
  for (int l = 0; l < c.n_layers && rc == CHATTS_OK; ++l) {
    rc = chatts_decoder_layer_part_batched(d, l, 0, batch, pos_dev, n_splits, stream);
    if (tp && rc == CHATTS_OK && !d->tp_fused) rc = chatts_allreduce(d->tp, d->b.delta, d->b.x, d->b.x, nx, stream);     // x += sum of the partial o_proj
    if (rc == CHATTS_OK) rc = chatts_decoder_layer_part_batched(d, l, 1, batch, pos_dev, n_splits, stream);
    if (tp && rc == CHATTS_OK && !d->tp_fused) rc = chatts_allreduce(d->tp, d->b.delta, d->b.x, d->b.x, nx, stream);     // ... and down_proj
  }
  d->chain = false;
  if (rc) { d->normed = false; d->normed_q = false; return rc; }
  if ((rc = chatts_decoder_logits_batched(d, batch, logits_all, stream)) != 0) return rc;
  return chatts_decoder_select_tokens(d, logits_all, batch, c.vocab_local, token_dev, token_logit_dev, out_tokens, out_stride, step_dev,
                                      pos_dev, c.max_ctx - 1, nullptr, stream);
}

// tensor parallel: can the [t, H] sums between the layer halves run as chatts_allreduce_bulk launches of the same call?
static bool tp_bulk_ready(const ChattsDecoder* d, int t) {
  return d->tp && chatts_tp_bulk_elems(d->tp) >= (int64_t)t * d->cfg.hidden;
}
#define CHATTS_REQUIRE_TP_BULK(d, t, name)                                                                                         \
  CHATTS_REQUIRE((d)->cfg.tp_world == 1 || tp_bulk_ready((d), (t)), CHATTS_E_BADARG,                                                \
                 name ": tp_world = %d needs an attached exchange whose bulk region holds t * hidden = %lld float32 (chatts_tp_buffer_bytes_bulk); " \
                 "otherwise drive chatts_decoder_layer_part_add and all-reduce buffers.delta yourself", (d)->cfg.tp_world,          \
                 (long long)(t) * (d)->cfg.hidden)
// x[0:t] += sum over the ranks of delta[0:t]
static int tp_sum_into_x(ChattsDecoder* d, int t, chatts_stream_t stream) {
  return chatts_allreduce_bulk(d->tp, d->b.delta, d->b.x, (int64_t)t * d->cfg.hidden, stream);
}

extern "C" int chatts_decoder_prefill(ChattsDecoder* d, int t, int pos0, chatts_stream_t stream) {
  CHATTS_REQUIRE(d, CHATTS_E_BADARG, "decoder_prefill: null decoder");
  CHATTS_REQUIRE(t >= 1 && t <= d->b.t_max, CHATTS_E_SHAPE, "decoder_prefill: t=%d exceeds buffers (%d)", t, d->b.t_max);
  CHATTS_REQUIRE_TP_BULK(d, t, "decoder_prefill");
  StageRange stage("chatts.prefill");
  const bool tp = d->cfg.tp_world > 1;
  d->chain = true;
  d->normed = false; d->normed_q = false;
  int rc = CHATTS_OK;
  for (int l = 0; l < d->cfg.n_layers && rc == CHATTS_OK; ++l) {
    rc = chatts_decoder_layer_part(d, l, 0, t, pos0, nullptr, 1, stream);
    if (tp && rc == CHATTS_OK) rc = tp_sum_into_x(d, t, stream);
    if (rc == CHATTS_OK) rc = chatts_decoder_layer_part(d, l, 1, t, pos0, nullptr, 1, stream);
    if (tp && rc == CHATTS_OK) rc = tp_sum_into_x(d, t, stream);
  }
  d->chain = false;
  d->normed = false; d->normed_q = false;
  return rc;
}

// The LAST layer of a prefill whose only consumers are the next token and the KV cache: K / V of every row are needed
// (cache), everything else only for the last row.  So: RMSNorm + qkv GEMM + RoPE / cache write for all t rows as usual, then
// the last row alone is attended (t = 1 against the cache), moved to row 0 of the residual stream and taken through o_proj
// and the MLP as weight-streaming GEMVs.  Saves one layer's attention + o / gate_up / down GEMMs (~0.9 of 44 ms at the
// benchmark prompt).  The result lives in row 0 of x (rows 1.. are stale).
static int layer_last_row(ChattsDecoder* d, int layer, int t, int pos0, chatts_stream_t stream) {
  d->normed_q = false;      // (f16q mode: the last layer runs the default kernels)
  const ChattsDecoderConfig& c = d->cfg;
  const ChattsLayerWeights& lw = d->layers[layer];
  const int H = c.hidden, qkv_n = (c.n_q + 2 * c.n_kv) * kHeadDim;
  int rc;
  ChattsLinearArgs la{};
  la.w = lw.qkv; la.w_tiled = lw.qkv_t; la.bias = lw.qkv_bias; la.c = d->b.qkv; la.m = t; la.n = qkv_n; la.k = H;
  la.lda = H; la.ldw = H; la.ldc = qkv_n; la.epilogue = CHATTS_EPI_NONE;
  la.workspace = d->b.workspace; la.workspace_bytes = d->b.workspace_bytes;
  if ((rc = norm_into(d, lw.input_norm, &la, stream)) != 0) return rc;
  if ((rc = chatts_linear(&la, stream)) != 0) return rc;
  ChattsKvCache kc = layer_cache(d, layer, d->cur_seq);
  if ((rc = chatts_rope_kv_write(d->b.qkv, t, c.n_q, c.n_kv, lw.q_norm, lw.k_norm, c.rms_eps, d->w.cos_tab, d->w.sin_tab, pos0,
                                 nullptr, &kc, stream)) != 0) return rc;
  // one query row against pos0 + t keys: spread the 64-key tiles over up to 16 workgroups per kv head (one workgroup walking 800
  // keys alone took 137 us in the round-2 trace)
  int ks = (pos0 + t + 63) / 64;
  if (ks > 16) ks = 16;
  if (ks < 1 || chatts_attn_workspace(1, c.n_q, ks) > d->b.workspace_bytes) ks = 1;
  if ((rc = chatts_attention(d->b.qkv + (size_t)(t - 1) * qkv_n, 1, c.n_q, c.n_kv, pos0 + t - 1, nullptr, &kc, d->b.attn, ks,
                             d->b.workspace, d->b.workspace_bytes, stream)) != 0) return rc;
  if (t > 1) {
    const hipError_t e = hipMemcpyAsync(d->b.x, d->b.x + (size_t)(t - 1) * H, (size_t)H * sizeof(float), hipMemcpyDeviceToDevice,
                                        as_stream(stream));
    CHATTS_REQUIRE(e == hipSuccess, CHATTS_E_LAUNCH, "prefill_last: row copy: %s", hipGetErrorString(e));
  }
  la = ChattsLinearArgs{};
  la.a = d->b.attn; la.w = lw.o; la.w_tiled = lw.o_t; la.m = 1; la.n = H; la.k = c.n_q * kHeadDim; la.lda = la.k; la.ldw = la.k; la.ldc = H;
  la.w8 = lw.o8; la.w8_scale = lw.o8_scale; la.ldw8 = la.k; la.w8_format = d->cfg.w8_format;
  la.w4 = lw.o4; la.w4_sz = lw.o4_sz; la.ldw4 = la.k / 2; la.w4_group = lw.w4_group;
  la.c = d->b.x; la.resid = d->b.x; la.epilogue = CHATTS_EPI_RESID;
  const bool tp = c.tp_world > 1;
  if (tp && !la.w4) la.tp_reduce = d->tp;                  // row 0 of x += the sum over the ranks, inside the GEMV launch
  else if (tp) { la.c = d->b.delta; la.resid = nullptr; la.epilogue = CHATTS_EPI_NONE; }
  if ((rc = chatts_linear(&la, stream)) != 0) return rc;
  if (tp && !la.tp_reduce && (rc = chatts_allreduce(d->tp, d->b.delta, d->b.x, d->b.x, H, stream)) != 0) return rc;
  d->fuse_tp = tp;
  rc = chatts_decoder_layer_part(d, layer, 1, 1, 0, nullptr, 1, stream);      // MLP on row 0: the decode path's GEMVs
  d->fuse_tp = false;
  if (rc == CHATTS_OK && tp && !d->tp_fused) rc = chatts_allreduce(d->tp, d->b.delta, d->b.x, d->b.x, H, stream);
  return rc;
}

extern "C" int chatts_decoder_prefill_last(ChattsDecoder* d, int t, int pos0, chatts_stream_t stream) {
  CHATTS_REQUIRE(d, CHATTS_E_BADARG, "decoder_prefill_last: null decoder");
  CHATTS_REQUIRE(t >= 1 && t <= d->b.t_max, CHATTS_E_SHAPE, "decoder_prefill_last: t=%d exceeds buffers (%d)", t, d->b.t_max);
  CHATTS_REQUIRE_TP_BULK(d, t, "decoder_prefill_last");
  StageRange stage("chatts.prefill_last");
  const bool tp = d->cfg.tp_world > 1;
  d->chain = true;
  d->normed = false; d->normed_q = false;
  int rc = CHATTS_OK;
  const int L = d->cfg.n_layers;
  for (int l = 0; l + 1 < L && rc == CHATTS_OK; ++l) {
    rc = chatts_decoder_layer_part(d, l, 0, t, pos0, nullptr, 1, stream);
    if (tp && rc == CHATTS_OK) rc = tp_sum_into_x(d, t, stream);
    if (rc == CHATTS_OK) rc = chatts_decoder_layer_part(d, l, 1, t, pos0, nullptr, 1, stream);
    if (tp && rc == CHATTS_OK) rc = tp_sum_into_x(d, t, stream);
  }
  if (rc == CHATTS_OK && d->prefill_fp8 && t >= 16) {      // speed mode: the final layer at full width too, then the last row moves to row 0
    rc = chatts_decoder_layer_part(d, L - 1, 0, t, pos0, nullptr, 1, stream);
    if (tp && rc == CHATTS_OK) rc = tp_sum_into_x(d, t, stream);
    if (rc == CHATTS_OK) rc = chatts_decoder_layer_part(d, L - 1, 1, t, pos0, nullptr, 1, stream);
    if (tp && rc == CHATTS_OK) rc = tp_sum_into_x(d, t, stream);
    if (rc == CHATTS_OK && t > 1) {
      const hipError_t e = hipMemcpyAsync(d->b.x, d->b.x + (size_t)(t - 1) * d->cfg.hidden, (size_t)d->cfg.hidden * sizeof(float),
                                          hipMemcpyDeviceToDevice, as_stream(stream));
      if (e != hipSuccess) { set_error("prefill_last: row copy: %s", hipGetErrorString(e)); rc = CHATTS_E_LAUNCH; }
    }
  } else if (rc == CHATTS_OK) rc = layer_last_row(d, L - 1, t, pos0, stream);
  d->chain = false;
  d->normed = false; d->normed_q = false;
  // the last row's o_proj / down_proj carried their exchange without advancing the device-side call counter (host-counted epochs):
  // settle it here, so that whatever runs next on this communicator - a replayed decode graph above all - starts from a flushed count
  if (tp && d->tp) { const int frc = chatts_tp_flush_epochs(d->tp, stream); if (rc == CHATTS_OK) rc = frc; }
  return rc;
}

// Packed multi-prompt prefill: the rows of several prompts (or of their not-yet-cached tails) sit
// back to back in x; everything row-wise - norms, the four projections, SwiGLU, residuals - runs ONCE over all rows (one big-M
// GEMM instead of one ragged small-M GEMM per prompt), only RoPE / cache write / attention run per segment, each against
// its own cache slot and positions.
static int layer_part0_packed(ChattsDecoder* d, int layer, int t, const ChattsPrefillSegment* segs, int n_segs,
                              chatts_stream_t stream) {
  const ChattsDecoderConfig& c = d->cfg;
  const ChattsLayerWeights& lw = d->layers[layer];
  const int H = c.hidden, qkv_n = (c.n_q + 2 * c.n_kv) * kHeadDim, na = c.n_q * kHeadDim;
  int rc;
  d->normed_q = false;      // (f16q mode: this half runs the default kernels and is about to overwrite planes 0 - the next MLP half recomputes its norm)
  const bool f8 = d->prefill_fp8 && t >= 16;          // speed mode: qkv and o_proj as fp8 x fp8 GEMMs (layer_part_fp8's arithmetic)
  ChattsLinearArgs la{};
  if (f8) {
    d->normed = false; d->normed_q = false;
    uint8_t* a8 = reinterpret_cast<uint8_t*>(d->b.planes_hi);
    if ((rc = chatts_quantize_rows_fp8(d->b.x, t, H, H, lw.input_norm, c.rms_eps, a8, H, d->b.xn, stream)) != 0) return rc;
    ChattsLinearFp8Args f{};
    f.a8 = a8; f.a_scale = d->b.xn; f.w8 = lw.qkv8; f.w_scale = lw.qkv8_scale; f.bias = lw.qkv_bias; f.c = d->b.qkv;
    f.m = t; f.n = qkv_n; f.k = H; f.lda8 = H; f.ldw8 = H; f.ldc = qkv_n; f.epilogue = CHATTS_EPI_NONE;
    if ((rc = chatts_linear_fp8(&f, stream)) != 0) return rc;
  } else {
    la.w = lw.qkv; la.w_tiled = lw.qkv_t; la.bias = lw.qkv_bias; la.c = d->b.qkv; la.m = t; la.n = qkv_n; la.k = H;
    la.lda = H; la.ldw = H; la.ldc = qkv_n; la.epilogue = CHATTS_EPI_NONE;
    la.workspace = d->b.workspace; la.workspace_bytes = d->b.workspace_bytes;
    if ((rc = norm_into(d, lw.input_norm, &la, stream)) != 0) return rc;
    if ((rc = chatts_linear(&la, stream)) != 0) return rc;
  }
  for (int i = 0; i < n_segs; ++i) {
    const ChattsPrefillSegment& sg = segs[i];
    ChattsKvCache kc = layer_cache(d, layer, sg.slot);
    float* q = d->b.qkv + (size_t)sg.row0 * qkv_n;
    if ((rc = chatts_rope_kv_write(q, sg.t, c.n_q, c.n_kv, lw.q_norm, lw.k_norm, c.rms_eps, d->w.cos_tab, d->w.sin_tab, sg.pos0,
                                   nullptr, &kc, stream)) != 0) return rc;
    const int ks = 1;
    if ((rc = attention_impl(q, sg.t, c.n_q, c.n_kv, sg.pos0, nullptr, &kc, d->b.attn + (size_t)sg.row0 * na, nullptr, nullptr, ks,
                             d->b.workspace, d->b.workspace_bytes, stream)) != 0) return rc;
  }
  if (f8) {
    uint8_t* b8 = reinterpret_cast<uint8_t*>(d->b.planes_lo);
    float* sb = d->b.xn + d->b.t_max;
    if ((rc = chatts_quantize_rows_fp8(d->b.attn, t, na, na, nullptr, 0.f, b8, na, sb, stream)) != 0) return rc;
    ChattsLinearFp8Args f{};
    f.a8 = b8; f.a_scale = sb; f.w8 = lw.o8; f.w_scale = lw.o8_scale; f.m = t; f.n = H; f.k = na; f.lda8 = na; f.ldw8 = na; f.ldc = H;
    f.c = d->b.x; f.resid = d->b.x; f.epilogue = CHATTS_EPI_RESID;
    return chatts_linear_fp8(&f, stream);
  }
  la = ChattsLinearArgs{};
  la.a = d->b.attn; la.w = lw.o; la.w_tiled = lw.o_t; la.m = t; la.n = H; la.k = na; la.lda = na; la.ldw = na; la.ldc = H;
  la.workspace = d->b.workspace; la.workspace_bytes = d->b.workspace_bytes;
  if (t > 1 && planes_path(d, t, na)) {
    if ((rc = chatts_split_bf16x2(d->b.attn, t, na, na, d->b.planes_hi, d->b.planes_lo, na, stream)) != 0) return rc;
    la.a_hi = d->b.planes_hi; la.a_lo = d->b.planes_lo; la.ld_planes = na;
  }
  la.c = d->b.x; la.resid = d->b.x; la.epilogue = CHATTS_EPI_RESID;
  request_post_norm(d, &la, lw.post_norm, false);
  return chatts_linear(&la, stream);
}

extern "C" int chatts_decoder_prefill_packed(ChattsDecoder* d, const ChattsPrefillSegment* segs, int n_segs, chatts_stream_t stream) {
  CHATTS_REQUIRE(d && segs && n_segs >= 1, CHATTS_E_BADARG, "decoder_prefill_packed: bad arguments");
  CHATTS_REQUIRE(d->cfg.tp_world == 1, CHATTS_E_BADARG, "decoder_prefill_packed: tensor_parallel_size 1 only");
  const int maxb = d->b.max_batch > 0 ? d->b.max_batch : 1;
  int t = 0;
  for (int i = 0; i < n_segs; ++i) {
    const ChattsPrefillSegment& sg = segs[i];
    CHATTS_REQUIRE(sg.t >= 1 && sg.row0 == t && sg.pos0 >= 0 && sg.pos0 + sg.t <= d->cfg.max_ctx && sg.slot >= 0 && sg.slot < maxb,
                   CHATTS_E_SHAPE, "decoder_prefill_packed: segment %d (row0 %d, t %d, pos0 %d, slot %d) is not contiguous / out of range",
                   i, sg.row0, sg.t, sg.pos0, sg.slot);
    for (int j = 0; j < i; ++j)
      CHATTS_REQUIRE(segs[j].slot != sg.slot, CHATTS_E_BADARG, "decoder_prefill_packed: cache slot %d appears twice", sg.slot);
    t += sg.t;
  }
  CHATTS_REQUIRE(t <= d->b.t_max, CHATTS_E_SHAPE, "decoder_prefill_packed: %d rows exceed buffers (%d)", t, d->b.t_max);
  StageRange stage("chatts.prefill_packed");
  d->chain = true;
  d->normed = false; d->normed_q = false;
  int rc = CHATTS_OK;
  for (int l = 0; l < d->cfg.n_layers && rc == CHATTS_OK; ++l) {
    rc = layer_part0_packed(d, l, t, segs, n_segs, stream);
    if (rc == CHATTS_OK) rc = chatts_decoder_layer_part(d, l, 1, t, 0, nullptr, 1, stream);
  }
  d->chain = false;
  d->normed = false; d->normed_q = false;
  return rc;
}

extern "C" int chatts_decoder_logits(ChattsDecoder* d, int row, chatts_stream_t stream) {
  CHATTS_REQUIRE(d && row >= 0 && row < d->b.t_max, CHATTS_E_BADARG, "decoder_logits: bad row");
  StageRange stage("chatts.logits");
  const ChattsDecoderConfig& c = d->cfg;
  ChattsLinearArgs la{};
  la.a = d->b.x + (size_t)row * c.hidden; la.w = d->w.lm_head; la.c = d->b.logits;
  la.m = 1; la.n = (int)c.vocab_local; la.k = c.hidden; la.lda = c.hidden; la.ldw = c.hidden; la.ldc = (int)c.vocab_local;
  la.epilogue = CHATTS_EPI_NONE; la.norm_w = d->w.final_norm; la.norm_eps = c.rms_eps;
  la.w8 = d->w.lm_head8; la.w8_scale = d->w.lm_head8_scale; la.ldw8 = c.hidden; la.w8_format = d->cfg.w8_format;
  return chatts_linear(&la, stream);
}

extern "C" int chatts_decoder_decode_step(ChattsDecoder* d, int32_t* pos_dev, int32_t* step_dev, int64_t* token_dev,
                                          float* token_logit_dev, int64_t* out_tokens, int n_splits,
                                          chatts_stream_t stream) {
  CHATTS_REQUIRE(d && pos_dev && step_dev && token_dev, CHATTS_E_BADARG, "decode_step: null argument");
  StageRange stage("chatts.decode_step");
  const bool tp = d->cfg.tp_world > 1;
  CHATTS_REQUIRE(!tp || d->tp, CHATTS_E_BADARG, "decode_step: tp_world = %d but no exchange attached (chatts_decoder_set_tp)", d->cfg.tp_world);
  CHATTS_REQUIRE(!tp || chatts_tp_pending(d->tp) == 0, CHATTS_E_BADARG, "decode_step: the exchange has %d collectives whose epochs were never settled "
                 "(chatts_tp_flush_epochs): a step enqueued or captured now would reuse their epochs", tp ? chatts_tp_pending(d->tp) : 0);
  int rc;
  const int H = d->cfg.hidden;
  // Tensor parallel: the two exchanges of a layer ride in the o_proj / down_proj GEMV launches (6 launches per layer instead of 8;
  // CHATTS_TP_FUSE=0 keeps the stand-alone chatts_allreduce kernels: same bits - tests/test_gpu_tp_p2p.py)
  d->fuse_tp = tp && opt_get(OPT_TP_FUSE, 1) != 0;
  rc = CHATTS_OK;
  for (int l = 0; l < d->cfg.n_layers && rc == CHATTS_OK; ++l) {
    rc = chatts_decoder_layer_part(d, l, 0, 1, 0, pos_dev, n_splits, stream);
    if (rc == CHATTS_OK && tp && !d->tp_fused) rc = chatts_allreduce(d->tp, d->b.delta, d->b.x, d->b.x, H, stream);    // x += sum of the partial o_proj
    if (rc == CHATTS_OK) rc = chatts_decoder_layer_part(d, l, 1, 1, 0, pos_dev, n_splits, stream);
    if (rc == CHATTS_OK && tp && !d->tp_fused) rc = chatts_allreduce(d->tp, d->b.delta, d->b.x, d->b.x, H, stream);    // ... and down_proj
  }
  d->fuse_tp = false;
  if (rc) {      // a refused launch part-way: settle the epochs of the exchange-carrying GEMVs that WERE issued (every rank fails alike)
    if (tp) (void)chatts_tp_flush_epochs(d->tp, stream);
    return rc;
  }
  if ((rc = chatts_decoder_logits(d, 0, stream)) != 0) return rc;
  if ((rc = chatts_decoder_select_tokens(d, d->b.logits, 1, d->cfg.vocab_local, token_dev, token_logit_dev, out_tokens, 0, step_dev,
                                         pos_dev, 0, nullptr, stream)) != 0) return rc;
  return chatts_embed_token(token_dev, d->w.embed, embed_offset(d), embed_rows(d), d->cfg.hidden, d->b.x, stream);
}
