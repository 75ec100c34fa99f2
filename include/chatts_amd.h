/*
 * chatts_amd.h - C-ABI of libchatts_amd.so: the MI355X (gfx950) ChatTS inference hot path.
 *
 * Every entry point is extern "C", takes plain pointers + sizes + a hipStream_t passed as void*,
 * NEVER allocates, frees or synchronises (graph-capturable), and returns 0 or a negative
 * CHATTS_E_* code; chatts_last_error() returns a thread-local message for the last failure.
 * Device pointers are owned by the caller (PyTorch-ROCm is used purely as the allocator).
 *
 * Each function cites the reference interface (file:line under NetManAIOps/ChatTS) it replaces.
 * Number formats: activations / residual stream / KV cache / logits are float32; weights are
 * bfloat16 (uint16_t bit patterns) streamed from HBM; every weight x activation product is formed
 * either exactly on the f32 VALU (decode, M == 1) or by two bf16 MFMA passes over a hi/lo split of the
 * f32 activation ("bf16x2": batched decode M = 2..16, prefill, TS MLP) - see DESIGN.md section 3 for why (1e-3 logit tolerance).
 */
#ifndef CHATTS_AMD_H
#define CHATTS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHATTS_OK 0
#define CHATTS_E_BADARG (-1)          /* null pointer, negative size, unsupported dimension          */
#define CHATTS_E_SHAPE (-2)           /* shape constraint violated (alignment, head_dim != 128, ...) */
#define CHATTS_E_COUNT_MISMATCH (-3)  /* #<ts> placeholders != #TS rows (vLLM raises ValueError)     */
#define CHATTS_E_LAUNCH (-4)          /* hipLaunchKernel / hipGetLastError failed                    */
#define CHATTS_E_WORKSPACE (-5)       /* caller-supplied workspace too small                         */

typedef void* chatts_stream_t; /* hipStream_t */
typedef uint16_t chatts_bf16;  /* raw bfloat16 bits */
typedef uint16_t chatts_f16;   /* raw IEEE binary16 bits */
typedef struct ChattsTpComm ChattsTpComm;     /* tensor-parallel exchange (section below); opaque, host memory only */

const char* chatts_last_error(void);
/* ABI version of this header: bumped whenever a struct grows or a signature changes
 * (2: plane operands, sampler, decoder plane buffers; 3: post-norm planes of chatts_linear;
 *  4: tensor-parallel exchange chatts_tp_* / chatts_allreduce, decoder embed_rows + TP buffers, chatts_decoder_select_tokens;
 *  6: persistent decode step chatts_decoder_mega_*; the attention 'parts' form and its ChattsLinearArgs fields removed;
 *  7: ChattsLinearArgs.tp_reduce (exchange inside the projection's launch), chatts_tp_init_loopback, chatts_allreduce_bulk +
 *     chatts_decoder_prefill(_last) under tensor parallelism, chatts_decoder_logits_batched; the persistent decode step
 *     chatts_decoder_mega_* is gone - built and measured slower than the captured multi-kernel step in round 3);
 *  8: chatts_set_option / chatts_unset_option / chatts_get_option / chatts_option_name (tuning knobs: the library no longer reads the
 *     environment); ChattsLinearArgs.tile_counters, ChattsDecoderBuffers.tile_counters and CHATTS_TILE_COUNTERS removed (the in-launch
 *     split-K fix-up they served was measured slower twice); chatts_tp_flush_epochs;
 *  9: the f16q operand format of the prefill projections (chatts_split_f16q / chatts_weights_f16q / chatts_rmsnorm_f16q /
 *     chatts_linear_f16q, ChattsLayerWeights.*16 / *_q8 / *_q8e, chatts_decoder_set_prefill_f16q: an opt-in, parity-grade, SLOWER prefill mode - DESIGN.md section 14);
 *     the tensor-parallel exchange's release form per communicator (chatts_tp_cross_device / chatts_tp_set_cross_device /
 *     chatts_tp_set_bulk_release / chatts_tp_bulk_release); tiled prefill weights (chatts_tile_bf16, ChattsLinearArgs.w_tiled /
 *     planes_tiled, ChattsLayerWeights.*_t). */
#define CHATTS_ABI_VERSION 9
int chatts_abi_version(void);
/* Number of CUs of the current device (grid sizing), or <0. */
int chatts_device_cus(void);

/* Tuning options: named integer knobs, process-wide, all UNSET by default (= the shipped, measured-best choice).  They select
 * alternative paths that were built and measured (A/B runs, the bit-identity tests between kernel forms) and launch geometries for
 * sweeps; none is a fallback.  The library never reads the environment: a host that wants CHATTS_<NAME>=v honoured calls
 * chatts_set_option("<NAME>", v) (chatts_amd/_lib.py does so once, at load).  Names (a leading "CHATTS_" is ignored): GEMM_SK GEMM_T
 * GEMM_BM GEMM_PRECISION (1 = the bf16 speed mode) GEMM_PLANES_MIN_M GEMM_STREAM GEMM_STREAM_STAGES GEMM_STREAM_WAVES GEMM_STREAM_MB
 * GEMM_STREAM_MB_WAVES EPI_V4 EPI_NORM_Q EPI_NORM_Q_GROUPS EPI_NORM_REG POST_NORM_SMALL_M ROPE_FUSE ATTN_BF16X3 ATTN_EXACT ATTN_PLANES ATTN_XCD ATTN_ROWS ATTN_KSPLIT ARGMAX_2STAGE
 * TS_F32_PATH KV_ROUND TP_FUSE TP_FUSE_BLOCKS TP_BULK_BLOCKS TP_BULK_THREADS TP_BULK_FENCE TP_AR_BLOCKS GEMV_ROWS GEMV_UNR GEMV_NW GEMV_OCC GEMV_BLOCKS GEMV_LDSPAD GEMV_KS
 * FP8_BM FP8_ORDER GEMV8_ROWS GEMV8_UNR GEMV8_NW GEMV8_OCC TS_L0_FUSED (DESIGN.md section 11 says what each selects and where it was measured).  A set / unset is a relaxed atomic store: safe
 * beside running calls, which see either value.  chatts_unset_option(NULL) clears all.  chatts_option_name(i) enumerates (NULL at the end). */
int chatts_set_option(const char* name, int value);
int chatts_unset_option(const char* name);
int chatts_get_option(const char* name, int* value, int* is_set);
const char* chatts_option_name(int index);

/* ---------------------------------------------------------------------------------------------
 * Synthetic weights.  No reference twin: checkpoints cannot travel to the GPU box, so weights are
 * DEFINED by a counter-based integer hash that oracle/synth.py evaluates bit-identically on the host.
 *   value(i) = bf16_rne(base + n(i) * 2^-shift),  n(i) = b0+b1+b2+b3-510 over the 4 bytes of
 *   mix32(lo32(i) ^ mix32(key + hi32(i)*0x85ebca6b)),  i = (row0+r)*full_cols + (col0+c)
 * Writes a [rows, cols] block (leading dimension ld elements) of the full tensor; out_f32 != 0 stores
 * the bf16-rounded value widened to float32.
 * ------------------------------------------------------------------------------------------- */
int chatts_fill_hash(void* dst, int out_f32, uint32_t key, float base, int shift, int64_t rows,
                     int64_t cols, int64_t ld, int64_t row0, int64_t col0, int64_t full_cols,
                     chatts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Time-series encoder front end.
 * Replaces TimeSeriesEmbedding.forward's per-series Python loop (chatts/vllm/chatts_vllm.py:93-183)
 * and get_patch_cnt (:198-207).
 * ------------------------------------------------------------------------------------------- */

/* Value-preserved normalisation ON THE DEVICE = sp_encoding (chatts/utils/encoding_utils.py:23-37) + the pad / stack of
 * eval_prompt_to_encoding (:65-86) for a whole batch: raw [N, Lmax] float64 (rows padded arbitrarily beyond lengths[i]) ->
 * enc [N, 2*Lmax] float32 (value, mask) interleaved, zero padded, and stats [N, 6] float64 = (mean, scale factor, max, min,
 * first, last) - what the prompt prefix prints.  float64 throughout like numpy; the sum has a fixed order of its own, so the
 * mean may differ from np.mean in the last bit (documented in DESIGN.md section 7); max / min / ends are exact. */
int chatts_ts_normalise(const double* raw, const int32_t* lengths, int n_series, int lmax, float* enc, double* stats,
                        chatts_stream_t stream);

/* valid_len[i] = sum(long(mask_i)), patch_cnt[i] = ceil(valid_len/patch) from the padded
 * [N, 2*Lmax] (value, mask)-interleaved float32 tensor (chatts_vllm.py:94-100 / :198-207).
 * One wave per series, wave-shuffle reduction.  patch_cnt is int64 like the reference's. */
int chatts_ts_patch_cnt(const float* series, int n_series, int lmax, int patch_size,
                        int32_t* valid_len, int64_t* patch_cnt, chatts_stream_t stream);

/* mode: 0 = raw patches [P,16]; 1 = use_position_embedding [P,16+16*emb]; 2 = use_position_idx [P,32] */
typedef struct ChattsPatchifyArgs {
  const float* series;     /* [N, 2*Lmax] float32, (value, mask) interleaved, zero padded (:78-84) */
  const int32_t* row_off;  /* device [N+1]: exclusive scan of patch_cnt (first output row of series i) */
  const int32_t* valid_len;/* device [N] (from chatts_ts_patch_cnt or the host processor)            */
  const float* pos_table;  /* mode 1: [max_seq_len+1, emb_dim] float32 (position_embedding.weight)   */
  float* out;              /* [P, ld_out] float32; columns [feat, ld_out) are zero filled            */
  int n_series, lmax, patch_size, mode, emb_dim, max_seq_len;
  int max_valid_len;       /* mode 2 only: max_i valid_len (chatts_vllm.py:146)                      */
  int total_patches;       /* P = row_off[N], known on the host                                      */
  int ld_out;              /* >= feature count, multiple of 32 (K padding for the MFMA GEMM)         */
  /* optional: write the rows as bf16 hi / lo planes [P, ld_out] (hi = bf16(v), lo = bf16(v - hi): the operand format of
   * chatts_linear's plane path) instead of float32 `out`, which may then be NULL */
  chatts_bf16* out_hi;
  chatts_bf16* out_lo;
} ChattsPatchifyArgs;
/* Builds the MLP input rows: values of patch p of series i, tail padded with the LAST VALID value
 * (:121-125), followed by the position-embedding rows of indices 16p..16p+15 with padding_idx =
 * max_seq_len for pad slots (:119,128-129,161-181).  One wave per patch, coalesced row reads. */
int chatts_ts_patchify(const ChattsPatchifyArgs* args, chatts_stream_t stream);

/* Whole encoder in one call = _parse_and_validate_ts_input + TimeSeriesEmbedding.forward (chatts_vllm.py:493-536,
 * 93-193) after the host computed the row offsets: patchify into feat [P, in_features_pad], then the MLP
 * (Linear + exact-erf GELU) x (n-1) + Linear through the ping-pong buffers h0/h1 [P, hidden] into out [P, hidden].
 * feat / h0 / h1 are SCRATCH (float32-sized; with P > 1 and in_features_pad, hidden multiples of 64 they hold bf16 hi / lo
 * planes and every GEMM runs on the LDS-DMA plane kernels).  workspace >= max_l chatts_linear_workspace(P, hidden, K_l). */
typedef struct ChattsTsWeights {
  int patch_size, num_layers, hidden, mode, emb_dim, max_seq_len;
  int in_features_pad;         /* K of layer 0, zero padded to a multiple of 32 (64 for the plane path: 320 for 16 + 16*16) */
  const float* pos_table;      /* mode 1 */
  const chatts_bf16* w[8];     /* [hidden, K_l] bf16 */
  const float* b[8];           /* [hidden] */
} ChattsTsWeights;
int chatts_ts_encode(const float* series, const int32_t* row_off, const int32_t* valid_len, int n_series, int lmax,
                     int max_valid_len, int total_patches, const ChattsTsWeights* weights, float* feat, float* h0,
                     float* h1, float* out, void* workspace, size_t workspace_bytes, chatts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Linear layers:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)   A,C float32, W bfloat16 row-major [N,K]
 * (the HF nn.Linear.weight layout).  Replaces every nn.Linear the path executes:
 * TimeSeriesEmbedding.mlp (chatts_vllm.py:83-91,188), Qwen2 q/k/v/o/gate/up/down/lm_head
 * (vLLM / transformers, NOT IN REFERENCE; math per oracle/qwen_decoder.py).
 * ------------------------------------------------------------------------------------------- */
#define CHATTS_EPI_NONE 0     /* C = acc (+bias)                                                    */
#define CHATTS_EPI_GELU 1     /* C = gelu_erf(acc + bias)                (nn.GELU(), :87)           */
#define CHATTS_EPI_RESID 2    /* C = resid + acc (+bias); C may alias resid                         */
#define CHATTS_EPI_SWIGLU 3   /* W rows are gate/up interleaved in blocks of 16 rows;               */
                              /* C[M,N/2] = silu(gate) * up            (Qwen2MLP.forward)           */

typedef struct ChattsLinearArgs {
  const float* a;          /* [M, lda] */
  const chatts_bf16* w;    /* [N, ldw] */
  const float* bias;       /* [N] or NULL (for SWIGLU: interleaved like the rows) */
  const float* resid;      /* EPI_RESID: [M, ldc] */
  float* c;                /* [M, ldc]  (SWIGLU: [M, N/2]) */
  /* optional fused RMSNorm on the rows of A (decode path, M == 1 only):
   *   a_used[m,k] = norm_w[k] * (a[m,k] * rsqrt(mean_k(a[m,:]^2) + eps))   (Qwen2RMSNorm.forward) */
  const float* norm_w;     /* [K] or NULL */
  float norm_eps;
  int m, n, k, lda, ldw, ldc, epilogue;
  void* workspace;         /* split-K partials; may be NULL when chatts_linear_workspace() == 0 */
  size_t workspace_bytes;
  /* optional fp8 copy of W (OCP e4m3fn, [N, ldw8]) with a per-row power-of-two scale, w = w8_scale[n] * fp8: a
   * lossless encoding of the bf16 matrix (so both copies describe the same weights).  When set it is the copy that
   * is streamed (BASELINE.json config 5): the decode GEMV widens it on the VALU, the MFMA GEMM widens it to bf16
   * while staging to LDS; the row scale is applied in the epilogue.  The prefill path passes NULL (MFMA-bound). */
  const uint8_t* w8;
  const float* w8_scale;
  int ldw8;
  /* optional pre-split A (see chatts_split_bf16x2): a == a_hi + a_lo exactly to 16 mantissa bits, both planes bf16
   * [M, ld_planes].  When set (and M > 1, fp8 copy absent) the GEMM stages all three operand tiles with LDS-DMA
   * (global_load_lds) through a 3-deep ring instead of splitting A on the VALU per tile; `a` may then be NULL.
   * Same products, same accumulation order -> bit-identical results to the `a` path. */
  const chatts_bf16* a_hi;
  const chatts_bf16* a_lo;
  int ld_planes;
  /* optional (M > 1): write the result as bf16 hi / lo planes [M, ld_cplanes] - the operand format of the next
   * projection - instead of float32 `c` (which may then be NULL).  Same values as chatts_split_bf16x2(c). */
  chatts_bf16* c_hi;
  chatts_bf16* c_lo;
  int ld_cplanes;
  /* optional (M > 1, EPI_NONE / EPI_RESID, float32 c): additionally write RMSNorm(c) - rows of width N, weight
   * post_norm_w, chatts_rmsnorm_planes arithmetic - as bf16 hi / lo planes [M, ld_post]: the operand of the projection that
   * follows a residual update.  Fused into the split-K epilogue when there is one, otherwise an extra launch. */
  const float* post_norm_w;
  float post_norm_eps;
  chatts_bf16* post_hi;
  chatts_bf16* post_lo;
  int ld_post;
  /* optional 4-bit copy of W (GPTQ-Int4 checkpoints): codes row-major [N, ldw4 bytes], byte j of a row = code 2j | code 2j+1 << 4;
   * w4_sz [N, K / w4_group, 2] float32 = (scale, scale * zero) per group of w4_group (a power of two >= 16) weights of a row.  The
   * weight it encodes is bf16_rne(code * scale - scale * zero) and MUST equal `w`: the M == 1 GEMV streams the codes (a quarter
   * of the bytes), rebuilds exactly that bf16 value and multiplies it with the bf16 hi / lo split of x (the bf16x2 scheme of the
   * MFMA GEMMs: exact products, float32 accumulate, x to 2^-17); every other kernel keeps streaming `w`. */
  const uint8_t* w4;
  const float* w4_sz;
  int ldw4, w4_group;
  /* encoding of `w8`: 0 = OCP fp8 e4m3fn (above); 1 = int8 (two's complement), w = w8_scale[n] * int8 with the same per-row
   * power-of-two scale - |int8| <= 127 has 7 significant bits, so the dequantised weight is again exactly a bf16 number and the
   * int8 tensor is a lossless encoding of the bf16 matrix the other kernels stream (the weight-only 8-bit format in the role of
   * the HF demo's `load_in_8bit`, demo/demo_hf.ipynb:78-80; bitsandbytes' own outlier decomposition is not reproduced).  Streamed by
   * the M == 1 GEMV and the M <= 16 weight-streaming GEMM; every other kernel keeps streaming `w`. */
  int w8_format;
  /* optional (M == 1, EPI_RESID, no bias): W is this rank's K-slice of a row-parallel projection (o_proj / down_proj under tensor
   * parallelism, vLLM's RowParallelLinear + all-reduce, demo/demo_vllm.py:30).  The GEMV launch then carries the exchange itself: a
   * wave pushes its finished rows to every rank as tagged granules, polls the same rows of all ranks from the local exchange buffer
   * and writes c = resid + (sum over ranks in rank order) - the arithmetic of chatts_allreduce(partial, c, resid), bit for bit,
   * without the partial vector's round trip and without the stand-alone exchange launch.  n <= chatts_tp_max_elems(comm).  Every rank
   * must issue the same call (same N, same launch geometry) - like any collective. */
  ChattsTpComm* tp_reduce;
  /* optional (prefill kernel only, i.e. planes given and M >= 96): the same bf16 matrix as `w` in the tiled layout of chatts_tile_bf16
   * (ldw == K).  The prefill kernel is bound by its operand feed, and an LDS-DMA piece of 16 rows x 64 bytes uses half of each of the 16
   * cache lines it touches (profiles/r6_feed_probe.txt: 18 B/clk/CU against 27-30 for 1 KB of consecutive memory).  Same products, same
   * order: bit-identical results.  planes_tiled != 0: a_hi / a_lo are in that layout as well (ld_planes ignored; every kernel other than
   * the prefill kernel refuses them). */
  const chatts_bf16* w_tiled;
  int planes_tiled;
} ChattsLinearArgs;
#define CHATTS_W8_FP8 0
#define CHATTS_W8_INT8 1
size_t chatts_linear_workspace(int m, int n, int k);
/* Host-only query (no device work): how many waves share a row group of the M == 1 bf16 GEMV for this shape on the current device
 * (1 = the whole-K kernel every TP = 1 projection of the supported models runs; > 1 = the K-split form of shard-sized column-parallel
 * projections, which sums a row in another order).  Lets a host-side test pin which shapes take which kernel. */
int chatts_gemv_ksplit(int n, int k, int epilogue, int has_norm);
/* hi = bf16(x) (RNE), lo = bf16(x - hi): the operand split of the bf16x2 GEMM, done once per activation matrix. */
int chatts_split_bf16x2(const float* x, int m, int k, int ldx, chatts_bf16* hi, chatts_bf16* lo, int ld_planes,
                        chatts_stream_t stream);
/* bf16 [rows, ld] row-major (K % 32 == 0) -> the tiled layout the prefill kernel's LDS-DMA pieces read as consecutive memory:
 * block (b = row / 16, t = k / 32) is the 1 KB at ((b * K / 32) + t) * 1 KB; inside it the 16-byte chunk at position l (0 .. 63) holds
 * row 16 b + (l >> 2), K-values 32 t + 8 c .. 32 t + 8 c + 7 with c = (l & 3) ^ ((l >> 5) << 1).  Rows beyond the matrix (the last
 * block of a ragged row count) repeat the last row.  dst holds chatts_tile_bf16_elems(rows, k) = ceil(rows / 16) * 16 * k elements. */
size_t chatts_tile_bf16_elems(int rows, int k);
int chatts_tile_bf16(const chatts_bf16* src, int rows, int k, int ld, chatts_bf16* dst, chatts_stream_t stream);
/* Dispatch: M == 1       -> weight-streaming GEMV (gemv_ldsx_kernel: exact f32 VALU products, x staged in LDS, HBM-bound);
 *           2 <= M <= 16 -> with pre-split planes: gemm_stream_kernel (W and the A planes staged by whole-line LDS-DMA through a
 *                           4-deep ring, split-K over workgroups); with float32 A: the register-staged gemm_bf16x2_kernel;
 *           larger M     -> gemm_dma_kernel (LDS-DMA, loader + compute waves) on pre-split planes for M >= 96 and K % 64 == 0,
 *                           otherwise gemm_bf16x2_kernel (register-staged LDS tiles);
 *           all MFMA paths: v_mfma_f32_16x16x32_bf16 with the bf16x2 split of A. */
int chatts_linear(const ChattsLinearArgs* args, chatts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The f16q operand format of the prefill projections (round 6; csrc/f16q.h, csrc/gemm_f16q.hip).  No reference twin: the reference
 * multiplies fp16 x fp16 inside vLLM / flash-attn kernels (NOT IN REFERENCE); this is HOW the float32-grade projection
 * (oracle: float32 matmul, oracle/qwen_decoder.py) is issued to the CDNA4 matrix pipes at 1.5 instead of 2 pass-equivalents per product.
 *   x = hi + lo,  hi = f16_rne(clamp(x, +-65504)) [M, ld_planes],  lo = x - hi (exact) stored as OCP e4m3fn q = rne(lo * 2^-E) [M, ld_planes],
 *   E + 127 = one e8m0 byte per row and 128 consecutive K-values [M, ld_scale]: the smallest power of two with max|lo| / 2^E <= 448 (127 for
 *   an all-zero block).  Weights: an f16 copy of the bf16 matrix (exact for 2^-17 <= |w| <= 65504) and an e4m3 copy with ONE power-of-two
 *   scale per row (byte E + 127, the same rule over the row).  C = epilogue(hi . W16^T + (q 2^E) . (W8 2^Ew)^T), float32 accumulate.
 * chatts_split_f16q: float32 [M, K] -> planes (K % 128 == 0); tiled != 0 (also chatts_rmsnorm_f16q's): hi / lo8 in the TILED plane layout
 * (ChattsLinearF16qArgs.planes_tiled; ld_planes == K; buffers of ceil(M / 16) * 16 rows).  chatts_weights_f16q: bf16 [N, K] -> (w16, w8, w8_exp), ld_out % 16 == 0;
 * chatts_tile_bf16 (on w16, as 16-bit elements) and chatts_tile_e4m3 (on w8) turn them into the kernel's block order (w_tiled).
 * chatts_rmsnorm_f16q: chatts_rmsnorm's arithmetic with the row written as planes.  chatts_linear_f16q: the GEMM (M >= 1; meant for
 * prefill chunks); epilogues of chatts_linear; SWIGLU may write its result as planes (c_hi / c_lo8 / c_scale: the next projection's
 * operand) instead of float32 c; EPI_NONE / EPI_RESID may additionally write RMSNorm(c) as planes (post_*), fused into the split-K
 * epilogue when there is one.  Workspace: chatts_linear_f16q_workspace(m, n, k) bytes.
 * ------------------------------------------------------------------------------------------- */
typedef struct ChattsLinearF16qArgs {
  const chatts_f16* a_hi;    /* [M, ld_a] */
  const uint8_t* a_lo8;      /* [M, ld_a] e4m3 */
  const uint8_t* a_scale;    /* [M, ld_scale] e8m0 */
  int ld_a, ld_scale;
  const chatts_f16* w16;     /* [N, ldw] */
  const uint8_t* w8;         /* [N, ldw] e4m3 */
  const uint8_t* w8_exp;     /* [N] e8m0 */
  int ldw;
  const float* bias;         /* [N] or NULL (SWIGLU: interleaved like the rows) */
  const float* resid;        /* EPI_RESID: [M, ldc] */
  float* c;                  /* [M, ldc] (SWIGLU: [M, N / 2]); may be NULL when c_hi is set */
  int m, n, k, ldc, epilogue;
  chatts_f16* c_hi;          /* optional (SWIGLU only): the result as f16q planes [M, ld_cplanes], scales [M, ld_cscale] */
  uint8_t* c_lo8;
  uint8_t* c_scale;
  int ld_cplanes, ld_cscale;
  const float* post_norm_w;  /* optional (EPI_NONE / EPI_RESID): RMSNorm(c) with this weight as f16q planes */
  float post_norm_eps;
  chatts_f16* post_hi;
  uint8_t* post_lo8;
  uint8_t* post_scale;
  int ld_post, ld_pscale;
  void* workspace;
  size_t workspace_bytes;
  int planes_tiled;          /* EVERY plane set of this call - a_hi / a_lo8, the SwiGLU output c_hi / c_lo8, post_hi / post_lo8 - is in the TILED plane
                              * layout (csrc/f16q.h: blocks of 16 rows x 32 K-values, hi 1 KB in chatts_tile_bf16's order, lo8 512 B = two 16-byte
                              * halves per row; leading dimensions = the matrix widths; buffers hold ceil(M / 16) * 16 rows); scales stay row-major */
  int w_tiled;               /* w16 / w8 are in the tiled layouts of chatts_tile_bf16 (as 16-bit elements) / chatts_tile_e4m3 (ldw == K): every
                              * LDS-DMA piece of the weights is 1 KB of consecutive memory; bit-identical results */
} ChattsLinearF16qArgs;
int chatts_split_f16q(const float* x, int m, int k, int ldx, chatts_f16* hi, uint8_t* lo8, uint8_t* scale, int ld_planes, int ld_scale,
                      int tiled, chatts_stream_t stream);
int chatts_weights_f16q(const chatts_bf16* w, int n, int k, int ldw, chatts_f16* w16, uint8_t* w8, uint8_t* w8_exp, int ld_out,
                        chatts_stream_t stream);
int chatts_rmsnorm_f16q(const float* x, const float* w, chatts_f16* hi, uint8_t* lo8, uint8_t* scale, int ld_planes, int ld_scale, int t,
                        int hidden, float eps, int tiled, chatts_stream_t stream);
/* e4m3 [rows, ld] row-major (K % 32 == 0) -> the tiled layout of the f16q kernel's e4m3 pieces: block (b = row / 32, t = k / 32) is the 1 KB at
 * ((b * K / 32) + t) * 1 KB; the 16-byte chunk at position l holds row 32 b + (l >> 5) * 16 + (l & 15), bytes 32 t + ((l >> 4) & 1) * 16 .. + 15.
 * dst holds chatts_tile_e4m3_bytes(rows, k) = ceil(rows / 32) * 32 * k bytes; rows beyond the matrix repeat the last row. */
size_t chatts_tile_e4m3_bytes(int rows, int k);
int chatts_tile_e4m3(const uint8_t* src, int rows, int k, int ld, uint8_t* dst, chatts_stream_t stream);
size_t chatts_linear_f16q_workspace(int m, int n, int k);
int chatts_linear_f16q(const ChattsLinearF16qArgs* a, chatts_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * SPEED MODE, not parity grade: fp8 x fp8 projections on the CDNA4 block-scaled matrix pipe (v_mfma_scale_f32_16x16x128_f8f6f4 with
 * unit block scales: twice the bf16 MFMA rate, and ONE pass instead of the two of the bf16x2 split).  What vLLM does with an fp8
 * quant_config (the reference passes it through: chatts_vllm.py:475,481; BASELINE.json config 5 "fp8 weights (CDNA4 fp8 MFMA)"):
 * activations are quantised per row (token) on the fly, weights are the per-row power-of-two-scaled e4m3 copies of
 * ChattsLinearArgs.w8.  Results differ from the float32-activation default by ~1e-2 (3-bit mantissas): selected only by
 * precision="fp8" for the MFMA-bound stages (prefill chunks, the TS encoder at thousands of patches), labelled wherever it is printed.
 * ------------------------------------------------------------------------------------------- */
/* q[m, :] = e4m3fn(x'[m, :] / scale[m]), scale[m] = max|x'[m, :]| / 448 (1 for an all-zero row); x' = x, or RMSNorm(x) with weight
 * norm_w when norm_w != NULL (Qwen2RMSNorm.forward).  K % 4 == 0; q rows ldq bytes apart. */
int chatts_quantize_rows_fp8(const float* x, int m, int k, int ldx, const float* norm_w, float norm_eps, uint8_t* q, int ldq,
                             float* scale, chatts_stream_t stream);
typedef struct ChattsLinearFp8Args {
  const uint8_t* a8;       /* [M, lda8] e4m3fn activations (chatts_quantize_rows_fp8) */
  const float* a_scale;    /* [M] */
  const uint8_t* w8;       /* [N, ldw8] e4m3fn weights, row-major like ChattsLinearArgs.w8 */
  const float* w_scale;    /* [N] */
  const float* bias;       /* [N] or NULL (SWIGLU: interleaved like the rows) */
  const float* resid;      /* EPI_RESID: [M, ldc]; may alias c */
  float* c;                /* [M, ldc] float32 (SWIGLU: [M, N / 2]) */
  int m, n, k, lda8, ldw8, ldc, epilogue;      /* K % 128 == 0 (zero pad both operands); CHATTS_EPI_* */
} ChattsLinearFp8Args;
/* C = epilogue(a_scale[m] * w_scale[n] * sum_k a8[m, k] * w8[n, k]), float32 accumulate */
int chatts_linear_fp8(const ChattsLinearFp8Args* args, chatts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Embedding gather + TS merge.  Replaces get_input_embeddings (chatts_vllm.py:564-574) =
 * embed_tokens(ids) then vLLM merge_multimodal_embeddings: rows whose id == ts_token_id are
 * overwritten, in order, by the TS rows.  ids_host (optional) lets the count check run on the host
 * and return CHATTS_E_COUNT_MISMATCH without a device sync; with ids_host == NULL a mismatch is
 * recorded in *dev_status (bit 0) instead.  ts_rows == NULL / n_ts_rows == 0: plain gather.
 * ------------------------------------------------------------------------------------------- */
int chatts_embed_merge(const int64_t* ids_dev, const int64_t* ids_host, int t, const chatts_bf16* table,
                       int64_t vocab, int hidden, const float* ts_rows, int n_ts_rows,
                       int64_t ts_token_id, float* out, int32_t* scratch /* [t+1] */,
                       int32_t* dev_status, chatts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Decoder pieces (Qwen2 / Qwen3; NOT IN REFERENCE, selected at chatts_vllm.py:483-488,:664-669).
 * ------------------------------------------------------------------------------------------- */

/* y[t,:] = w * (x[t,:] * rsqrt(mean(x[t,:]^2) + eps))            (Qwen2RMSNorm.forward) */
int chatts_rmsnorm(const float* x, const float* w, float* y, int t, int hidden, float eps,
                   chatts_stream_t stream);
/* The same, with the result written as bf16 hi / lo planes (== chatts_split_bf16x2(chatts_rmsnorm(x))). */
int chatts_rmsnorm_planes(const float* x, const float* w, chatts_bf16* hi, chatts_bf16* lo, int ld_planes, int t, int hidden,
                          float eps, chatts_stream_t stream);

typedef struct ChattsKvCache {
  float* k;                /* [n_kv, max_ctx, 128] float32 of ONE layer of ONE sequence */
  float* v;
  int max_ctx;             /* positions the sequence may reach */
  /* Block-paged form (the vLLM engine's KV layout), selected by block_table != NULL: k / v then
   * point at ONE layer's block POOL [n_blocks, n_kv, block_size, 128] shared by all sequences, and position j of this
   * sequence lives in block block_table[j / block_size], row j % block_size.  block_size: a power of two, 64..32768;
   * max_ctx must be a multiple of it; the table holds max_ctx / block_size entries (entries past the sequence's length are
   * never read).  Batched entry points: sequence b uses block_table + b * table_stride and seq_stride is ignored. */
  const int32_t* block_table;
  int block_size;
  int table_stride;
} ChattsKvCache;

/* In-place on qkv [T, (n_q+2*n_kv)*128] (after bias): optional per-head RMSNorm of q and k (Qwen3
 * q_norm/k_norm, NULL for Qwen2), NeoX rotate-half RoPE at positions pos0..pos0+T-1 (pos0 read from
 * *pos0_dev when pos0_dev != NULL, so a captured graph can be replayed), then k and v rows are
 * written into the cache at those positions.  cos/sin from the float32 table [max_pos, 64]. */
int chatts_rope_kv_write(float* qkv, int t, int n_q, int n_kv, const float* q_norm_w,
                         const float* k_norm_w, float norm_eps, const float* cos_tab,
                         const float* sin_tab, int pos0, const int32_t* pos0_dev,
                         const ChattsKvCache* cache, chatts_stream_t stream);

/* Causal GQA attention of T query rows (positions pos0..pos0+T-1) against cache rows [0, pos].
 * q is read from qkv [T, (n_q+2*n_kv)*128]; out [T, n_q*128].  float32 scores, softmax and PV.
 * n_splits > 1: key tiles are strided over n_splits workgroups per (row, kv-head), partials go to workspace and
 * are combined by a second kernel.  workspace >= chatts_attn_workspace() (needed whenever n_splits > 1).
 * n_splits == 1, T >= 64, host-side pos0: if the (optional) workspace holds ceil((pos0 + T) / 32) * n_kv * 32 KB, the call first splits the
 * sequence's K / V rows into bf16 hi / lo planes there (kv_planes_kernel) and runs the prefill kernel that reads them
 * (attn_prefill_planes_kernel); a smaller or NULL workspace selects the kernel that splits while it stages - same products, sums in
 * another order (both <= 3e-5 from float64).  The workspace contents are scratch either way. */
size_t chatts_attn_workspace(int t, int n_q, int n_splits);
int chatts_attention(const float* qkv, int t, int n_q, int n_kv, int pos0, const int32_t* pos0_dev,
                     const ChattsKvCache* cache, float* out, int n_splits, void* workspace,
                     size_t workspace_bytes, chatts_stream_t stream);

/* Decode form (T = 1) that also replaces chatts_rope_kv_write: qkv_raw holds the un-rotated projections
 * (after bias); the kernel applies the optional per-head q/k RMSNorm + RoPE at position pos (or *pos_dev),
 * stores the new K/V row into the cache and attends over cache rows [0, pos].  One wave per (kv head, 16-key
 * tile); n_splits (<= 64) is the number of tile slots, slot s walks tiles s, s+n_splits, ...; the workspace
 * (chatts_attn_workspace(1, n_q, n_splits) bytes) is always required. */
int chatts_attention_decode_fused(const float* qkv_raw, int n_q, int n_kv, const float* q_norm_w,
                                  const float* k_norm_w, float norm_eps, const float* cos_tab,
                                  const float* sin_tab, int pos, const int32_t* pos_dev,
                                  const ChattsKvCache* cache, float* out, int n_splits, void* workspace,
                                  size_t workspace_bytes, chatts_stream_t stream);

/* Batched form of the above: `batch` sequences, one decode token each; row b of qkv_raw / out and pos_dev[b] belong
 * to sequence b, whose cache is `cache` advanced by b * seq_stride floats (same layer). */
int chatts_attention_decode_batched(const float* qkv_raw, int batch, int n_q, int n_kv, const float* q_norm_w,
                                    const float* k_norm_w, float norm_eps, const float* cos_tab,
                                    const float* sin_tab, int pos, const int32_t* pos_dev,
                                    const ChattsKvCache* cache, size_t seq_stride, float* out, int n_splits,
                                    void* workspace, size_t workspace_bytes, chatts_stream_t stream);
/* logits [V] float32 -> *token (first index of the maximum, like torch.argmax); optionally also
 * appends the token to out_tokens[*step_dev] and increments *step_dev and *pos_dev (decode loop
 * state kept on the device so that a hipGraph of one decode step is replayable). */
int chatts_argmax(const float* logits, int64_t vocab, int64_t vocab_offset, int64_t* token,
                  float* token_logit, int64_t* out_tokens, int32_t* step_dev, int32_t* pos_dev,
                  chatts_stream_t stream);

/* Per-sequence form: logits [batch, logits_stride], token/token_logit/step_dev/pos_dev are arrays of `batch`,
 * out_tokens is [batch, out_stride]; sequence b appends at out_tokens[b][step_dev[b]]. */
int chatts_argmax_batched(const float* logits, int batch, int64_t logits_stride, int64_t vocab, int64_t vocab_offset,
                          int64_t* token, float* token_logit, int64_t* out_tokens, int64_t out_stride,
                          int32_t* step_dev, int32_t* pos_dev, int pos_limit /* pos saturates here; 0 = none */,
                          chatts_stream_t stream);
/* The same selection and side effects in two launches when the caller lends scratch (>= chatts_argmax_workspace(batch) bytes of device
 * memory): 64 workgroups per row + one merging wave per row instead of one workgroup per row (24 us for 152 k logits - the tail of
 * every decode step; the decoder passes its idle split-K workspace).  Same token: maximum, first index on ties.  Without scratch (NULL
 * / too small) it IS chatts_argmax_batched. */
size_t chatts_argmax_workspace(int batch);
int chatts_argmax_batched_ws(const float* logits, int batch, int64_t logits_stride, int64_t vocab, int64_t vocab_offset,
                             int64_t* token, float* token_logit, int64_t* out_tokens, int64_t out_stride,
                             int32_t* step_dev, int32_t* pos_dev, int pos_limit, void* workspace, size_t workspace_bytes,
                             chatts_stream_t stream);
int chatts_embed_token_batched(const int64_t* token_dev, int batch, const chatts_bf16* table, int64_t vocab_offset,
                               int64_t vocab_rows, int hidden, float* out /* [batch, hidden] */, chatts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Sampling (temperature / top-k / top-p): what the reference's evaluation drivers ask of vLLM / HF -
 * SamplingParams(temperature=0.2) (chatts/utils/inference_tsmllm_vllm.py:43-46), temperature=0.5 + top_p=0.95
 * (chatts/utils/llm_utils.py:94,153).  Rule (vLLM's, restated in oracle/sampler.py): logits / temperature; keep the
 * top_k largest; of those the smallest set of most probable tokens whose mass reaches top_p; renormalise; draw.
 * Ties at either cut are kept.  The uniform variate is a counter-based hash of (seed, sequence, step_dev[sequence]),
 * so a captured decode step replays with fresh draws and the same seed reproduces the same tokens.
 * Same side effects as chatts_argmax_batched (token, logit, out_tokens[step], ++step, ++pos).
 * temperature == 0 is greedy decoding: call chatts_argmax.
 * ------------------------------------------------------------------------------------------- */
typedef struct ChattsSamplingArgs {
  float temperature;   /* > 0 */
  int top_k;           /* <= 0 or >= vocab: off */
  float top_p;         /* (0, 1]; 1 = off */
  uint32_t seed;
  int32_t* n_kept;     /* optional diagnostics [batch]: size of the renormalised set ...            */
  float* kept_mass;    /* ... and its share of the (top-k) probability mass                          */
  /* Per-row mode (optional): device arrays [batch], read by the kernel at run time - a captured decode step serves requests with
   * DIFFERENT sampling settings, and a setting changes without re-capturing.  temperature_rows != NULL switches it on (the scalars
   * above are then ignored): row b uses (temperature_rows[b], top_k_rows[b], top_p_rows[b], seed_rows[b]); temperature 0 = that row
   * decodes greedily (torch.argmax's token); top_k_rows / top_p_rows / seed_rows may be NULL (off / off / 0).  The variate of a row
   * is a hash of (seed_rows[b], step_dev[b]) only - not of b - so a request's tokens do not depend on the cache slot it landed in. */
  const float* temperature_rows;
  const int32_t* top_k_rows;
  const float* top_p_rows;
  const uint32_t* seed_rows;
} ChattsSamplingArgs;
int chatts_sample_batched(const float* logits, int batch, int64_t logits_stride, int64_t vocab, int64_t vocab_offset,
                          const ChattsSamplingArgs* args, int64_t* token, float* token_logit, int64_t* out_tokens,
                          int64_t out_stride, int32_t* step_dev, int32_t* pos_dev, int pos_limit, chatts_stream_t stream);

/* x[h] = float(table[*token - vocab_offset, h]) : next-step input embedding, token id read on the device. */
int chatts_embed_token(const int64_t* token_dev, const chatts_bf16* table, int64_t vocab_offset,
                       int64_t vocab_rows, int hidden, float* out, chatts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Tensor-parallel exchange (one process per GPU).  Replaces what the reference gets from vLLM's
 * tensor_parallel_size=k (demo/demo_vllm.py:30, chatts/utils/llm_utils.py:154): the NCCL all-reduce after o_proj and
 * down_proj, and the logits gather / greedy-token agreement of a vocab-parallel lm_head.  The decode-sized messages
 * (H float32 per sequence, 96 per token) are pure latency, so they are one-shot peer-to-peer kernels over IPC-mapped
 * exchange buffers (8-byte {epoch tag, value} granules pushed over every xGMI link in parallel; sums in rank order, hence
 * bit-identical on all ranks; epochs live on the device, so the collectives are hipGraph-capturable); prefill-sized
 * all-reduces stay with RCCL (torch.distributed) in the host loop.  Bootstrap: every rank allocates a buffer and exports
 * a handle, the host exchanges the W handles (any out-of-band channel: torch.distributed all_gather_object), then
 * chatts_tp_init maps the peers.  The ONLY entry points that allocate device memory are the two buffer calls below:
 * IPC-exportable fine-grained memory cannot come from the caller's caching allocator.
 * ------------------------------------------------------------------------------------------- */
#define CHATTS_TP_HANDLE_BYTES 64
#define CHATTS_TP_MAX_WORLD 8
/* bytes of one rank's exchange buffer for collectives of up to max_elems float32 per rank */
size_t chatts_tp_buffer_bytes(int world, int64_t max_elems);
/* the same plus a BULK region for prefill-sized sums of up to bulk_elems float32 (chatts_allreduce_bulk); the init calls below
 * recognise the region from the buffer's size */
size_t chatts_tp_buffer_bytes_bulk(int world, int64_t max_elems, int64_t bulk_elems);
/* hipExtMallocWithFlags(uncached) + zero fill + (handle != NULL) hipIpcGetMemHandle into handle[CHATTS_TP_HANDLE_BYTES] */
int chatts_tp_buffer_alloc(size_t bytes, void** dev_ptr, uint8_t* handle);
/* Returns the buffer to the library's free list (a later chatts_tp_buffer_alloc re-uses it, zero-filled); the driver gets the
 * memory back at process exit.  Uncached allocations are deliberately never hipFree'd mid-process: see csrc/tp.hip. */
int chatts_tp_buffer_free(void* dev_ptr);
/* handles: [world][CHATTS_TP_HANDLE_BYTES] in rank order (entry `rank` is ignored); maps every peer buffer (hipIpcOpenMemHandle). */
ChattsTpComm* chatts_tp_init(int rank, int world, void* local_buf, const uint8_t* handles, size_t bytes, int64_t max_elems);
/* the same for ranks that live in ONE process (bufs[r] = rank r's buffer, plain device pointers): single-GPU emulation, tests */
ChattsTpComm* chatts_tp_init_local(int rank, int world, void* const* bufs, size_t bytes, int64_t max_elems);
/* ONE rank of a `world`-rank group alone on its device (measurement aid: tools/tp_shard_step.py): every push lands in the LOCAL
 * buffer, in the slot of the peer it would have gone to, carrying 0.0 for the absent peers - the same stores and polls per element
 * as a real step at zero link latency, so a single GPU can time one rank's decode step at the shard shapes of TP = 2 / 4 / 8.  The
 * sums are this rank's partials alone: timing only, never a result. */
ChattsTpComm* chatts_tp_init_loopback(int rank, int world, void* local_buf, size_t bytes, int64_t max_elems);
void chatts_tp_destroy(ChattsTpComm*);
int chatts_tp_rank(const ChattsTpComm*);
int chatts_tp_world(const ChattsTpComm*);
int64_t chatts_tp_max_elems(const ChattsTpComm*);
/* diagnostic (synchronises: one hipMemcpy): >= 0 status bits - bit 0 = a peer's contribution did not arrive within ~2 s,
 * the results since then are garbage; call chatts_tp_reset on every rank before re-using the comm */
int chatts_tp_status(ChattsTpComm*);
/* Release form of the prefill-sized sums (chatts_allreduce_bulk), per communicator.  Two forms publish a workgroup's stores to its peers:
 * the system-scope fence (__threadfence_system(), round 4) and the LIGHT form (s_waitcnt vmcnt(0): the exchange buffers are uncached,
 * round 5: 38.5 vs 50.4 us per sum).  The light form has only ever been exercised with all ranks on one device; that a store has reached
 * a peer's memory ACROSS a link when vmcnt returns is an unproven assumption, and a reordering there is silent wrong sums.  Hence:
 *   chatts_tp_cross_device   1 = some peer buffer lives on another device than the local one, or its device could not be determined
 *                            (hipPointerGetAttributes at init); chatts_tp_set_cross_device lets the host say so itself (it knows the ranks'
 *                            devices; chatts_amd/tp.py gathers their PCI ids).
 *   chatts_tp_bulk_release   the form the next sum uses: 1 = fence, 0 = light.  Cross-device communicators use the FENCE unless the host
 *                            has validated the light form on these very links and called chatts_tp_set_bulk_release(c, 0)
 *                            (P2PExchange.create's first-contact test: >= 64 sums of a rank-dependent pattern, light == fenced element
 *                            for element on every rank); same-device and loop-back communicators use the light form.
 *   chatts_tp_set_bulk_release(c, mode)   -1 = by device (default), 0 = light, 1 = fence.  The TP_BULK_FENCE option, when SET,
 *                            overrides the communicator's mode either way (A/B runs and tests).  The forms interoperate (same
 *                            flag protocol: they differ in what precedes a rank's flag stores); the host sets one form on all ranks. */
int chatts_tp_cross_device(const ChattsTpComm*);
int chatts_tp_set_cross_device(ChattsTpComm*, int cross);
int chatts_tp_set_bulk_release(ChattsTpComm*, int mode);
int chatts_tp_bulk_release(const ChattsTpComm*);
/* zero the local buffer and the call counter (all ranks, then a host barrier, before re-using a comm after an error) */
int chatts_tp_reset(ChattsTpComm*, chatts_stream_t stream);
/* The exchange-carrying projections (ChattsLinearArgs.tp_reduce) do not advance the device-resident call counter themselves: the host
 * counts them (their epoch = counter + 1 + number issued since the last advancing collective) and the next stand-alone collective
 * advances the counter for all of them.  chatts_tp_flush_epochs makes a sequence that ENDS with such projections self-contained: it
 * enqueues a one-thread kernel that adds the pending count to the device counter and zeroes the host count (no-op when nothing is
 * pending; every rank must call it at the same point).  chatts_decoder_prefill_last does so itself; a captured decode step requires a
 * flushed communicator (chatts_decoder_decode_step* return CHATTS_E_BADARG otherwise), because its epochs are baked in relative to
 * the counter at replay time.  chatts_tp_pending: the host count (0 = flushed). */
int chatts_tp_flush_epochs(ChattsTpComm*, chatts_stream_t stream);
int chatts_tp_pending(const ChattsTpComm*);
/* out[i] = (resid ? resid[i] : 0) + sum over ranks of in[i], ranks added in rank order; n <= max_elems; out may alias resid */
int chatts_allreduce(ChattsTpComm*, const float* in, float* out, const float* resid, int64_t n, chatts_stream_t stream);
/* x[i] += sum over ranks of in[i] for PREFILL-sized vectors ([T, H] partial sums, SURVEY.md section 5.8 "large"): a two-shot
 * all-reduce - direct reduce-scatter into the owner's buffer, the owner adds the W contributions in rank order, direct all-gather -
 * as ONE kernel over the bulk region of the exchange buffers; n <= chatts_tp_bulk_elems(comm) (0 = the buffer has no bulk region).
 * Bit-identical results on all ranks; graph-capturable like the other collectives.  Replaces the host-enqueued RCCL all-reduce
 * between the layer halves of a prefill chunk (vLLM's RowParallelLinear all-reduce, demo/demo_vllm.py:30). */
int64_t chatts_tp_bulk_elems(const ChattsTpComm*);
int chatts_allreduce_bulk(ChattsTpComm*, const float* in, float* x, int64_t n, chatts_stream_t stream);
/* in [rows, row_len] per rank -> out [rows, world * row_len] (row b = the ranks' rows b concatenated in rank order) */
int chatts_allgather(ChattsTpComm*, const float* in, float* out, int64_t rows, int64_t row_len, chatts_stream_t stream);
/* per sequence b < batch: the global greedy token from every rank's local (max logit, global token id); ties -> lowest id;
 * side effects of chatts_argmax_batched on every rank */
int chatts_tp_argmax(ChattsTpComm*, int batch, const float* local_logit, const int64_t* local_token, int64_t* token,
                     float* token_logit, int64_t* out_tokens, int64_t out_stride, int32_t* step_dev, int32_t* pos_dev,
                     int pos_limit, chatts_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole decoder (TP rank-local).  Replaces Qwen2TSForCausalLM.forward / compute_logits
 * (chatts_vllm.py:576-610).  Weight pointers are BORROWED for the life of the handle.
 * Packed layouts (cf. packed_modules_mapping, chatts_vllm.py:454-464):
 *   qkv  [ (n_q + 2*n_kv)*128, H ]   rows = q heads | k heads | v heads of THIS rank
 *   gate_up [ 2*I_local, H ]          gate/up interleaved in blocks of 16 rows
 *   o    [ H, n_q*128 ]  down [ H, I_local ]   (row-parallel: this rank's K-slice, ld = local K)
 * ------------------------------------------------------------------------------------------- */
typedef struct ChattsLayerWeights {
  const float* input_norm;      /* [H] */
  const chatts_bf16* qkv;       /* packed */
  const float* qkv_bias;        /* [(n_q+2n_kv)*128] or NULL (Qwen3) */
  const float* q_norm;          /* [128] or NULL (Qwen2) */
  const float* k_norm;
  const chatts_bf16* o;
  const float* post_norm;       /* [H] */
  const chatts_bf16* gate_up;
  const chatts_bf16* down;
  /* optional fp8 copies (see ChattsLinearArgs.w8): NULL = stream bf16 in decode too */
  const uint8_t* qkv8; const float* qkv8_scale;
  const uint8_t* o8; const float* o8_scale;
  const uint8_t* gate_up8; const float* gate_up8_scale;
  const uint8_t* down8; const float* down8_scale;
  /* optional 4-bit copies (see ChattsLinearArgs.w4; row length K/2 bytes): NULL = stream bf16 / fp8 in decode */
  const uint8_t* qkv4; const float* qkv4_sz;
  const uint8_t* o4; const float* o4_sz;
  const uint8_t* gate_up4; const float* gate_up4_sz;
  const uint8_t* down4; const float* down4_sz;
  int w4_group;
  /* optional copies of the four bf16 matrices in the tiled layout of chatts_tile_bf16 (ChattsLinearArgs.w_tiled): the operand feed of the
   * prefill kernel reads them as consecutive memory; every other kernel keeps streaming the row-major tensors.  NULL = row-major only. */
  const chatts_bf16* qkv_t; const chatts_bf16* o_t; const chatts_bf16* gate_up_t; const chatts_bf16* down_t;
  /* optional f16q copies of the four matrices, TILED (chatts_weights_f16q, then chatts_tile_bf16 on the f16 copy and chatts_tile_e4m3 on the
   * e4m3 copy; e8m0 row exponent [N]): the operands of chatts_decoder_set_prefill_f16q.  NULL otherwise. */
  const chatts_f16* qkv16; const uint8_t* qkv_q8; const uint8_t* qkv_q8e;
  const chatts_f16* o16; const uint8_t* o_q8; const uint8_t* o_q8e;
  const chatts_f16* gate_up16; const uint8_t* gate_up_q8; const uint8_t* gate_up_q8e;
  const chatts_f16* down16; const uint8_t* down_q8; const uint8_t* down_q8e;
} ChattsLayerWeights;

typedef struct ChattsDecoderConfig {
  int hidden, n_layers, n_q, n_kv, head_dim, inter;   /* n_q, n_kv, inter are RANK-LOCAL sizes */
  int64_t vocab_local, vocab_offset;                  /* this rank's slice of lm_head / embed   */
  float rms_eps;
  int max_ctx, max_pos;
  int tp_world;                                       /* > 1: o/down outputs are partial sums   */
  int64_t embed_rows, embed_offset;                   /* rows of `embed` this rank holds and the id of row 0; 0 rows = the lm_head
                                                         slice (vocab_local, vocab_offset).  Replicated table: (vocab, 0)          */
  int w8_format;                                      /* encoding of the optional 8-bit weight copies (ChattsLinearArgs.w8_format)   */
} ChattsDecoderConfig;

typedef struct ChattsDecoderWeights {
  const ChattsLayerWeights* layers;  /* host array [n_layers] */
  const float* final_norm;           /* [H] */
  const chatts_bf16* lm_head;        /* [vocab_local, H] */
  const uint8_t* lm_head8;           /* optional fp8 copy + per-row scale */
  const float* lm_head8_scale;
  const chatts_bf16* embed;          /* [vocab_local, H] */
  const float* cos_tab;              /* [max_pos, 64] */
  const float* sin_tab;
} ChattsDecoderWeights;

typedef struct ChattsDecoderBuffers {
  float* kv_k;        /* [n_layers, n_kv, max_ctx, 128] per sequence (or block pools, see kv_block_table below) */
  float* kv_v;
  float* x;           /* [T_max, H] residual stream */
  float* xn;          /* [T_max, H] normed scratch (prefill) */
  float* qkv;         /* [T_max, (n_q+2n_kv)*128] */
  float* attn;        /* [T_max, n_q*128] */
  float* act;         /* [T_max, inter] */
  float* delta;       /* [T_max, H] partial sums when tp_world > 1 */
  float* logits;      /* [vocab_local] */
  void* workspace;    /* split-K + attention partials */
  size_t workspace_bytes;
  int t_max;
  int max_batch;      /* KV caches are [max_batch, n_layers, n_kv, max_ctx, 128]; 0 or 1 = single sequence */
  /* optional: two pairs of bf16 hi / lo planes, each plane [T_max, max(H, n_q*128, inter)] (ChattsLinearArgs.a_hi /
   * a_lo / c_hi / c_lo).  When all four are set, prefill chunks of >= 96 rows run the LDS-DMA GEMM: the RMSNorms and
   * the SwiGLU epilogue write planes directly (pair 0 = projection input, pair 1 = gate_up output). */
  chatts_bf16* planes_hi;
  chatts_bf16* planes_lo;
  chatts_bf16* planes2_hi;
  chatts_bf16* planes2_lo;
  /* tensor parallel only: scratch of the token agreement - local (max logit, token) per sequence, and (sampling) the
   * gathered full-vocabulary logits [max(max_batch,1), tp_world * vocab_local] */
  float* tp_pair_logit;    /* [max(max_batch,1)] */
  int64_t* tp_pair_token;  /* [max(max_batch,1)] */
  float* logits_full;      /* or NULL: sampling under TP is then refused */
  /* block-paged KV cache (optional; ChattsKvCache's paged form): when kv_block_table != NULL, kv_k / kv_v are block pools
   * [n_layers, kv_pool_blocks, n_kv, kv_block_size, 128] and sequence (cache slot) s reads its blocks from
   * kv_block_table[s * kv_table_stride ...] (int32, device memory, max_ctx / kv_block_size entries used; the caller keeps the
   * table current - a captured decode graph reads it at replay time). */
  const int32_t* kv_block_table;
  int kv_block_size;
  int kv_table_stride;
  int kv_pool_blocks;
} ChattsDecoderBuffers;

typedef struct ChattsDecoder ChattsDecoder;  /* opaque; host memory only */
ChattsDecoder* chatts_decoder_create(const ChattsDecoderConfig*, const ChattsDecoderWeights*,
                                     const ChattsDecoderBuffers*);
void chatts_decoder_destroy(ChattsDecoder*);
size_t chatts_decoder_workspace(const ChattsDecoderConfig*, int t_max, int n_splits_max);

/* One layer, split at the two tensor-parallel exchange points (vLLM: all-reduce after o_proj and
 * after down_proj).  part 0: x -> norm -> qkv -> rope/cache -> attention -> o_proj;
 *                    part 1: x -> norm -> gate_up/SwiGLU -> down_proj.
 * tp_world == 1: the projection adds into x in place.  tp_world > 1: it writes this rank's partial
 * sum to buffers.delta; the caller all-reduces delta (RCCL) and calls chatts_residual_add. */
int chatts_decoder_layer_part(ChattsDecoder*, int layer, int part, int t, int pos0,
                              const int32_t* pos0_dev, int n_splits, chatts_stream_t stream);
int chatts_residual_add(float* x, const float* delta, int64_t n, chatts_stream_t stream);
/* The same layer part preceded by x[0:t] += delta[0:t] (buffers.delta = the all-reduced partial sum of the PREVIOUS part):
 * one host call per exchange point instead of two.  add_delta == 0: identical to chatts_decoder_layer_part. */
int chatts_decoder_layer_part_add(ChattsDecoder*, int add_delta, int layer, int part, int t, int pos0,
                                  const int32_t* pos0_dev, int n_splits, chatts_stream_t stream);

/* SPEED MODE switch for the prefill chunks (t >= 16 rows) of this decoder: on != 0 runs their four projections as fp8 x fp8 GEMMs
 * (chatts_quantize_rows_fp8 + chatts_linear_fp8, RMSNorm fused into the quantisation) on the fp8 weight copies; needs every layer's
 * qkv8 / o8 / gate_up8 / down8 in e4m3 (w8_format FP8) and K multiples of 128.  NOT parity grade (~1e-2 on logits); decode steps,
 * attention, norms, KV cache and logits are unchanged.  Returns CHATTS_E_BADARG when the copies are missing. */
int chatts_decoder_set_prefill_fp8(ChattsDecoder*, int on);
/* OPT-IN, PARITY GRADE, SLOWER than the default: prefill chunks of >= 96 rows run their four projections on the f16q
 * operand split (chatts_linear_f16q: f16 high part on v_mfma_f32_16x16x32_f16 + e4m3 residual x e4m3 weights on the CDNA4 block-scaled
 * v_mfma_scale_f32_16x16x128_f8f6f4) instead of the bf16 hi / lo split - the fp8 matrix pipe at float32 grade (logits ~1.3e-4 of the
 * float32 oracle at full depth against ~5e-5 for the default; bar 1e-3).  Needs every layer's f16q weight copies, hidden / inter / n_q * 128
 * multiples of 128, the plane buffers as scratch (each holding ceil(t_max / 32) * 32 rows of max(hidden, inter, n_q * 128) bf16 elements).  RoPE, cache, attention, residual stream and all decode steps are unchanged. */
int chatts_decoder_set_prefill_f16q(ChattsDecoder* d, int on);

/* Attach the tensor-parallel exchange (tp_world > 1): chatts_decoder_decode_step(_batched) then run whole TP steps on the
 * stream - partial o_proj / down_proj sums are all-reduced into the residual stream by chatts_allreduce, tokens are agreed on
 * by chatts_tp_argmax (greedy) or by sampling from the chatts_allgather-ed logits with the shared seed - and stay
 * graph-capturable.  The comm must outlive the decoder; NULL detaches. */
int chatts_decoder_set_tp(ChattsDecoder*, ChattsTpComm*);
/* Token selection for `batch` rows of this rank's logits [batch, logits_stride] (greedy, or the sampler configured with
 * chatts_decoder_set_sampling / `override`), agreed across the TP ranks when a comm is attached.  Side effects as
 * chatts_argmax_batched. */
int chatts_decoder_select_tokens(ChattsDecoder*, const float* logits, int batch, int64_t logits_stride, int64_t* token,
                                 float* token_logit, int64_t* out_tokens, int64_t out_stride, int32_t* step_dev,
                                 int32_t* pos_dev, int pos_limit, const ChattsSamplingArgs* override_or_null,
                                 chatts_stream_t stream);

/* Token selection of chatts_decoder_decode_step(_batched): NULL (default) = greedy argmax; otherwise the sampler above
 * with these parameters (copied).  Changing it invalidates any hipGraph captured over a decode step. */
int chatts_decoder_set_sampling(ChattsDecoder*, const ChattsSamplingArgs* args_or_null);
/* KV-cache slot (sequence) used by the single-sequence entry points (layer_part, prefill, decode_step). */
int chatts_decoder_select_sequence(ChattsDecoder*, int seq);
/* Batched decode (continuous batching): `batch` sequences advance one token each; row b of
 * the activation buffers and pos_dev[b] belong to cache slot b.  Projections run as M = batch MFMA GEMMs. */
int chatts_decoder_layer_part_batched(ChattsDecoder*, int layer, int part, int batch, const int32_t* pos_dev,
                                      int n_splits, chatts_stream_t stream);
int chatts_decoder_decode_step_batched(ChattsDecoder*, int batch, int32_t* pos_dev, int32_t* step_dev,
                                       int64_t* token_dev, float* token_logit_dev, int64_t* out_tokens,
                                       int64_t out_stride, float* logits_all /* [batch, vocab_local] */, int n_splits,
                                       chatts_stream_t stream);

/* final norm + lm_head for rows 0 .. batch-1 of x -> logits_all [batch, vocab_local] (this rank's vocabulary slice): the tail of
 * chatts_decoder_decode_step_batched as a call of its own (compute_logits for a batch, chatts_vllm.py:603-610). */
int chatts_decoder_logits_batched(ChattsDecoder*, int batch, float* logits_all, chatts_stream_t stream);

/* All layers back to back on the stream (no host round trip, graph-capturable).  tp_world == 1, or tp_world > 1 with an exchange
 * attached (chatts_decoder_set_tp) whose bulk region holds t * hidden float32: the [t, H] sums between the layer halves are then
 * chatts_allreduce_bulk launches of the same call - one host call per prefill chunk on every rank. */
int chatts_decoder_prefill(ChattsDecoder*, int t, int pos0, chatts_stream_t stream);
/* The same for a chunk whose hidden states nobody reads except to pick the NEXT token (the last chunk of a prompt): the final
 * layer writes K / V of all t rows into the cache but runs attention, o_proj and the MLP for the last row only (as GEMVs).
 * The row the next token is computed from ends up in ROW 0 of x (chatts_decoder_logits(d, 0, ..)); rows 1.. are stale. */
int chatts_decoder_prefill_last(ChattsDecoder*, int t, int pos0, chatts_stream_t stream);
/* Packed multi-prompt prefill (the vLLM scheduler's batched prefill; demo/demo_vllm.py:55 submits 100 prompts at once): the rows of
 * n_segs prompts - or of the tails that prefix reuse left to compute - lie back to back in x (segment i = rows row0 .. row0 + t - 1,
 * positions pos0 .. of cache slot `slot`; row0 of segment 0 is 0, segments are contiguous, slots distinct).  Row-wise work runs
 * once over all rows, RoPE / cache write / attention per segment.  Afterwards x holds every segment's hidden states;
 * chatts_decoder_logits(d, row0 + t - 1, ..) gives a segment's next-token logits.  `segs` is a HOST array. */
typedef struct ChattsPrefillSegment { int row0, t, pos0, slot; } ChattsPrefillSegment;
int chatts_decoder_prefill_packed(ChattsDecoder*, const ChattsPrefillSegment* segs, int n_segs, chatts_stream_t stream);
/* final norm + lm_head on row `row` of x -> buffers.logits */
int chatts_decoder_logits(ChattsDecoder*, int row, chatts_stream_t stream);
/* one greedy decode step: x[0,:] holds the input embedding; reads the position from *pos_dev;
 * runs all layers, logits, argmax (-> out_tokens[*step_dev], step/pos incremented), then loads the
 * next input embedding into x[0,:].  Fully device-driven: capture once, replay per token. */
int chatts_decoder_decode_step(ChattsDecoder*, int32_t* pos_dev, int32_t* step_dev,
                               int64_t* token_dev, float* token_logit_dev, int64_t* out_tokens,
                               int n_splits, chatts_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CHATTS_AMD_H */
