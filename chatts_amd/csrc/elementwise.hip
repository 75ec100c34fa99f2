// elementwise.hip - the small HBM/L2-bound kernels around the projections:
// RMSNorm, RoPE + KV-cache write, embedding gather + TS merge, argmax, residual add.
#include "common.h"

namespace chatts {

// ---- RMSNorm: one workgroup per row, float4 loads, wave-shuffle + LDS reduction --------------------
// y = w * (x * rsqrt(mean(x^2) + eps))     (Qwen2RMSNorm.forward; same op order as the oracle)
// PLANES: the row goes out as bf16 hi / lo planes (hi = bf16(y), lo = bf16(y - hi)), the LDS-DMA GEMM's operand format.
template <bool PLANES>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     float* __restrict__ y, int hidden, float eps,
                                                     uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int ldp) {
  __shared__ float red[4];
  const float* xr = x + (size_t)blockIdx.x * hidden;
  float* yr = PLANES ? nullptr : y + (size_t)blockIdx.x * hidden;
  float ss = 0.f;
  for (int k = threadIdx.x * 4; k < hidden; k += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + k);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = block_sum<4>(ss, red);
  const float rstd = rsqrtf(ss / (float)hidden + eps);
  for (int k = threadIdx.x * 4; k < hidden; k += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + k);
    const f32x4 g = *reinterpret_cast<const f32x4*>(w + k);
    f32x4 o;
    o.x = g.x * (v.x * rstd); o.y = g.y * (v.y * rstd); o.z = g.z * (v.z * rstd); o.w = g.w * (v.w * rstd);
    if (PLANES) {
      // no contraction: o is a product, and folding it into the subtraction as an FMA would make lo differ from the split of
      // the ROUNDED float32 value that the float32 variant stores
#pragma clang fp contract(off)
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      bf16x4_t hv, lv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16 h = (__bf16)o[j];
        hv[j] = h;
        lv[j] = (__bf16)(o[j] - (float)h);
      }
      *reinterpret_cast<bf16x4_t*>(hi + (size_t)blockIdx.x * ldp + k) = hv;
      *reinterpret_cast<bf16x4_t*>(lo + (size_t)blockIdx.x * ldp + k) = lv;
    } else {
      *reinterpret_cast<f32x4*>(yr + k) = o;
    }
  }
}

// ---- RoPE + KV write: one wave per (token, head); lane l owns dims (l, l+64) = a rotate-half pair ----
// q/k heads: optional RMSNorm over the 128 dims (Qwen3 q_norm/k_norm), then
//   out[l]    = x[l]    * cos[l] - x[l+64] * sin[l]
//   out[l+64] = x[l+64] * cos[l] + x[l]    * sin[l]      (rotate_half / apply_rotary_pos_emb)
// k (rotated) and v rows go to the cache at position pos; q is rotated in place.
__global__ __launch_bounds__(256) void rope_kv_kernel(float* __restrict__ qkv, int t, RopeFuse r) {
  const int heads = r.n_q + 2 * r.n_kv;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (gw >= t * heads) return;
  const int tok = gw / heads, h = gw - tok * heads;
  const int pos = (r.pos0_dev ? *r.pos0_dev : r.pos0) + tok;
  float* row = qkv + ((size_t)tok * heads + h) * kHeadDim;
  rope_kv_head(row[lane], row[lane + 64], h, pos, lane, row, r);
}

// ---- embedding gather + TS merge -------------------------------------------------------------------
// pass 1 (one workgroup): rank[t] = number of placeholder ids before position t (exclusive scan).
__global__ __launch_bounds__(1024) void placeholder_scan_kernel(const int64_t* __restrict__ ids, int t,
                                                               int64_t ts_id, int32_t* __restrict__ rank,
                                                               int n_ts_rows, int32_t* __restrict__ status) {
  __shared__ int wsum[16];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < t; base += 1024) {
    const int i = base + threadIdx.x;
    const int f = (i < t && ids[i] == ts_id) ? 1 : 0;
    int incl = f;   // inclusive wave scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    if (i < t) rank[i] = carry + woff + incl - f;
    __syncthreads();
    if (threadIdx.x == 1023) carry += woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    rank[t] = carry;
    if (status && carry != n_ts_rows) atomicOr(status, 1);
  }
}

// pass 2: one workgroup per output row; float4 stores; bf16 table rows widened, TS rows copied.
__global__ __launch_bounds__(256) void embed_merge_kernel(const int64_t* __restrict__ ids, const uint16_t* __restrict__ table,
                                                         int64_t vocab, int hidden, const float* __restrict__ ts_rows,
                                                         int n_ts_rows, int64_t ts_id, const int32_t* __restrict__ rank,
                                                         float* __restrict__ out) {
  const int t = blockIdx.x;
  const int64_t id = ids[t];
  float* o = out + (size_t)t * hidden;
  if (ts_rows != nullptr && id == ts_id) {
    const int r = rank[t];
    if (r < n_ts_rows) {
      const float* src = ts_rows + (size_t)r * hidden;
      for (int k = threadIdx.x * 4; k < hidden; k += 1024)
        *reinterpret_cast<f32x4*>(o + k) = *reinterpret_cast<const f32x4*>(src + k);
      return;
    }
  }
  const int64_t cid = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const uint16_t* src = table + (size_t)cid * hidden;
  for (int k = threadIdx.x * 4; k < hidden; k += 1024) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(src + k);
    f32x4 f = {bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y)};
    *reinterpret_cast<f32x4*>(o + k) = f;
  }
}

__global__ __launch_bounds__(256) void embed_token_kernel(const int64_t* __restrict__ token, const uint16_t* __restrict__ table,
                                                         int64_t vocab_offset, int64_t vocab_rows, int hidden,
                                                         float* __restrict__ out) {
  const int seq = blockIdx.y;                       // batched decode: one token / output row per sequence
  const int64_t id = token[seq] - vocab_offset;
  const int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (k >= hidden) return;
  f32x4 f = {0.f, 0.f, 0.f, 0.f};
  if (id >= 0 && id < vocab_rows) {   // vocab-parallel: rows of other ranks contribute zeros (summed by all-reduce)
    const u32x2 v = *reinterpret_cast<const u32x2*>(table + (size_t)id * hidden + k);
    f = (f32x4){bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y)};
  }
  *reinterpret_cast<f32x4*>(out + (size_t)seq * hidden + k) = f;
}

// ---- argmax: one workgroup of 1024; first index of the maximum (torch.argmax tie rule) -------------
// One workgroup per sequence (blockIdx.x): batched decode keeps token / step / pos / out_tokens per sequence.
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits_all, int64_t logits_stride,
                                                     int64_t vocab, int64_t vocab_offset, int64_t* __restrict__ token_all,
                                                     float* __restrict__ token_logit_all,
                                                     int64_t* __restrict__ out_tokens_all, int64_t out_stride,
                                                     int32_t* __restrict__ step_all, int32_t* __restrict__ pos_all,
                                                     int pos_limit) {
  __shared__ float sv[16];
  __shared__ int64_t si[16];
  const int seq = blockIdx.x;
  const float* logits = logits_all + (size_t)seq * logits_stride;
  int64_t* token = token_all ? token_all + seq : nullptr;
  float* token_logit = token_logit_all ? token_logit_all + seq : nullptr;
  int64_t* out_tokens = out_tokens_all ? out_tokens_all + (size_t)seq * out_stride : nullptr;
  int32_t* step_dev = step_all ? step_all + seq : nullptr;
  int32_t* pos_dev = pos_all ? pos_all + seq : nullptr;
  float best;
  int64_t bi;
  block_argmax_first(logits, vocab, sv, si, best, bi);
  if (threadIdx.x == 0) {
    const int64_t tok = bi + vocab_offset;
    if (token) *token = tok;
    if (token_logit) *token_logit = best;
    if (out_tokens && step_dev && (out_stride == 0 || *step_dev < out_stride)) out_tokens[*step_dev] = tok;
    if (step_dev) *step_dev += 1;
    // a parked slot (pos < 0: inactive sequence of a batched step) stays parked; live ones saturate, never overflow
    if (pos_dev && *pos_dev >= 0 && (pos_limit <= 0 || *pos_dev < pos_limit)) *pos_dev += 1;
  }
}

// ---- the same selection in two launches: 64 workgroups per sequence scan 1/64 of the row each (one workgroup needs 24 us for the
// 152 k logits of a row - the tail of every decode step), one wave per sequence merges the 64 (value, first index) pairs and applies
// the side effects.  Maximum and first-index tie rule are order-free: the token is the one argmax_kernel picks.
constexpr int kArgmaxParts = 64;
__global__ __launch_bounds__(256) void argmax_part_kernel(const float* __restrict__ logits_all, int64_t logits_stride, int64_t vocab,
                                                         float* __restrict__ part_v, int64_t* __restrict__ part_i) {
  __shared__ float sv[4];
  __shared__ int64_t si[4];
  const int seq = blockIdx.y, blk = blockIdx.x;
  const float* logits = logits_all + (size_t)seq * logits_stride;
  const int64_t chunk = ((vocab + kArgmaxParts - 1) / kArgmaxParts + 3) / 4 * 4;
  const int64_t lo = blk * chunk, hi = lo + chunk < vocab ? lo + chunk : vocab;
  float best = -INFINITY;
  int64_t bi = 0x7fffffffffffffffLL;
  const bool vec = (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
  const int64_t body_end = vec ? lo + (hi > lo ? (hi - lo) / 4 * 4 : 0) : lo;
  for (int64_t i = lo + threadIdx.x * 4; i < body_end; i += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(logits + i);
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (e[j] > best) { best = e[j]; bi = i + j; }        // ascending i within a thread: '>' keeps the first
  }
  for (int64_t i = body_end + threadIdx.x; i < hi; i += 256) {
    const float v = logits[i];
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int64_t oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sv[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    part_v[seq * kArgmaxParts + blk] = best;
    part_i[seq * kArgmaxParts + blk] = bi;
  }
}

__global__ __launch_bounds__(64) void argmax_final_kernel(const float* __restrict__ part_v, const int64_t* __restrict__ part_i,
                                                         int64_t vocab_offset, int64_t* __restrict__ token_all,
                                                         float* __restrict__ token_logit_all, int64_t* __restrict__ out_tokens_all,
                                                         int64_t out_stride, int32_t* __restrict__ step_all,
                                                         int32_t* __restrict__ pos_all, int pos_limit) {
  const int seq = blockIdx.x, lane = threadIdx.x;
  float best = part_v[seq * kArgmaxParts + lane];
  int64_t bi = part_i[seq * kArgmaxParts + lane];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int64_t oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) {
    int64_t* token = token_all ? token_all + seq : nullptr;
    float* token_logit = token_logit_all ? token_logit_all + seq : nullptr;
    int64_t* out_tokens = out_tokens_all ? out_tokens_all + (size_t)seq * out_stride : nullptr;
    int32_t* step_dev = step_all ? step_all + seq : nullptr;
    int32_t* pos_dev = pos_all ? pos_all + seq : nullptr;
    const int64_t tok = bi + vocab_offset;
    if (token) *token = tok;
    if (token_logit) *token_logit = best;
    if (out_tokens && step_dev && (out_stride == 0 || *step_dev < out_stride)) out_tokens[*step_dev] = tok;
    if (step_dev) *step_dev += 1;
    if (pos_dev && *pos_dev >= 0 && (pos_limit <= 0 || *pos_dev < pos_limit)) *pos_dev += 1;
  }
}

// chatts_argmax_batched with `scratch` (>= argmax_scratch_bytes(batch)): the two-launch form; without: the single workgroup per row.
size_t argmax_scratch_bytes(int batch) { return (size_t)batch * kArgmaxParts * (sizeof(float) + sizeof(int64_t)) + 16; }
int argmax_batched_scratch(const float* logits, int batch, int64_t logits_stride, int64_t vocab, int64_t vocab_offset, int64_t* token,
                           float* token_logit, int64_t* out_tokens, int64_t out_stride, int32_t* step_dev, int32_t* pos_dev,
                           int pos_limit, void* scratch, size_t scratch_bytes, hipStream_t s) {
  const bool off = opt_get(OPT_ARGMAX_2STAGE, 1) == 0;
  if (!scratch || scratch_bytes < argmax_scratch_bytes(batch) || vocab < 4096 || off)
    return chatts_argmax_batched(logits, batch, logits_stride, vocab, vocab_offset, token, token_logit, out_tokens, out_stride, step_dev,
                                 pos_dev, pos_limit, reinterpret_cast<chatts_stream_t>(s));
  CHATTS_REQUIRE(logits && vocab > 0 && batch >= 1 && logits_stride >= vocab, CHATTS_E_BADARG, "argmax: bad arguments");
  int64_t* part_i = reinterpret_cast<int64_t*>((reinterpret_cast<uintptr_t>(scratch) + 15) & ~(uintptr_t)15);
  float* part_v = reinterpret_cast<float*>(part_i + (size_t)batch * kArgmaxParts);
  hipLaunchKernelGGL(argmax_part_kernel, dim3(kArgmaxParts, batch), dim3(256), 0, s, logits, logits_stride, vocab, part_v, part_i);
  CHATTS_CHECK_LAUNCH("argmax_part");
  hipLaunchKernelGGL(argmax_final_kernel, dim3(batch), dim3(64), 0, s, part_v, part_i, vocab_offset, token, token_logit, out_tokens,
                     out_stride, step_dev, pos_dev, pos_limit);
  CHATTS_CHECK_LAUNCH("argmax_final");
  return CHATTS_OK;
}

__global__ __launch_bounds__(256) void residual_add_kernel(float* __restrict__ x, const float* __restrict__ d, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 a = reinterpret_cast<f32x4*>(x)[i];
    const f32x4 b = reinterpret_cast<const f32x4*>(d)[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    reinterpret_cast<f32x4*>(x)[i] = a;
  }
}

}  // namespace chatts

using namespace chatts;

extern "C" int chatts_rmsnorm(const float* x, const float* w, float* y, int t, int hidden, float eps,
                              chatts_stream_t stream) {
  CHATTS_REQUIRE(t >= 0 && hidden > 0, CHATTS_E_BADARG, "rmsnorm: bad sizes t=%d hidden=%d", t, hidden);
  if (t == 0) return CHATTS_OK;
  CHATTS_REQUIRE(x && w && y, CHATTS_E_BADARG, "rmsnorm: null pointer");
  CHATTS_REQUIRE(hidden % 4 == 0, CHATTS_E_SHAPE, "rmsnorm: hidden %d not a multiple of 4", hidden);
  hipLaunchKernelGGL(rmsnorm_kernel<false>, dim3(t), dim3(256), 0, as_stream(stream), x, w, y, hidden, eps, nullptr, nullptr, 0);
  CHATTS_CHECK_LAUNCH("rmsnorm");
  return CHATTS_OK;
}

extern "C" int chatts_rmsnorm_planes(const float* x, const float* w, chatts_bf16* hi, chatts_bf16* lo, int ld_planes, int t,
                                     int hidden, float eps, chatts_stream_t stream) {
  CHATTS_REQUIRE(t >= 0 && hidden > 0, CHATTS_E_BADARG, "rmsnorm_planes: bad sizes t=%d hidden=%d", t, hidden);
  if (t == 0) return CHATTS_OK;
  CHATTS_REQUIRE(x && w && hi && lo, CHATTS_E_BADARG, "rmsnorm_planes: null pointer");
  CHATTS_REQUIRE(hidden % 4 == 0 && ld_planes >= hidden && ld_planes % 4 == 0, CHATTS_E_SHAPE,
                 "rmsnorm_planes: hidden %d / ld_planes %d must be multiples of 4, ld_planes >= hidden", hidden, ld_planes);
  hipLaunchKernelGGL(rmsnorm_kernel<true>, dim3(t), dim3(256), 0, as_stream(stream), x, w, nullptr, hidden, eps, hi, lo,
                     ld_planes);
  CHATTS_CHECK_LAUNCH("rmsnorm_planes");
  return CHATTS_OK;
}

namespace chatts {
int rope_fuse_prepare(int t, int n_q, int n_kv, const float* q_norm_w, const float* k_norm_w, float norm_eps, const float* cos_tab,
                      const float* sin_tab, int pos0, const int32_t* pos0_dev, const ChattsKvCache* cache, RopeFuse* out) {
  CHATTS_REQUIRE(t >= 0 && n_q > 0 && n_kv > 0, CHATTS_E_BADARG, "rope_kv_write: bad sizes");
  CHATTS_REQUIRE(cos_tab && sin_tab && cache && cache->k && cache->v, CHATTS_E_BADARG, "rope_kv_write: null pointer");
  CHATTS_REQUIRE((q_norm_w == nullptr) == (k_norm_w == nullptr), CHATTS_E_BADARG,
                 "rope_kv_write: q_norm and k_norm must both be set or both be null");
  if (!pos0_dev)
    CHATTS_REQUIRE(pos0 >= 0 && pos0 + t <= cache->max_ctx, CHATTS_E_SHAPE,
                   "rope_kv_write: positions %d..%d exceed the cache (%d)", pos0, pos0 + t, cache->max_ctx);
  KvLayout kvl{cache->block_table, n_kv, cache->max_ctx, 0};
  if (cache->block_table) {
    kvl.log_block = kv_log_block(cache->block_size);
    CHATTS_REQUIRE(kvl.log_block > 0 && cache->max_ctx % cache->block_size == 0, CHATTS_E_BADARG,
                   "rope_kv_write: block_size %d must be a power of two in 64..32768 that divides max_ctx %d", cache->block_size,
                   cache->max_ctx);
  }
  *out = RopeFuse{n_q, n_kv, q_norm_w, k_norm_w, norm_eps, cos_tab, sin_tab, pos0, pos0_dev, cache->k, cache->v, kvl, kv_round_mode()};
  return CHATTS_OK;
}
}  // namespace chatts

extern "C" int chatts_rope_kv_write(float* qkv, int t, int n_q, int n_kv, const float* q_norm_w,
                                    const float* k_norm_w, float norm_eps, const float* cos_tab,
                                    const float* sin_tab, int pos0, const int32_t* pos0_dev,
                                    const ChattsKvCache* cache, chatts_stream_t stream) {
  CHATTS_REQUIRE(t >= 0 && n_q > 0 && n_kv > 0, CHATTS_E_BADARG, "rope_kv_write: bad sizes");
  if (t == 0) return CHATTS_OK;
  CHATTS_REQUIRE(qkv != nullptr, CHATTS_E_BADARG, "rope_kv_write: null pointer");
  RopeFuse r;
  if (const int rc = rope_fuse_prepare(t, n_q, n_kv, q_norm_w, k_norm_w, norm_eps, cos_tab, sin_tab, pos0, pos0_dev, cache, &r)) return rc;
  const int waves = t * (n_q + 2 * n_kv);
  hipLaunchKernelGGL(rope_kv_kernel, dim3((waves + 3) / 4), dim3(256), 0, as_stream(stream), qkv, t, r);
  CHATTS_CHECK_LAUNCH("rope_kv_write");
  return CHATTS_OK;
}

extern "C" int chatts_embed_merge(const int64_t* ids_dev, const int64_t* ids_host, int t, const chatts_bf16* table,
                                  int64_t vocab, int hidden, const float* ts_rows, int n_ts_rows,
                                  int64_t ts_token_id, float* out, int32_t* scratch, int32_t* dev_status,
                                  chatts_stream_t stream) {
  CHATTS_REQUIRE(t >= 0 && hidden > 0 && vocab > 0 && n_ts_rows >= 0, CHATTS_E_BADARG, "embed_merge: bad sizes");
  CHATTS_REQUIRE(hidden % 4 == 0, CHATTS_E_SHAPE, "embed_merge: hidden %d not a multiple of 4", hidden);
  if (ids_host) {   // the reference raises ValueError here (vLLM merge_multimodal_embeddings)
    int cnt = 0;
    for (int i = 0; i < t; ++i) cnt += ids_host[i] == ts_token_id;
    if (ts_rows != nullptr || n_ts_rows > 0)
      CHATTS_REQUIRE(cnt == n_ts_rows, CHATTS_E_COUNT_MISMATCH,
                     "Attempted to assign %d multimodal tokens to %d placeholders", n_ts_rows, cnt);
  }
  if (t == 0) return CHATTS_OK;
  CHATTS_REQUIRE(ids_dev && table && out, CHATTS_E_BADARG, "embed_merge: null pointer");
  const bool merge = ts_rows != nullptr && n_ts_rows > 0;
  if (merge) {
    CHATTS_REQUIRE(scratch != nullptr, CHATTS_E_WORKSPACE, "embed_merge: scratch [t+1] int32 required");
    hipLaunchKernelGGL(placeholder_scan_kernel, dim3(1), dim3(1024), 0, as_stream(stream), ids_dev, t, ts_token_id,
                       scratch, n_ts_rows, dev_status);
    CHATTS_CHECK_LAUNCH("placeholder_scan");
  }
  hipLaunchKernelGGL(embed_merge_kernel, dim3(t), dim3(256), 0, as_stream(stream), ids_dev, table, vocab, hidden,
                     merge ? ts_rows : nullptr, n_ts_rows, ts_token_id, scratch, out);
  CHATTS_CHECK_LAUNCH("embed_merge");
  return CHATTS_OK;
}

extern "C" int chatts_argmax_batched(const float* logits, int batch, int64_t logits_stride, int64_t vocab,
                                     int64_t vocab_offset, int64_t* token, float* token_logit, int64_t* out_tokens,
                                     int64_t out_stride, int32_t* step_dev, int32_t* pos_dev, int pos_limit,
                                     chatts_stream_t stream) {
  CHATTS_REQUIRE(logits && vocab > 0 && batch >= 1 && logits_stride >= vocab, CHATTS_E_BADARG, "argmax: bad arguments");
  hipLaunchKernelGGL(argmax_kernel, dim3(batch), dim3(1024), 0, as_stream(stream), logits, logits_stride, vocab,
                     vocab_offset, token, token_logit, out_tokens, out_stride, step_dev, pos_dev, pos_limit);
  CHATTS_CHECK_LAUNCH("argmax");
  return CHATTS_OK;
}

extern "C" size_t chatts_argmax_workspace(int batch) { return chatts::argmax_scratch_bytes(batch); }
extern "C" int chatts_argmax_batched_ws(const float* logits, int batch, int64_t logits_stride, int64_t vocab, int64_t vocab_offset,
                                        int64_t* token, float* token_logit, int64_t* out_tokens, int64_t out_stride, int32_t* step_dev,
                                        int32_t* pos_dev, int pos_limit, void* workspace, size_t workspace_bytes, chatts_stream_t stream) {
  return chatts::argmax_batched_scratch(logits, batch, logits_stride, vocab, vocab_offset, token, token_logit, out_tokens, out_stride,
                                        step_dev, pos_dev, pos_limit, workspace, workspace_bytes, as_stream(stream));
}

extern "C" int chatts_argmax(const float* logits, int64_t vocab, int64_t vocab_offset, int64_t* token,
                             float* token_logit, int64_t* out_tokens, int32_t* step_dev, int32_t* pos_dev,
                             chatts_stream_t stream) {
  return chatts_argmax_batched(logits, 1, vocab, vocab, vocab_offset, token, token_logit, out_tokens, 0, step_dev, pos_dev,
                               0, stream);
}

extern "C" int chatts_embed_token_batched(const int64_t* token_dev, int batch, const chatts_bf16* table,
                                          int64_t vocab_offset, int64_t vocab_rows, int hidden, float* out,
                                          chatts_stream_t stream) {
  CHATTS_REQUIRE(token_dev && table && out && hidden > 0 && hidden % 4 == 0 && batch >= 1, CHATTS_E_BADARG,
                 "embed_token: bad arguments");
  hipLaunchKernelGGL(embed_token_kernel, dim3((hidden / 4 + 255) / 256, batch), dim3(256), 0, as_stream(stream), token_dev,
                     table, vocab_offset, vocab_rows, hidden, out);
  CHATTS_CHECK_LAUNCH("embed_token");
  return CHATTS_OK;
}

extern "C" int chatts_embed_token(const int64_t* token_dev, const chatts_bf16* table, int64_t vocab_offset,
                                  int64_t vocab_rows, int hidden, float* out, chatts_stream_t stream) {
  return chatts_embed_token_batched(token_dev, 1, table, vocab_offset, vocab_rows, hidden, out, stream);
}

extern "C" int chatts_residual_add(float* x, const float* delta, int64_t n, chatts_stream_t stream) {
  CHATTS_REQUIRE(x && delta && n >= 0 && n % 4 == 0, CHATTS_E_BADARG, "residual_add: bad arguments");
  if (n == 0) return CHATTS_OK;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(residual_add_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, delta, n / 4);
  CHATTS_CHECK_LAUNCH("residual_add");
  return CHATTS_OK;
}
