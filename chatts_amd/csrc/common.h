// common.h - shared device helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/chatts_amd.h"

namespace chatts {

void set_error(const char* fmt, ...);

#define CHATTS_REQUIRE(cond, code, ...)        \
  do {                                         \
    if (!(cond)) {                             \
      ::chatts::set_error(__VA_ARGS__);        \
      return (code);                           \
    }                                          \
  } while (0)

#define CHATTS_CHECK_LAUNCH(name)                                                  \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      ::chatts::set_error("%s: launch failed: %s", (name), hipGetErrorString(e_)); \
      return CHATTS_E_LAUNCH;                                                      \
    }                                                                              \
  } while (0)

constexpr int kWave = 64;
constexpr int kHeadDim = 128;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// round-to-nearest-even float32 -> bfloat16 bits (finite inputs)
__device__ __host__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  union { float f; uint32_t u; } v;
  v.f = f;
  uint32_t u = v.u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// x = hi + lo (+ eps, |eps| <= 2^-17 |x|): the bf16x2 split of an f32 activation
__device__ __forceinline__ void split_bf16x2(float x, uint16_t& hi, uint16_t& lo) {
  hi = f32_to_bf16_rne(x);
  lo = f32_to_bf16_rne(x - bf16_to_f32(hi));
}

// Cross-lane moves on the VALU (DPP) instead of the LDS crossbar (__shfl_xor lowers to ds_bpermute_b32: an LDS
// instruction and an lgkmcnt wait per step).  quad_perm covers lane ^ 1 and lane ^ 2, row_ror:4 / row_ror:8 rotate within a
// row of 16 lanes: after the four steps every lane of the row holds the row's reduction.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float lane_xor1(float v) { return dpp_mov<0xB1>(v); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor2(float v) { return dpp_mov<0x4E>(v); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ float row_ror4(float v) { return dpp_mov<0x124>(v); }
__device__ __forceinline__ float row_ror8(float v) { return dpp_mov<0x128>(v); }
__device__ __forceinline__ float half_mirror(float v) { return dpp_mov<0x141>(v); }  // row_half_mirror: lane i <-> 7 - i within 8 lanes
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, lane_xor1(v)); v = fmaxf(v, lane_xor2(v)); v = fmaxf(v, row_ror4(v)); v = fmaxf(v, row_ror8(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {     // ((v0+v1)+(v2+v3)) per quad, then quads 0+1.. : fixed order
  v += lane_xor1(v); v += lane_xor2(v); v += row_ror4(v); v += row_ror8(v);
  return v;
}

// Whole-wave reductions of a value that is already constant within quads or rows: the 16-lane rows are reduced with DPP,
// the four row results are fetched with v_readlane (wave-uniform) and combined in a fixed order.
__device__ __forceinline__ float readlane_bits(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float rows4_max(float v) {    // v: per-row value -> max over the 4 rows, in every lane
  return fmaxf(fmaxf(readlane_bits(v, 0), readlane_bits(v, 16)), fmaxf(readlane_bits(v, 32), readlane_bits(v, 48)));
}
__device__ __forceinline__ float rows4_sum(float v) {
  return (readlane_bits(v, 0) + readlane_bits(v, 16)) + (readlane_bits(v, 32) + readlane_bits(v, 48));
}

__device__ __forceinline__ float wave_sum(float v) { return rows4_sum(row16_sum(v)); }   // every lane gets the total
__device__ __forceinline__ float wave_max(float v) { return rows4_max(row16_max(v)); }
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blockDim.x == 64*NW; scratch must hold NW floats; all threads get the result
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += scratch[i];
  __syncthreads();
  return t;
}

// ---- KV cache addressing ------------------------------------------------------------------------------------------------
// Contiguous form: one sequence's layer cache is [n_kv, max_ctx, 128].  Block-paged form (table != nullptr,
// ChattsKvCache.block_table): the layer's pool is [n_blocks, n_kv, 2^log_block, 128] and logical key j of the sequence lives
// in block table[j >> log_block], row j & (2^log_block - 1).  Every kernel walks the keys in tiles of 16 / 32 / 64 consecutive
// keys that start at a multiple of the tile size, and a block holds a multiple of 64 keys: ONE lookup per tile gives the
// tile's first row, the rows of a tile are contiguous in both forms.
struct KvLayout {
  const int32_t* table;
  int n_kv, max_ctx, log_block;
};
// float offset (from the cache / pool base) of row `key` of kv head `hk`
__device__ __forceinline__ size_t kv_tile_off(const KvLayout& L, int hk, int key) {
  if (L.table == nullptr) return ((size_t)hk * L.max_ctx + key) * kHeadDim;
  const int blk = L.table[key >> L.log_block];
  return ((((size_t)blk * L.n_kv + hk) << L.log_block) + (key & ((1 << L.log_block) - 1))) * kHeadDim;
}
// host: log2 of a valid block size (power of two, 64 .. 32768), -1 otherwise
inline int kv_log_block(int block_size) {
  for (int l = 6; l <= 15; ++l)
    if (block_size == (1 << l)) return l;
  return -1;
}

// What rope_kv_kernel does to the raw qkv rows of a prefill chunk: per head optional RMSNorm (Qwen3 q_norm / k_norm), rotation by the
// token's position, q back in place, K / V rows into the cache.  Also carried by the qkv projection's split-K epilogue
// (splitk_epilogue_rope_kernel): ONE definition of the arithmetic, contraction off, so both forms write the same bits.
struct RopeFuse {
  int n_q, n_kv;
  const float* q_norm_w;
  const float* k_norm_w;
  float eps;
  const float* cos_tab;
  const float* sin_tab;
  int pos0;
  const int32_t* pos0_dev;
  float* kc;
  float* vc;
  KvLayout kvl;
  int kv_round;
};
// one 128-wide head of one token, held by a wave as (a, b) = elements (lane, lane + 64); h = head index in [q | k | v] order;
// `row` = this head's 128 floats of the qkv buffer (q is written back there)
__device__ __forceinline__ void rope_kv_head(float a, float b, int h, int pos, int lane, float* row, const RopeFuse& r);

inline hipStream_t as_stream(chatts_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- stage ranges for traces (api.hip): roctx push / pop around the host-side enqueue of a stage, resolved with dlopen at first use
// (librocprofiler-sdk-roctx.so; absent library = no-ops), so that `rocprofv3 --marker-trace --kernel-trace` reads by stage - ts_encode,
// prefill, layer.attn / layer.mlp, decode_step, logits, select_tokens - instead of by kernel name.  Ranges mark where work is ENQUEUED;
// a captured decode step shows its range at capture time only.
void stage_push(const char* name);
void stage_pop();
struct StageRange {
  explicit StageRange(const char* name) { stage_push(name); }
  ~StageRange() { stage_pop(); }
  StageRange(const StageRange&) = delete;
  StageRange& operator=(const StageRange&) = delete;
};

// ---- tuning options (chatts_set_option, api.hip) ---------------------------------------------------------------------------------
// A fixed table of named integers, process-wide, all UNSET by default = the shipped, measured-best choice.  The entry points read
// the table (one relaxed load); nothing in the library reads the environment.  The host sets an option explicitly (A/B runs, the
// bit-identity tests of alternative paths, geometry sweeps): chatts_amd._lib.set_option / _lib.options(...).
#define CHATTS_OPTIONS(X)                                                                                                        \
  X(GEMM_SK) X(GEMM_T) X(GEMM_BM) X(GEMM_PRECISION) X(GEMM_PLANES_MIN_M) X(GEMM_STREAM) X(GEMM_STREAM_STAGES) X(GEMM_STREAM_WAVES)   \
  X(GEMM_STREAM_MB) X(GEMM_STREAM_MB_WAVES) X(EPI_V4) X(EPI_NORM_Q) X(EPI_NORM_REG) X(POST_NORM_SMALL_M) X(ROPE_FUSE) X(ATTN_BF16X3) X(ATTN_EXACT) X(EPI_NORM_Q_GROUPS)  \
  X(ATTN_PLANES) X(ATTN_XCD) X(ATTN_ROWS) X(ATTN_KSPLIT) X(ARGMAX_2STAGE) X(TS_F32_PATH) X(KV_ROUND) X(TP_FUSE) X(TP_FUSE_BLOCKS)       \
  X(TP_BULK_BLOCKS) X(TP_BULK_THREADS) X(TP_BULK_FENCE) X(TP_AR_BLOCKS) X(GEMV_ROWS) X(GEMV_UNR) X(GEMV_NW) X(GEMV_OCC) X(GEMV_BLOCKS) X(GEMV_LDSPAD) X(GEMV_KS) X(FP8_BM) \
  X(FP8_ORDER) X(GEMV8_ROWS) X(GEMV8_UNR) X(GEMV8_NW) X(GEMV8_OCC) X(TS_L0_FUSED)
enum ChattsOpt {
#define CHATTS_OPT_ENUM(name) OPT_##name,
  CHATTS_OPTIONS(CHATTS_OPT_ENUM)
#undef CHATTS_OPT_ENUM
  OPT_COUNT
};
int opt_get(ChattsOpt o, int dflt);      // the option's value, or dflt while it is unset

int device_cus();

// Maximum of a row with torch.argmax's tie rule (first index), by a workgroup of 1024 threads; sv / si: 16-entry shared scratch.
// The result is valid in thread 0 (argmax_kernel, and the greedy rows of sample_kernel's per-row mode).
__device__ __forceinline__ void block_argmax_first(const float* __restrict__ logits, int64_t vocab, float* sv, int64_t* si, float& best,
                                                   int64_t& bi) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  best = -INFINITY;
  bi = 0x7fffffffffffffffLL;
  const int64_t v4 = ((reinterpret_cast<uintptr_t>(logits) & 15) == 0) ? vocab / 4 : 0;   // float4 body, scalar tail
  for (int64_t q = threadIdx.x; q < v4; q += 1024) {
    const f32x4_t v = reinterpret_cast<const f32x4_t*>(logits)[q];
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (e[j] > best) { best = e[j]; bi = q * 4 + j; }      // ascending i within a thread: '>' keeps the first
  }
  for (int64_t i = v4 * 4 + threadIdx.x; i < vocab; i += 1024) {
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
    for (int w = 1; w < 16; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
  }
}

// greedy selection with scratch for the two-launch form (elementwise.hip); falls back to chatts_argmax_batched without it
size_t argmax_scratch_bytes(int batch);
int argmax_batched_scratch(const float* logits, int batch, int64_t logits_stride, int64_t vocab, int64_t vocab_offset, int64_t* token,
                           float* token_logit, int64_t* out_tokens, int64_t out_stride, int32_t* step_dev, int32_t* pos_dev,
                           int pos_limit, void* scratch, size_t scratch_bytes, hipStream_t s);

// EXPERIMENT knob (CHATTS_KV_ROUND, default 0 = off): K / V rows are rounded to a narrower format's VALUES as they enter the float32
// cache - 1 = bf16, 2 = bf16 hi + lo (16 mantissa bits), 3 = fp16 - to measure what such a cache would cost in logits error against
// the 1e-3 bar before any kernel is rewritten for it.  The cache layout and every kernel stay float32.
__device__ __forceinline__ float kv_round_f(float v, int mode) {
  if (mode == 1) return (float)(__bf16)v;
  if (mode == 2) { const __bf16 h = (__bf16)v; return (float)h + (float)(__bf16)(v - (float)h); }
  if (mode == 3) return (float)(_Float16)v;
  return v;
}
inline int kv_round_mode() { return opt_get(OPT_KV_ROUND, 0); }

__device__ __forceinline__ void rope_kv_head(float a, float b, int h, int pos, int lane, float* row, const RopeFuse& r) {
#pragma clang fp contract(off)
  if (h >= r.n_q + r.n_kv) {   // v head: straight copy into the cache
    float* dst = r.vc + kv_tile_off(r.kvl, h - r.n_q - r.n_kv, pos);
    dst[lane] = kv_round_f(a, r.kv_round);
    dst[lane + 64] = kv_round_f(b, r.kv_round);
    return;
  }
  const float* nw = h < r.n_q ? r.q_norm_w : r.k_norm_w;
  if (nw) {
    const float ss = wave_sum(a * a + b * b);
    const float rstd = rsqrtf(ss / (float)kHeadDim + r.eps);
    a = nw[lane] * (a * rstd);
    b = nw[lane + 64] * (b * rstd);
  }
  const float c = r.cos_tab[(size_t)pos * 64 + lane], s = r.sin_tab[(size_t)pos * 64 + lane];
  const float oa = a * c - b * s, ob = b * c + a * s;
  if (h < r.n_q) {
    row[lane] = oa;
    row[lane + 64] = ob;
  } else {
    float* dst = r.kc + kv_tile_off(r.kvl, h - r.n_q, pos);
    dst[lane] = kv_round_f(oa, r.kv_round);
    dst[lane + 64] = kv_round_f(ob, r.kv_round);
  }
}
// host: validate the arguments of chatts_rope_kv_write and fill a RopeFuse (elementwise.hip)
int rope_fuse_prepare(int t, int n_q, int n_kv, const float* q_norm_w, const float* k_norm_w, float norm_eps, const float* cos_tab,
                      const float* sin_tab, int pos0, const int32_t* pos0_dev, const ChattsKvCache* cache, RopeFuse* out);

}  // namespace chatts
