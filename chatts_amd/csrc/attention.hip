// attention.hip - causal grouped-query attention over the float32 KV cache (head_dim 128).
// Math per transformers eager_attention_forward (oracle/qwen_decoder.py): scores = q.k^T * d^-1/2,
// float32 softmax, out = P.V; GQA: the n_q/n_kv query heads of a group share one K/V head (repeat_kv).
//
// One workgroup = (kv head, query row, key split): the group's G query heads are processed together
// so each K/V tile is read once per group.  Keys are walked in tiles of 64 with an online softmax:
//   phase 1  scores: 4 lanes per key (32 dims each, 64-B contiguous per 4 lanes), xor-shuffle reduce
//   phase 2  one wave per head: tile max / rescale factor / p = exp(s - m) written back to LDS
//   phase 3  P.V: thread = (dim, key half), V rows read fully coalesced (512 B per 128 lanes),
//            8 independent loads in flight per thread (addresses clamped, masked keys carry p = 0)
// Decode (T = 1) runs n_splits > 1 workgroups per kv head to put more CUs on the cache stream and a
// second kernel merges the (m, l, o) partials.  In the FUSED decode form the kernel also applies the
// per-head q/k RMSNorm (Qwen3) + RoPE to the raw projections and writes the new K/V row into the cache
// (replacing a separate launch): every workgroup rotates the group's q itself; the one workgroup whose
// tile contains `pos` rotates k, stores k/v to the cache and uses them from LDS.
// The KV cache is float32 on purpose: the parity target is the float32 reference path and the cache
// is ~1% of decode traffic at the benchmark context.
#include <math.h>

#include "common.h"

namespace chatts {

constexpr int kMaxGroup = 8;
constexpr int kTile = 64;

struct AttnParams {
  const float* qkv;   // [T, (n_q + 2 n_kv) * 128]; q already rotated unless fused
  float* kc;          // [n_kv, max_ctx, 128]
  float* vc;
  float* out;         // [T, n_q * 128]
  float* part_ml;     // [T, n_q, n_splits, 2]
  float* part_o;      // [T, n_q, n_splits, 128]
  const int32_t* pos0_dev;
  int pos0, t, n_q, n_kv, max_ctx, n_splits;
  // fused decode only
  const float* q_norm_w;
  const float* k_norm_w;
  const float* cos_tab;
  const float* sin_tab;
  float eps;
};

// rotate one 128-wide head held as (a = x[lane], b = x[lane+64]) by the wave; optional RMSNorm first
__device__ __forceinline__ void norm_rope(float& a, float& b, const float* nw, float eps, float c, float s, int lane) {
  if (nw) {
    const float ss = wave_sum(a * a + b * b);
    const float rstd = rsqrtf(ss / (float)kHeadDim + eps);
    a = nw[lane] * (a * rstd);
    b = nw[lane + 64] * (b * rstd);
  }
  const float oa = a * c - b * s, ob = b * c + a * s;
  a = oa;
  b = ob;
}

template <bool FUSED>
__global__ __launch_bounds__(256) void attn_rows_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) float q_s[kMaxGroup * kHeadDim];
  __shared__ __attribute__((aligned(16))) float knew_s[kHeadDim];
  __shared__ float vnew_s[kHeadDim];
  __shared__ float s_s[kMaxGroup * kTile];
  __shared__ float alpha_s[kMaxGroup];
  __shared__ float o_s[kMaxGroup * kHeadDim];

  const int hk = blockIdx.x, row = blockIdx.y, split = blockIdx.z;
  const int G = p.n_q / p.n_kv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pos = (p.pos0_dev ? *p.pos0_dev : p.pos0) + row;
  const int heads = p.n_q + 2 * p.n_kv;
  const float scale = 0.08838834764831845f;   // 128^-1/2
  const int ntiles = pos / kTile + 1;
  if (split >= ntiles) {                       // nothing to do for this split (uniform exit, before any barrier)
    if (p.n_splits > 1 && tid < G) {
      const size_t pi = ((size_t)row * p.n_q + hk * G + tid) * p.n_splits + split;
      p.part_ml[pi * 2] = -INFINITY;
      p.part_ml[pi * 2 + 1] = 0.f;
    }
    return;
  }
  const bool owner = FUSED && ((pos / kTile) % p.n_splits) == split;

  if (FUSED) {
    const float c = p.cos_tab[(size_t)pos * 64 + lane], s = p.sin_tab[(size_t)pos * 64 + lane];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int g = wave + gi * 4;
      if (g < G) {
        const float* src = p.qkv + ((size_t)row * heads + hk * G + g) * kHeadDim;
        float a = src[lane], b = src[lane + 64];
        norm_rope(a, b, p.q_norm_w, p.eps, c, s, lane);
        q_s[g * kHeadDim + lane] = a;
        q_s[g * kHeadDim + lane + 64] = b;
      }
    }
    if (owner && wave == 3) {                  // new K row: norm + rope, to the cache and to LDS
      const float* src = p.qkv + ((size_t)row * heads + p.n_q + hk) * kHeadDim;
      float a = src[lane], b = src[lane + 64];
      norm_rope(a, b, p.k_norm_w, p.eps, c, s, lane);
      knew_s[lane] = a;
      knew_s[lane + 64] = b;
      float* dst = p.kc + ((size_t)hk * p.max_ctx + pos) * kHeadDim;
      dst[lane] = a;
      dst[lane + 64] = b;
    }
    if (owner && wave == 2) {                  // new V row: straight copy
      const float* src = p.qkv + ((size_t)row * heads + p.n_q + p.n_kv + hk) * kHeadDim;
      const float a = src[lane], b = src[lane + 64];
      vnew_s[lane] = a;
      vnew_s[lane + 64] = b;
      float* dst = p.vc + ((size_t)hk * p.max_ctx + pos) * kHeadDim;
      dst[lane] = a;
      dst[lane + 64] = b;
    }
  } else {
    for (int i = tid; i < G * kHeadDim; i += 256)
      q_s[i] = p.qkv[((size_t)row * heads + hk * G) * kHeadDim + i];
  }
  __syncthreads();

  const float* kbase = p.kc + (size_t)hk * p.max_ctx * kHeadDim;
  const float* vbase = p.vc + (size_t)hk * p.max_ctx * kHeadDim;

  // per-head running statistics live in the wave that owns the head in phase 2 (heads wave, wave+4)
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  float acc[kMaxGroup];
#pragma unroll
  for (int g = 0; g < kMaxGroup; ++g) acc[g] = 0.f;

  const int key_l = tid >> 2, quarter = tid & 3;   // phase 1 mapping
  const int d_o = tid & 127, half = tid >> 7;      // phase 3 mapping

  for (int tile = split; tile < ntiles; tile += p.n_splits) {
    const int j0 = tile * kTile;
    // V of this thread's (dim, key half): all 32 loads are issued now and land while the scores are computed
    float vv[32];
    {
      const int jb = j0 + half * 32;
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const int j = jb + u;
        const int jc = j <= pos ? j : pos;       // clamped address; masked keys carry p = 0
        vv[u] = vbase[(size_t)jc * kHeadDim + d_o];
      }
    }
    // ---- phase 1: scores --------------------------------------------------------------------
    {
      const int j = j0 + key_l;
      const int jc = j <= pos ? j : pos;           // clamped: loads are unconditional, masked below
      const bool from_lds = owner && jc == pos;
      const float* kr = kbase + (size_t)jc * kHeadDim + quarter * 4;
      f32x4 kv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) kv[i] = *reinterpret_cast<const f32x4*>(kr + i * 16);
      if (from_lds) {
#pragma unroll
        for (int i = 0; i < 8; ++i) kv[i] = *reinterpret_cast<const f32x4*>(knew_s + quarter * 4 + i * 16);
      }
      float dot[kMaxGroup];
#pragma unroll
      for (int g = 0; g < kMaxGroup; ++g) dot[g] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int g = 0; g < kMaxGroup; ++g) {
          if (g < G) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(q_s + g * kHeadDim + quarter * 4 + i * 16);
            dot[g] = fmaf(kv[i].x, qv.x, dot[g]);
            dot[g] = fmaf(kv[i].y, qv.y, dot[g]);
            dot[g] = fmaf(kv[i].z, qv.z, dot[g]);
            dot[g] = fmaf(kv[i].w, qv.w, dot[g]);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < kMaxGroup; ++g) {
        if (g < G) {
          float v = dot[g];
          v += __shfl_xor(v, 1, 64);
          v += __shfl_xor(v, 2, 64);
          if (quarter == 0) s_s[g * kTile + key_l] = j <= pos ? v * scale : -INFINITY;
        }
      }
    }
    __syncthreads();
    // ---- phase 2: online softmax statistics, one wave per head ------------------------------------
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int g = wave + gi * 4;
      if (g < G) {
        const float s = s_s[g * kTile + lane];
        const float m_new = fmaxf(m_run[gi], wave_max(s));   // finite: key j0 <= pos is always valid
        const float pr = expf(s - m_new);
        const float a = expf(m_run[gi] - m_new);             // exp(-inf) = 0 on the first tile
        l_run[gi] = l_run[gi] * a + wave_sum(pr);
        m_run[gi] = m_new;
        s_s[g * kTile + lane] = pr;
        if (lane == 0) alpha_s[g] = a;
      }
    }
    __syncthreads();
    // ---- phase 3: o = o * alpha + P.V ---------------------------------------------------------
    {
#pragma unroll
      for (int g = 0; g < kMaxGroup; ++g)
        if (g < G) acc[g] *= alpha_s[g];
      if (owner) {                               // the new V row is not in the cache yet for this workgroup
        const int jb = j0 + half * 32;
#pragma unroll
        for (int u = 0; u < 32; ++u)
          if (jb + u >= pos) vv[u] = vnew_s[d_o];
      }
#pragma unroll
      for (int u = 0; u < 32; ++u) {
#pragma unroll
        for (int g = 0; g < kMaxGroup; ++g)
          if (g < G) acc[g] = fmaf(s_s[g * kTile + half * 32 + u], vv[u], acc[g]);   // p = 0 for masked keys
      }
    }
    __syncthreads();
  }

  // ---- merge the two key halves, then write ------------------------------------------------------
  if (half == 1) {
#pragma unroll
    for (int g = 0; g < kMaxGroup; ++g)
      if (g < G) o_s[g * kHeadDim + d_o] = acc[g];
  }
  if (lane == 0) {   // publish m, l (every lane of the owning wave holds the same value)
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int g = wave + gi * 4;
      if (g < G) { s_s[g * 2] = m_run[gi]; s_s[g * 2 + 1] = l_run[gi]; }
    }
  }
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int g = 0; g < kMaxGroup; ++g) {
      if (g < G) {
        const float o = acc[g] + o_s[g * kHeadDim + d_o];
        const int hq = hk * G + g;
        if (p.n_splits == 1) {
          p.out[((size_t)row * p.n_q + hq) * kHeadDim + d_o] = o / s_s[g * 2 + 1];
        } else {
          const size_t pi = ((size_t)row * p.n_q + hq) * p.n_splits + split;
          p.part_o[pi * kHeadDim + d_o] = o;
          if (d_o == 0) { p.part_ml[pi * 2] = s_s[g * 2]; p.part_ml[pi * 2 + 1] = s_s[g * 2 + 1]; }
        }
      }
    }
  }
}

// Merge split partials: out = sum_s e^{m_s - M} o_s / sum_s e^{m_s - M} l_s over the splits that saw keys.
// One workgroup (2 waves) per (row, head).  Lane s of each wave loads (m_s, l_s) once; weights are broadcast
// by shuffle, so the o_s[d] loads of all splits are independent and stay in flight together.
__global__ __launch_bounds__(128) void attn_combine_kernel(AttnParams p) {
  const int hq = blockIdx.x, row = blockIdx.y, d = threadIdx.x, lane = threadIdx.x & 63;
  const int pos = (p.pos0_dev ? *p.pos0_dev : p.pos0) + row;
  const int ntiles = pos / kTile + 1;
  const int ns = ntiles < p.n_splits ? ntiles : p.n_splits;     // splits >= ntiles saw no key (n_splits <= 64)
  const size_t base = ((size_t)row * p.n_q + hq) * p.n_splits;
  float m = -INFINITY, l = 0.f;
  if (lane < ns) { m = p.part_ml[(base + lane) * 2]; l = p.part_ml[(base + lane) * 2 + 1]; }
  const float M = wave_max(m);
  const float w = lane < ns ? expf(m - M) : 0.f;
  const float den = wave_sum(w * l);
  float num = 0.f;
  for (int s0 = 0; s0 < ns; s0 += 8) {
    float o[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u < ns ? s0 + u : ns - 1;
      o[u] = p.part_o[(base + s) * kHeadDim + d];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float ws = __shfl(w, s0 + u < 64 ? s0 + u : 63, 64);
      if (s0 + u < ns) num = fmaf(ws, o[u], num);
    }
  }
  p.out[((size_t)row * p.n_q + hq) * kHeadDim + d] = num / den;
}

}  // namespace chatts

using namespace chatts;

extern "C" size_t chatts_attn_workspace(int t, int n_q, int n_splits) {
  if (n_splits <= 1) return 0;
  return (size_t)t * n_q * n_splits * (kHeadDim + 2) * sizeof(float);
}

static int attention_common(AttnParams& p, bool fused, void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (p.n_splits > 1) {
    const size_t need = chatts_attn_workspace(p.t, p.n_q, p.n_splits);
    CHATTS_REQUIRE(workspace && workspace_bytes >= need, CHATTS_E_WORKSPACE,
                   "attention: needs %zu workspace bytes, got %zu", need, workspace_bytes);
    p.part_o = reinterpret_cast<float*>(workspace);
    p.part_ml = p.part_o + (size_t)p.t * p.n_q * p.n_splits * kHeadDim;
  }
  if (fused)
    hipLaunchKernelGGL(attn_rows_kernel<true>, dim3(p.n_kv, p.t, p.n_splits), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(attn_rows_kernel<false>, dim3(p.n_kv, p.t, p.n_splits), dim3(256), 0, s, p);
  CHATTS_CHECK_LAUNCH("attn_rows");
  if (p.n_splits > 1) {
    hipLaunchKernelGGL(attn_combine_kernel, dim3(p.n_q, p.t), dim3(128), 0, s, p);
    CHATTS_CHECK_LAUNCH("attn_combine");
  }
  return CHATTS_OK;
}

extern "C" int chatts_attention(const float* qkv, int t, int n_q, int n_kv, int pos0, const int32_t* pos0_dev,
                                const ChattsKvCache* cache, float* out, int n_splits, void* workspace,
                                size_t workspace_bytes, chatts_stream_t stream) {
  CHATTS_REQUIRE(t >= 0 && n_q > 0 && n_kv > 0 && n_splits >= 1 && n_splits <= 64, CHATTS_E_BADARG, "attention: bad sizes");
  if (t == 0) return CHATTS_OK;
  CHATTS_REQUIRE(qkv && out && cache && cache->k && cache->v, CHATTS_E_BADARG, "attention: null pointer");
  CHATTS_REQUIRE(n_q % n_kv == 0 && n_q / n_kv <= kMaxGroup, CHATTS_E_SHAPE,
                 "attention: GQA group %d/%d unsupported (max %d)", n_q, n_kv, kMaxGroup);
  if (!pos0_dev)
    CHATTS_REQUIRE(pos0 >= 0 && pos0 + t <= cache->max_ctx, CHATTS_E_SHAPE, "attention: positions exceed the cache");
  AttnParams p{};
  p.qkv = qkv; p.kc = cache->k; p.vc = cache->v; p.out = out; p.pos0_dev = pos0_dev; p.pos0 = pos0;
  p.t = t; p.n_q = n_q; p.n_kv = n_kv; p.max_ctx = cache->max_ctx; p.n_splits = n_splits;
  return attention_common(p, false, workspace, workspace_bytes, as_stream(stream));
}

extern "C" int chatts_attention_decode_fused(const float* qkv_raw, int n_q, int n_kv, const float* q_norm_w,
                                             const float* k_norm_w, float norm_eps, const float* cos_tab,
                                             const float* sin_tab, int pos, const int32_t* pos_dev,
                                             const ChattsKvCache* cache, float* out, int n_splits, void* workspace,
                                             size_t workspace_bytes, chatts_stream_t stream) {
  CHATTS_REQUIRE(n_q > 0 && n_kv > 0 && n_splits >= 1 && n_splits <= 64, CHATTS_E_BADARG, "attention_decode_fused: bad sizes");
  CHATTS_REQUIRE(qkv_raw && out && cos_tab && sin_tab && cache && cache->k && cache->v, CHATTS_E_BADARG,
                 "attention_decode_fused: null pointer");
  CHATTS_REQUIRE((q_norm_w == nullptr) == (k_norm_w == nullptr), CHATTS_E_BADARG,
                 "attention_decode_fused: q_norm and k_norm must both be set or both be null");
  CHATTS_REQUIRE(n_q % n_kv == 0 && n_q / n_kv <= kMaxGroup, CHATTS_E_SHAPE,
                 "attention: GQA group %d/%d unsupported (max %d)", n_q, n_kv, kMaxGroup);
  if (!pos_dev)
    CHATTS_REQUIRE(pos >= 0 && pos < cache->max_ctx, CHATTS_E_SHAPE, "attention_decode_fused: position exceeds the cache");
  AttnParams p{};
  p.qkv = qkv_raw; p.kc = cache->k; p.vc = cache->v; p.out = out; p.pos0_dev = pos_dev; p.pos0 = pos;
  p.t = 1; p.n_q = n_q; p.n_kv = n_kv; p.max_ctx = cache->max_ctx; p.n_splits = n_splits;
  p.q_norm_w = q_norm_w; p.k_norm_w = k_norm_w; p.cos_tab = cos_tab; p.sin_tab = sin_tab; p.eps = norm_eps;
  return attention_common(p, true, workspace, workspace_bytes, as_stream(stream));
}
