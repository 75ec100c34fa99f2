// attn_decode.h - the batch-1 decode attention as per-wave device functions, shared by the stand-alone kernels (attention.hip)
// and the persistent decode step (decode_mega.hip): same instructions, same summation order, bit-identical results.
#pragma once
#include <math.h>

#include "common.h"

namespace chatts {

// write-through stores for data another workgroup reads inside the SAME launch (agent-scope relaxed atomic store = `sc1`)
__device__ __forceinline__ void wt_store1(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned int*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wt_store2(float* p, float a, float b) {      // p 8-byte aligned
  const unsigned long long bits = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kMaxGroup = 8;
constexpr int kTile = 64;    // prefill kernel: keys per tile
constexpr int kDTile = 16;   // decode kernel: keys per wave-tile
constexpr int kMaxSlots = 64;    // decode: tile slots per kv head (one lane of the combine wave each)

struct AttnParams {
  const float* qkv;   // [T, (n_q + 2 n_kv) * 128]; q already rotated for attn_rows, raw for attn_decode
  float* kc;          // [n_kv, max_ctx, 128]
  float* vc;
  float* out;         // [T, n_q * 128]
  float* part_ml;     // [T, n_q, n_splits, 2]
  float* part_o;      // [T, n_q, n_splits, 128]
  const int32_t* pos0_dev;
  int pos0, t, n_q, n_kv, max_ctx, n_splits;
  // decode only
  const float* q_norm_w;
  const float* k_norm_w;
  const float* cos_tab;
  const float* sin_tab;
  float eps;
  size_t seq_stride;   // batched decode: floats between the caches of consecutive sequences (same layer)
  // block-paged cache (ChattsKvCache.block_table != NULL): kc / vc are the layer's pool [n_blocks, n_kv, 2^log_block, 128];
  // sequence b looks its blocks up in table + b * table_stride, seq_stride is not used
  const int32_t* table;
  int log_block, table_stride;
  uint16_t* out_hi;    // decode combine, optional: the output as bf16 hi / lo planes [T, n_q * 128] (the o_proj operand of the
  uint16_t* out_lo;    // weight-streaming kernel) instead of float32 `out`
  int kv_round;        // experiment knob, see kv_round_f (0 = off)
};

// rotate one 128-wide head held as (a = x[lane], b = x[lane+64]) by the wave; optional RMSNorm first
__device__ __forceinline__ void norm_rope(float& a, float& b, const float* nw, float eps, float c, float s, int lane) {
  // every instantiation of the decode attention (stand-alone, persistent step) must produce the SAME bits here: no contraction
  // decisions left to the optimiser (seen: an unrelated statement after this call flipped one instantiation's a * c - b * s to an fma)
#pragma clang fp contract(off)
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

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// ---------------------------------------------------------------------------------------------------
// decode: grid (n_kv, n_slots), 64 threads.  Slot s walks tiles s, s + n_slots, ... of 16 keys.
// ---------------------------------------------------------------------------------------------------
// One wave's work: kv head hk, tile slot `slot` of sequence `seq`, in two stages so that a caller which knows its task before
// the projections are ready (decode_mega.hip) can have the cache rows in registers by then:
//   attn_decode_preload: everything that does not depend on this step's qkv - the position, the slot's first K / V tile, cos / sin;
//   attn_decode_finish : q prologue, new K / V row, scores, softmax, P.V, the partial.
// q_s [kMaxGroup * 128], knew_s / vnew_s [128] are this wave's private LDS scratch (16-byte aligned).  WT: the partials are
// stored write-through (agent-scope relaxed atomics = `sc1` stores) for consumers inside the SAME launch; the stand-alone kernel
// uses plain stores.
struct AttnTileRegs {       // what a wave keeps between the two stages: the slot's first K tile, cos / sin, the position
  f32x4 kv[8];
  float c, s;
  int pos;
};

__device__ __forceinline__ void attn_load_k(const KvLayout& kvl, const float* kcache, const int hk, const int tile, const int pos,
                                            const int lane, f32x4 (&kv)[8]) {
  const int key_l = lane >> 2, quarter = lane & 3;
  const int j0 = tile * kDTile;
  const int j = j0 + key_l;
  const int jc = j <= pos ? j : pos;          // clamped address (stays inside this tile: pos lies in it); masked below
  const size_t toff = kv_tile_off(kvl, hk, j0);   // the tile's first row: one table lookup per tile when the cache is paged
  const float* kr = kcache + toff + (size_t)(jc - j0) * kHeadDim + quarter * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) kv[i] = *reinterpret_cast<const f32x4*>(kr + i * 16);
}
__device__ __forceinline__ void attn_load_v(const KvLayout& kvl, const float* vcache, const int hk, const int tile, const int pos,
                                            const int lane, float2 (&vv)[kDTile]) {
  const int j0 = tile * kDTile;
  const size_t toff = kv_tile_off(kvl, hk, j0);
#pragma unroll
  for (int u = 0; u < kDTile; ++u) {
    const int ju = j0 + u <= pos ? j0 + u : pos;
    vv[u] = *reinterpret_cast<const float2*>(vcache + toff + (size_t)(ju - j0) * kHeadDim + lane * 2);
  }
}

// -> false: nothing to do for this slot (parked sequence, or the slot lies beyond the context)
__device__ __forceinline__ bool attn_decode_preload(const AttnParams& p, const int hk, const int slot, const int seq, const int lane,
                                                    AttnTileRegs& t) {
  const int pos = p.pos0_dev ? p.pos0_dev[seq] : p.pos0;     // batched decode: one position per sequence
  t.pos = pos;
  if (pos < 0) return false;                   // parked slot of a batched step: no cache write, no attention (its row is ignored)
  const int ntiles = pos / kDTile + 1;
  if (slot >= ntiles) return false;            // the combine only reads slots < min(ntiles, NS)
  const float* kcache = p.kc + (p.table ? 0 : (size_t)seq * p.seq_stride);
  const KvLayout kvl{p.table ? p.table + (size_t)seq * p.table_stride : nullptr, p.n_kv, p.max_ctx, p.log_block};
  // K of the first tile only depends on `pos`: issued before the q prologue so that the cache rows, q, cos / sin all travel in
  // the same memory round trip (or, preloaded, in an earlier one); V follows at the head of the second stage - it is not
  // needed before the softmax is done
  attn_load_k(kvl, kcache, hk, slot, pos, lane, t.kv);
  t.c = p.cos_tab[(size_t)pos * 64 + lane];
  t.s = p.sin_tab[(size_t)pos * 64 + lane];
  return true;
}

// GMAX: query heads of the group this wave handles at most; it takes heads g0 .. g0 + gn - 1 of kv head hk's group (the stand-alone
// kernel: all of them, GMAX = kMaxGroup; the persistent step splits a group over two waves).  Heads are independent of each
// other, so the split does not change a single operation of any head.  The wave with g0 == 0 stores the new K / V row.
template <bool WT, int GMAX, bool PF>
__device__ __forceinline__ void attn_decode_finish(const AttnParams& p, const int hk, const int slot, const int seq, const int lane,
                                                   AttnTileRegs& t, float* q_s, float* knew_s, float* vnew_s, const int g0, const int gn) {
  const int NS = p.n_splits;
  const int G = p.n_q / p.n_kv;
  const int pos = t.pos;
  const int ntiles = pos / kDTile + 1;
  const float* qkv = p.qkv + (size_t)seq * (p.n_q + 2 * p.n_kv) * kHeadDim;
  float* kcache = p.kc + (p.table ? 0 : (size_t)seq * p.seq_stride);
  float* vcache = p.vc + (p.table ? 0 : (size_t)seq * p.seq_stride);
  const KvLayout kvl{p.table ? p.table + (size_t)seq * p.table_stride : nullptr, p.n_kv, p.max_ctx, p.log_block};
  const bool owner = ((pos / kDTile) % NS) == slot;
  const float scale = 0.08838834764831845f * 1.4426950408889634f;    // 128^-1/2 * log2(e): the softmax runs on v_exp_f32 (exp2)
  const int key_l = lane >> 2, quarter = lane & 3;
  f32x4 (&kv)[8] = t.kv;
  float2 vv[kDTile];
  attn_load_v(kvl, vcache, hk, slot, pos, lane, vv);

  {
    const float c = t.c, s = t.s;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
      if (g < gn) {
        const float* src = qkv + (size_t)(hk * G + g0 + g) * kHeadDim;
        float a = src[lane], b = src[lane + 64];
        norm_rope(a, b, p.q_norm_w, p.eps, c, s, lane);
        q_s[g * kHeadDim + lane] = a;
        q_s[g * kHeadDim + lane + 64] = b;
      }
    }
    if (owner) {                                // the new K/V row: to the cache and to LDS
      const float* ks = qkv + (size_t)(p.n_q + hk) * kHeadDim;
      float a = ks[lane], b = ks[lane + 64];
      norm_rope(a, b, p.k_norm_w, p.eps, c, s, lane);
      a = kv_round_f(a, p.kv_round); b = kv_round_f(b, p.kv_round);
      knew_s[lane] = a;
      knew_s[lane + 64] = b;
      const size_t noff = kv_tile_off(kvl, hk, pos);      // row of the new token
      float* kd = kcache + noff;
      if (g0 == 0) { kd[lane] = a; kd[lane + 64] = b; }
      const float* vs = qkv + (size_t)(p.n_q + p.n_kv + hk) * kHeadDim;
      const float va = kv_round_f(vs[lane], p.kv_round), vb = kv_round_f(vs[lane + 64], p.kv_round);
      vnew_s[lane] = va;
      vnew_s[lane + 64] = vb;
      float* vd = vcache + noff;
      if (g0 == 0) { vd[lane] = va; vd[lane + 64] = vb; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // single wave: orders the LDS writes above (in-order LDS, no barrier needed)

  float m_run[GMAX], l_run[GMAX], acc0[GMAX], acc1[GMAX];
#pragma unroll
  for (int g = 0; g < GMAX; ++g) { m_run[g] = -INFINITY; l_run[g] = 0.f; acc0[g] = 0.f; acc1[g] = 0.f; }

  // one 16-key tile: scores, online softmax, P.V - on the K / V rows held in (kvb, vvb)
  auto tile_body = [&](const int tile, f32x4 (&kvb)[8], float2 (&vvb)[kDTile]) __attribute__((always_inline)) {
    const int j0 = tile * kDTile;
    const int j = j0 + key_l;
    const int jc = j <= pos ? j : pos;
    if (owner && jc == pos) {                   // the row just produced is not in the cache for this wave yet
#pragma unroll
      for (int i = 0; i < 8; ++i) kvb[i] = *reinterpret_cast<const f32x4*>(knew_s + quarter * 4 + i * 16);
    }
    if (owner) {
      const float2 vn = *reinterpret_cast<const float2*>(vnew_s + lane * 2);
#pragma unroll
      for (int u = 0; u < kDTile; ++u)
        if (j0 + u >= pos) vvb[u] = vn;
    }
    float dot[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) dot[g] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int g = 0; g < GMAX; ++g) {
        if (g < gn) {
          const f32x4 qv = *reinterpret_cast<const f32x4*>(q_s + g * kHeadDim + quarter * 4 + i * 16);
          dot[g] = fmaf(kvb[i].x, qv.x, dot[g]);
          dot[g] = fmaf(kvb[i].y, qv.y, dot[g]);
          dot[g] = fmaf(kvb[i].z, qv.z, dot[g]);
          dot[g] = fmaf(kvb[i].w, qv.w, dot[g]);
        }
      }
    }
    float pr[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
      pr[g] = 0.f;
      if (g < gn) {
        float sc = dot[g];
        sc += lane_xor1(sc);
        sc += lane_xor2(sc);                    // all 4 lanes of a key now hold its score (DPP, no LDS round trip)
        sc = j <= pos ? sc * scale : -INFINITY;
        // max / sum over the 16 keys: keys of one 16-lane row with row_ror (DPP), the four rows with v_readlane
        const float mt = rows4_max(fmaxf(fmaxf(sc, row_ror4(sc)), row_ror8(fmaxf(sc, row_ror4(sc)))));
        const float m_new = fmaxf(m_run[g], mt);  // finite: key j0 <= pos is always valid
        const float e = __builtin_amdgcn_exp2f(sc - m_new);
        float es = e + row_ror4(e);
        es = rows4_sum(es + row_ror8(es));
        const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_new);
        l_run[g] = l_run[g] * alpha + es;
        m_run[g] = m_new;
        acc0[g] *= alpha;
        acc1[g] *= alpha;
        pr[g] = e;
      }
    }
#pragma unroll
    for (int u = 0; u < kDTile; ++u) {
#pragma unroll
      for (int g = 0; g < GMAX; ++g) {
        if (g < gn) {
          const float pu = readlane_f(pr[g], u * 4);   // p of key u, wave-uniform (0 for masked keys)
          acc0[g] = fmaf(pu, vvb[u].x, acc0[g]);
          acc1[g] = fmaf(pu, vvb[u].y, acc1[g]);
        }
      }
    }
  };
  if constexpr (PF) {
    // a slot that walks several tiles (long contexts, batched decode with few slots per sequence) fetches tile t + NS while it
    // works on tile t: two register sets in ping-pong, one memory round trip per PAIR of steps instead of one per tile
    f32x4 kv2[8];
    float2 vv2[kDTile];
    int tile = slot;
    while (true) {
      const int n1 = tile + NS;
      const bool h1 = n1 < ntiles;
      if (h1) { attn_load_k(kvl, kcache, hk, n1, pos, lane, kv2); attn_load_v(kvl, vcache, hk, n1, pos, lane, vv2); }
      tile_body(tile, kv, vv);
      if (!h1) break;
      const int n2 = n1 + NS;
      const bool h2 = n2 < ntiles;
      if (h2) { attn_load_k(kvl, kcache, hk, n2, pos, lane, kv); attn_load_v(kvl, vcache, hk, n2, pos, lane, vv); }
      tile_body(n1, kv2, vv2);
      if (!h2) break;
      tile = n2;
    }
  } else {
    for (int tile = slot; tile < ntiles; tile += NS) {
      if (tile != slot) {
        attn_load_k(kvl, kcache, hk, tile, pos, lane, kv);
        attn_load_v(kvl, vcache, hk, tile, pos, lane, vv);
      }
      tile_body(tile, kv, vv);
    }
  }
#pragma unroll
  for (int g = 0; g < GMAX; ++g) {
    if (g < gn) {
      const size_t pi = ((size_t)seq * p.n_q + hk * G + g0 + g) * NS + slot;
      const float mn = m_run[g] * 0.6931471805599453f;       // m back to nats for the combine
      if (WT) {
        wt_store2(p.part_o + pi * kHeadDim + lane * 2, acc0[g], acc1[g]);
        if (lane == 0) wt_store2(p.part_ml + pi * 2, mn, l_run[g]);
      } else {
        *reinterpret_cast<float2*>(p.part_o + pi * kHeadDim + lane * 2) = make_float2(acc0[g], acc1[g]);
        if (lane == 0) { p.part_ml[pi * 2] = mn; p.part_ml[pi * 2 + 1] = l_run[g]; }
      }
    }
  }
}


template <bool WT>
__device__ __forceinline__ void attn_decode_wave(const AttnParams& p, const int hk, const int slot, const int seq, const int lane,
                                                 float* q_s, float* knew_s, float* vnew_s) {
  AttnTileRegs t;
  if (!attn_decode_preload(p, hk, slot, seq, lane, t)) return;
  attn_decode_finish<WT, kMaxGroup, true>(p, hk, slot, seq, lane, t, q_s, knew_s, vnew_s, 0, p.n_q / p.n_kv);
}

// out[h] = sum_s e^{m_s - M} o_s / sum_s e^{m_s - M} l_s over the slots that saw keys (<= 64 slots).  One workgroup
// (2 waves) per head; lane s of each wave holds (m_s, l_s), weights are broadcast by shuffle, and 16 independent
// o_s[d] loads are in flight per thread: no LDS, no barrier.
// One wave's half of a head: d = half * 64 + lane (the stand-alone kernel runs two waves per head).
template <bool WT>
__device__ __forceinline__ void attn_combine_wave(const AttnParams& p, const int hq, const int seq, const int d, const int lane) {
  const int pos = p.pos0_dev ? p.pos0_dev[seq] : p.pos0;
  const int ntiles = pos < 0 ? 0 : pos / kDTile + 1;         // parked slot: no partials exist, the row is written as zeros
  const int ns = ntiles < p.n_splits ? ntiles : p.n_splits;
  const size_t base = ((size_t)seq * p.n_q + hq) * p.n_splits;
  float m = -INFINITY, l = 0.f;
  if (lane < ns) { m = p.part_ml[(base + lane) * 2]; l = p.part_ml[(base + lane) * 2 + 1]; }
  const float M = wave_max(m);
  const float w = lane < ns ? expf(m - M) : 0.f;
  const float den = wave_sum(w * l);
  float num = 0.f;
  for (int s0 = 0; s0 < ns; s0 += 32) {       // 32 independent loads in flight per lane: at most two dependent rounds for 64 slots
    float o[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int s = s0 + u < ns ? s0 + u : ns - 1;
      o[u] = p.part_o[(base + s) * kHeadDim + d];
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const float ws = __shfl(w, (s0 + u) & 63, 64);     // 0 for slots >= ns
      num = fmaf(ws, o[u], num);
    }
  }
  const size_t oi = ((size_t)seq * p.n_q + hq) * kHeadDim + d;
  const float v = ns > 0 ? num / den : 0.f;
  if (p.out_hi) {
    const __bf16 h = (__bf16)v;
    p.out_hi[oi] = __builtin_bit_cast(uint16_t, h);
    p.out_lo[oi] = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)h));
  } else if (WT) {
    wt_store1(p.out + oi, v);
  } else {
    p.out[oi] = v;
  }
}


}  // namespace chatts
