// attn_decode.h - the decode attention as per-wave device functions (attention.hip: one wave per (kv head, tile slot, sequence), the
// partials merged by a second launch).  Two other forms were built on these functions, measured and removed: one workgroup per (kv
// head, sequence) merging in LDS (round 4, profiles/r4_attn_wg_ab.txt) and the merge by the last-arriving waves of the same launch
// (round 5, profiles/r5_attn_fold_ab.txt: 16.1 us against 8.95 + 4.88 - five dependent round trips behind the last arriver).
#pragma once
#include <math.h>

#include "common.h"

namespace chatts {

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
  // prefill (bf16x3 kernel), optional: this sequence's K / V rows 0 .. kv_plane_keys-1 already split into bf16 hi / lo planes by
  // kv_planes_kernel, per (kv head, 32-key tile) 32 KB: K_hi [32][128], K_lo, V^T_hi [128][32], V^T_lo.  NULL: split while staging
  const uint16_t* kv_planes;
  int kv_plane_tiles;  // tiles per kv head
  // attn_prefill_planes_kernel's 1-D grid: query tiles per head and, when n_kv divides 8, the number of XCDs that share one kv head
  // (8 / n_kv; 0 = no XCD-aware order) - see the kernel
  int units, xcd_share;
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
// One wave's work: kv head hk, tile slot `slot` of sequence `seq`, in two stages:
//   attn_decode_preload: everything that does not depend on this step's qkv - the position, the slot's first K tile, cos / sin;
//   attn_decode_finish : q prologue, new K / V row, scores, softmax, P.V, the partial.
// q_s [GMAX * 128], knew_s / vnew_s [128] are LDS scratch (16-byte aligned) - private to the wave, or shared by the waves of a
// workgroup that all work on the same (kv head, sequence): they write the same q values, and only the owner wave touches knew_s /
// vnew_s.  The partial of head g goes to part_o[(head_base + g0 + g) * NS + slot] / part_ml[..] - global workspace or LDS.
struct AttnTileRegs {       // what a wave keeps between the two stages: the slot's first K tile, cos / sin, the position
  f32x4 kv[8];
  float c, s;
  int pos;
};

// Register layout of a 16-key tile (both loads ask for WHOLE 128-byte lines: a request of 16 rows x 64 bytes - the layout this
// kernel had before, and the MFMA operand shape - streams at 4.1 TB/s on this part, whole lines at 6.2, profiles/r3_row_stream_probe.txt):
//   K: lane = (kg = lane >> 3, c = lane & 7) holds dims c*4 + 32 i (i < 4) of key kg in kv[i] and of key 8 + kg in kv[4 + i]:
//      one instruction = 8 key rows x 128 contiguous bytes;
//   V: lane = (h = lane >> 5, c = lane & 31) holds dims c*4 .. c*4+3 of key 2 i + h in vv[i] (i < 8):
//      one instruction = 2 key rows x 512 contiguous bytes.
// A decode step reads every cache row exactly once: the tile loads are non-temporal like the weight stream (they do not displace the
// activations and partials the step re-reads from L2).  Measured A/B (-DCHATTS_KV_NT=0 builds the plain loads): config 5 6.51 -> 6.24 ms
// per step, config 4 162.4 -> 164.1 tok/s, headline 180.4 -> 181.7 tok/s.
#ifndef CHATTS_KV_NT
#define CHATTS_KV_NT 1
#endif
__device__ __forceinline__ f32x4 kv_stream_load(const float* p) {
#if CHATTS_KV_NT
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
  return *reinterpret_cast<const f32x4*>(p);
#endif
}
__device__ __forceinline__ void attn_load_k(const KvLayout& kvl, const float* kcache, const int hk, const int tile, const int pos,
                                            const int lane, f32x4 (&kv)[8]) {
  const int kg = lane >> 3, c = lane & 7;
  const int j0 = tile * kDTile;
  const int ja = j0 + kg <= pos ? j0 + kg : pos;          // clamped addresses (stay inside this tile: pos lies in it); masked below
  const int jb = j0 + 8 + kg <= pos ? j0 + 8 + kg : pos;
  const size_t toff = kv_tile_off(kvl, hk, j0);           // the tile's first row: one table lookup per tile when the cache is paged
  const float* ka = kcache + toff + (size_t)(ja - j0) * kHeadDim + c * 4;
  const float* kb = kcache + toff + (size_t)(jb - j0) * kHeadDim + c * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) kv[i] = kv_stream_load(ka + i * 32);
#pragma unroll
  for (int i = 0; i < 4; ++i) kv[4 + i] = kv_stream_load(kb + i * 32);
}
__device__ __forceinline__ void attn_load_v(const KvLayout& kvl, const float* vcache, const int hk, const int tile, const int pos,
                                            const int lane, f32x4 (&vv)[8]) {
  const int j0 = tile * kDTile, h = lane >> 5, c = lane & 31;
  const size_t toff = kv_tile_off(kvl, hk, j0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ju = j0 + 2 * i + h <= pos ? j0 + 2 * i + h : pos;
    vv[i] = kv_stream_load(vcache + toff + (size_t)(ju - j0) * kHeadDim + c * 4);
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
// EXACT (round 5): the group IS GMAX heads - `g < gn` is decided by the compiler and the tile body is one straight-line block (the
// run-time form compiled to ~400 basic blocks, every head's every step behind a scalar branch, no load / FMA interleaving across them).
template <int GMAX, bool PF, bool EXACT>
__device__ __forceinline__ void attn_decode_finish(const AttnParams& p, const int hk, const int slot, const int seq, const int lane,
                                                   AttnTileRegs& t, float* q_s, float* knew_s, float* vnew_s, const int g0, const int gn_rt,
                                                   float* part_o, float* part_ml, const size_t head_base) {
  const int gn = EXACT ? GMAX : gn_rt;
  const int NS = p.n_splits;
  const int G = EXACT ? GMAX : p.n_q / p.n_kv;
  const int pos = t.pos;
  const int ntiles = pos / kDTile + 1;
  const int qkv_n = (p.n_q + 2 * p.n_kv) * kHeadDim;
  const float* qkv = p.qkv + (size_t)seq * qkv_n;
  auto qkv_at = [&](const int col) __attribute__((always_inline)) -> float { return qkv[col]; };
  float* kcache = p.kc + (p.table ? 0 : (size_t)seq * p.seq_stride);
  float* vcache = p.vc + (p.table ? 0 : (size_t)seq * p.seq_stride);
  const KvLayout kvl{p.table ? p.table + (size_t)seq * p.table_stride : nullptr, p.n_kv, p.max_ctx, p.log_block};
  const bool owner = ((pos / kDTile) % NS) == slot;
  const float scale = 0.08838834764831845f * 1.4426950408889634f;    // 128^-1/2 * log2(e): the softmax runs on v_exp_f32 (exp2)
  const int kg = lane >> 3, kc = lane & 7, vh = lane >> 5, vc = lane & 31;      // see attn_load_k / attn_load_v
  f32x4 (&kv)[8] = t.kv;
  f32x4 vv[8];
  attn_load_v(kvl, vcache, hk, slot, pos, lane, vv);

  {
    const float c = t.c, s = t.s;
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
      if (g < gn) {
        const int src = (hk * G + g0 + g) * kHeadDim;
        float a = qkv_at(src + lane), b = qkv_at(src + lane + 64);
        norm_rope(a, b, p.q_norm_w, p.eps, c, s, lane);
        q_s[g * kHeadDim + lane] = a;
        q_s[g * kHeadDim + lane + 64] = b;
      }
    }
    if (owner) {                                // the new K/V row: to the cache and to LDS
      const int ks = (p.n_q + hk) * kHeadDim;
      float a = qkv_at(ks + lane), b = qkv_at(ks + lane + 64);
      norm_rope(a, b, p.k_norm_w, p.eps, c, s, lane);
      a = kv_round_f(a, p.kv_round); b = kv_round_f(b, p.kv_round);
      knew_s[lane] = a;
      knew_s[lane + 64] = b;
      const size_t noff = kv_tile_off(kvl, hk, pos);      // row of the new token
      float* kd = kcache + noff;
      if (g0 == 0) { kd[lane] = a; kd[lane + 64] = b; }
      const int vs = (p.n_q + p.n_kv + hk) * kHeadDim;
      const float va = kv_round_f(qkv_at(vs + lane), p.kv_round), vb = kv_round_f(qkv_at(vs + lane + 64), p.kv_round);
      vnew_s[lane] = va;
      vnew_s[lane + 64] = vb;
      float* vd = vcache + noff;
      if (g0 == 0) { vd[lane] = va; vd[lane + 64] = vb; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // single wave: orders the LDS writes above (in-order LDS, no barrier needed)

  float m_run[GMAX], l_run[GMAX];
  f32x4 acc[GMAX];                              // this lane's 4 dims, summed over the keys of its half (even / odd keys of a tile)
#pragma unroll
  for (int g = 0; g < GMAX; ++g) { m_run[g] = -INFINITY; l_run[g] = 0.f; acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  // one 16-key tile: scores, online softmax, P.V - on the K / V rows held in (kvb, vvb)
  auto tile_body = [&](const int tile, f32x4 (&kvb)[8], f32x4 (&vvb)[8]) __attribute__((always_inline)) {
    const int j0 = tile * kDTile;
    const int ja = j0 + kg, jb = j0 + 8 + kg;
    if (owner) {                                // the row just produced is not in the cache for this wave yet
      if ((ja <= pos ? ja : pos) == pos) {
#pragma unroll
        for (int i = 0; i < 4; ++i) kvb[i] = *reinterpret_cast<const f32x4*>(knew_s + kc * 4 + i * 32);
      }
      if ((jb <= pos ? jb : pos) == pos) {
#pragma unroll
        for (int i = 0; i < 4; ++i) kvb[4 + i] = *reinterpret_cast<const f32x4*>(knew_s + kc * 4 + i * 32);
      }
      const f32x4 vn = *reinterpret_cast<const f32x4*>(vnew_s + vc * 4);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (j0 + 2 * i + vh >= pos) vvb[i] = vn;
    }
    float da[GMAX], db[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) { da[g] = 0.f; db[g] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int g = 0; g < GMAX; ++g) {
        if (g < gn) {
          const f32x4 qv = *reinterpret_cast<const f32x4*>(q_s + g * kHeadDim + kc * 4 + i * 32);      // shared by both keys
          da[g] = fmaf(kvb[i].x, qv.x, da[g]);
          da[g] = fmaf(kvb[i].y, qv.y, da[g]);
          da[g] = fmaf(kvb[i].z, qv.z, da[g]);
          da[g] = fmaf(kvb[i].w, qv.w, da[g]);
          db[g] = fmaf(kvb[4 + i].x, qv.x, db[g]);
          db[g] = fmaf(kvb[4 + i].y, qv.y, db[g]);
          db[g] = fmaf(kvb[4 + i].z, qv.z, db[g]);
          db[g] = fmaf(kvb[4 + i].w, qv.w, db[g]);
        }
      }
      // EXACT: keep the q fragments of one 32-dim slice live at a time (unfenced, the scheduler hoists all GMAX x 4 LDS reads of the
      // tile to its top and the kernel no longer fits the 256 registers of two waves per SIMD)
      if constexpr (EXACT) __builtin_amdgcn_sched_barrier(0);
    }
    float pa[GMAX], pb[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
      pa[g] = 0.f; pb[g] = 0.f;
      if (g < gn) {
        float sa = da[g], sb = db[g];
        sa += lane_xor1(sa); sb += lane_xor1(sb);
        sa += lane_xor2(sa); sb += lane_xor2(sb);
        sa += half_mirror(sa); sb += half_mirror(sb);      // all 8 lanes of a key pair now hold both scores (DPP, no LDS round trip)
        sa = ja <= pos ? sa * scale : -INFINITY;
        sb = jb <= pos ? sb * scale : -INFINITY;
        // max / sum over the 16 keys: a 16-lane row holds 4 of them (two lane groups x two keys), the four rows via v_readlane
        const float m2 = fmaxf(sa, sb);
        const float mt = rows4_max(fmaxf(m2, row_ror8(m2)));
        const float m_new = fmaxf(m_run[g], mt);  // finite: key j0 <= pos is always valid
        const float ea = __builtin_amdgcn_exp2f(sa - m_new), eb = __builtin_amdgcn_exp2f(sb - m_new);
        const float e2 = ea + eb;
        const float es = rows4_sum(e2 + row_ror8(e2));
        const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_new);
        l_run[g] = l_run[g] * alpha + es;
        m_run[g] = m_new;
        acc[g].x *= alpha; acc[g].y *= alpha; acc[g].z *= alpha; acc[g].w *= alpha;
        pa[g] = ea; pb[g] = eb;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {               // keys 2 i (lanes 0-31) and 2 i + 1 (lanes 32-63)
#pragma unroll
      for (int g = 0; g < GMAX; ++g) {
        if (g < gn) {
          // p of key u lives in lanes 8 u .. 8 u + 7 of pa (u < 8) / 8 (u - 8) .. of pb (u >= 8): wave-uniform reads (0 for masked keys)
          const float pe = i < 4 ? readlane_f(pa[g], 16 * i) : readlane_f(pb[g], 16 * (i - 4));
          const float po = i < 4 ? readlane_f(pa[g], 16 * i + 8) : readlane_f(pb[g], 16 * (i - 4) + 8);
          const float pu = vh ? po : pe;
          acc[g].x = fmaf(pu, vvb[i].x, acc[g].x);
          acc[g].y = fmaf(pu, vvb[i].y, acc[g].y);
          acc[g].z = fmaf(pu, vvb[i].z, acc[g].z);
          acc[g].w = fmaf(pu, vvb[i].w, acc[g].w);
        }
      }
    }
  };
  if constexpr (PF) {
    // a slot that walks several tiles (long contexts, batched decode with few slots per sequence) fetches tile t + NS while it
    // works on tile t: two register sets in ping-pong, one memory round trip per PAIR of steps instead of one per tile
    f32x4 kv2[8];
    f32x4 vv2[8];
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
      // even keys were summed by lanes 0-31, odd keys by lanes 32-63: one exchange gives the totals (written by the lower half)
      f32x4 o = acc[g];
      o.x += __shfl_xor(o.x, 32, 64); o.y += __shfl_xor(o.y, 32, 64); o.z += __shfl_xor(o.z, 32, 64); o.w += __shfl_xor(o.w, 32, 64);
      const size_t pi = (head_base + g0 + g) * NS + slot;
      const float mn = m_run[g] * 0.6931471805599453f;       // m back to nats for the combine
      if (lane < 32) *reinterpret_cast<f32x4*>(part_o + pi * kHeadDim + vc * 4) = o;
      if (lane == 0) { part_ml[pi * 2] = mn; part_ml[pi * 2 + 1] = l_run[g]; }
    }
  }
}


// GMAX >= the group size: the per-head registers are sized by it (a Qwen2-14B group of 5 in registers for 8 costs the third wave per SIMD)
template <int GMAX = kMaxGroup, bool PF = true, bool EXACT = false>
__device__ __forceinline__ void attn_decode_wave(const AttnParams& p, const int hk, const int slot, const int seq, const int lane,
                                                 float* q_s, float* knew_s, float* vnew_s, float* part_o, float* part_ml,
                                                 const size_t head_base) {
  AttnTileRegs t;
  if (!attn_decode_preload(p, hk, slot, seq, lane, t)) return;
  attn_decode_finish<GMAX, PF, EXACT>(p, hk, slot, seq, lane, t, q_s, knew_s, vnew_s, 0, p.n_q / p.n_kv, part_o, part_ml, head_base);
}

// out[h] = sum_s e^{m_s - M} o_s / sum_s e^{m_s - M} l_s over the slots that saw keys (<= 64 slots).  One workgroup
// (2 waves) per head; lane s of each wave holds (m_s, l_s), weights are broadcast by shuffle, and 16 independent
// o_s[d] loads are in flight per thread: no LDS, no barrier.
// One wave's half of a head: d = half * 64 + lane (two waves per head).  The partials of head hq are part_o / part_ml rows
// base .. base + n_splits - 1 (global workspace: base = (seq * n_q + hq) * n_splits; LDS of the workgroup form: its local head index).
__device__ __forceinline__ void attn_combine_wave(const AttnParams& p, const int hq, const int seq, const int d, const int lane,
                                                  const float* part_o, const float* part_ml, const size_t base) {
  const int pos = p.pos0_dev ? p.pos0_dev[seq] : p.pos0;
  const int ntiles = pos < 0 ? 0 : pos / kDTile + 1;         // parked slot: no partials exist, the row is written as zeros
  const int ns = ntiles < p.n_splits ? ntiles : p.n_splits;
  // every load of this wave is requested before the first use (round 5): (m, l) of the slots and all <= 64 partial rows of this lane's
  // dim travel in ONE memory round trip instead of three dependent ones (the weights do not decide WHAT is loaded, only how it is summed)
  float m = -INFINITY, l = 0.f;
  if (lane < ns) { m = part_ml[(base + lane) * 2]; l = part_ml[(base + lane) * 2 + 1]; }
  float o[kMaxSlots];
#pragma unroll
  for (int u = 0; u < kMaxSlots; ++u) {
    const int sidx = u < ns ? u : (ns > 0 ? ns - 1 : 0);      // (clamped: its weight is 0)
    o[u] = (u < 32 || ns > 32) ? part_o[(base + sidx) * kHeadDim + d] : 0.f;
  }
  const float M = wave_max(m);
  const float w = lane < ns ? expf(m - M) : 0.f;
  const float den = wave_sum(w * l);
  float num = 0.f;
#pragma unroll
  for (int u = 0; u < kMaxSlots; ++u) {
    if (u < 32 || ns > 32) {                                   // (same fmaf chain over the slots in ascending order as before)
      const float ws = __shfl(w, u, 64);                       // 0 for slots >= ns
      num = fmaf(ws, o[u], num);
    }
  }
  const size_t oi = ((size_t)seq * p.n_q + hq) * kHeadDim + d;
  const float v = ns > 0 ? num / den : 0.f;
  if (p.out_hi) {
    const __bf16 h = (__bf16)v;
    p.out_hi[oi] = __builtin_bit_cast(uint16_t, h);
    p.out_lo[oi] = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)h));
  } else {
    p.out[oi] = v;
  }
}

}  // namespace chatts
