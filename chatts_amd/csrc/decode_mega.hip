// decode_mega.hip - ONE persistent launch per generated token (batch 1, tensor_parallel_size 1, bf16 weights).
//
// What it replaces: chatts_decoder_decode_step's chain of 6 kernels per layer (qkv GEMV, attention, combine, o GEMV, gate_up
// GEMV, down GEMV) + lm_head + argmax + embedding = 291 launches for ChatTS-14B.  Those kernels stream at 7.1 TB/s while they
// stream, but every one of them pays ~4.2 us of ramp + drain + boundary during which HBM is idle (round-2 fit: t = 4.2 us +
// bytes / 7.09 TB/s), and the attention island streams nothing: 31 % of a layer.  The weights do not depend on the activations,
// so here the weight stream never stops:
//
//   * one workgroup per CU, resident for the whole step; 10 COMPUTE waves stream weights, 6 HELPER waves do everything else;
//   * a compute wave owns a fixed list of row-pair tasks per projection and walks them as one stream of ELEMENTS (2 rows x 512
//     columns = two 16-byte non-temporal loads per lane) that always runs kDepth = 8 elements ahead of the FMAs - across task
//     boundaries and across PHASE boundaries: while the grid barrier of phase p completes and the helpers stage the next
//     activation vector into LDS, the first 16 KB per wave of phase p + 1 are already in flight or in registers (the o_proj
//     weights arrive during the attention island).  Registers, not LDS, are the prefetch buffer: 10 waves x 16 KB = 160 KB per CU;
//   * phases meet at an XCD-sharded grid barrier (per-group arrival counters, one top counter, per-group generation words;
//     MI355X_MICROARCH.md "barrier-xcd"); data crosses workgroups by write-through (sc1) stores, a drained vmcnt, a relaxed
//     flag, ONE agent-scope acquire per CU, then plain loads (cdna_hip_programming.md Guideline 16, recipe R1);
//   * only ONE lane per workgroup ever polls (helper wave 15); every other wave waits in s_barrier, which costs no issue slots;
//   * compute waves never touch global memory except for the weight stream (their results go through LDS to the publishing
//     helper), so the in-order vmcnt of a wave that is 16 loads deep never delays a hand-off.
//
// Arithmetic is the stand-alone kernels' (gemv_common.h / attn_decode.h are shared): every row is accumulated in the same
// order, the fused RMSNorm reproduces the stand-alone GEMV's partial-sum geometry, attention and its combine are the same
// per-wave functions - the step is bit-identical to the multi-kernel path (tests/test_gpu_decode_mega.py).
//
// Every spin is bounded; a timeout sets a STICKY status word (chatts_decoder_mega_status) and every later launch returns at once.
#include <vector>

#include "common.h"
#include "gemv_common.h"
#include "attn_decode.h"
#include "decode_mega.h"

namespace chatts {

constexpr int kMegaThreads = 1024;
constexpr int kCompute = 10;          // waves 0..9
constexpr int kHelpers = 6;           // waves 10..15
constexpr int kMaster = 15;           // publishes the workgroup's results and runs the grid barrier
constexpr int kDepth = 8;             // stream elements in flight per compute wave
constexpr int kOutSlots = 1024;       // floats of per-workgroup results (lm_head: 2 x 297)
constexpr unsigned kSpinLimit = 1u << 19;

enum { PH_QKV = 0, PH_ATTN = 1, PH_COMBINE = 2, PH_O = 3, PH_GATE_UP = 4, PH_DOWN = 5, PH_LM_HEAD = 6 };
enum { G_QKV = 0, G_O = 1, G_GATE_UP = 2, G_DOWN = 3, G_LM_HEAD = 4 };

// ---- grid barrier -----------------------------------------------------------------------------------------------------
// Monotonic counters, zeroed by a memset node before every launch.  epoch = 1, 2, ...  ONE lane per workgroup calls this after
// its workgroup's write-through stores have been drained (s_waitcnt vmcnt(0)).
__device__ __forceinline__ bool spin_until(unsigned* word, unsigned target, unsigned* status) {
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if (spins > kSpinLimit) {
      __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

__device__ __forceinline__ bool grid_barrier(MegaSync* s, unsigned epoch, int grp, unsigned grp_size, unsigned n_groups) {
  const unsigned old = __hip_atomic_fetch_add(&s->grp_count[grp * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old + 1 == grp_size * epoch) {          // the group's last arriver carries it to the top and releases the group
    __hip_atomic_fetch_add(&s->top_count[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = spin_until(&s->top_count[0], n_groups * epoch, &s->status[0]);
    __hip_atomic_store(&s->grp_gen[grp * 32], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ok;
  }
  return spin_until(&s->grp_gen[grp * 32], epoch, &s->status[0]);
}

// Pointers read from the device-side layer table are generic to the compiler: a load through them would be a FLAT load, which
// counts on lgkmcnt as well as vmcnt - every LDS wait would then drain the whole weight stream.  Device code therefore reads
// the table through this view of MegaLayer (same layout) whose members are global-address-space pointers.
#define MEGA_G __attribute__((address_space(1)))
struct MegaLayerDev {
  const MEGA_G float* input_norm;
  const MEGA_G uint16_t* qkv;
  const MEGA_G float* qkv_bias;
  const MEGA_G float* q_norm;
  const MEGA_G float* k_norm;
  const MEGA_G uint16_t* o;
  const MEGA_G float* post_norm;
  const MEGA_G uint16_t* gate_up;
  const MEGA_G uint16_t* down;
  MEGA_G float* kc;
  MEGA_G float* vc;
};
static_assert(sizeof(MegaLayerDev) == sizeof(MegaLayer), "device view of the layer table");
typedef const __attribute__((address_space(1))) u32x4* gw_ptr;

// One 8-byte field of the layer table by SCALAR load (a vector load + s_waitcnt vmcnt(0) here would stall the compute wave
// behind its own 16 outstanding weight loads at every phase change).
__device__ __forceinline__ const uint16_t* sload_weight_ptr(const MegaLayer* layers, int layer, int field_off) {
  unsigned long long r;
  const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane(layer * (int)sizeof(MegaLayer) + field_off);
  asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(layers), "s"(off) : "memory");
  return reinterpret_cast<const uint16_t*>(r);
}

// ---- the weight stream of a compute wave ------------------------------------------------------------------------------
struct WDesc {             // one weight phase as seen by one compute wave (all wave-uniform)
  const uint16_t* w;
  int k, nchunks, n, swiglu;
  int task0;               // first global task of this wave, then + nact
  int lt0;                 // its index within the workgroup's task range
  int nact;
  int n_elems;             // tasks of this wave x nchunks
};

__device__ __forceinline__ WDesc weight_desc(const MegaParams& p, int wph, int b, int wave) {
  WDesc d;
  const int last = 4 * p.n_layers;
  const int kind = wph >= last ? G_LM_HEAD : (wph & 3);
  const MegaGeom g = p.geom[kind];
  const uint16_t* w = p.lm_head;
  if (wph < last) {
    const int foff = kind == G_QKV ? (int)offsetof(MegaLayer, qkv) : kind == G_O ? (int)offsetof(MegaLayer, o) :
                     kind == G_GATE_UP ? (int)offsetof(MegaLayer, gate_up) : (int)offsetof(MegaLayer, down);
    w = sload_weight_ptr(p.layers, wph >> 2, foff);
  }
  d.w = w; d.k = g.k; d.nchunks = (g.k + 511) >> 9; d.n = g.n; d.swiglu = g.swiglu; d.nact = g.nact;
  int nt = g.tasks - b * g.tpw;
  nt = nt > g.tpw ? g.tpw : nt;
  int ntw = (wave < g.nact && nt > wave) ? (nt - wave + g.nact - 1) / g.nact : 0;
  if (wph > last) ntw = 0;                  // beyond the step: nothing left to prefetch
  d.task0 = b * g.tpw + wave; d.lt0 = wave;
  d.n_elems = ntw * d.nchunks;
  return d;
}

__device__ __forceinline__ int first_row(const WDesc& d, int task) {      // second row: + 1 (plain) or + 16 (gate / up pair)
  return d.swiglu ? (task >> 4) * 32 + (task & 15) : task * 2;
}

// ---- activation staging (helper waves) ----------------------------------------------------------------------------------
// The next phase's input vector -> LDS in the GEMV's permuted layout ([chunk][half][lane] float4: conflict-free ds_read_b128),
// optionally RMS-normalised.  The sum of squares is formed exactly like the stand-alone gemv_ldsx_kernel forms it with
// `vthreads` threads: virtual thread vt accumulates k4 = 4 vt, 4 vt + 4 vthreads, ...; each virtual wave is reduced by wave_sum;
// the wave results are added in wave order.
__device__ __forceinline__ void stage_sumsq(const float* src, int K, int vthreads, int hw, int lane, float* red) {
  const int nvw = vthreads >> 6;
  for (int vw = hw; vw < nvw; vw += kHelpers) {
    float ss = 0.f;
    for (int k4 = (vw * 64 + lane) * 4; k4 < K; k4 += vthreads * 4) ss = sumsq4(ss, *reinterpret_cast<const f32x4*>(src + k4));
    ss = wave_sum(ss);
    if (lane == 0) red[vw] = ss;
  }
}

__device__ __forceinline__ void stage_write(const float* src, const float* norm_w, int K, int nchunks, int vthreads, float eps,
                                            int htid, const float* red, f32x4* xs4) {
  float rstd = 1.f;
  if (norm_w) {
    float t = 0.f;
    const int nvw = vthreads >> 6;
    for (int i = 0; i < nvw; ++i) t += red[i];
    rstd = rsqrtf(t / (float)K + eps);
  }
  for (int q = htid; q < nchunks * 128; q += kHelpers * 64) {
    const int k4 = q * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k4 < K) {
      v = *reinterpret_cast<const f32x4*>(src + k4);
      if (norm_w) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(norm_w + k4);
        v.x = g.x * (v.x * rstd); v.y = g.y * (v.y * rstd); v.z = g.z * (v.z * rstd); v.w = g.w * (v.w * rstd);
      }
    }
    const int chunk = k4 >> 9, within = k4 & 511;
    xs4[chunk * 128 + ((within >> 2) & 1) * 64 + (within >> 3)] = v;
  }
}

#define MEGA_LDS_SYNC()                                     \
  do {                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                           \
    asm volatile("" ::: "memory");                          \
  } while (0)

// geometry index of the phase after `ph` if it needs a staged activation vector, else -1
__device__ __forceinline__ int stage_kind_after(int ph, int n_ph, int L) {
  if (ph + 1 >= n_ph) return -1;
  const int nk = ph + 1 < 6 * L ? (ph + 1) % 6 : PH_LM_HEAD;
  return nk == PH_QKV ? G_QKV : nk == PH_O ? G_O : nk == PH_GATE_UP ? G_GATE_UP : nk == PH_DOWN ? G_DOWN : nk == PH_LM_HEAD ? G_LM_HEAD : -1;
}

// ---- compute waves: nothing but the weight stream, the FMAs and s_barriers -------------------------------------------------
// Barrier schedule (must match helper_main exactly): 2 at entry; per phase A, B and - when the next phase stages a vector - 2
// more; 1 in the greedy tail of workgroup 0.
__device__ __forceinline__ void compute_main(const MegaParams& p, const int b, const int wave, const int lane, const f32x4* xs4,
                                             float* out_s) {
  const int L = p.n_layers;
  const int n_ph = 6 * L + 1;
  u32x4 buf[kDepth][2];
  WDesc dn = weight_desc(p, 0, b, wave);        // the phase being ISSUED (at most one weight phase ahead of the FMAs)
  int it = 0, ic = 0, ie = 0;                   // its cursor: task ordinal, chunk, element
  int wph = 0;
  auto issue = [&](int j) __attribute__((always_inline)) {
    const bool valid = ie < dn.n_elems;
    const int task = dn.task0 + it * dn.nact;
    int r0 = first_row(dn, task), r1 = r0 + (dn.swiglu ? 16 : 1);
    if (!valid || r0 >= dn.n) r0 = 0;           // always a finite, in-bounds address: dummies are never used or meet x = 0
    if (!valid || r1 >= dn.n) r1 = 0;
    const int col = ic * 512 + lane * 8;
    const int off = (valid && col < dn.k) ? col : 0;
    buf[j][0] = __builtin_nontemporal_load((gw_ptr)(dn.w + (size_t)r0 * dn.k + off));
    buf[j][1] = __builtin_nontemporal_load((gw_ptr)(dn.w + (size_t)r1 * dn.k + off));
    ++ie;
    if (++ic == dn.nchunks) { ic = 0; ++it; }
  };
#pragma unroll
  for (int j = 0; j < kDepth; ++j) issue(j);
  MEGA_LDS_SYNC();
  MEGA_LDS_SYNC();                              // layer 0's normalised input is staged
  for (int ph = 0; ph < n_ph; ++ph) {
    const int kind = ph < 6 * L ? ph % 6 : PH_LM_HEAD;
    if (kind != PH_ATTN && kind != PH_COMBINE) {
      const int nchunks = dn.nchunks, n_elems = dn.n_elems, swiglu = dn.swiglu, lt0 = dn.lt0, nact = dn.nact;
      int ct = 0, cc = 0, ce = 0;
      float acc0 = 0.f, acc1 = 0.f;
      const int nblk = n_elems > 0 ? (n_elems + kDepth - 1) / kDepth : 1;
      for (int blk = 0; blk < nblk; ++blk) {
        if (blk == nblk - 1) {                  // everything of this phase has been issued: run ahead into the next weight phase
          ++wph;
          dn = weight_desc(p, wph, b, wave);
          it = 0; ic = 0; ie = 0;
        }
#pragma unroll
        for (int j = 0; j < kDepth; ++j) {
          if (ce < n_elems) {
            const f32x4 xa = xs4[cc * 128 + lane], xb = xs4[cc * 128 + 64 + lane];
            acc0 = dot8(buf[j][0], xa, xb, acc0);
            acc1 = dot8(buf[j][1], xa, xb, acc1);
            ++ce;
            if (++cc == nchunks) {              // a row pair is complete
              const float r0 = wave_sum(acc0), r1 = wave_sum(acc1);
              const int lt = lt0 + ct * nact;
              if (lane == 0) {
                if (swiglu) out_s[lt] = silu_f(r0) * r1;
                else { out_s[2 * lt] = r0; out_s[2 * lt + 1] = r1; }
              }
              acc0 = 0.f; acc1 = 0.f; cc = 0; ++ct;
            }
          }
          issue(j);
        }
      }
    }
    MEGA_LDS_SYNC();                            // A
    MEGA_LDS_SYNC();                            // B
    if (stage_kind_after(ph, n_ph, L) >= 0) {
      MEGA_LDS_SYNC();
      MEGA_LDS_SYNC();                          // C: xs4 holds this phase's successor's input
    }
  }
  if (p.greedy_tail && b == 0) MEGA_LDS_SYNC();
}

// ---- helper waves: staging, attention, publishing, the grid barrier ------------------------------------------------------
__device__ __forceinline__ void helper_main(const MegaParams& p, const int b, const int wave, const int lane, const int htid, char* smem,
                                            f32x4* xs4, float* out_s, float* red, long long* tok_s) {
  const int hw = wave - kCompute;
  MegaSync* sync = p.sync;
  const int grp = b & 7;
  const unsigned n_groups = p.nwg < 8 ? p.nwg : 8;
  const unsigned grp_size = (p.nwg - grp + 7) >> 3;
  const int L = p.n_layers;
  const int n_ph = 6 * L + 1;
  unsigned epoch = 0;
  bool alive = true;
  {
    const MegaGeom g = p.geom[G_QKV];
    stage_sumsq(p.x, g.k, g.vthreads, hw, lane, red);
    MEGA_LDS_SYNC();
    stage_write(p.x, (const float*)reinterpret_cast<const MegaLayerDev*>(p.layers)[0].input_norm, g.k, (g.k + 511) >> 9, g.vthreads, p.eps, htid, red, xs4);
    MEGA_LDS_SYNC();
  }
  for (int ph = 0; ph < n_ph; ++ph) {
    const int layer = ph < 6 * L ? ph / 6 : L - 1;
    const int kind = ph < 6 * L ? ph - layer * 6 : PH_LM_HEAD;
    const MegaLayerDev* ML = reinterpret_cast<const MegaLayerDev*>(p.layers) + layer;
    const bool weight_phase = kind != PH_ATTN && kind != PH_COMBINE;
    if (!weight_phase) {
      AttnParams ap;
      ap.qkv = p.qkv; ap.kc = (float*)ML->kc; ap.vc = (float*)ML->vc; ap.out = p.attn; ap.part_ml = p.part_ml; ap.part_o = p.part_o;
      ap.pos0_dev = p.pos_dev; ap.pos0 = 0; ap.t = 1; ap.n_q = p.n_q; ap.n_kv = p.n_kv; ap.max_ctx = p.max_ctx;
      ap.n_splits = p.n_splits; ap.q_norm_w = (const float*)ML->q_norm; ap.k_norm_w = (const float*)ML->k_norm; ap.cos_tab = p.cos_tab; ap.sin_tab = p.sin_tab;
      ap.eps = p.eps; ap.seq_stride = 0; ap.table = p.kv_table; ap.log_block = p.kv_log_block; ap.table_stride = 0;
      ap.out_hi = nullptr; ap.out_lo = nullptr;
      if (kind == PH_ATTN) {
        float* scratch = reinterpret_cast<float*>(smem) + hw * (kMaxGroup * kHeadDim + 2 * kHeadDim);
        const int n_items = p.n_kv * p.n_splits;
        for (int i = b + p.nwg * hw; i < n_items; i += p.nwg * kHelpers)
          attn_decode_wave<true>(ap, i % p.n_kv, i / p.n_kv, 0, lane, scratch, scratch + kMaxGroup * kHeadDim,
                                 scratch + kMaxGroup * kHeadDim + kHeadDim);
      } else {
        const int n_items = p.n_q * 2;
        for (int i = b + p.nwg * hw; i < n_items; i += p.nwg * kHelpers)
          attn_combine_wave<true>(ap, i >> 1, 0, (i & 1) * 64 + lane, lane);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its write-through stores
    }
    MEGA_LDS_SYNC();                                          // A: the workgroup's part of the phase is done

    if (wave == kMaster) {                                    // publish + grid barrier
      if (weight_phase) {
        const int gk = kind == PH_QKV ? G_QKV : kind == PH_O ? G_O : kind == PH_GATE_UP ? G_GATE_UP : kind == PH_DOWN ? G_DOWN : G_LM_HEAD;
        const MegaGeom g = p.geom[gk];
        int nt = g.tasks - b * g.tpw;
        nt = nt > g.tpw ? g.tpw : (nt < 0 ? 0 : nt);
        if (g.swiglu) {                                       // gate_up: one value per task
          for (int i = lane; i < nt; i += 64) wt_store1(p.act + (size_t)b * g.tpw + i, out_s[i]);
        } else {
          const float* bias = kind == PH_QKV ? (const float*)ML->qkv_bias : nullptr;
          float* dst = kind == PH_QKV ? p.qkv : kind == PH_LM_HEAD ? p.logits : p.x;
          const bool resid = kind == PH_O || kind == PH_DOWN;
          float best = -INFINITY;
          int bi = 0x7fffffff;
          for (int i = lane; i < 2 * nt; i += 64) {
            const int row = b * g.tpw * 2 + i;
            if (row < g.n) {
              float v = out_s[i];
              if (bias) v += bias[row];
              if (resid) v = dst[row] + v;
              wt_store1(dst + row, v);
              if (v > best) { best = v; bi = row; }            // ascending rows within a lane: '>' keeps the first
            }
          }
          if (kind == PH_LM_HEAD && p.greedy_tail) {          // this workgroup's (max logit, first index): torch.argmax's tie rule
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
              const float ov = __shfl_xor(best, o, 64);
              const int oi = __shfl_xor(bi, o, 64);
              if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0)
              __hip_atomic_store(p.argmax_pairs + b, ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bi,
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      ++epoch;
      if (lane == 0 && alive) alive = grid_barrier(sync, epoch, grp, grp_size, n_groups);
      alive = __builtin_amdgcn_readfirstlane((int)alive) != 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE buffer_inv sc1 per CU: later plain loads see the other CUs' stores
    }
    MEGA_LDS_SYNC();                                          // B: everybody's results of this phase are visible

    const int gk = stage_kind_after(ph, n_ph, L);
    if (gk >= 0) {                                            // stage the next weight phase's input
      const MegaGeom g = p.geom[gk];
      const MegaLayerDev* NL = reinterpret_cast<const MegaLayerDev*>(p.layers) + (ph + 1 < 6 * L ? (ph + 1) / 6 : L - 1);
      const float* src = gk == G_O ? p.attn : gk == G_DOWN ? p.act : p.x;
      const float* nw = gk == G_QKV ? (const float*)NL->input_norm : gk == G_GATE_UP ? (const float*)NL->post_norm : gk == G_LM_HEAD ? p.final_norm : nullptr;
      if (nw) stage_sumsq(src, g.k, g.vthreads, hw, lane, red);
      MEGA_LDS_SYNC();
      stage_write(src, nw, g.k, (g.k + 511) >> 9, g.vthreads, p.eps, htid, red, xs4);
      MEGA_LDS_SYNC();                                        // C: the compute waves may read xs4
    }
  }

  // greedy tail: token, decode-loop state, next input embedding (workgroup 0)
  if (p.greedy_tail && b == 0) {
    if (wave == kMaster) {
      float best = -INFINITY;
      long long bi = 0x7fffffffffffffffLL;
      for (int i = lane; i < p.nwg; i += 64) {
        const unsigned long long pr = p.argmax_pairs[i];
        const float v = __uint_as_float((unsigned)(pr >> 32));
        const long long idx = (long long)(unsigned)pr;
        if (v > best || (v == best && idx < bi)) { best = v; bi = idx; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const long long oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (lane == 0) {
        const long long tok = bi + p.vocab_offset;
        if (p.token_dev) *p.token_dev = tok;
        if (p.token_logit_dev) *p.token_logit_dev = best;
        if (p.out_tokens && p.step_dev) p.out_tokens[*p.step_dev] = tok;
        if (p.step_dev) *p.step_dev += 1;
        if (p.pos_dev && *p.pos_dev >= 0) *p.pos_dev += 1;
        tok_s[0] = tok;
      }
    }
    MEGA_LDS_SYNC();
    const long long id = tok_s[0] - p.embed_offset;
    for (int k = htid * 4; k < p.hidden; k += kHelpers * 64 * 4) {
      f32x4 f = {0.f, 0.f, 0.f, 0.f};
      if (id >= 0 && id < p.embed_rows) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(p.embed + (size_t)id * p.hidden + k);
        f = (f32x4){bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y)};
      }
      *reinterpret_cast<f32x4*>(p.x + k) = f;
    }
  }
}

__global__ __launch_bounds__(kMegaThreads) void decode_mega_kernel(MegaParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* xs4 = reinterpret_cast<f32x4*>(smem);                                   // staged activation vector / attention scratch
  float* out_s = reinterpret_cast<float*>(smem + p.xs_bytes);                    // this workgroup's results of the phase
  float* red = out_s + kOutSlots;                                                // RMSNorm partial sums (<= 16 virtual waves)
  long long* tok_s = reinterpret_cast<long long*>(red + 32);                     // greedy tail: the token, for the embedding
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (__hip_atomic_load(&p.sync->status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;     // a previous step hung
  if (wave < kCompute) compute_main(p, blockIdx.x, wave, lane, xs4, out_s);
  else helper_main(p, blockIdx.x, wave, lane, tid - kCompute * 64, smem, xs4, out_s, red, tok_s);
}

// ---- host ---------------------------------------------------------------------------------------------------------------
static int pick_nact(int tpw) {
  for (int w = kCompute; w >= 5; --w)
    if (tpw % w == 0) return w;
  return tpw < kCompute ? (tpw > 0 ? tpw : 1) : kCompute;
}

size_t mega_state_bytes(int n_layers, int nwg) {        // MegaSync | argmax pairs [nwg] (64-byte padded) | MegaLayer [n_layers]
  return sizeof(MegaSync) + (((size_t)nwg * 8 + 63) / 64) * 64 + (size_t)n_layers * sizeof(MegaLayer) + 256;
}

int mega_lds_bytes(const MegaHost& h) {
  int kmax = 0;
  for (int i = 0; i < 5; ++i) kmax = h.geom[i].k > kmax ? h.geom[i].k : kmax;
  int xs = ((kmax + 511) / 512) * 512 * 4;
  const int attn = kHelpers * (kMaxGroup * kHeadDim + 2 * kHeadDim) * 4;
  if (xs < attn) xs = attn;
  return xs;
}

// Fill the geometry for the decoder's shapes.  Returns false when a shape does not fit the kernel (caller keeps the old path).
bool mega_plan(MegaHost* h, int hidden, int n_q, int n_kv, int inter, int64_t vocab_local, int cus) {
  const int qkv_n = (n_q + 2 * n_kv) * kHeadDim;
  const int shapes[5][3] = {{qkv_n, hidden, CHATTS_EPI_NONE}, {hidden, n_q * kHeadDim, CHATTS_EPI_RESID},
                            {2 * inter, hidden, CHATTS_EPI_SWIGLU}, {hidden, inter, CHATTS_EPI_RESID},
                            {(int)vocab_local, hidden, CHATTS_EPI_NONE}};
  h->nwg = cus;
  for (int i = 0; i < 5; ++i) {
    MegaGeom& g = h->geom[i];
    g.n = shapes[i][0]; g.k = shapes[i][1]; g.swiglu = shapes[i][2] == CHATTS_EPI_SWIGLU;
    if (g.k % 8 != 0 || g.n % 2 != 0 || (g.swiglu && g.n % 32 != 0)) return false;
    g.tasks = g.swiglu ? g.n / 2 : (g.n + 1) / 2;
    g.tpw = (g.tasks + h->nwg - 1) / h->nwg;
    g.nact = pick_nact(g.tpw);
    int nw = 4, occ = 0;
    gemv_default_geometry(g.n, g.k, shapes[i][2], cus, &nw, &occ);
    g.vthreads = nw * 64;
    if (2 * g.tpw > kOutSlots) return false;
    if ((size_t)((g.k + 511) / 512) * 512 * 4 > 120 * 1024) return false;
  }
  return true;
}

int mega_launch(const MegaParams& p, const MegaHost& h, hipStream_t s) {
  static bool attr_set = false;
  const int lds_used = mega_lds_bytes(h) + (kOutSlots + 32 + 16) * 4;
  int lds = lds_used < 96 * 1024 ? 96 * 1024 : lds_used;       // > half of a CU's LDS: exactly one workgroup per CU
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_mega_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CHATTS_REQUIRE(e == hipSuccess, CHATTS_E_LAUNCH, "decode_mega: cannot reserve LDS: %s", hipGetErrorString(e));
    attr_set = true;
  }
  // counters of the grid barrier: zeroed on the stream before every launch (a memset node under graph capture); the sticky
  // status word behind them is not touched
  const hipError_t e = hipMemsetAsync(p.sync, 0, offsetof(MegaSync, status), s);
  CHATTS_REQUIRE(e == hipSuccess, CHATTS_E_LAUNCH, "decode_mega: memset: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(decode_mega_kernel, dim3(h.nwg), dim3(kMegaThreads), lds, s, p);
  CHATTS_CHECK_LAUNCH("decode_mega");
  return CHATTS_OK;
}

}  // namespace chatts
