// decode_mega.hip - ONE persistent launch per generated token (batch 1, tensor_parallel_size 1, bf16 weights).
//
// What it replaces: chatts_decoder_decode_step's chain of 6 kernels per layer (qkv GEMV, attention, combine, o GEMV, gate_up
// GEMV, down GEMV) + lm_head + argmax + embedding = 291 launches for ChatTS-14B.  Those kernels stream at 7.1 TB/s while they
// stream, but every one of them pays ~4.2 us of ramp + drain + boundary during which HBM is idle (round-2 fit: t = 4.2 us +
// bytes / 7.09 TB/s), and the attention island streams nothing: 31 % of a layer.  The weights do not depend on the activations,
// so here the weight stream never stops:
//
//   * one workgroup per CU, resident for the whole step; 8 COMPUTE waves stream weights, 4 HELPER waves do everything else;
//   * a compute wave owns a fixed list of row-pair tasks per projection and walks them as one stream of ELEMENTS (2 rows x 512
//     columns = two 16-byte non-temporal loads per lane) that always runs kDepth elements ahead of the FMAs - across task
//     boundaries and across PHASE boundaries: while the grid barrier of phase p completes and the helpers stage the next
//     activation vector into LDS, the first 16 KB per wave of phase p + 1 are already in flight or in registers (the o_proj
//     weights arrive during the attention island).  Registers, not LDS, are the prefetch buffer (up to 8 waves x 32 KB per CU);
//   * phases meet at an XCD-sharded grid barrier (per-group arrival counters, one top counter, per-group generation words;
//     MI355X_MICROARCH.md "barrier-xcd"); data crosses workgroups by write-through (sc1) stores, a drained vmcnt, a relaxed
//     flag, ONE agent-scope acquire per CU, then plain loads (cdna_hip_programming.md Guideline 16, recipe R1);
//   * only ONE lane per workgroup ever polls (the master helper wave); every other wave waits in s_barrier, which costs no issue slots;
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

// 12 waves per workgroup = 3 per SIMD -> 168 VGPRs each: room for a 16-element-deep stream in the compute waves and for the
// attention / staging register sets of the helpers without spills (16 waves x 128 VGPRs spilled hundreds of values)
constexpr int kMegaThreads = 768;
constexpr int kCompute = 8;           // waves 0..7
constexpr int kHelpers = 4;           // waves 8..11
constexpr int kMaster = 11;           // publishes the workgroup's results and runs the grid barrier
constexpr int kOutSlots = 1024;       // floats of per-workgroup results (lm_head: 2 x 297)
constexpr unsigned kSpinLimit = 1u << 19;

enum { PH_QKV = 0, PH_ATTN = 1, PH_COMBINE = 2, PH_O = 3, PH_GATE_UP = 4, PH_DOWN = 5, PH_LM_HEAD = 6 };
enum { G_QKV = 0, G_O = 1, G_GATE_UP = 2, G_DOWN = 3, G_LM_HEAD = 4 };

// ---- grid barrier -----------------------------------------------------------------------------------------------------
// Monotonic counters, zeroed by a memset node before every launch.  epoch = 1, 2, ...  ONE lane per workgroup calls this after
// its workgroup's write-through stores have been drained (s_waitcnt vmcnt(0)).
// Two fabric hops: lane 0 adds one to its group's counter (b & 7: observed to be the XCD; 32 arrivals per word instead of 256 on
// one) WITHOUT waiting for the returned value, then lanes 0..7 of the same wave poll the eight counters until every group has
// counted all of its workgroups for this epoch.  (The hierarchical form - group leader -> top counter -> generation word - costs
// four dependent hops; measured 2.5-3.6 us per barrier in this kernel.)
__device__ __forceinline__ bool grid_barrier(MegaSync* s, const unsigned epoch, const int grp, const int nwg, const int lane) {
  if (lane == 0) __hip_atomic_fetch_add(&s->grp_count[grp * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned n_groups = nwg < 8 ? nwg : 8;
  const unsigned want = lane < (int)n_groups ? (unsigned)((nwg - lane + 7) >> 3) * epoch : 0u;
  for (unsigned spins = 0;; ++spins) {
    unsigned v = ~0u;
    if (lane < (int)n_groups) v = __hip_atomic_load(&s->grp_count[lane * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__all(v >= want)) return true;
    if (spins > kSpinLimit) {
      if (lane == 0) __hip_atomic_fetch_or(&s->status[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
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
  int ntw;                 // row-pair tasks of this wave in the phase
  int n_elems;             // ntw x nchunks
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
  d.ntw = ntw;
  d.n_elems = ntw * d.nchunks;
  return d;
}

__device__ __forceinline__ int first_row(const WDesc& d, int task) {      // second row: + 1 (plain) or + 16 (gate / up pair)
  return d.swiglu ? (task >> 4) * 32 + (task & 15) : task * 2;
}

// ---- activation staging (helper waves) ----------------------------------------------------------------------------------
// The next phase's input vector -> LDS in the GEMV's permuted layout ([chunk][half][lane] float4: conflict-free ds_read_b128),
// optionally RMS-normalised.  Three steps with a workgroup barrier between them (the compute waves just pass the barriers):
//   1  every helper thread loads ITS float4s (index htid + 384 m, m < kStageMax) in one round trip - the vector was written by
//      other CUs a moment ago, so a load costs a fabric round trip and nine dependent ones were the old 5 us - and, when a norm
//      follows, parks them in LDS in natural order;
//   2  the sum of squares, formed exactly like the stand-alone gemv_ldsx_kernel forms it with `vthreads` threads: virtual thread
//      vt accumulates elements 4 vt + 4 vthreads j, each virtual wave is reduced by wave_sum, wave results are added in order;
//   3  x * rstd * weight from the registers into the permuted layout.
constexpr int kStageMax = 14;         // float4 per helper thread: K <= 14 * 256 * 4 = 14336 (checked by mega_plan)
constexpr int kStageHid = 8;          // ... of a [hidden] / [n_q * 128] vector: <= 8 * 256 * 4 = 8192 (checked by mega_plan)

template <int NM, bool NORM>
struct StageRegs {
  f32x4 v[NM];
  f32x4 g[NORM ? NM : 1];
};

template <int NM, bool NORM>
__device__ __forceinline__ void stage_load(const float* src, const float* norm_w, int K, int htid, StageRegs<NM, NORM>& r, f32x4* raw4) {
  const int nk4 = K >> 2;
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int q = htid + kHelpers * 64 * m;
    const int qc = q < nk4 ? q : 0;             // clamped, always in bounds: every load is issued unconditionally, masked afterwards
    r.v[m] = *reinterpret_cast<const f32x4*>(src + (size_t)qc * 4);
    if (NORM) r.g[m] = *reinterpret_cast<const f32x4*>(norm_w + (size_t)qc * 4);
  }
  if (NORM) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int q = htid + kHelpers * 64 * m;
      if (q < nk4) raw4[q] = r.v[m];
    }
  }
}

__device__ __forceinline__ void stage_sumsq(const f32x4* raw4, int K, int vthreads, int hw, int lane, float* red) {
  const int nvw = vthreads >> 6;
  const int nk4 = K >> 2;
  for (int vw = hw; vw < nvw; vw += kHelpers) {
    float ss = 0.f;
    for (int q = vw * 64 + lane; q < nk4; q += vthreads) ss = sumsq4(ss, raw4[q]);
    ss = wave_sum(ss);
    if (lane == 0) red[vw] = ss;
  }
}

template <int NM, bool NORM>
__device__ __forceinline__ void stage_write(int K, int nchunks, int vthreads, float eps, int htid, const StageRegs<NM, NORM>& r,
                                            const float* red, f32x4* xs4) {
  float rstd = 1.f;
  if (NORM) {
    float t = 0.f;
    const int nvw = vthreads >> 6;
    for (int i = 0; i < nvw; ++i) t += red[i];
    rstd = rsqrtf(t / (float)K + eps);
  }
  const int nk4 = K >> 2;
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int q = htid + kHelpers * 64 * m;
    if (q < nchunks * 128) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < nk4) {
        v = r.v[m];
        if (NORM) {
          const f32x4 g = r.g[m];
          v.x = g.x * (v.x * rstd); v.y = g.y * (v.y * rstd); v.z = g.z * (v.z * rstd); v.w = g.w * (v.w * rstd);
        }
      }
      const int k4 = q * 4;
      const int chunk = k4 >> 9, within = k4 & 511;
      xs4[chunk * 128 + ((within >> 2) & 1) * 64 + (within >> 3)] = v;
    }
  }
}

// ---- synchronisation inside a workgroup --------------------------------------------------------------------------------
// No s_barrier after the entry: a hardware barrier needs EVERY wave, and the compute waves must never stand in one while there
// is room in their registers for more of the weight stream (loads are only in flight while somebody is issuing them: what a
// wave issued before it parked drains in ~2 us, then HBM idles for the rest of the phase edge).  Three LDS words instead:
//   flags[0]  results ready: every compute wave adds one when its row pairs of the phase are in out_s;
//   flags[1]  activation ready: set to the ordinal of the staged vector when xs4 holds the next weight phase's input;
//   flags[2]  the helpers' own barrier (four waves, monotonic counter).
__device__ __forceinline__ unsigned lds_load(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_wait_ge(unsigned* p, unsigned want) {
  while (lds_load(p) < want) __builtin_amdgcn_s_sleep(1);
}
// all four helper waves call this the same number of times; gen counts the calls
__device__ __forceinline__ void hsync(unsigned* flags, unsigned& gen, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's LDS writes are done
  ++gen;
  if (lane == 0) {
    __hip_atomic_fetch_add(&flags[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    lds_wait_ge(&flags[2], gen * kHelpers);
  }
  asm volatile("" ::: "memory");
}

// geometry index of the phase after `ph` if it needs a staged activation vector, else -1
__device__ __forceinline__ int stage_kind_after(int ph, int n_ph, int L) {
  if (ph + 1 >= n_ph) return -1;
  const int nk = ph + 1 < 6 * L ? (ph + 1) % 6 : PH_LM_HEAD;
  return nk == PH_QKV ? G_QKV : nk == PH_O ? G_O : nk == PH_GATE_UP ? G_GATE_UP : nk == PH_DOWN ? G_DOWN : nk == PH_LM_HEAD ? G_LM_HEAD : -1;
}

// ---- compute waves: nothing but the weight stream and the FMAs ---------------------------------------------------------
// Per weight phase: wait for the staged activation (flags[1]); blocks of kDepth elements - FMAs on slot j, then the load of the
// element kDepth further on into slot j - and a last block that only consumes; results to out_s, one add to flags[0]; then,
// WITHOUT waiting for anybody, the first kDepth elements of the next weight phase are issued.  That last step is where the
// persistent kernel earns its keep: it happens while the master publishes, the grid barrier completes and the helpers stage the
// next vector (and, after qkv, during the whole attention island), and the issue itself is paced by the memory system (a load
// instruction waits while the CU's request queue is full), so HBM keeps streaming through the edge for as long as there are
// registers to fill: kDepth x 2 KB per wave.
//
// The per-element instruction count decides whether 8 waves per CU keep up with HBM: an element costs two loads through a
// wave-uniform row pointer (SGPR pair) + a per-lane byte offset that grows by 1 KB, two ds_read_b128, 24 VALU for the two dot
// products and a handful of scalar ops (~50 instructions; ~0.17 us per element when nothing else paces it).
template <int kDepth>
__device__ __forceinline__ void compute_main(const MegaParams& p, const int b, const int wave, const int lane, const f32x4* xs4,
                                             float* out_s, unsigned* flags, unsigned long long* prof_s) {
  const int n_wph = 4 * p.n_layers + 1;
  const unsigned lane16 = (unsigned)lane * 16u;
  u32x4 buf[kDepth][2];
  // ---- issue side: one weight phase ahead of the FMAs while in the edge
  WDesc dn = weight_desc(p, 0, b, wave);
  int it = 0, ic = 0;                            // task ordinal (dn.ntw and beyond: padding), chunk within the task
  unsigned voff = lane16;                        // lane * 16 + chunk * 1024: byte offset into both rows
  const char* row0 = nullptr;
  const char* row1 = nullptr;
  auto next_rows = [&]() __attribute__((always_inline)) {
    // beyond the wave's tasks (the padding of the last block): a row of its own per wave - in bounds, finite, never used, and
    // not one address for the whole chip (thousands of waves padding on row 0 of a matrix hammer a single L2 channel)
    const int task = dn.task0 + it * dn.nact;
    int r0 = first_row(dn, task), r1 = r0 + (dn.swiglu ? 16 : 1);
    const int pad_row = ((b * kCompute + wave) * 2) % dn.n;
    if (it >= dn.ntw || r0 >= dn.n) r0 = pad_row;
    if (it >= dn.ntw || r1 >= dn.n) r1 = pad_row;
    row0 = reinterpret_cast<const char*>(dn.w) + (size_t)r0 * dn.k * 2;
    row1 = reinterpret_cast<const char*>(dn.w) + (size_t)r1 * dn.k * 2;
  };
  next_rows();
  auto issue = [&](int j) __attribute__((always_inline)) {
    const unsigned vo = voff < (unsigned)dn.k * 2u ? voff : 0u;           // ragged K: lanes past the row's end re-read its head (x is 0 there)
    buf[j][0] = __builtin_nontemporal_load((gw_ptr)(row0 + vo));
    buf[j][1] = __builtin_nontemporal_load((gw_ptr)(row1 + vo));
    voff += 1024u;
    if (++ic == dn.nchunks) {
      ic = 0; ++it; voff = lane16;
      next_rows();
    }
  };
  // The loads must be ISSUED in slot order everywhere: the s_waitcnt the compiler places before a slot's FMAs counts the loads
  // issued after that slot's.  (Left to the scheduler, a prologue's loads came out in another order, the loop-entry state said
  // "slot 0 is the youngest", and every block began by draining the whole queue.)
#pragma unroll
  for (int j = 0; j < kDepth; ++j) {
    issue(j);
    __builtin_amdgcn_sched_barrier(0x86);       // VALU / SALU / LDS may be scheduled across; the loads keep their order
  }
  const char* xs_b = reinterpret_cast<const char*>(xs4);
  const bool prof_c = p.prof != nullptr && wave == 0 && lane == 0;
  for (int wph = 0; wph < n_wph; ++wph) {
    lds_wait_ge(&flags[1], (unsigned)wph + 1u);                  // xs4 holds this phase's input
    asm volatile("" ::: "memory");
    if (prof_c) prof_s[0] = __builtin_amdgcn_s_memtime();
    // ---- consume side of this phase
    const int nchunks = dn.nchunks, n_elems = dn.n_elems, swiglu = dn.swiglu, lt0 = dn.lt0, nact = dn.nact;
    int ct = 0, cc = 0;
    unsigned xoff = lane16;                     // lane * 16 + chunk * 2048: byte offset of this lane's first float4 of the chunk
    float acc0 = 0.f, acc1 = 0.f;
    auto consume = [&](int j) __attribute__((always_inline)) {
      const f32x4 xa = *reinterpret_cast<const f32x4*>(xs_b + xoff), xb = *reinterpret_cast<const f32x4*>(xs_b + xoff + 1024);
      acc0 = dot8(buf[j][0], xa, xb, acc0);
      acc1 = dot8(buf[j][1], xa, xb, acc1);
      xoff += 2048u;
      if (++cc == nchunks) {                    // a row pair is complete
        const float r0 = wave_sum(acc0), r1 = wave_sum(acc1);
        const int lt = lt0 + ct * nact;
        if (lane == 0) {
          if (swiglu) out_s[lt] = silu_f(r0) * r1;
          else { out_s[2 * lt] = r0; out_s[2 * lt + 1] = r1; }
        }
        acc0 = 0.f; acc1 = 0.f; cc = 0; xoff = lane16; ++ct;
      }
    };
    const int nblk = n_elems > 0 ? (n_elems + kDepth - 1) / kDepth : 1;
    for (int blk = 0; blk + 1 < nblk; ++blk) {  // every element of these blocks is real; each slot is refilled kDepth elements on
#pragma unroll
      for (int j = 0; j < kDepth; ++j) {
        consume(j);
        __builtin_amdgcn_sched_barrier(0x86);
        issue(j);
        __builtin_amdgcn_sched_barrier(0x86);
      }
    }
    {
      const int left = n_elems - (nblk - 1) * kDepth;             // 0 .. kDepth real elements; nothing is issued here
#pragma unroll
      for (int j = 0; j < kDepth; ++j) {
        if (j < left) consume(j);
        __builtin_amdgcn_sched_barrier(0x86);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the results are in out_s
    if (lane == 0) __hip_atomic_fetch_add(&flags[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (prof_c) prof_s[1] = __builtin_amdgcn_s_memtime();
    // ---- the edge: run ahead into the next weight phase
    dn = weight_desc(p, wph + 1, b, wave);
    it = 0; ic = 0; voff = lane16;
    next_rows();
    if (dn.n_elems > 0) {
#pragma unroll
      for (int j = 0; j < kDepth; ++j) {
        issue(j);
        __builtin_amdgcn_sched_barrier(0x86);
      }
    }
  }
}

// ---- helper waves: staging, attention, publishing, the grid barrier ------------------------------------------------------
// Written as straight-line phases per layer (not one loop over phase indices) so that every piece of wave state lives only as
// long as it is needed: the preloaded attention tile from the qkv phase to the attention phase, the staging registers within
// one staging step.  Every helper wave makes the same sequence of hsync calls.
struct HelperCtx {
  const MegaParams& p;
  const int b, wave;
  int lane, htid;           // re-laundered every layer (see helper_main): keeps per-lane address arithmetic inside the layer loop
  const int hw;
  char* smem;
  f32x4* xs4;
  float* out_s;
  float* red;
  f32x4* raw4;
  unsigned long long* prof_s;
  unsigned* flags;
  unsigned hgen;            // calls of hsync so far
  unsigned wdone;           // weight phases whose results the master has collected
  unsigned staged;          // activation vectors staged so far
  int grp;
  unsigned epoch;
  bool alive;
  int prof_slot;            // 0 / 1: this workgroup records stamps, -1: it does not
  int n_ph;
};

// end of phase `ph` of kind KIND: barrier A, the master publishes the workgroup's results (weight phases), grid barrier, barrier B
template <int KIND>
__device__ __forceinline__ void helper_phase_end(HelperCtx& c, const MegaLayerDev* ML, const int ph, unsigned long long* pr) {
  const MegaParams& p = c.p;
  const int b = c.b, lane = c.lane;
  float* out_s = c.out_s;
  constexpr bool weight_phase = KIND != PH_ATTN && KIND != PH_COMBINE;
  if (weight_phase) {                                         // A: every compute wave has put its row pairs into out_s
    ++c.wdone;
    if (c.wave == kMaster) {
      if (lane == 0) lds_wait_ge(&c.flags[0], c.wdone * kCompute);
      asm volatile("" ::: "memory");
    }
  } else {
    hsync(c.flags, c.hgen, lane);                             // A: every helper has stored (and drained) its attention results
  }
  if (p.prof != nullptr && c.wave == kMaster && lane == 0 && ph >= 6 && ph < 12)      // layer 1: every workgroup's finishing time
    p.prof[(size_t)2 * c.n_ph * 16 + (size_t)(ph - 6) * p.nwg + b] = __builtin_amdgcn_s_memtime();
  if (pr) {
    pr[1] = __builtin_amdgcn_s_memtime(); pr[6] = c.prof_s[0]; pr[7] = c.prof_s[1];
#pragma unroll
    for (int i = 0; i < 8; ++i) pr[8 + i] = c.prof_s[2 + i];
  }
  if (c.wave == kMaster) {
    if (weight_phase) {
      constexpr int gk = KIND == PH_QKV ? G_QKV : KIND == PH_O ? G_O : KIND == PH_GATE_UP ? G_GATE_UP : KIND == PH_DOWN ? G_DOWN : G_LM_HEAD;
      const MegaGeom g = p.geom[gk];
      int nt = g.tasks - b * g.tpw;
      nt = nt > g.tpw ? g.tpw : (nt < 0 ? 0 : nt);
      if (KIND == PH_GATE_UP) {                               // one value per task
        for (int i = lane; i < nt; i += 64) wt_store1(p.act + (size_t)b * g.tpw + i, out_s[i]);
      } else {
        const float* bias = KIND == PH_QKV ? (const float*)ML->qkv_bias : nullptr;
        float* dst = KIND == PH_QKV ? p.qkv : KIND == PH_LM_HEAD ? p.logits : p.x;
        constexpr bool resid = KIND == PH_O || KIND == PH_DOWN;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = lane; i < 2 * nt; i += 64) {
          const int row = b * g.tpw * 2 + i;
          if (row < g.n) {
            float v = out_s[i];
            if (bias) v += bias[row];
            if (resid) v = dst[row] + v;
            wt_store1(dst + row, v);
            if (v > best) { best = v; bi = row; }              // ascending rows within a lane: '>' keeps the first
          }
        }
        if (KIND == PH_LM_HEAD && p.greedy_tail) {            // this workgroup's (max logit, first index): torch.argmax's tie rule
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
    if (pr) pr[2] = __builtin_amdgcn_s_memtime();
    ++c.epoch;
    if (c.alive) c.alive = grid_barrier(p.sync, c.epoch, c.grp, p.nwg, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // ONE buffer_inv sc1 per CU: later plain loads see the other CUs' stores
    if (pr) pr[3] = __builtin_amdgcn_s_memtime();
  }
  hsync(c.flags, c.hgen, lane);                               // B: everybody's results of this phase are visible
  if (pr) pr[4] = __builtin_amdgcn_s_memtime();
}

// stage the input of the weight phase with geometry GK
template <int GK>
__device__ __forceinline__ void helper_stage(HelperCtx& c, const MegaLayerDev* NL) {
  const MegaParams& p = c.p;
  const MegaGeom g = p.geom[GK];
  const float* src = GK == G_O ? p.attn : GK == G_DOWN ? p.act : p.x;
  const float* nw = GK == G_QKV ? (const float*)NL->input_norm : GK == G_GATE_UP ? (const float*)NL->post_norm : GK == G_LM_HEAD ? p.final_norm : nullptr;
  constexpr bool norm = GK == G_QKV || GK == G_GATE_UP || GK == G_LM_HEAD;
  constexpr int NM = GK == G_DOWN ? kStageMax : kStageHid;
  StageRegs<NM, norm> sr;
  stage_load<NM, norm>(src, nw, g.k, c.htid, sr, c.raw4);
  if (norm) {
    hsync(c.flags, c.hgen, c.lane);
    stage_sumsq(c.raw4, g.k, g.vthreads, c.hw, c.lane, c.red);
    hsync(c.flags, c.hgen, c.lane);
  }
  stage_write<NM, norm>(g.k, (g.k + 511) >> 9, g.vthreads, p.eps, c.htid, sr, c.red, c.xs4);
  hsync(c.flags, c.hgen, c.lane);                             // every helper's part of xs4 is written
  ++c.staged;
  if (c.wave == kMaster && c.lane == 0)                       // C: the compute waves may read xs4
    __hip_atomic_store(&c.flags[1], c.staged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void helper_main(const MegaParams& p, const int b, const int wave, const int lane_in, const int htid_in, char* smem,
                                            f32x4* xs4, float* out_s, float* red, long long* tok_s, unsigned long long* prof_s,
                                            f32x4* raw4, unsigned* flags) {
  const int L = p.n_layers;
  const int grp = b & 7;
  HelperCtx c{p, b, wave, lane_in, htid_in, wave - kCompute, smem, xs4, out_s, red, raw4, prof_s, flags, 0u, 0u, 0u, grp, 0u, true,
              b == 0 ? 0 : (b == p.nwg / 2 + 3 ? 1 : -1), 6 * L + 1};
  const bool prof_m = p.prof != nullptr && c.prof_slot >= 0 && wave == kMaster && lane_in == 0;
  const MegaLayerDev* layers = reinterpret_cast<const MegaLayerDev*>(p.layers);
  auto stamps = [&](int ph) -> unsigned long long* {
    if (p.prof != nullptr && wave == kMaster && lane_in == 0 && ph >= 6 && ph < 12)      // layer 1: every workgroup's phase start
      p.prof[(size_t)2 * c.n_ph * 16 + (size_t)(6 + ph - 6) * p.nwg + b] = __builtin_amdgcn_s_memtime();
    unsigned long long* pr = prof_m ? p.prof + ((size_t)c.prof_slot * c.n_ph + ph) * 16 : nullptr;
    if (pr) pr[0] = __builtin_amdgcn_s_memtime();
    return pr;
  };
  helper_stage<G_QKV>(c, layers);                            // RMSNorm(x) for layer 0's qkv

  const int hw = c.hw;
  float* scratch = reinterpret_cast<float*>(smem) + hw * (kMaxGroup * kHeadDim + 2 * kHeadDim);
  // attention items: (kv head, key slot, half of the head group) - the group's query heads are independent, two waves share them
  const int G = p.n_q / p.n_kv;
  const int gsplit = (G + 1) >> 1;
  const int n_items = p.n_kv * p.n_splits * 2;
  const int i0 = b + p.nwg * hw;
  auto run_item = [&](const AttnParams& ap, int i, AttnTileRegs& t, int lane) __attribute__((always_inline)) {
    const int sub = i & 1, hs = i >> 1;
    const int g0 = sub ? gsplit : 0, gn = sub ? G - gsplit : gsplit;
    if (gn > 0)
      attn_decode_finish<true, (kMaxGroup + 1) / 2, false>(ap, hs % p.n_kv, hs / p.n_kv, 0, lane, t, scratch, scratch + kMaxGroup * kHeadDim,
                                                    scratch + kMaxGroup * kHeadDim + kHeadDim, g0, gn);
  };
  for (int layer = 0; layer < L; ++layer) {
    // The optimiser would hoist every per-lane address of every phase (hundreds of 64-bit values) out of this loop and then
    // spill them to scratch; making the lane ids opaque once per layer keeps that arithmetic - a few VALU ops - where it is used.
    int lane = lane_in, htid = htid_in;
    asm volatile("" : "+v"(lane), "+v"(htid));
    c.lane = lane; c.htid = htid;
    const MegaLayerDev* ML = layers + layer;
    AttnParams ap;
    ap.qkv = p.qkv; ap.kc = (float*)ML->kc; ap.vc = (float*)ML->vc; ap.out = p.attn; ap.part_ml = p.part_ml; ap.part_o = p.part_o;
    ap.pos0_dev = p.pos_dev; ap.pos0 = 0; ap.t = 1; ap.n_q = p.n_q; ap.n_kv = p.n_kv; ap.max_ctx = p.max_ctx;
    ap.n_splits = p.n_splits; ap.q_norm_w = (const float*)ML->q_norm; ap.k_norm_w = (const float*)ML->k_norm; ap.cos_tab = p.cos_tab; ap.sin_tab = p.sin_tab;
    ap.eps = p.eps; ap.seq_stride = 0; ap.table = p.kv_table; ap.log_block = p.kv_log_block; ap.table_stride = 0;
    ap.out_hi = nullptr; ap.out_lo = nullptr; ap.kv_round = 0;
    const int ph0 = layer * 6;
    unsigned long long* pr;
    {
      // ---- qkv: the helpers idle while the compute waves stream - fetch the K / V rows of this wave's attention item now (they do
      // not depend on this step's projections), so that after the barrier only q and the new row are a round trip away
      pr = stamps(ph0 + PH_QKV);
      AttnTileRegs pre;
      const bool pre_ok = i0 < n_items && attn_decode_preload(ap, (i0 >> 1) % p.n_kv, (i0 >> 1) / p.n_kv, 0, lane, pre);
      helper_phase_end<PH_QKV>(c, ML, ph0 + PH_QKV, pr);
      if (pr) pr[5] = pr[4];
      // ---- attention
      pr = stamps(ph0 + PH_ATTN);
      bool ok = pre_ok;
      for (int i = i0; i < n_items; i += p.nwg * kHelpers) {       // (one item per wave at ChatTS sizes: 8 x 64 x 2 <= 256 x 4)
        if (i != i0) ok = attn_decode_preload(ap, (i >> 1) % p.n_kv, (i >> 1) / p.n_kv, 0, lane, pre);
        if (ok) run_item(ap, i, pre, lane);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains its write-through stores
    helper_phase_end<PH_ATTN>(c, ML, ph0 + PH_ATTN, pr);
    if (pr) pr[5] = pr[4];
    // ---- combine
    pr = stamps(ph0 + PH_COMBINE);
    for (int i = b + p.nwg * hw; i < p.n_q * 2; i += p.nwg * kHelpers)
      attn_combine_wave<true>(ap, i >> 1, 0, (i & 1) * 64 + lane, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    helper_phase_end<PH_COMBINE>(c, ML, ph0 + PH_COMBINE, pr);
    helper_stage<G_O>(c, ML);
    if (pr) pr[5] = __builtin_amdgcn_s_memtime();
    // ---- o_proj
    pr = stamps(ph0 + PH_O);
    helper_phase_end<PH_O>(c, ML, ph0 + PH_O, pr);
    helper_stage<G_GATE_UP>(c, ML);
    if (pr) pr[5] = __builtin_amdgcn_s_memtime();
    // ---- gate_up
    pr = stamps(ph0 + PH_GATE_UP);
    helper_phase_end<PH_GATE_UP>(c, ML, ph0 + PH_GATE_UP, pr);
    helper_stage<G_DOWN>(c, ML);
    if (pr) pr[5] = __builtin_amdgcn_s_memtime();
    // ---- down_proj
    pr = stamps(ph0 + PH_DOWN);
    helper_phase_end<PH_DOWN>(c, ML, ph0 + PH_DOWN, pr);
    if (layer + 1 < L) helper_stage<G_QKV>(c, ML + 1);
    else helper_stage<G_LM_HEAD>(c, ML);
    if (pr) pr[5] = __builtin_amdgcn_s_memtime();
  }
  {
    unsigned long long* pr = stamps(6 * L);
    helper_phase_end<PH_LM_HEAD>(c, layers + (L - 1), 6 * L, pr);
    if (pr) pr[5] = __builtin_amdgcn_s_memtime();
  }

  // greedy tail: token, decode-loop state, next input embedding (workgroup 0)
  const int lane = lane_in, htid = htid_in;
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
    hsync(c.flags, c.hgen, lane);
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

template <int kDepth>
__global__ __launch_bounds__(kMegaThreads) void decode_mega_kernel(MegaParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* xs4 = reinterpret_cast<f32x4*>(smem);                                   // staged activation vector / attention scratch
  float* out_s = reinterpret_cast<float*>(smem + p.xs_bytes);                    // this workgroup's results of the phase
  float* red = out_s + kOutSlots;                                                // RMSNorm partial sums (<= 16 virtual waves)
  long long* tok_s = reinterpret_cast<long long*>(red + 32);                     // greedy tail: the token, for the embedding
  unsigned long long* prof_s = reinterpret_cast<unsigned long long*>(red + 36);  // profiling: compute wave 0's stamps [10]
  unsigned* flags = reinterpret_cast<unsigned*>(red + 56);                       // results ready | activation ready | helper barrier
  f32x4* raw4 = reinterpret_cast<f32x4*>(red + 64);                              // staging: the un-normalised vector, natural order
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (__hip_atomic_load(&p.sync->status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;     // a previous step hung
  if (tid < 4) flags[tid] = 0u;
  __syncthreads();                                            // the only hardware barrier of the step
  if (wave < kCompute) compute_main<kDepth>(p, blockIdx.x, wave, lane, xs4, out_s, flags, prof_s);
  else helper_main(p, blockIdx.x, wave, lane, tid - kCompute * 64, smem, xs4, out_s, red, tok_s, prof_s, raw4, flags);
}

// ---- host ---------------------------------------------------------------------------------------------------------------
// Every compute wave takes tasks (wave w: local tasks w, w + 8, ...): the longest wave is never longer than with fewer, evenly
// loaded waves (14 tasks: 2 either way; 54 tasks: 7 instead of 9), and no wave sits out a phase issuing padding loads.
static int pick_nact(int tpw) { return tpw < kCompute ? (tpw > 0 ? tpw : 1) : kCompute; }

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
    if (((g.k + 511) / 512) * 128 > (i == G_DOWN ? kStageMax : kStageHid) * kHelpers * 64) return false;      // one staging round per vector
    if ((size_t)((g.k + 511) / 512) * 512 * 4 > 120 * 1024) return false;
  }
  return true;
}

// Stream depth: the elements of a wave's share of a phase are padded to whole blocks of `depth`, and every padding element is
// two wasted 1-KB loads - pick, among the instantiated depths, the one that pads least over a layer (ties: the deeper one).
static int pick_depth(const MegaHost& h) {
  const int cands[4] = {16, 12, 10, 8};
  int best = 16;
  long best_pad = -1;
  for (int d : cands) {
    long pad = 0;
    for (int i = 0; i < 4; ++i) {
      const MegaGeom& g = h.geom[i];
      const int nch = (g.k + 511) / 512;
      for (int w = 0; w < g.nact; ++w) {             // wave w of a full workgroup: tasks w, w + nact, ...
        const int per_wave = (g.tpw - w + g.nact - 1) / g.nact * nch;
        pad += (per_wave + d - 1) / d * d - per_wave;
      }
    }
    if (best_pad < 0 || pad < best_pad) { best_pad = pad; best = d; }
  }
  return best;
}

template <int D>
static int mega_launch_depth(const MegaParams& p, const MegaHost& h, int lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_mega_kernel<D>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    CHATTS_REQUIRE(e == hipSuccess, CHATTS_E_LAUNCH, "decode_mega: cannot reserve LDS: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(decode_mega_kernel<D>, dim3(h.nwg), dim3(kMegaThreads), lds, s, p);
  CHATTS_CHECK_LAUNCH("decode_mega");
  return CHATTS_OK;
}

int mega_launch(const MegaParams& p, const MegaHost& h, hipStream_t s) {
  const int lds_used = mega_lds_bytes(h) + (kOutSlots + 64) * 4 + h.geom[G_QKV].k * 4 + 64;      // + the raw copy of a [hidden] vector
  const int lds = lds_used < 96 * 1024 ? 96 * 1024 : lds_used;       // > half of a CU's LDS: exactly one workgroup per CU
  // counters of the grid barrier: zeroed on the stream before every launch (a memset node under graph capture); the sticky
  // status word behind them is not touched
  const hipError_t e = hipMemsetAsync(p.sync, 0, offsetof(MegaSync, status), s);
  CHATTS_REQUIRE(e == hipSuccess, CHATTS_E_LAUNCH, "decode_mega: memset: %s", hipGetErrorString(e));
  static const int forced = getenv("CHATTS_MEGA_DEPTH") ? atoi(getenv("CHATTS_MEGA_DEPTH")) : 0;
  const int depth = (forced == 8 || forced == 10 || forced == 12 || forced == 16) ? forced : pick_depth(h);
  switch (depth) {
    case 8: return mega_launch_depth<8>(p, h, lds, s);
    case 10: return mega_launch_depth<10>(p, h, lds, s);
    case 12: return mega_launch_depth<12>(p, h, lds, s);
    default: return mega_launch_depth<16>(p, h, lds, s);
  }
}

}  // namespace chatts
