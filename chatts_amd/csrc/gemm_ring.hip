// gemm_ring.hip - the prefill projection (M >= 96 rows on pre-split bf16 hi / lo planes of A):  C = epilogue(A . W^T), bf16x2.
// Round 5 rebuild of the LDS-DMA kernel (gemm_dma_kernel, round 1-4).  What the round-5 timeline probe of that kernel showed
// (tools/gemm_probe.py, profiles/r5_gemm_probe_before.txt) and what this kernel does about it:
//
//  1. The K loop was LOADER-bound, not MFMA-bound.  With two 64 KB stages a stage can only be requested after the barrier that
//     retires its slot and must have landed by the next barrier: 16 pieces per loader wave at ~100 cycles of issue each, then the
//     drain (vmcnt(0)), once per K-step - 2400-2500 cycles per step whatever the clock, against 2048 cycles of MFMA (a ragged tile
//     with a quarter of the MFMAs still took 1500).  HERE: the ring holds FOUR half-stages (32 K-values each, rows of 64 bytes); a
//     half-stage is requested three half-steps before it is needed, the loaders wait with a COUNTED vmcnt (two younger half-stages
//     stay in flight across every barrier) and never drain.  One barrier per half-step, placed between the lo and the hi sweep.
//  2. Every workgroup paid dispatch + prologue (~4.5 us per tile) and the epilogues ran as chains of dependent memory round trips
//     (direct stores 14 us, residual epilogue 25 us per tile: one `s_waitcnt vmcnt(0)` per element).  HERE: 256 PERSISTENT
//     workgroups walk their units; the loader waves run ahead into the next unit while the compute waves store; the MFMA operands
//     are swapped (D = W . A^T), so a lane holds 4 consecutive output columns of one token and the epilogue is branch-free 16-byte
//     loads / stores selected once per unit.
//  3. Tiles did not fill rounds of workgroups: 798 rows = 6 tiles of 128 + a ragged one of 30 (its second wave row idle, DMA-bound);
//     o_proj / down_proj ran 420 units (split-K 3) as two rounds on 256 CUs with a third of the CU-time idle.  HERE: the 16-row
//     fragments of M are dealt EVENLY over T M-tiles (50 fragments: 9,9,8,8,8,8 or 8,7,7,7,7,7,7 or 6,6,6,6,6,5,5,5,5), a tile's
//     fragments evenly over the two wave rows, and (T, split-K) are chosen per shape so that the units fill whole rounds
//     (o / down: 6 x 20 x 2 = 240 units; qkv: 9 x 28 = 252; gate_up: 7 x 108 = 756 = 2.95 rounds).
//
// Geometry: unit = (K split, W panel of 256 rows, M-tile of f <= 10 fragments); 12 waves = 8 compute (2 x 4: wave row r holds
// ceil / floor of f / 2 fragments x 64 columns) + 4 loaders.  Half-stage slot (36 KB): A_hi [160][64 B] | A_lo | W [256][64 B]; 16-byte
// chunk c of row r lives at chunk c ^ (((r >> 3) & 1) << 1) (conflict-free for the ds_read_b128 fragment reads; applied on the SOURCE
// side of the DMA: lane l of a 16-row piece fetches global chunk (l & 3) ^ ((l >> 5) << 1)).
// Same products and the same per-element summation order over K as the kernels in gemm.hip at equal split-K.
#include <type_traits>

#include "gemm_common.h"
#include "ring_store.h"

namespace chatts {

constexpr int kRingBN = kRingPanel, kRingMaxF = 10, kRingSlots = 4;
constexpr int kRingAPlane = kRingMaxF * 16 * 64;             // 10 KB: one plane of a half-stage
constexpr int kRingWOff = 2 * kRingAPlane;
constexpr int kRingHalf = kRingWOff + kRingBN * 64;          // 36 KB
constexpr int kRingLds = kRingSlots * kRingHalf;             // 144 KB
constexpr int kRingThreads = 768, kRingCompute = 8, kRingLoaders = 4;


// SINGLE: the "bf16" speed mode - the lo plane is neither staged nor multiplied (activations rounded to bf16: NOT parity grade)
template <bool SINGLE>
__global__ __launch_bounds__(kRingThreads) void gemm_ring_kernel(GemmParams p, const uint16_t* __restrict__ a_hi,
                                                                 const uint16_t* __restrict__ a_lo, int ldp, RingGeom g) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this workgroup's units: XCD x (= blockIdx % 8, observed; speed only) owns a contiguous eighth of the unit sequence and its
  // workgroups take every wpx-th unit of it, so that at any time an XCD works on ~wpx consecutive units: the M-tiles of a few W panels
  // at the same K offset - each K-slice of a panel is fetched into that L2 once for all of them
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int share = g.units >> 3, rem = g.units & 7;
  const int ubeg = xcd * share + (xcd < rem ? xcd : rem), ucnt = share + (xcd < rem);
  if (local >= ucnt) return;                                   // padding workgroup (fewer units than workgroups): uniform exit
  const int uend = ubeg + ucnt, u0 = ubeg + local, ustep = g.wpx;
  int G = 0;                                                   // half-steps of this workgroup = barriers after the first
  for (int u = u0; u < uend; u += ustep) G += ring_unit(p, g, u).nh;

  if (wave >= kRingCompute) {
    // ---- loader wave L: pieces of 16 rows x 64 bytes (1 KB per wave instruction).  W pieces {L, L + 4, L + 8, L + 12}; of the unit's
    // A pieces q = 0 .. 2 f - 1 (q < f: hi plane fragment q, else lo plane fragment q - f) those with q % 4 == L.
    const int L = wave - kRingCompute;
    const int prow = lane >> 2, gchunk = (lane & 3) ^ ((lane >> 5) << 1);
    // tiled operands (chatts_tile_bf16): piece (16-row block b, half-stage t) is the 1 KB at ((b * K / 32) + t) * 1 KB, already in LDS
    // order - whole cache lines per request instead of 16 half lines (profiles/r6_feed_probe.txt)
    const bool wtiled = p.wt != nullptr, atiled = p.planes_tiled != 0;
    const int nkt = p.k >> 5, wstep = wtiled ? 1024 : 64, astep = atiled ? 1024 : 64;
    const char* src[9];
    int dst[9];
    int np = 0, nh_cur = 0, hh = 0, ucur = u0;
    auto setup = [&](int u) {
      const RingUnit r = ring_unit(p, g, u);
      nh_cur = r.nh;
      np = 0;
      const int n0 = r.panel * kRingBN;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        if (wtiled) {
          int rb = (n0 >> 4) + L + 4 * h;
          const int rbmax = ((p.n + 15) >> 4) - 1;
          if (rb > rbmax) rb = rbmax;
          src[h] = reinterpret_cast<const char*>(p.wt + ((size_t)rb * nkt + (r.kbeg >> 5)) * 512) + lane * 16;
        } else {
          int wr = n0 + (L + 4 * h) * 16 + prow;
          if (wr > p.n - 1) wr = p.n - 1;
          src[h] = reinterpret_cast<const char*>(p.w + (size_t)wr * p.ldw + r.kbeg + gchunk * 8);
        }
        dst[h] = kRingWOff + (L + 4 * h) * 1024;
      }
      np = 4;
      const int na = SINGLE ? r.f : 2 * r.f;
#pragma unroll
      for (int h = 0; h < 5; ++h) {
        const int q = L + 4 * h;
        if (q < na) {
          const int plane = q >= r.f, frag = plane ? q - r.f : q;
          if (atiled) {
            src[4 + h] = reinterpret_cast<const char*>((plane ? a_lo : a_hi) + ((size_t)((r.m0 >> 4) + frag) * nkt + (r.kbeg >> 5)) * 512) + lane * 16;
          } else {
            int am = r.m0 + frag * 16 + prow;
            if (am > p.m - 1) am = p.m - 1;
            src[4 + h] = reinterpret_cast<const char*>((plane ? a_lo : a_hi) + (size_t)am * ldp + r.kbeg + gchunk * 8);
          }
          dst[4 + h] = plane * kRingAPlane + frag * 1024;
          np = 5 + h;
        }
      }
    };
    int gi = 0;                                                // global index of the next half-stage to request
    auto issue = [&]() -> int {                                // request the next half-stage (if any); -> its piece count
      if (ucur >= uend) return 0;
      char* base = smem + (gi & (kRingSlots - 1)) * kRingHalf;
      const size_t koffw = (size_t)hh * wstep, koffa = (size_t)hh * astep;
#pragma unroll
      for (int h = 0; h < 9; ++h)
        if (h < np) __builtin_amdgcn_global_load_lds((gptr_t)(src[h] + (h < 4 ? koffw : koffa)), (lptr_t)(base + dst[h]), 16, 0, 0);
      const int n = np;
      ++gi;
      if (++hh == nh_cur) {
        ucur += ustep;
        hh = 0;
        if (ucur < uend) setup(ucur);
      }
      return n;
    };
    setup(u0);
    issue();
    int q1 = issue(), q2 = issue();
    ring_wait_vmcnt(q1 + q2);                                  // half-stage 0 landed
    __builtin_amdgcn_s_barrier();                              // ... published
    int q3 = issue();                                          // (slot 3 has never been read)
    for (int gh = 0; gh < G; ++gh) {
      ring_wait_vmcnt(q2 + q3);                                // half-stage gh + 1 landed; gh + 2, gh + 3 stay in flight
      __builtin_amdgcn_s_barrier();                            // publishes it; every read of half-stage gh has returned
      const int q4 = issue();                                  // gh + 4 -> the slot of gh
      q1 = q2; q2 = q3; q3 = q4;
    }
    (void)q1;
    return;
  }

  // ---- compute wave (wm, wn): wave row wm holds f0 (wm = 0) or f - f0 fragments of the tile, 64 columns wn * 64 ..
  const int wm = wave >> 2, wn = wave & 3;
  const int lane_off = (lane & 15) * 64 + ((((lane >> 4) ^ (((lane >> 3) & 1) << 1))) << 4);      // byte offset of this lane's chunk in a 16-row block
  __builtin_amdgcn_s_barrier();                                // half-stage 0 published
  int gh = 0;
  for (int u = u0; u < uend; u += ustep) {
    const RingUnit r = ring_unit(p, g, u);
    const int fml = wm ? r.f - r.f0 : r.f0;
    const int rowblk = wm ? r.f0 : 0;                          // first fragment of this wave within the tile
    if (fml == 0) {
      for (int h = 0; h < r.nh; ++h) __builtin_amdgcn_s_barrier();
      gh += r.nh;
      continue;
    }
    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto k_loop = [&](auto fml_c) {
      constexpr int FML = decltype(fml_c)::value;
      bf16x8_t wf[2][4], alo[FML], ahi[FML];
      const int a_off = rowblk * 1024 + lane_off, w_off = kRingWOff + wn * 4096 + lane_off;
      auto slot = [&](int h) { return smem + ((gh + h) & (kRingSlots - 1)) * kRingHalf; };
      auto read_w = [&](const char* base, bf16x8_t (&dst)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const bf16x8_t*>(base + w_off + j * 1024);
      };
      auto read_a = [&](const char* base, int plane, bf16x8_t (&dst)[FML]) {
#pragma unroll
        for (int i = 0; i < FML; ++i) dst[i] = *reinterpret_cast<const bf16x8_t*>(base + plane * kRingAPlane + a_off + i * 1024);
      };
      auto sweep = [&](const bf16x8_t (&af)[FML], const bf16x8_t (&wfr)[4]) {      // D = W . A^T: operands swapped
#pragma unroll
        for (int i = 0; i < FML; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[j], af[i], acc[i][j], 0, 0, 0);
      };
      // half-step h (operands in registers) with the fragments of h + 1 requested behind the barrier
      auto half = [&](int h, const bf16x8_t (&wcur)[4], bf16x8_t (&wnxt)[4], bool more) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!SINGLE) sweep(alo, wcur);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every fragment read of half-stage h has returned
        __builtin_amdgcn_s_barrier();                          // h + 1 is published; the loaders refill the slot of h
        if (more) {
          const char* nb = slot(h + 1);
          read_w(nb, wnxt);
          if constexpr (!SINGLE) read_a(nb, 1, alo);
        }
        __builtin_amdgcn_sched_barrier(0);
        sweep(ahi, wcur);
        __builtin_amdgcn_sched_barrier(0);
        if (more) read_a(slot(h + 1), 0, ahi);
      };
      {
        const char* b0 = slot(0);
        read_w(b0, wf[0]);
        if constexpr (!SINGLE) read_a(b0, 1, alo);
        read_a(b0, 0, ahi);
      }
      // (nh is even: K splits are multiples of 64; the last pair is peeled so that the loop body has no `more` test)
      int h = 0;
      for (; h + 2 < r.nh; h += 2) {
        half(h, wf[0], wf[1], true);
        half(h + 1, wf[1], wf[0], true);
      }
      half(h, wf[0], wf[1], true);
      half(h + 1, wf[1], wf[0], false);
    };
    switch (fml) {
      case 1: k_loop(std::integral_constant<int, 1>{}); break;
      case 2: k_loop(std::integral_constant<int, 2>{}); break;
      case 3: k_loop(std::integral_constant<int, 3>{}); break;
      case 4: k_loop(std::integral_constant<int, 4>{}); break;
      default: k_loop(std::integral_constant<int, 5>{}); break;
    }
    gh += r.nh;

    const int tok0 = r.m0 + rowblk * 16, fb0 = r.panel * kRingBN + wn * 64;
    const bool planes = p.c_hi != nullptr;
    if (!p.direct) ring_store<kStRaw, false>(p, acc, fml, tok0, fb0, lane, r.split);
    else if (p.epilogue == CHATTS_EPI_SWIGLU) {
      if (planes) ring_store<kStSwiglu, true>(p, acc, fml, tok0, fb0, lane, 0);
      else ring_store<kStSwiglu, false>(p, acc, fml, tok0, fb0, lane, 0);
    } else if (p.epilogue == CHATTS_EPI_RESID) {
      if (planes) ring_store<kStResid, true>(p, acc, fml, tok0, fb0, lane, 0);
      else ring_store<kStResid, false>(p, acc, fml, tok0, fb0, lane, 0);
    } else if (p.epilogue == CHATTS_EPI_GELU) {
      if (planes) ring_store<kStGelu, true>(p, acc, fml, tok0, fb0, lane, 0);
      else ring_store<kStGelu, false>(p, acc, fml, tok0, fb0, lane, 0);
    } else {
      if (planes) ring_store<kStNone, true>(p, acc, fml, tok0, fb0, lane, 0);
      else ring_store<kStNone, false>(p, acc, fml, tok0, fb0, lane, 0);
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// (T, sk) per shape: a cost model in cycles of the slowest workgroup - rounds x (K-steps x step + a per-unit constant) + the split-K
// epilogue launch.  Fitted to the probe build's numbers (profiles/r5_ring_probe_first.txt): a 64-deep step of a tile with f fragments
// takes ~235 f + 900 cycles (f = 6: 2330, 7-8: 2650-2680, 9: 3020 - the 16 f MFMAs per SIMD and the (32 f + 256) x 128 bytes of LDS-DMA
// overlap only partly), a unit costs ~9000 cycles of prologue + epilogue, a split-K epilogue ~6 us + its slab traffic at ~5 TB/s.
void ring_pick(int m, int n, int k, int cus, int force_t, int force_sk, RingGeom& g) {
  const int F = (m + 15) / 16, P = (n + kRingBN - 1) / kRingBN, nk = k / 64;
  g.F = F; g.P = P;
  const int slots = cus >= 8 ? (cus / 8) * 8 : 8;
  double best = 1e30;
  const int tmin = (F + kRingMaxF - 1) / kRingMaxF;
  int bt = tmin, bs = 1;
  int tmax = (F + 2) / 3;                              // >= 3 fragments per tile
  if (tmax < tmin) tmax = tmin;
  for (int T = tmin; T <= tmax; ++T) {
    const int fmax = (F + T - 1) / T;
    const double step = 235.0 * fmax + 900.0;
    for (int sk = 1; sk <= 16; ++sk) {
      if (sk > 1 && nk / sk < 4) break;
      const long long units = (long long)T * P * sk;
      const long long rounds = (units + slots - 1) / slots;
      const int steps = (nk + sk - 1) / sk;
      double t = (double)rounds * (steps * step + 9000.0);
      // split-K epilogue: <= 4 slabs take the all-round-trips-at-once form (splitk_epilogue_norm_reg_kernel), more the row-walking one -
      // measured at down_proj, M = 798 (profiles/r6_ring_geometry_sweep.txt): (T, sk) = (6, 2) 210 us against 224-227 for the (5, 5) the
      // flat estimate used to pick
      if (sk > 1) t += sk <= 4 ? 12000.0 + (double)(sk + 2) * m * n * 4.0 / 2500.0 : 24000.0 + (double)(sk + 2) * m * n * 4.0 / 1800.0;
      if (t < best) { best = t; bt = T; bs = sk; }
    }
  }
  if (force_t >= tmin && force_t <= F) bt = force_t;
  if (force_sk >= 1 && force_sk <= 16 && nk / force_sk >= 1) bs = force_sk;
  g.T = bt; g.sk = bs;
  g.units = bt * P * bs;
  int wpx = (g.units + 7) / 8;
  if (wpx > slots / 8) wpx = slots / 8;
  g.wpx = wpx;
}

template <bool SINGLE>
static int launch_ring_t(const GemmParams& p, const uint16_t* a_hi, const uint16_t* a_lo, int ldp, const RingGeom& g, hipStream_t s) {
  static bool configured = false;     // > 64 KB of dynamic LDS must be opted into once
  if (!configured) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring_kernel<SINGLE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             kRingLds);
    CHATTS_REQUIRE(e == hipSuccess, CHATTS_E_LAUNCH, "gemm_ring: cannot reserve %d bytes of LDS: %s", kRingLds, hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL(gemm_ring_kernel<SINGLE>, dim3(8 * g.wpx), dim3(kRingThreads), kRingLds, s, p, a_hi, a_lo, ldp, g);
  return CHATTS_OK;
}
int launch_ring(const GemmParams& p, const uint16_t* a_hi, const uint16_t* a_lo, int ldp, const RingGeom& g, bool single, hipStream_t s) {
  return single ? launch_ring_t<true>(p, a_hi, a_lo, ldp, g, s) : launch_ring_t<false>(p, a_hi, a_lo, ldp, g, s);
}

// bf16 [rows, k] row-major -> the tiled layout: block (b = row / 16, t = k / 32) is 1 KB at ((b * k / 32) + t) * 1 KB; inside it the 16-byte
// chunk at position l (0 .. 63) holds row 16 b + (l >> 2), K-values 32 t + 8 c .. + 7 with c = (l & 3) ^ ((l >> 5) << 1) - the order in
// which the lanes of an LDS-DMA piece deposit a half-stage block (the header comment's swizzle).  Rows beyond the matrix repeat the last row.
__global__ __launch_bounds__(256) void tile_bf16_kernel(const uint16_t* __restrict__ src, int rows, int k, int ld, uint16_t* __restrict__ dst) {
  const int nkt = k >> 5;
  const size_t chunks = (size_t)((rows + 15) >> 4) * nkt * 64;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (size_t)gridDim.x * 256) {
    const int l = (int)(i & 63);
    const size_t blk = i >> 6;
    const int t = (int)(blk % nkt);
    int row = (int)(blk / nkt) * 16 + (l >> 2);
    if (row > rows - 1) row = rows - 1;
    const int c = (l & 3) ^ ((l >> 5) << 1);
    *reinterpret_cast<u32x4*>(dst + i * 8) = *reinterpret_cast<const u32x4*>(src + (size_t)row * ld + t * 32 + c * 8);
  }
}

}  // namespace chatts

extern "C" size_t chatts_tile_bf16_elems(int rows, int k) { return rows > 0 && k > 0 ? (size_t)((rows + 15) / 16) * 16 * k : 0; }

extern "C" int chatts_tile_bf16(const chatts_bf16* src, int rows, int k, int ld, chatts_bf16* dst, chatts_stream_t stream) {
  using namespace chatts;
  CHATTS_REQUIRE(rows >= 0 && k > 0 && k % 32 == 0 && ld >= k && ld % 8 == 0, CHATTS_E_SHAPE, "tile_bf16: rows=%d k=%d ld=%d (K %% 32 == 0, ld %% 8 == 0)", rows, k, ld);
  if (rows == 0) return CHATTS_OK;
  CHATTS_REQUIRE(src && dst && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, CHATTS_E_BADARG, "tile_bf16: null / unaligned pointer");
  const size_t chunks = chatts_tile_bf16_elems(rows, k) / 8;
  const unsigned blocks = (unsigned)((chunks + 255) / 256 < 16384 ? (chunks + 255) / 256 : 16384);
  hipLaunchKernelGGL(tile_bf16_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, rows, k, ld, dst);
  CHATTS_CHECK_LAUNCH("tile_bf16");
  return CHATTS_OK;
}

