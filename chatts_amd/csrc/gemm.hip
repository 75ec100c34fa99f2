// gemm.hip - MFMA projections for M > 1:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
// A float32 activations (or their bf16 hi / lo planes), W bfloat16 weights (HF nn.Linear layout, K contiguous for both).
//
// Precision scheme "bf16x2": the reference path we must match is float32 (1e-3 relative on logits,
// identical greedy tokens).  Weights are bf16 by definition, so only A needs care: a = hi + lo,
// hi = bf16(a), lo = bf16(a - hi), and the wave issues two v_mfma_f32_16x16x32_bf16 per fragment pair
// (lo.W, hi.W) into ONE f32 accumulator.  Products are exact (8b x 8b mantissas), accumulation is f32.
//
// Three kernels share the epilogue (gemm_store: bias / GELU / residual / SwiGLU / fp8 row scale / plane output / split-K
// partials) and therefore produce bit-identical results at equal split-K:
//   gemm_bf16x2_kernel  register-staged, BM x 128 x 32 tiles, A split on the VALU while staging; any M, float32 A;
//                       TS-encoder MLP, prefill chunks below 96 rows, generic callers
//   gemm_dma_kernel     prefill (M >= 96) on pre-split planes: 128 x 256 x 64, LDS-DMA whole-line staging by 4 loader
//                       waves, 8 compute waves
//   gemm_stream_kernel  batched decode (2 <= M <= 16) on planes: 16 x 128 x 64 (128 for fp8 W), weight-streaming shape
// The comments above each kernel say which measurement made it look the way it does.
//
// Register-staged kernel geometry: 256 threads = 4 waves (WM x WN), double-buffered LDS, global loads issued one
// K-step ahead, one barrier per K-step.  LDS rows are 64 B (32 bf16); 16-byte chunk c of row r lives at chunk
// c ^ (((r>>3)&1)<<1), which makes every ds_read_b128 fragment read conflict-free (checked exhaustively over the 4
// lane groups).  Split-K (grid.z) fills the 256 CUs when M is small; partials go to a workspace and a second kernel
// applies the epilogue.
#include <string.h>
#include <type_traits>

#include "gemm_common.h"

// cache-policy bits of the prefill GEMM's LDS-DMA loads (A/B knobs, compile time): 0 = default, 2 = non-temporal
#ifndef CHATTS_DMA_A_AUX
#define CHATTS_DMA_A_AUX 0
#endif
#ifndef CHATTS_DMA_W_AUX
#define CHATTS_DMA_W_AUX 0
#endif

namespace chatts {

// ---- timeline probe (diagnostic builds only: -DCHATTS_GEMM_PROBE, tools/build_variant.py) --------------------------------------
// One 16-word record per workgroup of gemm_dma_kernel: compute wave 0 stamps entry / first stage published / K loop done / stores
// drained, loader wave 0 stamps entry / prologue issued / first stage landed, with the 100 MHz s_memrealtime counter (comparable
// across CUs) plus s_memtime at both ends (shader clock -> effective frequency) and the CU the workgroup ran on.
#ifdef CHATTS_GEMM_PROBE
constexpr int kProbeRecs = 1 << 16;
__device__ unsigned long long g_gemm_probe[kProbeRecs * 16];
__device__ __forceinline__ unsigned long long probe_rt() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ unsigned long long probe_clk() { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ unsigned probe_cu() {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
  return ((xcc & 15u) << 16) | ((hw >> 8) & 0xffffu);
}
#define PROBE(slot, expr) do { if (lane == 0 && prec) prec[slot] = (expr); } while (0)
#else
#define PROBE(slot, expr) do { } while (0)
#endif

// ---- stream-K decomposition (gemm_dma_kernel) -------------------------------------------------------------------
// The (tile, K-step) space of a GEMM is linear: tile t owns steps [t * nk, (t + 1) * nk), tiles in (N-panel major, M-tile
// minor) order.  It is cut into kSkRanges = 256 equal contiguous ranges (range i = [b_i, b_{i+1}), b_i = floor(i * T / 256)),
// one per CU, so every CU gets the same number of K-steps whatever the tile count (140 tiles of o_proj on 256 CUs: 43.75 steps
// each instead of two rounds of 27).  A range shorter than a tile crosses at most one tile boundary, i.e. it is one or two
// PIECES, each a (tile, K sub-range) the unchanged kernel body can run; a tile collects its <= S pieces as split-K slabs
// (slab = ordinal of the piece within the tile) and the split-K epilogue sums the slabs a tile really has.
// Nothing is sorted or tabulated: a workgroup derives its piece from blockIdx by integer arithmetic (below).
constexpr int kSkRanges = 256;
__device__ __host__ __forceinline__ int sk_bound(int i, int T) { return (int)(((long long)i * T) / kSkRanges); }
__device__ __host__ __forceinline__ int sk_range_of(int step, int T) {          // the range that contains `step`
  return (int)((((long long)(step + 1)) * kSkRanges + T - 1) / T) - 1;
}
__device__ __host__ __forceinline__ int sk_tile_nseg(int tile, int T, int nk) {
  return sk_range_of((tile + 1) * nk - 1, T) - sk_range_of(tile * nk, T) + 1;
}

__device__ __forceinline__ int lds_off(int r, int c) {   // byte offset of 16-byte chunk c of row r
  return r * 64 + ((c ^ (((r >> 3) & 1) << 1)) << 4);
}

// Epilogue shared by the GEMM kernels.  C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
template <int FM, int FN, int TM, int TN, bool W8>
__device__ __forceinline__ void gemm_store(const GemmParams& p, const f32x4 (&acc)[FM][FN], int m0, int n0, int wm, int wn,
                                           int lane, int split) {
  // ---- epilogue.  C/D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg -----------
  const int ccol = lane & 15, crow0 = (lane >> 4) * 4;
  if (!p.direct) {
    float* ws = p.c + (size_t)split * p.m * p.n;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = n0 + wn * TN + j * 16 + ccol;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * TM + i * 16 + crow0 + r;
          if (row < p.m && col < p.n) ws[(size_t)row * p.n + col] = acc[i][j][r];
        }
      }
    return;
  }
  if (p.epilogue == CHATTS_EPI_SWIGLU) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; j += 2) {
        const int prow = n0 + wn * TN + j * 16 + ccol;   // packed gate row; up row = prow + 16
        const int ocol = (prow >> 5) * 16 + ccol;
        if (prow + 16 < p.n) {
          const float bg = p.bias ? p.bias[prow] : 0.f, bu = p.bias ? p.bias[prow + 16] : 0.f;
          const float sg = W8 ? p.w8_scale[prow] : 1.f, su = W8 ? p.w8_scale[prow + 16] : 1.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * TM + i * 16 + crow0 + r;
            if (row < p.m) {
              const float v = silu_g(acc[i][j][r] * sg + bg) * (acc[i][j + 1][r] * su + bu);
              if (p.c_hi) store_planes(p.c_hi, p.c_lo, (size_t)row * p.ldcp + ocol, v);
              else p.c[(size_t)row * p.ldc + ocol] = v;
            }
          }
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + wn * TN + j * 16 + ccol;
      if (col >= p.n) continue;
      const float b = p.bias ? p.bias[col] : 0.f;
      const float sc = W8 ? p.w8_scale[col] : 1.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * TM + i * 16 + crow0 + r;
        if (row >= p.m) continue;
        float v = acc[i][j][r] * sc + b;
        if (p.epilogue == CHATTS_EPI_GELU) v = gelu_erf_f(v);
        if (p.epilogue == CHATTS_EPI_RESID) v = p.resid[(size_t)row * p.ldc + col] + v;
        if (p.c_hi) store_planes(p.c_hi, p.c_lo, (size_t)row * p.ldcp + col, v);
        else p.c[(size_t)row * p.ldc + col] = v;
      }
    }
}

// The same epilogue for accumulators of the 32x32 MFMA (gemm_dma_kernel<.., W32 = true>): FM x FN blocks of 32 x 32, 16 values per
// lane; C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  A SwiGLU pair (gate rows 32q .. 32q+15, up
// rows 32q+16 .. 32q+31 of W) lies inside ONE block: lane c < 16 holds the gate, lane c + 16 the up value of unit c.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int FM, int FN, int TM, int TN>
__device__ __forceinline__ void gemm_store32(const GemmParams& p, const f32x16 (&acc)[FM][FN], int m0, int n0, int wm, int wn,
                                             int lane, int split) {
  const int ccol = lane & 31, rbase = 4 * (lane >> 5);
  if (!p.direct) {
    float* ws = p.c + (size_t)split * p.m * p.n;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = n0 + wn * TN + j * 32 + ccol;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
          if (row < p.m && col < p.n) ws[(size_t)row * p.n + col] = acc[i][j][r];
        }
      }
    return;
  }
  if (p.epilogue == CHATTS_EPI_SWIGLU) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int unit = ccol & 15;
        const int prow = n0 + wn * TN + j * 32 + unit;   // packed gate row; up row = prow + 16
        const int ocol = (prow >> 5) * 16 + unit;
        const bool live = ccol < 16 && prow + 16 < p.n;
        const float bg = (live && p.bias) ? p.bias[prow] : 0.f, bu = (live && p.bias) ? p.bias[prow + 16] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float g = acc[i][j][r];
          const float u = __shfl_xor(g, 16, 64);            // every lane takes part: lane c < 16 receives the up value of its unit
          const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
          if (live && row < p.m) {
            const float v = silu_g(g + bg) * (u + bu);
            if (p.c_hi) store_planes(p.c_hi, p.c_lo, (size_t)row * p.ldcp + ocol, v);
            else p.c[(size_t)row * p.ldc + ocol] = v;
          }
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + wn * TN + j * 32 + ccol;
      if (col >= p.n) continue;
      const float b = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + rbase;
        if (row >= p.m) continue;
        float v = acc[i][j][r] + b;
        if (p.epilogue == CHATTS_EPI_GELU) v = gelu_erf_f(v);
        if (p.epilogue == CHATTS_EPI_RESID) v = p.resid[(size_t)row * p.ldc + col] + v;
        if (p.c_hi) store_planes(p.c_hi, p.c_lo, (size_t)row * p.ldcp + col, v);
        else p.c[(size_t)row * p.ldc + col] = v;
      }
    }
}

// W8: W is streamed from its fp8 copy (compile-time switch: a runtime branch in the K-loop cost 18 % on the bf16 path)
template <int BM, int WM, int WN, bool W8>
__global__ __launch_bounds__(256) void gemm_bf16x2_kernel(GemmParams p) {
  constexpr int BN = 128, BK = 32;
  constexpr int TM = BM / WM, TN = BN / WN;     // wave tile
  constexpr int FM = TM / 16, FN = TN / 16;     // 16x16 fragments per wave
  constexpr int A_ITERS = (BM * 4 + 255) / 256; // each thread-slot stages 8 floats of one A row
  constexpr int STAGE = BM * 64 * 2 + BN * 64;  // bytes: A_hi, A_lo, B
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed; used for speed only).  All M-tiles of one
  // W panel (N-tile) are given to the same XCD, adjacent in dispatch order, so the panel is fetched from HBM once
  // into that XCD's L2 instead of once per M-tile (profiles/r1_bench_pmc_fetch_size.txt showed ~7x re-fetch).
  const int mt_count = (p.m + BM - 1) / BM, nt_count = (p.n + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int nt = (local / mt_count) * 8 + xcd, mt = local % mt_count;
  if (nt >= nt_count) return;                    // padding slot (N-tiles not a multiple of 8); uniform exit
  const int m0 = mt * BM, n0 = nt * BN;
  const int kbeg = blockIdx.z * p.k_per_split;
  int kend = kbeg + p.k_per_split;
  if (kend > p.k) kend = p.k;
  const int nk = (kend - kbeg) / BK;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging registers
  f32x4 ra[A_ITERS][2];
  u32x4 rb[2];
  const int srow = tid >> 2, schunk = tid & 3;

  auto load_global = [&](int kt) {
    const int k0 = kbeg + kt * BK + schunk * 8;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int r = srow + i * 64;
      ra[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ra[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (r < BM && m0 + r < p.m) {
        const float* src = p.a + (size_t)(m0 + r) * p.lda + k0;
        ra[i][0] = *reinterpret_cast<const f32x4*>(src);
        ra[i][1] = *reinterpret_cast<const f32x4*>(src + 4);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = srow + i * 64;
      rb[i] = (u32x4){0u, 0u, 0u, 0u};
      if (n0 + r < p.n) {
        if (W8) {
          const u32x2 q = *reinterpret_cast<const u32x2*>(p.w8 + (size_t)(n0 + r) * p.ldw8 + k0);   // 8 fp8 weights
          rb[i].x = q.x;
          rb[i].y = q.y;
        } else {
          rb[i] = *reinterpret_cast<const u32x4*>(p.w + (size_t)(n0 + r) * p.ldw + k0);
        }
      }
    }
  };

  auto store_lds = [&](int buf) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int r = srow + i * 64;
      if (r < BM) {
        const float v[8] = {ra[i][0].x, ra[i][0].y, ra[i][0].z, ra[i][0].w,
                            ra[i][1].x, ra[i][1].y, ra[i][1].z, ra[i][1].w};
        // native __bf16 conversions: the compiler selects v_cvt_pk_bf16_f32 (RNE), ~3 VALU per element instead of ~10
        bf16x8_t hv, lv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const __bf16 h = (__bf16)v[j];
          hv[j] = h;
          lv[j] = (__bf16)(v[j] - (float)h);
        }
        const int off = lds_off(r, schunk);
        *reinterpret_cast<bf16x8_t*>(base + off) = hv;
        *reinterpret_cast<bf16x8_t*>(base + BM * 64 + off) = lv;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = srow + i * 64;
      if (W8) {                                    // e4m3 -> f32 -> bf16: both steps exact
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t f0 = __builtin_amdgcn_cvt_pk_f32_fp8(rb[i].x, false), f1 = __builtin_amdgcn_cvt_pk_f32_fp8(rb[i].x, true);
        const f32x2_t f2 = __builtin_amdgcn_cvt_pk_f32_fp8(rb[i].y, false), f3 = __builtin_amdgcn_cvt_pk_f32_fp8(rb[i].y, true);
        bf16x8_t bv;
        bv[0] = (__bf16)f0.x; bv[1] = (__bf16)f0.y; bv[2] = (__bf16)f1.x; bv[3] = (__bf16)f1.y;
        bv[4] = (__bf16)f2.x; bv[5] = (__bf16)f2.y; bv[6] = (__bf16)f3.x; bv[7] = (__bf16)f3.y;
        *reinterpret_cast<bf16x8_t*>(base + BM * 128 + lds_off(r, schunk)) = bv;
      } else {
        *reinterpret_cast<u32x4*>(base + BM * 128 + lds_off(r, schunk)) = rb[i];
      }
    }
  };

  if (nk > 0) {
    load_global(0);
    store_lds(0);
  }
  __syncthreads();

  const int frow = lane & 15, fchunk = lane >> 4;   // fragment: row (A) / col (B) and 8-wide K chunk
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_global(kt + 1);
    const char* base = smem + (kt & 1) * STAGE;
    bf16x8_t bfrag[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j)
      bfrag[j] = *reinterpret_cast<const bf16x8_t*>(base + BM * 128 + lds_off(wn * TN + j * 16 + frow, fchunk));
    // Two sweeps over all FM x FN accumulators (lo pass, then hi pass): consecutive MFMAs never touch the same
    // accumulator, so none waits on the previous one's result (back-to-back lo/hi on one accumulator held the
    // matrix pipe at 41 % busy: SQ_WAIT_INST_ANY 44 %, profiles/r1_pmc_sq_gemm.txt).
    bf16x8_t afrag[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i)
      afrag[i] = *reinterpret_cast<const bf16x8_t*>(base + BM * 64 + lds_off(wm * TM + i * 16 + frow, fchunk));
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[i], bfrag[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < FM; ++i)
      afrag[i] = *reinterpret_cast<const bf16x8_t*>(base + lds_off(wm * TM + i * 16 + frow, fchunk));
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[i], bfrag[j], acc[i][j], 0, 0, 0);
    if (kt + 1 < nk) store_lds((kt + 1) & 1);
    __syncthreads();
  }

  gemm_store<FM, FN, TM, TN, W8>(p, acc, m0, n0, wm, wn, lane, blockIdx.z);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA kernel (prefill, M large): A arrives already split into bf16 hi / lo planes (split_bf16x2_kernel, or a
// producer that writes planes directly), so all three operand tiles are plain bf16 rows and are staged by
// global_load_lds_dwordx4: 16 B per lane from global memory straight into LDS - no staging registers, no VALU
// conversion, no ds_write.  One wave instruction fills 1 KB of LDS (M0 base + lane * 16) = 8 tile rows x 128 B, i.e.
// every lane-octet fetches one whole 128-byte line.
//
// Why this shape (measurements: tools/gemm_dma_sweep.py, tools/probes/*.hip, profiles/r1_gemm_dma_*.txt).
//  * The 128 x 128 tile of the register-staged kernel moves 24 KB per 32-deep K-step and CU-side fetch, not the MFMA
//    pipe or LDS, paces it.  Flops per fetched byte = 2*BM*BN / (2*BM + BN) (A counts twice: hi and lo): 85 for
//    128 x 128, 128 for 128 x 256.  Hence 128 x 256 x 64: eight compute waves as 2 x 4, each a 64 x 64 wave tile,
//    64 MFMAs per wave and K-step.
//  * An LDS-DMA instruction holds its wave's issue port for ~100 cycles.  When the compute waves issue their own
//    pieces right behind the step's barrier, both waves of every SIMD do so at the same moment and the matrix pipe
//    drains (57 % busy).  So the DMA is issued by four LOADER waves (one per SIMD, 16 pieces each per K-step): their
//    VMEM issue overlaps the compute waves' MFMAs, and the compute waves' stream is ds_read + MFMA only.
//    12 waves = 3 per SIMD -> 168 VGPRs per wave, which the read schedule below is built to fit.
//  * L2 prefetching of the W panel (by the compute waves, by a ninth wave, or by the panel's first M-tile only) was
//    measured and does not pay: the in-order VMEM return queue or the doubled L2 request count cost more than the
//    HBM latency they hide.
//
// LDS: two 64 KB stages (A_hi | A_lo | W, rows of 128 B); 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7),
// applied on the SOURCE side of the DMA (lane l of a piece fetches global chunk (l & 7) ^ swizzle(row)); conflict-free
// for the ds_read_b128 fragment reads (brute-forced over the four lane groups and both K halves).
//
// Compute-wave pipeline (K-step kt; W and A_lo/A_hi fragments of K-half 0 already in registers):
//   lo sweep h0 | read W,A_lo of h1 | hi sweep h0 | read A_hi of h1 | lo sweep h1 | lgkmcnt(0) + s_barrier |
//   read h0 of stage kt+1 | hi sweep h1
// The barrier sits INSIDE the MFMA stream: the loaders arrive at it once stage kt+1 has landed, the compute waves once
// they have read the last fragment of stage kt, whose slot the loaders refill right behind it.  The next step's
// first fragments land underneath the last 16 MFMAs.
// Rows past M / N are clamped to the last valid row (the epilogue masks them): the loop has no bounds checks.
//
// XCD mapping: workgroup b runs on XCD b % 8.  The (N-panel major, M-tile minor) tile sequence is cut into 8 contiguous,
// equally long ranges, one per XCD: the M-tiles of a W panel stay adjacent on one XCD (its L2 fetches the panel once),
// and no XCD gets a whole panel more than another (108 panels over 8 XCDs as 14/13 cost a 4th round of workgroups).
__device__ __forceinline__ int lds_off128(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

constexpr int kDmaBN = 256, kDmaBK = 64, kDmaLds = 2 * (2 * 128 * 128 + kDmaBN * 128), kDmaThreads = 768;

// SINGLE: the "bf16" speed mode (CHATTS_GEMM_PRECISION=bf16; SURVEY.md section 7's precision="bf16"): the lo plane is staged but
// neither read nor multiplied - activations rounded to bf16, one MFMA pass, half the matrix work; logits then sit at ~1e-2 of
// the float32 oracle instead of 5e-5, so this is never the parity-grade / headline path.  SINGLE = false is the kernel as it was
// (same instruction stream: the flag only removes code).
//
// W32 (round 3, opt-in - measured slower, see launch_dma): FOUR compute waves, one per SIMD, as 2 x 2 with 64 x 128 wave tiles on v_mfma_f32_32x32x16_bf16.  The speed-mode
// measurement of round 2 (half the MFMAs: -7 %) had shown the K loop to be paced by LDS operand delivery, not by the matrix pipe:
// eight waves x 24 ds_read_b128 = 192 KB of fragment reads per K-step next to the 64 KB of DMA writes.  A wave tile twice as wide
// reads A 8 + 8 and W 16 fragments for 64 MFMAs of twice the size: 128 KB per K-step and workgroup for the same matrix work (2048
// MFMA cycles per SIMD and K-step either way); accumulators 2 x 4 x 16 = 128 registers per wave, fragments double-buffered per
// 16-deep K sub-step.  Loader waves, LDS layout and swizzle are unchanged - the swizzle is conflict-free for the 32-row fragment
// reads as well (a 16-lane group of ds_read_b128 covers row pairs with all eight values of (r >> 1) & 7, in both K chunks).
template <bool SINGLE, bool W32>
__global__ __launch_bounds__(W32 ? 512 : kDmaThreads) void gemm_dma_kernel(GemmParams p, const uint16_t* __restrict__ a_hi,
                                                                           const uint16_t* __restrict__ a_lo, int ldp) {
  constexpr int BM = 128, BN = kDmaBN, BK = kDmaBK, WN = W32 ? 2 : 4, NCOMPUTE = W32 ? 4 : 8, NLOAD = 4;
  constexpr int TM = 64, TN = W32 ? 128 : 64, FM = 4, FN = 4;
  constexpr int A_PLANE = BM * 128, STAGE = 2 * A_PLANE + BN * 128;
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt_count = (p.m + BM - 1) / BM, nt_count = (p.n + BN - 1) / BN;
  const int total = mt_count * nt_count, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  int idx, kbeg, kend, slab = blockIdx.z;
  if (p.sk_T > 0) {
    // stream-K: grid = 8 XCDs x 64.  Workgroup (xcd, local < 32) runs the FIRST piece of range i = 32 xcd + local; workgroup
    // (xcd, 32 + r) runs the SECOND piece of the range whose first piece is the r-th shortest of this XCD's two-piece ranges:
    // workgroups start in blockIdx order, so the CU that finishes its short first piece first picks up exactly its own (long)
    // second piece - every CU ends up with its own range, i.e. the same number of K-steps, without a table or a sort.
    const int T = p.sk_T, nk_t = p.sk_nk;
    auto first_len = [&](int i, int& b0, int& b1) {       // length of the first piece of range i; == b1 - b0 when it has one piece
      b0 = sk_bound(i, T); b1 = sk_bound(i + 1, T);
      const int tile_end = (b0 / nk_t + 1) * nk_t;
      return tile_end < b1 ? tile_end - b0 : b1 - b0;
    };
    int i = xcd * 32 + (local & 31), b0, b1, s0, s1;
    if (local < 32) {
      const int a = first_len(i, b0, b1);
      if (b1 <= b0) return;
      s0 = b0; s1 = b0 + a;
    } else {
      // lane j < 32 evaluates range j of this XCD once; ranks by shuffle (every wave does the same: wave-uniform result)
      const int r = local - 32, j = lane & 31;
      int jb0, jb1;
      const int aj = first_len(xcd * 32 + j, jb0, jb1);
      const bool two = aj < jb1 - jb0;
      int rank = 0;
      for (int q = 0; q < 32; ++q) {
        const int aq = __shfl(aj, q, 64);
        const bool tq = __shfl((int)two, q, 64) != 0;
        if (tq && (aq < aj || (aq == aj && q < j))) ++rank;
      }
      const unsigned long long hit = __ballot(two && rank == r && lane < 32);
      const int found = __builtin_amdgcn_readfirstlane(hit ? (int)__builtin_ctzll(hit) : -1);     // wave-uniform -> scalar
      if (found < 0) return;                                             // fewer two-piece ranges than slots: uniform exit
      i = xcd * 32 + found;
      const int a = first_len(i, b0, b1);
      s0 = b0 + a; s1 = b1;
    }
    idx = s0 / nk_t;
    kbeg = (s0 - idx * nk_t) * BK;
    kend = (s1 - idx * nk_t) * BK;
    slab = i - sk_range_of(idx * nk_t, T);
  } else {
    const int share = total >> 3, rem = total & 7;
    if (local >= share + (xcd < rem)) return;                          // uniform exit of the padding workgroups
    idx = xcd * share + (xcd < rem ? xcd : rem) + local;
    kbeg = blockIdx.z * p.k_per_split;
    kend = kbeg + p.k_per_split;
    if (kend > p.k) kend = p.k;
  }
  const int nt = idx / mt_count, mt = idx - nt * mt_count;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nk = (kend - kbeg) / BK;           // the launcher makes k_per_split a multiple of 64; nk >= 1
#ifdef CHATTS_GEMM_PROBE
  const unsigned prec_i = blockIdx.z * gridDim.x + blockIdx.x;
  unsigned long long* prec = (prec_i < (unsigned)kProbeRecs && (wave == 0 || wave == NCOMPUTE)) ? g_gemm_probe + (size_t)prec_i * 16 + (wave ? 8 : 0) : nullptr;
  PROBE(0, probe_rt());
  PROBE(1, probe_clk());
  if (wave == 0) { PROBE(6, ((unsigned long long)probe_cu() << 32) | (unsigned)idx); PROBE(7, ((unsigned long long)nk << 32) | (unsigned)(p.m - m0)); }
#endif

  if (wave >= NCOMPUTE) {
    // ---- loader wave L: 1 KB pieces (8 rows x 128 B) {L, L+4, ...} of each plane, 16 per stage.  Piece q = rows
    // 8q .. 8q+7, so (r >> 1) & 7 = ((q & 1) << 2) | (lrow >> 1), and q & 1 == L & 1.
    constexpr int NA = 16 / NLOAD, NW = BN / 8 / NLOAD;
    const int L = wave - NCOMPUTE;
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((L & 1) << 2) | (lrow >> 1));
    const uint16_t* src[2 * NA + NW];
#pragma unroll
    for (int h = 0; h < NA; ++h) {
      int am = m0 + (L + NLOAD * h) * 8 + lrow;
      if (am > p.m - 1) am = p.m - 1;
      src[h] = a_hi + (size_t)am * ldp + kbeg + lchunk * 8;
      src[NA + h] = a_lo + (size_t)am * ldp + kbeg + lchunk * 8;
    }
#pragma unroll
    for (int h = 0; h < NW; ++h) {
      int wr = n0 + (L + NLOAD * h) * 8 + lrow;
      if (wr > p.n - 1) wr = p.n - 1;
      src[2 * NA + h] = p.w + (size_t)wr * p.ldw + kbeg + lchunk * 8;
    }
    auto issue = [&](int kt) {   // stage kt -> slot kt & 1
      char* base = smem + (kt & 1) * STAGE + L * 1024;
#pragma unroll
      for (int h = 0; h < NA; ++h) {
        if (m0 + (L + NLOAD * h) * 8 >= p.m) continue;     // ragged last M-tile: these 8 rows lie past M, nobody reads them
        __builtin_amdgcn_global_load_lds((gptr_t)(src[h] + (size_t)kt * BK), (lptr_t)(base + h * NLOAD * 1024), 16, 0, CHATTS_DMA_A_AUX);
        __builtin_amdgcn_global_load_lds((gptr_t)(src[NA + h] + (size_t)kt * BK), (lptr_t)(base + A_PLANE + h * NLOAD * 1024),
                                         16, 0, CHATTS_DMA_A_AUX);
      }
#pragma unroll
      for (int h = 0; h < NW; ++h)
        __builtin_amdgcn_global_load_lds((gptr_t)(src[2 * NA + h] + (size_t)kt * BK),
                                         (lptr_t)(base + 2 * A_PLANE + h * NLOAD * 1024), 16, 0, CHATTS_DMA_W_AUX);
    };
    static_assert(NA == 4 && NW == 8, "the vmcnt literals below count NW + 2 * (live A piece pairs) pieces per stage");
    issue(0);
    if (nk > 1) {
      issue(1);
      PROBE(2, probe_rt());                                         // prologue pieces issued
      int a_live = 0;                                               // a ragged tile issues fewer than 16 pieces per stage
#pragma unroll
      for (int h = 0; h < NA; ++h) a_live += m0 + (L + NLOAD * h) * 8 < p.m;
      switch (a_live) {                                             // stage 0 landed, stage 1 (NW + 2 a_live pieces) may be in flight
        case 0: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PROBE(3, probe_rt());                                           // stage 0 landed
    __builtin_amdgcn_s_barrier();                                   // publishes stage 0
    for (int kt = 0; kt + 1 < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // stage kt+1 landed (issued a whole K-step ago)
      __builtin_amdgcn_s_barrier();                                 // ... published; slot of stage kt retired
      if (kt + 2 < nk) issue(kt + 2);
    }
    return;
  }

  // ---- compute wave
  const int wm = wave / WN, wn = wave % WN;
  if constexpr (W32) {
    constexpr int GM = 2, GN = 4;                 // 32 x 32 blocks of the 64 x 128 wave tile
    f32x16 acc[GM][GN];
#pragma unroll
    for (int i = 0; i < GM; ++i)
#pragma unroll
      for (int j = 0; j < GN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // ragged last M-tile: 32-row blocks of this wave wholly past M are neither read nor multiplied
    const int rows_left = p.m - (m0 + wm * TM);
    const int gm_live = rows_left <= 0 ? 0 : (rows_left > 32 ? 2 : 1);
    if (gm_live == 0) {
      for (int kt = 0; kt < nk; ++kt) __builtin_amdgcn_s_barrier();    // nk barriers, like the live waves
      return;
    }
    const int frow = lane & 31, fhalf = lane >> 5;                      // fragment row (A) / column (W), 8-wide K chunk of the 16
    auto k_loop32 = [&](auto gml_c) {
      constexpr int GML = decltype(gml_c)::value;
      bf16x8_t bfrag[2][GN], alo[2][GML], ahi[2][GML];
      auto read_sub = [&](int kt, int sub, int buf) {                   // fragments of K sub-step `sub` (16 deep) of stage kt
        const char* base = smem + (kt & 1) * STAGE;
        const int chunk = sub * 2 + fhalf;
#pragma unroll
        for (int j = 0; j < GN; ++j)
          bfrag[buf][j] = *reinterpret_cast<const bf16x8_t*>(base + 2 * A_PLANE + lds_off128(wn * TN + j * 32 + frow, chunk));
#pragma unroll
        for (int i = 0; i < GML; ++i) {
          if constexpr (!SINGLE) alo[buf][i] = *reinterpret_cast<const bf16x8_t*>(base + A_PLANE + lds_off128(wm * TM + i * 32 + frow, chunk));
          ahi[buf][i] = *reinterpret_cast<const bf16x8_t*>(base + lds_off128(wm * TM + i * 32 + frow, chunk));
        }
      };
      auto sweep = [&](const bf16x8_t (&af)[GML], const bf16x8_t (&bf)[GN]) {
#pragma unroll
        for (int i = 0; i < GML; ++i)
#pragma unroll
          for (int j = 0; j < GN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      };
      __builtin_amdgcn_s_barrier();                // stage 0 published
      read_sub(0, 0, 0);
      auto step = [&](int kt, bool more) {
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          const int cur = sub & 1, nxt = cur ^ 1;
          __builtin_amdgcn_sched_barrier(0);
          if (sub < 3) {
            read_sub(kt, sub + 1, nxt);            // the next sub-step's fragments land under this one's MFMAs
          } else if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment read of slot kt has returned
            __builtin_amdgcn_s_barrier();          // stage kt + 1 is published; the loaders refill slot kt behind this
            read_sub(kt + 1, 0, nxt);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!SINGLE) sweep(alo[cur], bfrag[cur]);
          sweep(ahi[cur], bfrag[cur]);
        }
      };
      for (int kt = 0; kt + 1 < nk; ++kt) step(kt, true);
      step(nk - 1, false);
    };
    if (gm_live == 1) k_loop32(std::integral_constant<int, 1>{});
    else k_loop32(std::integral_constant<int, 2>{});
    gemm_store32<GM, GN, TM, TN>(p, acc, m0, n0, wm, wn, lane, slab);
    return;
  }
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Ragged last M-tile (798 = 6 x 128 + 30): the 16-row fragments of this wave that lie wholly past M are not read and not
  // multiplied - the K loop is instantiated for 1..4 live fragments - and a wave with none only keeps the barrier cadence.
  // One in seven workgroups of the benchmark prompt is such a tile, and the kernel is MFMA-bound.
  const int rows_left = p.m - (m0 + wm * TM);
  const int fm_live = rows_left <= 0 ? 0 : (rows_left >= TM ? FM : (rows_left + 15) / 16);
  if (fm_live == 0) {
    for (int kt = 0; kt < nk; ++kt) __builtin_amdgcn_s_barrier();    // nk barriers, like the live waves
    return;
  }
  const int frow = lane & 15, fchunk = lane >> 4;
  auto k_loop = [&](auto fml_c) {
    constexpr int FML = decltype(fml_c)::value;
    bf16x8_t bfrag[2][FN], alo[2][FML], ahi[2][FML];
    auto read_b = [&](int kt, int h) {
      const char* base = smem + (kt & 1) * STAGE + 2 * A_PLANE;
#pragma unroll
      for (int j = 0; j < FN; ++j)
        bfrag[h][j] = *reinterpret_cast<const bf16x8_t*>(base + lds_off128(wn * TN + j * 16 + frow, h * 4 + fchunk));
    };
    auto read_a = [&](int kt, int h, int plane, bf16x8_t (&dst)[FML]) {
      const char* base = smem + (kt & 1) * STAGE + plane * A_PLANE;
#pragma unroll
      for (int i = 0; i < FML; ++i)
        dst[i] = *reinterpret_cast<const bf16x8_t*>(base + lds_off128(wm * TM + i * 16 + frow, h * 4 + fchunk));
    };
    auto sweep = [&](const bf16x8_t (&af)[FML], const bf16x8_t (&bf)[FN]) {
#pragma unroll
      for (int i = 0; i < FML; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };

    __builtin_amdgcn_s_barrier();                // stage 0 published
    PROBE(2, probe_rt());
    read_b(0, 0);
    if constexpr (!SINGLE) read_a(0, 0, 1, alo[0]);
    read_a(0, 0, 0, ahi[0]);
    // (the last K-step is peeled: with the `more` test inside the loop the register allocator stops accumulating in place
    // and spills fragments)
    auto step = [&](int kt, bool more) {
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!SINGLE) sweep(alo[0], bfrag[0]);
      __builtin_amdgcn_sched_barrier(0);
      read_b(kt, 1);
      if constexpr (!SINGLE) read_a(kt, 1, 1, alo[1]);
      __builtin_amdgcn_sched_barrier(0);
      sweep(ahi[0], bfrag[0]);
      __builtin_amdgcn_sched_barrier(0);
      read_a(kt, 1, 0, ahi[1]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!SINGLE) sweep(alo[1], bfrag[1]);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment read of slot kt has returned
        __builtin_amdgcn_s_barrier();
        read_b(kt + 1, 0);
        if constexpr (!SINGLE) read_a(kt + 1, 0, 1, alo[0]);
        read_a(kt + 1, 0, 0, ahi[0]);
      }
      __builtin_amdgcn_sched_barrier(0);
      sweep(ahi[1], bfrag[1]);
    };
    for (int kt = 0; kt + 1 < nk; ++kt) step(kt, true);
    step(nk - 1, false);
  };
  switch (fm_live) {
    case 1: k_loop(std::integral_constant<int, 1>{}); break;
    case 2: k_loop(std::integral_constant<int, 2>{}); break;
    case 3: k_loop(std::integral_constant<int, 3>{}); break;
    default: k_loop(std::integral_constant<int, 4>{}); break;
  }
  PROBE(3, probe_rt());                          // K loop done
  gemm_store<FM, FN, TM, TN, false>(p, acc, m0, n0, wm, wn, lane, slab);
#ifdef CHATTS_GEMM_PROBE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PROBE(4, probe_rt());                          // stores drained
  PROBE(5, probe_clk());
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-streaming kernel for 2 <= M <= 16 (batched decode: M = sequences in flight).  At this M the projection is an
// HBM stream of W, like the M = 1 GEMV, and what decides its rate is the SHAPE of the loads, not the MFMA work:
// fetching W in the MFMA fragment pattern (16 rows x 64 B per wave instruction - what a BK = 32 tile does, in registers
// or through LDS) tops out near 4.0-4.5 TB/s, the same bytes as 8 rows x 128 B (whole lines) stream at 5.2-5.7 TB/s
// (profiles/r1_probe_weight_stream_patterns.txt, DBG=1 vs DBG=3).  So: tile 16 x 128 x 64, every operand staged by
// LDS-DMA in whole-line pieces (the same 128-byte-row, XOR-swizzled layout as gemm_dma_kernel), a ring of NSTAGE
// 20 KB stages so that 2 workgroups per CU keep 60-80 KB of W in flight, split-K over workgroups to fill the chip
// (52 MB of o_proj weights are only 40 N-tiles).  A comes as bf16 hi / lo planes ([M, K], M <= 16).
constexpr int kStreamBN = 128;
constexpr int stream_bk(bool w8) { return w8 ? 128 : 64; }                       // K per stage = one 128-byte line of a W row
constexpr int stream_stage(bool w8, int mb = 1) { return 2 * mb * 16 * stream_bk(w8) * 2 + kStreamBN * 128; }   // A_hi | A_lo | W

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Wait until at most min(later, CAP) x PIECES of this wave's loads are outstanding (later = stages issued after the one about to be read).
template <int PIECES, int CAP>
__device__ __forceinline__ void wait_stages(int later) {
  if constexpr (CAP >= 1) {
    if (later >= CAP) { wait_vmcnt<PIECES * CAP>(); return; }
    wait_stages<PIECES, CAP - 1>(later);
  } else {
    wait_vmcnt<0>();
  }
}

// W8: W is the fp8 (e4m3fn) copy: a 128-byte line of a row holds 128 K-values, so a stage is 128 deep; the 8-byte B fragments
// are widened (exactly) to bf16 in registers and the per-row scale is applied in the epilogue.
// MB = 16-row blocks of A a workgroup carries (M <= 16 * MB).  MB = 1 is the batched-decode kernel described above.  MB = 2 / 4 / 8
// serve the skinny GEMMs with FEW N-tiles (the TS-encoder MLP: P <= 128 patches x 5120 columns; short prefill chunks): the
// LDS-DMA prefill kernel's 128 x 256 tiles give such a problem 20 tiles for 256 CUs and it needs ~10 K-splits - 26 MB of
// partials and 8 K-steps per workgroup, mostly pipeline fill (33.7 us per TS layer at P = 128, profiles/r2_ts_gemm_sweep.txt).
// Here the tile is (16 MB) x 128: 40 N-tiles x ~6 K-splits = one workgroup per CU, W streamed in whole lines exactly once per
// split, the MB A blocks of a K-step (MB x 4 KB) staged next to the W tile (16 KB), 8 MB MFMAs per wave and K-step.
// Split-K without the epilogue launch (2 <= M <= 16): every workgroup of a tile writes its partial slab with write-through
// stores, drains them and bumps the tile's arrival counter; the one that arrives LAST (whichever it is) acquires, sums the slabs
// in split order 0 .. sk-1 and applies the epilogue - the arithmetic of splitk_epilogue_kernel, element for element - then
// re-arms the counter.  The cross-workgroup hand-off is the guide's recipe (sc1 stores + vmcnt(0) + relaxed atomic / acquire
// fence + plain loads).
template <int FN, int NW, bool W8>
__device__ __forceinline__ void stream_fixup(const GemmParams& p, const f32x4 (&acc)[1][FN], int n0, int wave, int lane, char* smem) {
  const int sk = gridDim.z;
  const size_t plane = (size_t)p.m * p.n;
  float* ws = p.c + blockIdx.z * plane;
  const int ccol = lane & 15, crow0 = (lane >> 4) * 4;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int col = n0 + wave * (16 * FN) + j * 16 + ccol;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = crow0 + r;
      if (row < p.m && col < p.n)      // write-through (sc1): read by another workgroup inside this launch
        __hip_atomic_store(reinterpret_cast<unsigned int*>(ws + (size_t)row * p.n + col), __float_as_uint(acc[0][j][r]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                               // the slab is out; nobody reads the ring any more
  int* flag = reinterpret_cast<int*>(smem);
  if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(p.fix_cnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*flag != sk - 1) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (threadIdx.x == 0) __hip_atomic_store(p.fix_cnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float* __restrict__ all = p.c;
  const float* scale = W8 ? p.w8_scale : nullptr;
  // all loads of a thread first (8 elements x up to kMaxSk slabs in flight: one memory round trip, not one per element), sums in
  // split order afterwards
  constexpr int kMaxSk = 8, NE = 16 * 128 / (64 * NW);
  const bool swiglu = p.epilogue == CHATTS_EPI_SWIGLU;
  float part[NE][kMaxSk];
  size_t off[NE];
  bool live[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = threadIdx.x + i * 64 * NW;
    // SwiGLU: element (row, packed row n0 + c) of the slab, c = 0 .. 127 (gate and up rows alike: summed here, paired below)
    const int row = e >> 7, col = n0 + (e & 127);
    live[i] = row < p.m && col < p.n;
    off[i] = (size_t)row * p.n + col;
#pragma unroll
    for (int sp = 0; sp < kMaxSk; ++sp) part[i][sp] = (live[i] && sp < sk) ? all[sp * plane + off[i]] : 0.f;
  }
  float* xch = reinterpret_cast<float*>(smem) + 64;      // SwiGLU: the summed tile goes through LDS to pair gate with up
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = threadIdx.x + i * 64 * NW;
    const int row = e >> 7, col = n0 + (e & 127);
    float v = 0.f;
#pragma unroll
    for (int sp = 0; sp < kMaxSk; ++sp)
      if (sp < sk) v += part[i][sp];
    if (swiglu) { xch[e] = v; continue; }
    if (!live[i]) continue;
    if (scale) v *= scale[col];
    if (p.bias) v += p.bias[col];
    if (p.epilogue == CHATTS_EPI_GELU) v = gelu_erf_f(v);
    if (p.epilogue == CHATTS_EPI_RESID) v = p.resid[(size_t)row * p.ldc + col] + v;
    if (p.c_hi) store_planes(p.c_hi, p.c_lo, (size_t)row * p.ldcp + col, v);
    else p.c_out[(size_t)row * p.ldc + col] = v;
  }
  if (!swiglu) return;
  __syncthreads();
  for (int e = threadIdx.x; e < 16 * 64; e += 64 * NW) {
    const int row = e >> 6, cc = e & 63;
    const int loc = (cc >> 4) * 32 + (cc & 15), prow = n0 + loc;
    if (row >= p.m || prow + 16 >= p.n) continue;
    float g = xch[row * 128 + loc], u = xch[row * 128 + loc + 16];
    if (scale) { g *= scale[prow]; u *= scale[prow + 16]; }
    if (p.bias) { g += p.bias[prow]; u += p.bias[prow + 16]; }
    const int col = (prow >> 5) * 16 + (prow & 15);
    if (p.c_hi) store_planes(p.c_hi, p.c_lo, (size_t)row * p.ldcp + col, silu_g(g) * u);
    else p.c_out[(size_t)row * p.ldc + col] = silu_g(g) * u;
  }
}

// NW = waves per workgroup.  4: each wave owns 32 columns (two B fragments).  8 [fp8 W]: each owns 16 - the per-stage chain of a
// wave (LDS reads -> fp8 widening -> MFMAs) is what paces a workgroup when it is alone on its CU (a deeper ring changes nothing,
// profiles/r3_stream_sweep_fp8.txt), and two waves per SIMD overlap their chains.  Every output column still sees its K-steps in
// the same order with the same operands: bit-identical to NW = 4.
template <int NSTAGE, bool W8, int MB, int NW = 4, bool I8 = false>
__global__ __launch_bounds__(64 * NW) void gemm_stream_kernel(GemmParams p, const uint16_t* __restrict__ a_hi,
                                                              const uint16_t* __restrict__ a_lo, int ldp) {
  static_assert(!W8 || MB == 1, "the fp8 weight stream is a batched-decode format (M <= 16)");
  static_assert(NW == 4 || (NW == 8 && (W8 || MB % 2 == 0)), "8 waves: fp8 W (one A piece per wave), or bf16 W with an even block count");
  // wave grid inside the 16 MB x 128 tile: WM x WN.  4 waves: 1 x 4 (32 columns each).  8 waves, fp8 W (M <= 16): 1 x 8 (16 columns).
  // 8 waves, bf16 W with several row blocks (the TS-encoder MLP, short prefill chunks): 2 x 4 - each wave MB / 2 row blocks x 32 columns:
  // as many fragment reads and MFMAs per workgroup as with 4 waves, but the DMA pieces of a K-step (16 W + 4 MB A: 48 at MB = 8, ~150
  // cycles of issue apiece) are spread over twice the waves and two waves per SIMD overlap issue with MFMAs.
  constexpr int WM = (NW == 8 && !W8) ? 2 : 1, WN = NW / WM, MBW = MB / WM;
  constexpr int BN = kStreamBN, BK = stream_bk(W8), STAGE = stream_stage(W8, MB);
  constexpr int A_SUB = 16 * 128;                  // one A sub-block: 16 token rows x 64 K-values (128 B)
  constexpr int NSUB = BK / 64;                    // K sub-blocks per plane and stage (2 for fp8 W)
  constexpr int A_PLANE = MB * NSUB * A_SUB, W_OFF = 2 * A_PLANE;
  constexpr int NWP = 16 / NW;                     // W pieces per wave and stage
  constexpr int NAP = W8 ? (NW == 8 ? 1 : NSUB) : (NW == 8 ? MB / 2 : MB);   // A pieces per wave and stage
  constexpr int NPIECE = NWP + NAP;                // DMA pieces per wave and stage
  constexpr int FN = 8 / WN;                       // 16-column B fragments per wave
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * p.k_per_split;
  int kend = kbeg + p.k_per_split;
  if (kend > p.k) kend = p.k;
  const int nk = (kend - kbeg) / BK;

  // 1 KB DMA pieces (8 rows x 128 B): W pieces {wave + NW h}; of the A pieces (plane, block / K sub-block, token-row half) this
  // wave takes row half wave & 1 of plane wave >> 1 of EVERY block [bf16 W], or of both planes for K sub-block wave >> 1 [fp8 W,
  // 4 waves], or the one piece (plane wave >> 2, sub-block (wave >> 1) & 1) [fp8 W, 8 waves].  Every piece index this wave
  // touches has parity wave & 1.
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ (((wave & 1) << 2) | (lrow >> 1));
  const char* wsrc[NWP];
  const uint16_t* asrc[NAP];
  int adst[NAP];
#pragma unroll
  for (int h = 0; h < NWP; ++h) {
    int wr = n0 + (wave + NW * h) * 8 + lrow;
    if (wr > p.n - 1) wr = p.n - 1;
    wsrc[h] = W8 ? reinterpret_cast<const char*>(p.w8) + (size_t)wr * p.ldw8 + kbeg + lchunk * 16
                 : reinterpret_cast<const char*>(p.w + (size_t)wr * p.ldw + kbeg + lchunk * 8);
  }
#pragma unroll
  for (int q = 0; q < NAP; ++q) {
    // fp8 W: (plane, K sub-block) per wave as before.  bf16 W: 4 waves - plane wave >> 1 of EVERY block; 8 waves - plane (wave >> 1) & 1
    // of the blocks 2 q + (wave >> 2).  The row half is wave & 1 in all forms (the piece parity the swizzle above assumes).
    const int plane = W8 ? (NW == 8 ? (wave >> 2) : q) : ((wave >> 1) & 1);
    const int sub = W8 ? (NW == 8 ? ((wave >> 1) & 1) : (wave >> 1)) : 0;
    const int blk = W8 ? 0 : (NW == 8 ? 2 * q + (wave >> 2) : q);
    int am = blk * 16 + (wave & 1) * 8 + lrow;
    if (am > p.m - 1) am = p.m - 1;
    asrc[q] = (plane ? a_lo : a_hi) + (size_t)am * ldp + kbeg + sub * 64 + lchunk * 8;
    adst[q] = plane * A_PLANE + (blk * NSUB + sub) * A_SUB + (wave & 1) * 1024;
  }
  auto issue = [&](int kt) {
    char* base = smem + (kt % NSTAGE) * STAGE;
#pragma unroll
    for (int h = 0; h < NWP; ++h)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[h] + (size_t)kt * 128), (lptr_t)(base + W_OFF + (wave + NW * h) * 1024), 16, 0,
                                       2 /* nt: streamed once */);
#pragma unroll
    for (int q = 0; q < NAP; ++q)
      __builtin_amdgcn_global_load_lds((gptr_t)(asrc[q] + (size_t)kt * BK), (lptr_t)(base + adst[q]), 16, 0, 0);
  };

  f32x4 acc[MBW][FN];
#pragma unroll
  for (int b = 0; b < MBW; ++b)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) issue(s);

  const int frow = lane & 15, fchunk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int later = nk - 1 - kt;               // stages issued after kt that may still be in flight
    wait_stages<NPIECE, NSTAGE - 2>(later);
    __builtin_amdgcn_s_barrier();
    if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1);
    const char* base = smem + (kt % NSTAGE) * STAGE;
    constexpr int NH = BK / 32;                  // MFMA K-steps per stage
    bf16x8_t bfrag[NH][FN];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int r = wn * (16 * FN) + j * 16 + frow;
        if (W8) {   // 8 fp8 = 8 bytes at byte 32 h + 8 q of the row: 16-byte chunk 2 h + (q >> 1), half q & 1
          const u32x2 q8 = *reinterpret_cast<const u32x2*>(base + W_OFF + lds_off128(r, 2 * h + (fchunk >> 1)) + (fchunk & 1) * 8);
          typedef float f32x2_t __attribute__((ext_vector_type(2)));
          const f32x2_t f0 = __builtin_amdgcn_cvt_pk_f32_fp8(q8.x, false), f1 = __builtin_amdgcn_cvt_pk_f32_fp8(q8.x, true);
          const f32x2_t f2 = __builtin_amdgcn_cvt_pk_f32_fp8(q8.y, false), f3 = __builtin_amdgcn_cvt_pk_f32_fp8(q8.y, true);
          bf16x8_t bv;
          if (I8) {   // int8 copy: eight sign-extended bytes, exactly representable in bf16 (|q| <= 127)
            const int a0 = (int)q8.x, a1 = (int)q8.y;
            bv[0] = (__bf16)(float)((a0 << 24) >> 24); bv[1] = (__bf16)(float)((a0 << 16) >> 24);
            bv[2] = (__bf16)(float)((a0 << 8) >> 24);  bv[3] = (__bf16)(float)(a0 >> 24);
            bv[4] = (__bf16)(float)((a1 << 24) >> 24); bv[5] = (__bf16)(float)((a1 << 16) >> 24);
            bv[6] = (__bf16)(float)((a1 << 8) >> 24);  bv[7] = (__bf16)(float)(a1 >> 24);
          } else {
            bv[0] = (__bf16)f0.x; bv[1] = (__bf16)f0.y; bv[2] = (__bf16)f1.x; bv[3] = (__bf16)f1.y;
            bv[4] = (__bf16)f2.x; bv[5] = (__bf16)f2.y; bv[6] = (__bf16)f3.x; bv[7] = (__bf16)f3.y;
          }
          bfrag[h][j] = bv;
        } else {
          bfrag[h][j] = *reinterpret_cast<const bf16x8_t*>(base + W_OFF + lds_off128(r, h * 4 + fchunk));
        }
      }
    }
#pragma unroll
    for (int b = 0; b < MBW; ++b) {
      bf16x8_t alo[NH], ahi[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int aoff = ((wm * MBW + b) * NSUB + (h >> 1)) * A_SUB + lds_off128(frow, (h & 1) * 4 + fchunk);
        alo[h] = *reinterpret_cast<const bf16x8_t*>(base + A_PLANE + aoff);
        ahi[h] = *reinterpret_cast<const bf16x8_t*>(base + aoff);
      }
#pragma unroll
      for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[b][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[h], bfrag[h][j], acc[b][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[b][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[h], bfrag[h][j], acc[b][j], 0, 0, 0);
      }
    }
  }
  if constexpr (MB == 1) {
    if (!p.direct && p.fix_cnt) { stream_fixup<FN, NW, W8>(p, acc, n0, wave, lane, smem); return; }
  }
  if constexpr (FN == 1) {
    if (p.direct && p.epilogue == CHATTS_EPI_SWIGLU) {
      // a (gate, up) pair of 16-column fragments sits in waves (2 q, 2 q + 1): the odd wave hands its accumulator over through LDS
      // and the even one runs the two-fragment epilogue
      __syncthreads();                           // every wave is done reading the last stage
      f32x4* xch = reinterpret_cast<f32x4*>(smem);
      if (wave & 1) xch[(wave >> 1) * 64 + lane] = acc[0][0];
      __syncthreads();
      if (!(wave & 1)) {
        f32x4 pair[1][2] = {{acc[0][0], xch[(wave >> 1) * 64 + lane]}};
        gemm_store<1, 2, 16, 32, W8>(p, pair, 0, n0, 0, wave >> 1, lane, blockIdx.z);
      }
      return;
    }
    gemm_store<MBW, 1, 16 * MBW, 16, W8>(p, acc, 0, n0, wm, wn, lane, blockIdx.z);
  } else {
    gemm_store<MBW, 2, 16 * MBW, 32, W8>(p, acc, 0, n0, wm, wn, lane, blockIdx.z);
  }
}

// x = hi + lo (to 16 mantissa bits): one thread per 8 consecutive elements.
__global__ __launch_bounds__(256) void split_bf16x2_kernel(const float* __restrict__ x, int m, int k, int ldx,
                                                          uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int ldp) {
  const int kc = k >> 3;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)m * kc) return;
  const int row = (int)(idx / kc), c = (int)(idx % kc);
  const float* src = x + (size_t)row * ldx + c * 8;
  const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
  const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
  bf16x8_t hv, lv;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hv[j] = h;
    lv[j] = (__bf16)(v[j] - (float)h);
  }
  *reinterpret_cast<bf16x8_t*>(hi + (size_t)row * ldp + c * 8) = hv;
  *reinterpret_cast<bf16x8_t*>(lo + (size_t)row * ldp + c * 8) = lv;
}

int launch_split_bf16x2(const float* x, int m, int k, int ldx, uint16_t* hi, uint16_t* lo, int ldp, hipStream_t s) {
  const size_t total = (size_t)m * (k >> 3);
  hipLaunchKernelGGL(split_bf16x2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, m, k, ldx, hi, lo, ldp);
  CHATTS_CHECK_LAUNCH("split_bf16x2");
  return CHATTS_OK;
}

// Sum split-K partials in a fixed order and apply the epilogue.  One thread per output element.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ ws, int sk, int m, int n,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ resid, float* __restrict__ c,
                                                             int ldc, int epilogue, const float* __restrict__ scale,
                                                             uint16_t* __restrict__ c_hi, uint16_t* __restrict__ c_lo, int ldcp,
                                                             int sk_T, int sk_nk) {
  const int ncols = epilogue == CHATTS_EPI_SWIGLU ? n / 2 : n;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)m * ncols) return;
  const int row = (int)(idx / ncols), col = (int)(idx % ncols);
  const size_t plane = (size_t)m * n;
  if (sk_T > 0) {      // stream-K: a tile has as many slabs as pieces (the SwiGLU layout never takes this path)
    const int wcol = col;
    sk = sk_tile_nseg((wcol / 256) * ((m + 127) / 128) + row / 128, sk_T, sk_nk);
  }
  if (epilogue == CHATTS_EPI_SWIGLU) {
    const int prow = (col >> 4) * 32 + (col & 15);
    float g = 0.f, u = 0.f;
    for (int s = 0; s < sk; ++s) {
      g += ws[s * plane + (size_t)row * n + prow];
      u += ws[s * plane + (size_t)row * n + prow + 16];
    }
    if (scale) { g *= scale[prow]; u *= scale[prow + 16]; }
    if (bias) { g += bias[prow]; u += bias[prow + 16]; }
    if (c_hi) store_planes(c_hi, c_lo, (size_t)row * ldcp + col, silu_g(g) * u);
    else c[(size_t)row * ldc + col] = silu_g(g) * u;
    return;
  }
  float v = 0.f;
  for (int s = 0; s < sk; ++s) v += ws[s * plane + (size_t)row * n + col];
  if (scale) v *= scale[col];
  if (bias) v += bias[col];
  if (epilogue == CHATTS_EPI_GELU) v = gelu_erf_f(v);
  if (epilogue == CHATTS_EPI_RESID) v = resid[(size_t)row * ldc + col] + v;
  if (c_hi) store_planes(c_hi, c_lo, (size_t)row * ldcp + col, v);
  else c[(size_t)row * ldc + col] = v;
}

// The same epilogue, four columns per thread (N % 4 == 0, sk <= 8, no SwiGLU pairing, no stream-K): 16-byte loads instead of 4-byte
// ones and all slabs of a thread requested at once.  Same sums in the same order per element: bit-identical to the kernel above.
__global__ __launch_bounds__(256) void splitk_epilogue_v4_kernel(const float* __restrict__ ws, int sk, int m, int n,
                                                                const float* __restrict__ bias, const float* __restrict__ resid,
                                                                float* __restrict__ c, int ldc, int epilogue,
                                                                const float* __restrict__ scale, uint16_t* __restrict__ c_hi,
                                                                uint16_t* __restrict__ c_lo, int ldcp) {
  const int n4 = n >> 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)m * n4) return;
  const int row = (int)(idx / n4), col = (int)(idx % n4) * 4;
  const size_t plane = (size_t)m * n, at = (size_t)row * n + col;
  f32x4 t8[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) t8[s] = *reinterpret_cast<const f32x4*>(ws + (s < sk ? s : sk - 1) * plane + at);
  f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
  if (epilogue == CHATTS_EPI_RESID) r4 = *reinterpret_cast<const f32x4*>(resid + (size_t)row * ldc + col);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s)
    if (s < sk) { v[0] += t8[s].x; v[1] += t8[s].y; v[2] += t8[s].z; v[3] += t8[s].w; }
  const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (scale) v[j] *= scale[col + j];
    if (bias) v[j] += bias[col + j];
    if (epilogue == CHATTS_EPI_GELU) v[j] = gelu_erf_f(v[j]);
    if (epilogue == CHATTS_EPI_RESID) v[j] = rr[j] + v[j];
  }
  if (c_hi) {
    if ((ldcp & 3) == 0) {      // both planes as one 8-byte store each (store_planes' arithmetic)
#pragma clang fp contract(off)
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      bf16x4_t hv, lv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16 h = (__bf16)v[j];
        hv[j] = h;
        lv[j] = (__bf16)(v[j] - (float)h);
      }
      *reinterpret_cast<bf16x4_t*>(c_hi + (size_t)row * ldcp + col) = hv;
      *reinterpret_cast<bf16x4_t*>(c_lo + (size_t)row * ldcp + col) = lv;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) store_planes(c_hi, c_lo, (size_t)row * ldcp + col + j, v[j]);
    }
  } else {
    *reinterpret_cast<f32x4*>(c + (size_t)row * ldc + col) = (f32x4){v[0], v[1], v[2], v[3]};
  }
}

// Split-K epilogue fused with the RMSNorm that consumes its result: one workgroup per output row sums the partials
// (fixed order), applies bias / residual, writes the row, and - the row's sum of squares being at hand - writes
// norm_w * (row * rsqrt(mean(row^2) + eps)) as bf16 hi / lo planes, the next projection's operand.  Replaces the
// element-wise epilogue launch and the rmsnorm_planes launch behind it (same arithmetic, same order: bit-identical).
__global__ __launch_bounds__(256) void splitk_epilogue_norm_kernel(const float* __restrict__ ws, int sk, int m, int n,
                                                                  const float* __restrict__ bias, const float* __restrict__ resid,
                                                                  float* __restrict__ c, int ldc, int epilogue,
                                                                  const float* __restrict__ scale, const float* __restrict__ norm_w,
                                                                  float eps, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                                  int ldp, int sk_T, int sk_nk) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const size_t plane = (size_t)m * n;
  float ss = 0.f;
  for (int col = threadIdx.x * 4; col < n; col += 1024) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (sk_T > 0) sk = sk_tile_nseg((col / 256) * ((m + 127) / 128) + row / 128, sk_T, sk_nk);    // stream-K: slabs of THIS tile
    if (sk <= 4) {          // the usual split counts: all slabs requested at once, summed in split order (same sums, one round trip)
      f32x4 t4[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) t4[s] = *reinterpret_cast<const f32x4*>(ws + (s < sk ? s : sk - 1) * plane + (size_t)row * n + col);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s < sk) { v.x += t4[s].x; v.y += t4[s].y; v.z += t4[s].z; v.w += t4[s].w; }
    } else {
      for (int s = 0; s < sk; ++s) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(ws + s * plane + (size_t)row * n + col);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
    }
    if (scale) { const f32x4 t = *reinterpret_cast<const f32x4*>(scale + col); v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w; }
    if (bias) { const f32x4 t = *reinterpret_cast<const f32x4*>(bias + col); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (epilogue == CHATTS_EPI_RESID) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(resid + (size_t)row * ldc + col);
      v.x = t.x + v.x; v.y = t.y + v.y; v.z = t.z + v.z; v.w = t.w + v.w;
    }
    *reinterpret_cast<f32x4*>(c + (size_t)row * ldc + col) = v;
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;        // same accumulation pattern as rmsnorm_kernel
  }
  ss = block_sum<4>(ss, red);
  const float rstd = rsqrtf(ss / (float)n + eps);
  for (int col = threadIdx.x * 4; col < n; col += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(c + (size_t)row * ldc + col);    // this thread's own stores
    const f32x4 g = *reinterpret_cast<const f32x4*>(norm_w + col);
    const float o[4] = {g.x * (v.x * rstd), g.y * (v.y * rstd), g.z * (v.z * rstd), g.w * (v.w * rstd)};
    {
#pragma clang fp contract(off)   // lo must be the split of the ROUNDED product, as in rmsnorm_kernel<true>
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      bf16x4_t hv, lv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16 h = (__bf16)o[j];
        hv[j] = h;
        lv[j] = (__bf16)(o[j] - (float)h);
      }
      *reinterpret_cast<bf16x4_t*>(hi + (size_t)row * ldp + col) = hv;
      *reinterpret_cast<bf16x4_t*>(lo + (size_t)row * ldp + col) = lv;
    }
  }
}

// The same kernel for FEW rows (batched decode, M <= 16): one workgroup per row leaves the chip idle (16 workgroups), so gridDim.y = Q
// workgroups share a row.  Each of them sums the slabs of the WHOLE row - it needs the row's sum of squares, and forming it with the same
// thread-to-column mapping as above keeps every bit - but stores c and the planes only for its own 256 / Q threads' columns.  The
// re-read of the slabs by Q workgroups is L2 traffic (Q x sk x M x N x 4 bytes: 16 MB at Q = 8, sk = 6, 16 x 5120).  Because a
// workgroup reads resid columns that another one updates, c must NOT alias resid here (the batched decoder ping-pongs x / xn).
template <int kMaxIt>
__global__ __launch_bounds__(256) void splitk_epilogue_norm_q_kernel(const float* __restrict__ ws, int sk, int m, int n,
                                                                    const float* __restrict__ bias, const float* __restrict__ resid,
                                                                    float* __restrict__ c, int ldc, int epilogue,
                                                                    const float* __restrict__ scale, const float* __restrict__ norm_w,
                                                                    float eps, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                                    int ldp) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const bool mine = (int)(threadIdx.x * gridDim.y / 256) == (int)blockIdx.y;
  const size_t plane = (size_t)m * n;
  f32x4 keep[kMaxIt];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < kMaxIt; ++it) {
    const int col = threadIdx.x * 4 + it * 1024;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col < n) {
      // all slabs of this column group requested at once (sk <= 8, the clamp re-reads the last slab and the value is dropped): with
      // 16 workgroups' worth of rows there is no other wave to hide a chain of sk dependent round trips behind
      f32x4 t8[8];
#pragma unroll
      for (int s = 0; s < 8; ++s)
        t8[s] = *reinterpret_cast<const f32x4*>(ws + (s < sk ? s : sk - 1) * plane + (size_t)row * n + col);
      f32x4 tr = {0.f, 0.f, 0.f, 0.f};
      if (epilogue == CHATTS_EPI_RESID) tr = *reinterpret_cast<const f32x4*>(resid + (size_t)row * ldc + col);
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (s < sk) { v.x += t8[s].x; v.y += t8[s].y; v.z += t8[s].z; v.w += t8[s].w; }
      if (scale) { const f32x4 t = *reinterpret_cast<const f32x4*>(scale + col); v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w; }
      if (bias) { const f32x4 t = *reinterpret_cast<const f32x4*>(bias + col); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
      if (epilogue == CHATTS_EPI_RESID) { v.x = tr.x + v.x; v.y = tr.y + v.y; v.z = tr.z + v.z; v.w = tr.w + v.w; }
      if (mine) *reinterpret_cast<f32x4*>(c + (size_t)row * ldc + col) = v;
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;        // same accumulation pattern as rmsnorm_kernel
    }
    keep[it] = v;
  }
  ss = block_sum<4>(ss, red);
  const float rstd = rsqrtf(ss / (float)n + eps);
  if (!mine) return;
#pragma unroll
  for (int it = 0; it < kMaxIt; ++it) {
    const int col = threadIdx.x * 4 + it * 1024;
    if (col >= n) continue;
    const f32x4 v = keep[it];
    const f32x4 g = *reinterpret_cast<const f32x4*>(norm_w + col);
    const float o[4] = {g.x * (v.x * rstd), g.y * (v.y * rstd), g.z * (v.z * rstd), g.w * (v.w * rstd)};
    {
#pragma clang fp contract(off)   // lo must be the split of the ROUNDED product, as in rmsnorm_kernel<true>
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      bf16x4_t hv, lv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16 h = (__bf16)o[j];
        hv[j] = h;
        lv[j] = (__bf16)(o[j] - (float)h);
      }
      *reinterpret_cast<bf16x4_t*>(hi + (size_t)row * ldp + col) = hv;
      *reinterpret_cast<bf16x4_t*>(lo + (size_t)row * ldp + col) = lv;
    }
  }
}

// The qkv projection's split-K epilogue that also does rope_kv_kernel's work (prefill): one wave per (token, head) sums the head's
// 128 columns over the slabs in split order, adds the bias - splitk_epilogue_kernel's arithmetic - and hands the two values per lane
// to rope_kv_head: q rotated into c, K / V rows into the cache.  One pass over the [T, 7168] rows less per layer.
__global__ __launch_bounds__(256) void splitk_epilogue_rope_kernel(const float* __restrict__ ws, int sk, int m, int n,
                                                                  const float* __restrict__ bias, float* __restrict__ c, int ldc,
                                                                  RopeFuse r) {
  const int heads = r.n_q + 2 * r.n_kv;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (gw >= m * heads) return;
  const int tok = gw / heads, h = gw - tok * heads;
  const size_t plane = (size_t)m * n, at = (size_t)tok * n + h * kHeadDim + lane;
  float a = 0.f, b = 0.f;
  int s = 0;
  for (; s + 4 <= sk; s += 4) {                  // four slabs in flight, summed in split order
    float ta[4], tb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { ta[u] = ws[(s + u) * plane + at]; tb[u] = ws[(s + u) * plane + at + 64]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { a += ta[u]; b += tb[u]; }
  }
  for (; s < sk; ++s) { a += ws[s * plane + at]; b += ws[s * plane + at + 64]; }
  if (bias) { a += bias[h * kHeadDim + lane]; b += bias[h * kHeadDim + lane + 64]; }
  const int pos = (r.pos0_dev ? *r.pos0_dev : r.pos0) + tok;
  rope_kv_head(a, b, h, pos, lane, c + (size_t)tok * ldc + h * kHeadDim, r);
}

static int gemm_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

static void pick_geometry(int m, int n, int k, int& bm, int& sk, int slots_per_cu = 3, int bn = 128) {
  bm = m > 64 ? 128 : (m > 32 ? 64 : (m > 16 ? 32 : 16));
  const int force_bm = gemm_env_int("CHATTS_GEMM_BM", 0);          // tuning / tests only
  if (force_bm == 16 || force_bm == 32 || force_bm == 64 || force_bm == 128) bm = force_bm;
  // short K (no split-K possible, e.g. the first TS-MLP layer, K = 288) and few tiles: smaller M-tiles fill more CUs
  if (!force_bm && k < 512 && ((m + bm - 1) / bm) * ((n + 127) / 128) * 4 <= device_cus() && bm > 32) bm = 32;
  const int tiles = ((m + bm - 1) / bm) * ((n + bn - 1) / bn);
  // Split-K so that the workgroups fill whole "rounds" of the resident slots (3 workgroups of 48 KB LDS per CU):
  // efficiency of a launch = blocks / (slots * ceil(blocks / slots)); split-K costs a partials round trip + an
  // epilogue launch, hence the small penalty.  (tools/gemm_sweep.py: qkv @ M=798 wants 3, o/down 2, gate_up 1.)
  const int slots = slots_per_cu * device_cus();
  int max_sk = k / 256 > 0 ? (k / 256 < 16 ? k / 256 : 16) : 1;   // keep >= 8 K-steps per split
  const int traffic_cap = k / (2 * m) > 1 ? k / (2 * m) : 1;        // partials (sk*M*N*8 B) <= 2x the weight bytes
  if (max_sk > traffic_cap) max_sk = traffic_cap;
  sk = 1;
  float best = -1.f;
  for (int cand = 1; cand <= max_sk; ++cand) {
    const int blocks = tiles * cand;
    const int rounds = (blocks + slots - 1) / slots;
    const float eff = (float)blocks / (float)(slots * rounds);
    const float score = eff - 0.08f * (m < 800 ? (float)m / 800.f : 1.f) * (cand - 1);   // partials cost grows with M
    if (score > best) { best = score; sk = cand; }
  }
  const int force_sk = gemm_env_int("CHATTS_GEMM_SK", 0);
  if (force_sk > 0 && force_sk <= 16 && k / force_sk >= 32) sk = force_sk;
}

static int k_per_split(int k, int sk, int bk = 32) {
  int kps = (k + sk - 1) / sk;
  return ((kps + bk - 1) / bk) * bk;
}

// The LDS-DMA kernel runs when the caller supplies the pre-split planes and either M >= kDmaMinM (its M-tile is a
// fixed 128 rows) or there is no float32 A to fall back on.
constexpr int kDmaMinM = 96;

static bool use_dma(const ChattsLinearArgs* a) {
  if (!a->a_hi || !a->a_lo || a->k % kDmaBK != 0) return false;
  if (!a->a) return true;
  return a->m >= gemm_env_int("CHATTS_GEMM_DMA_MIN_M", kDmaMinM) && gemm_env_int("CHATTS_GEMM_DMA", 1) != 0;
}

// The round-5 prefill kernel (gemm_ring.hip) takes every call the LDS-DMA kernel took whose epilogue operands allow 16-byte accesses
// (its lanes hold 4 consecutive output columns).  CHATTS_GEMM_RING=0 keeps the round-4 kernel for A/B runs.
static bool use_ring(const ChattsLinearArgs* a) {
  if (gemm_env_int("CHATTS_GEMM_RING", 1) == 0) return false;
  if (a->w8 || a->k % 64 != 0 || a->ldc % 4 != 0 || a->n % 16 != 0) return false;
  if (((uintptr_t)a->bias % 16) || ((uintptr_t)a->resid % 16) || ((uintptr_t)a->c % 16)) return false;
  if (a->c_hi && (a->ld_cplanes % 4 != 0 || ((uintptr_t)a->c_hi % 8) || ((uintptr_t)a->c_lo % 8))) return false;
  return true;
}
static void pick_ring(const ChattsLinearArgs* a, RingGeom& g) {
  ring_pick(a->m, a->n, a->k, device_cus(), gemm_env_int("CHATTS_GEMM_T", 0), gemm_env_int("CHATTS_GEMM_SK", 0), g);
}

// DMA geometry: 128 x 256 tiles, one 8-wave workgroup per CU.
static void pick_dma_geometry(int m, int n, int k, int& sk) {
  // Few tiles (the TS-encoder MLP: P <= 128 patches x 5120 columns = 20 tiles): split K as far as ONE round of workgroups
  // allows - the grid is padded to 8 * ceil(tiles / 8) per split, and one workgroup more than the CUs costs a whole second
  // round (profiles/r2_ts_gemm_sweep.txt, P = 128: 10 splits 33.7 us, 12 splits 47.7 us, 1 split 133 us; K = 320: 2 splits
  // 16.4 us, 1 split 28.0 us).
  const int tiles = ((m + 127) / 128) * ((n + kDmaBN - 1) / kDmaBN), gx = 8 * ((tiles + 7) / 8), cus = device_cus();
  if (2 * gx <= cus && gemm_env_int("CHATTS_GEMM_SK", 0) == 0) {
    sk = cus / gx;
    const int nk = k / kDmaBK;
    if (sk > nk / 2) sk = nk / 2;          // >= 2 K-steps per split
    if (sk > 16) sk = 16;
    if (sk < 1) sk = 1;
    return;
  }
  int bm;
  pick_geometry(m < 128 ? 128 : m, n, k, bm, sk, 1, kDmaBN);
}

// Streaming kernel (2 <= M <= 16 with planes): split-K so that ~2 workgroups per CU are busy, >= 4 K-steps per split.
// ... and the multi-block form (17 <= M <= 128, bf16 W) when the 128 x 256 tiling would leave most CUs without a tile
// (fewer N-panels than half the CUs: the TS-encoder MLP, o / down / qkv of a short prefill chunk)
// Measured (round 2, one TS-MLP layer 5120 x 5120, profiles/r2_ts_gemm_sweep.txt + tools/jobs/r2_job15.sh): P = 32: 18.7 us (LDS-DMA
// kernel 23.7), P = 64: 23.5 (26.0), P = 128: 35.7 (33.9) - with 8 row blocks the four waves issue 12 DMA pieces each per K-step
// (~150 cycles apiece) in the same instruction stream as their 64 MFMAs, which is what the prefill kernel's loader waves exist
// to avoid; and a short K (layer 0: 5 K-steps) is served better by the 2-way split of the 128 x 256 tiles.  Hence M <= 64, K >= 1024.
static bool stream_multiblock(int m, int n, int k, bool w8) {
  // round 3: with 8 waves as 2 x 4 the streaming form also wins at 65 .. 128 rows on the 5120-column shapes (P = 128, one TS-MLP layer:
  // 30.1 us against 33.2 for the LDS-DMA kernel and 35.0 for the 4-wave form, epilogue included - profiles/r3_ts_gemm_sweep.txt)
  const int max_m = gemm_env_int("CHATTS_GEMM_STREAM_MB", n <= 8192 && gemm_env_int("CHATTS_GEMM_STREAM_MB_WAVES", 8) == 8 ? 128 : 64);
  return m > 16 && m <= max_m && m <= 128 && !w8 && k % 64 == 0 && k >= 1024 &&
         2 * 8 * (((n + kDmaBN - 1) / kDmaBN + 7) / 8) <= device_cus();
}
static bool use_stream(const ChattsLinearArgs* a) {
  if (!(a->a_hi && a->a_lo) || gemm_env_int("CHATTS_GEMM_STREAM", 1) == 0) return false;
  if (a->m >= 2 && a->m <= 16) return a->k % stream_bk(a->w8 != nullptr) == 0;
  return stream_multiblock(a->m, a->n, a->k, a->w8 != nullptr);
}

static int pick_stream_sk(int n, int k, bool w8) {
  const int tiles = (n + kStreamBN - 1) / kStreamBN, nk = k / stream_bk(w8);
  // ~one workgroup per CU (tools/stream_sweep.py: gate_up / lm_head want no split, qkv 4, o / down ~6); each split costs a
  // partials round trip and the epilogue launch
  const int cus = device_cus();
  int sk = 4 * tiles >= 3 * cus ? 1 : (cus + tiles / 2) / tiles;
  // fp8 W: a workgroup's 4 x 24 KB ring leaves room for ONE per CU, so a grid beyond the CU count runs a second, nearly empty round
  // (qkv, 56 tiles: 5 splits = 280 workgroups 19.3 us, 4 = 224 workgroups 13.9 us - profiles/r3_stream_sweep_fp8.txt)
  if (w8 && sk > 1 && tiles * sk > cus) sk = cus / tiles;
  if (sk > nk / 4) sk = nk / 4;
  if (sk < 1) sk = 1;
  const int force_sk = gemm_env_int("CHATTS_GEMM_SK", 0);
  if (force_sk > 0 && force_sk <= nk) sk = force_sk;
  return sk;
}

template <int NSTAGE, bool W8, int MB = 1, int NW = 4, bool I8 = false>
static int launch_stream_t(const GemmParams& p, const ChattsLinearArgs* a, int sk, hipStream_t s) {
  constexpr int LDS = NSTAGE * stream_stage(W8, MB);
  static bool configured = false;
  if (!configured) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_kernel<NSTAGE, W8, MB, NW, I8>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    CHATTS_REQUIRE(e == hipSuccess, CHATTS_E_LAUNCH, "gemm_stream: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(e));
    configured = true;
  }
  dim3 grid((a->n + kStreamBN - 1) / kStreamBN, 1, sk), block(64 * NW);
  hipLaunchKernelGGL((gemm_stream_kernel<NSTAGE, W8, MB, NW, I8>), grid, block, LDS, s, p, a->a_hi, a->a_lo, a->ld_planes);
  return CHATTS_OK;
}

// Stream-K for the LDS-DMA kernel (see sk_bound): worth it when the tiles do not fill whole rounds of workgroups.  Estimates
// in K-steps (+ kSkOv per workgroup for the pipeline fill / drain; + a partials round trip when the uniform choice would have
// needed none): o_proj @ M = 798: 2 rounds x 27 -> 43.75; down_proj 2 x 72 -> 118; qkv 80 -> 61; gate_up (756 tiles = 2.95
// rounds) keeps whole tiles.  Returns the slab count (max pieces per tile), 0 = keep the uniform split.
static int pick_streamk(int m, int n, int k, int uniform_sk) {
  // MEASURED (round 2, profiles/r2_streamk_ttft.txt): correct (tests) and balanced, but SLOWER - TTFT 54.6 ms against 45.6 with the
  // uniform split.  The uniform split runs the M-tiles of a W panel in lockstep on one XCD, so a K-slice of the panel is fetched
  // once into that L2 and reused by all 7 M-tiles; stream-K ranges walk the same panel at different K offsets at any moment,
  // and every tile re-fetches its W slices.  Hence OFF unless CHATTS_GEMM_STREAMK=1 (kept for the tests and for shapes with one
  // M-tile, where there is nothing to share).
  const int force = gemm_env_int("CHATTS_GEMM_STREAMK", 0);
  if (force <= 0 || m < 256 || k % kDmaBK != 0) return 0;
  const int mt = (m + 127) / 128, nt = (n + kDmaBN - 1) / kDmaBN, tiles = mt * nt, nk = k / kDmaBK;
  const long long T = (long long)tiles * nk;
  if (T > (1ll << 30) || T < 2 * kSkRanges) return 0;
  const double per = (double)T / kSkRanges;
  if (per >= nk) return 0;                                  // a range would span whole tiles: plain tiles are as good
  int slabs = 0;
  for (int t = 0; t < tiles; ++t) {
    const int ns = sk_tile_nseg(t, (int)T, nk);
    if (ns > slabs) slabs = ns;
  }
  if (slabs > 4) return 0;                                  // too many partials per tile
  if (force == 1) return slabs;
  const double kSkOv = 2.0;
  const int cus = device_cus();
  const long long units = (long long)tiles * uniform_sk;
  const double uniform = (double)((units + cus - 1) / cus) * ((double)nk / uniform_sk + kSkOv);
  const double stream = per + 2 * kSkOv + (uniform_sk == 1 ? 8.0 : 0.0);
  return stream < 0.93 * uniform ? slabs : 0;
}

size_t gemm_workspace(int m, int n, int k) {
  int bm, sk, sk2;
  pick_geometry(m, n, k, bm, sk);
  pick_dma_geometry(m, n, k, sk2);
  if (sk2 > sk) sk = sk2;
  if (m >= kDmaMinM && k % 64 == 0) {          // the ring kernel's own split choice
    RingGeom g;
    ring_pick(m, n, k, device_cus(), gemm_env_int("CHATTS_GEMM_T", 0), gemm_env_int("CHATTS_GEMM_SK", 0), g);
    if (g.sk > sk) sk = g.sk;
  }
  if (m >= 256 && k % kDmaBK == 0) {
    const int slabs = pick_streamk(m, n, k, sk2);
    if (slabs > sk) sk = slabs;
  }
  if ((m <= 16 && k % 64 == 0) || stream_multiblock(m, n, k, false)) {
    sk2 = pick_stream_sk(n, k, false);         // (the fp8 variant's K-steps are twice as long: never more splits)
    if (sk2 > sk) sk = sk2;
  }
  return sk > 1 ? (size_t)sk * m * n * sizeof(float) : 0;
}

// the speed mode is a process-wide choice read per call (tests and tools flip it between calls): CHATTS_GEMM_PRECISION=bf16
static bool dma_single_pass() {
  const char* v = getenv("CHATTS_GEMM_PRECISION");
  return v && strcmp(v, "bf16") == 0;
}

template <bool SINGLE, bool W32>
static int launch_dma_t(const GemmParams& p, const ChattsLinearArgs* a, int sk, hipStream_t s) {
  static bool configured = false;     // > 64 KB of dynamic LDS must be opted into once
  if (!configured) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dma_kernel<SINGLE, W32>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kDmaLds);
    CHATTS_REQUIRE(e == hipSuccess, CHATTS_E_LAUNCH, "gemm_dma: cannot reserve %d bytes of LDS: %s", kDmaLds, hipGetErrorString(e));
    configured = true;
  }
  const int tiles = ((a->n + kDmaBN - 1) / kDmaBN) * ((a->m + 127) / 128);
  dim3 grid(p.sk_T > 0 ? 8 * 64 : 8 * ((tiles + 7) / 8), 1, p.sk_T > 0 ? 1 : sk), block(W32 ? 512 : kDmaThreads);
  hipLaunchKernelGGL((gemm_dma_kernel<SINGLE, W32>), grid, block, kDmaLds, s, p, a->a_hi, a->a_lo, a->ld_planes);
  return CHATTS_OK;
}

// CHATTS_GEMM_DMA32=1: the 4 x (64 x 128) compute waves on the 32x32x16 MFMA instead of the 8 x (64 x 64) ones on 16x16x32.
// MEASURED (round 3, profiles/r3_gemm_dma32_sweep.txt, M = 798): correct (every GEMM test passes with it) and a third fewer LDS
// fragment bytes per K-step, but SLOWER - gate_up 537 against 363 us, down 272 / 212, qkv 168 / 112: with ONE compute wave per SIMD
// every s_waitcnt and every barrier idles that SIMD's matrix pipe (two co-resident waves cover each other's stalls), which costs
// more than the LDS traffic it saves.  Hence OFF by default; kept for the tests and as the starting point of a hand-scheduled version.
static int launch_dma(const GemmParams& p, const ChattsLinearArgs* a, int sk, hipStream_t s) {
  const bool w32 = gemm_env_int("CHATTS_GEMM_DMA32", 0) != 0;
  if (dma_single_pass()) return w32 ? launch_dma_t<true, true>(p, a, sk, s) : launch_dma_t<true, false>(p, a, sk, s);
  return w32 ? launch_dma_t<false, true>(p, a, sk, s) : launch_dma_t<false, false>(p, a, sk, s);
}

int launch_gemm(const ChattsLinearArgs* a_in, hipStream_t s, const RopeFuse* rope, bool* rope_done, SlabOut* slabs) {
  if (slabs) slabs->sk = 0;
  int bm, sk;
  ChattsLinearArgs a_copy;
  const ChattsLinearArgs* a = a_in;
  if (a_in->w8 && a_in->w8_format == CHATTS_W8_INT8 && !use_stream(a_in)) {      // int8 copies: the weight-streaming kernel only
    a_copy = *a_in;
    a_copy.w8 = nullptr; a_copy.w8_scale = nullptr;
    a = &a_copy;
  }
  const bool stream = use_stream(a);
  const bool dma = !stream && use_dma(a);
  bool ring = false;
  RingGeom rg{};
  if (stream) {
    sk = pick_stream_sk(a->n, a->k, a->w8 != nullptr);
    bm = 16;
  } else if (dma) {
    ring = use_ring(a);
    if (ring) { pick_ring(a, rg); sk = rg.sk; }
    else pick_dma_geometry(a->m, a->n, a->k, sk);
    bm = 128;
  } else {
    CHATTS_REQUIRE(a->a, CHATTS_E_SHAPE, "linear: a == NULL needs K %% %d == 0 (K=%d) for the plane path", kDmaBK, a->k);
    pick_geometry(a->m, a->n, a->k, bm, sk);
  }
  int kps = k_per_split(a->k, sk, stream ? stream_bk(a->w8 != nullptr) : (dma ? kDmaBK : 32));
  sk = (a->k + kps - 1) / kps;
  GemmParams p;
  p.sk_T = 0; p.sk_nk = 0;
  if (dma && !ring && a->epilogue != CHATTS_EPI_SWIGLU) {
    const int slabs = pick_streamk(a->m, a->n, a->k, sk);
    if (slabs > 0) {                     // stream-K: every tile goes through `slabs` split-K slabs (some tiles use fewer)
      const int tiles = ((a->n + kDmaBN - 1) / kDmaBN) * ((a->m + 127) / 128);
      p.sk_nk = a->k / kDmaBK;
      p.sk_T = tiles * p.sk_nk;
      sk = slabs > 1 ? slabs : 2;        // > 1: results are summed by the split-K epilogue
      kps = a->k;
    }
  }
  p.a = a->a; p.w = a->w; p.bias = a->bias; p.resid = a->resid;
  p.m = a->m; p.n = a->n; p.k = a->k; p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc;
  p.epilogue = a->epilogue; p.k_per_split = kps; p.direct = sk == 1;
  p.w8 = a->w8; p.w8_scale = a->w8_scale; p.ldw8 = a->ldw8; p.w8_format = a->w8_format;
  p.c_hi = a->c_hi; p.c_lo = a->c_lo; p.ldcp = a->ld_cplanes;
  p.fix_cnt = nullptr; p.c_out = a->c;
  if (stream && sk > 1 && sk <= 8 && a->m <= 16 && a->tile_counters && !a->post_norm_w && (a->n + kStreamBN - 1) / kStreamBN <= CHATTS_TILE_COUNTERS &&
      gemm_env_int("CHATTS_GEMM_FIXUP", 0) != 0)      // MEASURED slower than the epilogue launch (config 5: 7.32 against 6.78 ms per
                                                       // step, profiles/r3_cfg5_split_k_fixup_ab.txt): a captured launch costs ~2 us,
                                                       // the serial tail of the last workgroup + write-through slabs cost more

    p.fix_cnt = a->tile_counters;
  if (sk > 1) {
    const size_t need = (size_t)sk * a->m * a->n * sizeof(float);
    CHATTS_REQUIRE(a->workspace && a->workspace_bytes >= need, CHATTS_E_WORKSPACE,
                   "linear: split-K needs %zu workspace bytes, got %zu", need, a->workspace_bytes);
    p.c = reinterpret_cast<float*>(a->workspace);
  } else {
    p.c = a->c;
  }
  const int nt_count = (a->n + 127) / 128, mt_count = (a->m + bm - 1) / bm;
  dim3 grid(8 * ((nt_count + 7) / 8) * mt_count, 1, sk), block(256);
  if (stream) {
    int rc;
    const int stages = gemm_env_int("CHATTS_GEMM_STREAM_STAGES", 4);
    const bool w8x = gemm_env_int("CHATTS_GEMM_STREAM_MB_WAVES", 8) == 8;   // 8 waves as 2 x 4 for the multi-block forms (bit-identical to 4)
    if (a->m > 64) rc = w8x ? launch_stream_t<3, false, 8, 8>(p, a, sk, s) : launch_stream_t<3, false, 8>(p, a, sk, s);   // 3 x 48 KB stages: one workgroup per CU
    else if (a->m > 32) rc = w8x ? launch_stream_t<4, false, 4, 8>(p, a, sk, s) : launch_stream_t<4, false, 4>(p, a, sk, s);     // 4 x 32 KB
    else if (a->m > 16) rc = w8x ? launch_stream_t<4, false, 2, 8>(p, a, sk, s) : launch_stream_t<4, false, 2>(p, a, sk, s);     // 4 x 24 KB
    else if (a->w8 && a->w8_format == CHATTS_W8_INT8)
      rc = stages == 3 ? launch_stream_t<3, true, 1, 4, true>(p, a, sk, s) : launch_stream_t<4, true, 1, 4, true>(p, a, sk, s);
    else if (a->w8 && gemm_env_int("CHATTS_GEMM_STREAM_WAVES", 4) == 8)
      rc = stages == 3 ? launch_stream_t<3, true, 1, 8>(p, a, sk, s) : launch_stream_t<4, true, 1, 8>(p, a, sk, s);
    else if (a->w8) rc = stages == 3 ? launch_stream_t<3, true>(p, a, sk, s) : launch_stream_t<4, true>(p, a, sk, s);
    else if (stages == 3) rc = launch_stream_t<3, false>(p, a, sk, s);
    else if (stages == 5) rc = launch_stream_t<5, false>(p, a, sk, s);
    else rc = launch_stream_t<4, false>(p, a, sk, s);
    if (rc) return rc;
  } else if (ring) {
    rg.sk = sk;                      // (k_per_split was rounded to whole K-steps: the split count may have shrunk)
    rg.units = rg.T * rg.P * sk;
    int wpx = (rg.units + 7) / 8;
    if (wpx > device_cus() / 8) wpx = device_cus() / 8;
    rg.wpx = wpx < 1 ? 1 : wpx;
    const int rc = launch_ring(p, a->a_hi, a->a_lo, a->ld_planes, rg, dma_single_pass(), s);
    if (rc) return rc;
  } else if (dma) {
    const int rc = launch_dma(p, a, sk, s);
    if (rc) return rc;
  } else if (a->w8) {
    switch (bm) {
      case 128: hipLaunchKernelGGL((gemm_bf16x2_kernel<128, 2, 2, true>), grid, block, 0, s, p); break;
      case 64: hipLaunchKernelGGL((gemm_bf16x2_kernel<64, 2, 2, true>), grid, block, 0, s, p); break;
      case 32: hipLaunchKernelGGL((gemm_bf16x2_kernel<32, 1, 4, true>), grid, block, 0, s, p); break;
      default: hipLaunchKernelGGL((gemm_bf16x2_kernel<16, 1, 4, true>), grid, block, 0, s, p); break;
    }
  } else {
    switch (bm) {
      case 128: hipLaunchKernelGGL((gemm_bf16x2_kernel<128, 2, 2, false>), grid, block, 0, s, p); break;
      case 64: hipLaunchKernelGGL((gemm_bf16x2_kernel<64, 2, 2, false>), grid, block, 0, s, p); break;
      case 32: hipLaunchKernelGGL((gemm_bf16x2_kernel<32, 1, 4, false>), grid, block, 0, s, p); break;
      default: hipLaunchKernelGGL((gemm_bf16x2_kernel<16, 1, 4, false>), grid, block, 0, s, p); break;
    }
  }
  CHATTS_CHECK_LAUNCH("gemm_bf16x2");
  if (p.fix_cnt) return CHATTS_OK;        // the epilogue ran inside the launch
  const bool post_norm = a->post_norm_w != nullptr;
  if (sk > 1 && post_norm) {        // epilogue + the consumer's RMSNorm in one row-wise launch
    const int qsplit = gemm_env_int("CHATTS_EPI_NORM_Q", 8);
    if (a->m <= 16 && qsplit > 1 && qsplit <= 256 && p.sk_T == 0 && sk <= 8 && a->n <= 8192 && a->n % 4 == 0 &&
        (a->epilogue != CHATTS_EPI_RESID || a->c != a->resid)) {          // few rows: Q workgroups per row (c must not alias resid)
      hipLaunchKernelGGL((splitk_epilogue_norm_q_kernel<8>), dim3(a->m, qsplit), dim3(256), 0, s,
                         reinterpret_cast<const float*>(a->workspace), sk, a->m, a->n, a->bias, a->resid, a->c, a->ldc, a->epilogue,
                         a->w8 ? a->w8_scale : nullptr, a->post_norm_w, a->post_norm_eps, a->post_hi, a->post_lo, a->ld_post);
      CHATTS_CHECK_LAUNCH("splitk_epilogue_norm_q");
      return CHATTS_OK;
    }
    hipLaunchKernelGGL(splitk_epilogue_norm_kernel, dim3(a->m), dim3(256), 0, s, reinterpret_cast<const float*>(a->workspace), sk,
                       a->m, a->n, a->bias, a->resid, a->c, a->ldc, a->epilogue, a->w8 ? a->w8_scale : nullptr, a->post_norm_w,
                       a->post_norm_eps, a->post_hi, a->post_lo, a->ld_post, p.sk_T, p.sk_nk);
    CHATTS_CHECK_LAUNCH("splitk_epilogue_norm");
    return CHATTS_OK;
  }
  if (sk > 1 && rope && rope_done && !post_norm && a->epilogue == CHATTS_EPI_NONE && !a->c_hi && !a->w8 && p.sk_T == 0 &&
      a->n == (rope->n_q + 2 * rope->n_kv) * kHeadDim) {
    const int waves = a->m * (rope->n_q + 2 * rope->n_kv);
    hipLaunchKernelGGL(splitk_epilogue_rope_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, reinterpret_cast<const float*>(a->workspace),
                       sk, a->m, a->n, a->bias, a->c, a->ldc, *rope);
    CHATTS_CHECK_LAUNCH("splitk_epilogue_rope");
    *rope_done = true;
    return CHATTS_OK;
  }
  if (slabs && sk > 1 && sk <= 8 && a->epilogue == CHATTS_EPI_NONE && !post_norm && !a->c_hi && p.sk_T == 0) {
    // the consumer sums the slabs itself (SlabOut): no epilogue launch
    slabs->sk = sk; slabs->plane = (size_t)a->m * a->n; slabs->scale = a->w8 ? a->w8_scale : nullptr; slabs->bias = a->bias;
    return CHATTS_OK;
  }
  if (sk > 1 && sk <= 8 && a->epilogue != CHATTS_EPI_SWIGLU && p.sk_T == 0 && a->n % 4 == 0 && a->ldc % 4 == 0 &&
      ((uintptr_t)a->c % 16) == 0 && ((uintptr_t)a->resid % 16) == 0 && gemm_env_int("CHATTS_EPI_V4", 1) != 0) {
    const size_t total = (size_t)a->m * (a->n / 4);
    hipLaunchKernelGGL(splitk_epilogue_v4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float*>(a->workspace), sk, a->m, a->n, a->bias, a->resid, a->c, a->ldc, a->epilogue,
                       a->w8 ? a->w8_scale : nullptr, a->c_hi, a->c_lo, a->ld_cplanes);
    CHATTS_CHECK_LAUNCH("splitk_epilogue_v4");
  } else if (sk > 1) {
    const int ncols = a->epilogue == CHATTS_EPI_SWIGLU ? a->n / 2 : a->n;
    const size_t total = (size_t)a->m * ncols;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float*>(a->workspace), sk, a->m, a->n, a->bias, a->resid, a->c,
                       a->ldc, a->epilogue, a->w8 ? a->w8_scale : nullptr, a->c_hi, a->c_lo, a->ld_cplanes, p.sk_T, p.sk_nk);
    CHATTS_CHECK_LAUNCH("splitk_epilogue");
  }
  if (post_norm)                  // no split-K epilogue to fuse into: the contract still holds, as its own launch
    return chatts_rmsnorm_planes(a->c, a->post_norm_w, a->post_hi, a->post_lo, a->ld_post, a->m, a->n, a->post_norm_eps,
                                 reinterpret_cast<chatts_stream_t>(s));
  return CHATTS_OK;
}

}  // namespace chatts

#ifdef CHATTS_GEMM_PROBE
// diagnostic builds only (not in include/chatts_amd.h): copy the probe records to the host and clear them
extern "C" int chatts_debug_gemm_probe(void* dst, size_t bytes) {
  const size_t all = sizeof(unsigned long long) * chatts::kProbeRecs * 16;
  if (bytes > all) bytes = all;
  if (hipDeviceSynchronize() != hipSuccess) return CHATTS_E_LAUNCH;
  if (dst && hipMemcpyFromSymbol(dst, HIP_SYMBOL(chatts::g_gemm_probe), bytes) != hipSuccess) return CHATTS_E_LAUNCH;
  void* sym = nullptr;
  if (hipGetSymbolAddress(&sym, HIP_SYMBOL(chatts::g_gemm_probe)) != hipSuccess || hipMemset(sym, 0, all) != hipSuccess) return CHATTS_E_LAUNCH;
  return CHATTS_OK;
}
#endif
