// gemm.hip - MFMA projections for M > 1:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
// A float32 activations (or their bf16 hi / lo planes), W bfloat16 weights (HF nn.Linear layout, K contiguous for both).
//
// Precision scheme "bf16x2": the reference path we must match is float32 (1e-3 relative on logits,
// identical greedy tokens).  Weights are bf16 by definition, so only A needs care: a = hi + lo,
// hi = bf16(a), lo = bf16(a - hi), and the wave issues two v_mfma_f32_16x16x32_bf16 per fragment pair
// (lo.W, hi.W) into ONE f32 accumulator.  Products are exact (8b x 8b mantissas), accumulation is f32.
//
// Three kernels compute the same sums (per output element: K in ascending 32-deep blocks, lo pass before hi pass) and share the
// epilogue arithmetic (bias / GELU / residual / SwiGLU / fp8 row scale / plane output / split-K partials), and therefore produce
// bit-identical results at equal split-K:
//   gemm_bf16x2_kernel  register-staged, BM x 128 x 32 tiles, A split on the VALU while staging; any M, float32 A;
//                       TS-encoder MLP, prefill chunks below 96 rows, generic callers
//   gemm_ring_kernel    (gemm_ring.hip) prefill (M >= 96) on pre-split planes: persistent workgroups, balanced M-tiles x 256 columns,
//                       4 loader waves feeding a ring of four 32-deep half-stages by LDS-DMA, 8 compute waves
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

namespace chatts {

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

// 128-byte LDS rows (gemm_stream_kernel): 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7), applied on the SOURCE side of the
// LDS-DMA (lane l of an 8-row piece fetches global chunk (l & 7) ^ swizzle(row)); conflict-free for the ds_read_b128 fragment reads.
__device__ __forceinline__ int lds_off128(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

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
                                                             uint16_t* __restrict__ c_hi, uint16_t* __restrict__ c_lo, int ldcp) {
  const int ncols = epilogue == CHATTS_EPI_SWIGLU ? n / 2 : n;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)m * ncols) return;
  const int row = (int)(idx / ncols), col = (int)(idx % ncols);
  const size_t plane = (size_t)m * n;
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
  f32x4 r4 = {0.f, 0.f, 0.f, 0.f}, s4 = r4, b4 = r4;            // residual, scale, bias: requested with the slabs (one round trip)
  if (epilogue == CHATTS_EPI_RESID) r4 = *reinterpret_cast<const f32x4*>(resid + (size_t)row * ldc + col);
  if (scale) s4 = *reinterpret_cast<const f32x4*>(scale + col);
  if (bias) b4 = *reinterpret_cast<const f32x4*>(bias + col);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s)
    if (s < sk) { v[0] += t8[s].x; v[1] += t8[s].y; v[2] += t8[s].z; v[3] += t8[s].w; }
  const float rr[4] = {r4.x, r4.y, r4.z, r4.w}, sc[4] = {s4.x, s4.y, s4.z, s4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (scale) v[j] *= sc[j];
    if (bias) v[j] += bb[j];
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
                                                                  int ldp) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const size_t plane = (size_t)m * n;
  float ss = 0.f;
  for (int col = threadIdx.x * 4; col < n; col += 1024) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
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

// The same kernel with every memory round trip of a thread taken ONCE (round 5): sk <= 4 slabs, rows of <= 1024 kIt columns.  The kernel
// above walks a row in kIt dependent iterations - slab loads, wait, store - and then re-reads its own stores for the planes: ~2 kIt round
// trips per workgroup, 16.5 us per 798 x 5120 epilogue (two per layer in a prefill chunk).  Here a thread requests all slabs, the residual,
// bias and norm weights of all its kIt column groups up front, keeps the row values in registers, and writes c and the planes from them.
// Same sums in the same order (split order per element, the row's sum of squares in column-group order per thread): bit-identical.
template <int kIt>
__global__ __launch_bounds__(256) void splitk_epilogue_norm_reg_kernel(const float* __restrict__ ws, int sk, int m, int n,
                                                                      const float* __restrict__ bias, const float* __restrict__ resid,
                                                                      float* __restrict__ c, int ldc, int epilogue,
                                                                      const float* __restrict__ scale, const float* __restrict__ norm_w,
                                                                      float eps, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int ldp) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const size_t plane = (size_t)m * n;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 t[kIt][4], rs[kIt], gw[kIt], sc4[kIt], b4[kIt];
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int col = threadIdx.x * 4 + it * 1024;
    const bool live = col < n;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      t[it][s] = live ? *reinterpret_cast<const f32x4*>(ws + (s < sk ? s : sk - 1) * plane + (size_t)row * n + col) : zero;
    rs[it] = (live && epilogue == CHATTS_EPI_RESID) ? *reinterpret_cast<const f32x4*>(resid + (size_t)row * ldc + col) : zero;
    gw[it] = live ? *reinterpret_cast<const f32x4*>(norm_w + col) : zero;
    sc4[it] = (live && scale) ? *reinterpret_cast<const f32x4*>(scale + col) : zero;
    b4[it] = (live && bias) ? *reinterpret_cast<const f32x4*>(bias + col) : zero;
  }
  f32x4 keep[kIt];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int col = threadIdx.x * 4 + it * 1024;
    f32x4 v = zero;
    if (col < n) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s < sk) { v.x += t[it][s].x; v.y += t[it][s].y; v.z += t[it][s].z; v.w += t[it][s].w; }
      if (scale) { v.x *= sc4[it].x; v.y *= sc4[it].y; v.z *= sc4[it].z; v.w *= sc4[it].w; }
      if (bias) { v.x += b4[it].x; v.y += b4[it].y; v.z += b4[it].z; v.w += b4[it].w; }
      if (epilogue == CHATTS_EPI_RESID) { v.x = rs[it].x + v.x; v.y = rs[it].y + v.y; v.z = rs[it].z + v.z; v.w = rs[it].w + v.w; }
      *reinterpret_cast<f32x4*>(c + (size_t)row * ldc + col) = v;
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;        // same accumulation pattern as rmsnorm_kernel
    }
    keep[it] = v;
  }
  ss = block_sum<4>(ss, red);
  const float rstd = rsqrtf(ss / (float)n + eps);
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int col = threadIdx.x * 4 + it * 1024;
    if (col >= n) continue;
    const f32x4 v = keep[it], g = gw[it];
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
template <int kMaxIt, int kG>
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
  // Column groups are taken kG at a time (kG = 3 since round 5, option EPI_NORM_Q_GROUPS; 1 = the round-3 walk): all slabs (sk <= 8;
  // the clamp re-reads the last slab and the value is dropped), the residual and the scale / bias of kG groups are requested before
  // the first is summed - two dependent round trips for a 5120-wide row instead of five; with 16 workgroups' worth of rows there is no
  // other wave to hide a chain of trips behind.  Clamped addresses instead of branches inside a batch: a group past the row end
  // loads column 0 and is dropped; a batch that starts past the row end is skipped whole.
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g0 = 0; g0 < kMaxIt; g0 += kG) {
    if (g0 * 1024 >= n) {
#pragma unroll
      for (int u = 0; u < kG; ++u)
        if (g0 + u < kMaxIt) keep[g0 + u] = zero;
      continue;
    }
    f32x4 t8[kG][8], tr[kG], tsc[kG], tb[kG];
#pragma unroll
    for (int u = 0; u < kG; ++u) {
      const int it = g0 + u;
      const int col = threadIdx.x * 4 + it * 1024;
      const int cc = (it < kMaxIt && col < n) ? col : 0;
#pragma unroll
      for (int s = 0; s < 8; ++s)
        t8[u][s] = *reinterpret_cast<const f32x4*>(ws + (s < sk ? s : sk - 1) * plane + (size_t)row * n + cc);
      tr[u] = epilogue == CHATTS_EPI_RESID ? *reinterpret_cast<const f32x4*>(resid + (size_t)row * ldc + cc) : zero;
      tsc[u] = scale ? *reinterpret_cast<const f32x4*>(scale + cc) : zero;
      tb[u] = bias ? *reinterpret_cast<const f32x4*>(bias + cc) : zero;
    }
#pragma unroll
    for (int u = 0; u < kG; ++u) {
      const int it = g0 + u;
      if (it >= kMaxIt) continue;
      const int col = threadIdx.x * 4 + it * 1024;
      f32x4 v = zero;
      if (col < n) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
          if (s < sk) { v.x += t8[u][s].x; v.y += t8[u][s].y; v.z += t8[u][s].z; v.w += t8[u][s].w; }
        if (scale) { v.x *= tsc[u].x; v.y *= tsc[u].y; v.z *= tsc[u].z; v.w *= tsc[u].w; }
        if (bias) { v.x += tb[u].x; v.y += tb[u].y; v.z += tb[u].z; v.w += tb[u].w; }
        if (epilogue == CHATTS_EPI_RESID) { v.x = tr[u].x + v.x; v.y = tr[u].y + v.y; v.z = tr[u].z + v.z; v.w = tr[u].w + v.w; }
        if (mine) *reinterpret_cast<f32x4*>(c + (size_t)row * ldc + col) = v;
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;        // same accumulation pattern as rmsnorm_kernel
      }
      keep[it] = v;
    }
  }
  f32x4 gw[kMaxIt];                                  // the norm weights travel while the workgroup reduces
#pragma unroll
  for (int it = 0; it < kMaxIt; ++it) {
    const int col = threadIdx.x * 4 + it * 1024;
    gw[it] = (kG > 1 && mine && col < n) ? *reinterpret_cast<const f32x4*>(norm_w + col) : zero;
  }
  ss = block_sum<4>(ss, red);
  const float rstd = rsqrtf(ss / (float)n + eps);
  if (!mine) return;
#pragma unroll
  for (int it = 0; it < kMaxIt; ++it) {
    const int col = threadIdx.x * 4 + it * 1024;
    if (col >= n) continue;
    const f32x4 v = keep[it];
    const f32x4 g = kG > 1 ? gw[it] : *reinterpret_cast<const f32x4*>(norm_w + col);
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

static void pick_geometry(int m, int n, int k, int& bm, int& sk, int slots_per_cu = 3, int bn = 128) {
  bm = m > 64 ? 128 : (m > 32 ? 64 : (m > 16 ? 32 : 16));
  const int force_bm = opt_get(OPT_GEMM_BM, 0);          // tuning / tests only
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
  const int force_sk = opt_get(OPT_GEMM_SK, 0);
  if (force_sk > 0 && force_sk <= 16 && k / force_sk >= 32) sk = force_sk;
}

static int k_per_split(int k, int sk, int bk = 32) {
  int kps = (k + sk - 1) / sk;
  return ((kps + bk - 1) / bk) * bk;
}

// The prefill kernel (gemm_ring.hip) runs when the caller supplies the pre-split planes and either M >= kPlanesMinM or there is no
// float32 A to fall back on.  Its lanes hold 4 consecutive output columns: the epilogue operands must allow 16-byte accesses.
constexpr int kPlanesMinM = 96;
static bool use_ring(const ChattsLinearArgs* a) {
  if (!a->a_hi || !a->a_lo || a->k % 64 != 0) return false;
  if (!a->a) return true;
  return a->m >= opt_get(OPT_GEMM_PLANES_MIN_M, kPlanesMinM);
}
static int check_ring_operands(const ChattsLinearArgs* a) {
  CHATTS_REQUIRE(a->ldc % 4 == 0 && ((uintptr_t)a->bias % 16) == 0 && ((uintptr_t)a->resid % 16) == 0 && ((uintptr_t)a->c % 16) == 0,
                 CHATTS_E_SHAPE, "linear: the plane path needs ldc %% 4 == 0 (ldc=%d) and 16-byte aligned bias / resid / c", a->ldc);
  CHATTS_REQUIRE(!a->c_hi || (a->ld_cplanes % 4 == 0 && ((uintptr_t)a->c_hi % 8) == 0 && ((uintptr_t)a->c_lo % 8) == 0), CHATTS_E_SHAPE,
                 "linear: plane output needs ld_cplanes %% 4 == 0 (ld_cplanes=%d) and 8-byte aligned planes", a->ld_cplanes);
  return CHATTS_OK;
}
static void pick_ring(int m, int n, int k, RingGeom& g) {
  ring_pick(m, n, k, device_cus(), opt_get(OPT_GEMM_T, 0), opt_get(OPT_GEMM_SK, 0), g);
}

// Streaming kernel (2 <= M <= 16 with planes): split-K so that ~2 workgroups per CU are busy, >= 4 K-steps per split.
// ... and the multi-block form (17 <= M <= 128, bf16 W) when a 256-column tiling would leave most CUs without a tile
// (fewer N-panels than half the CUs: the TS-encoder MLP, o / down / qkv of a short prefill chunk)
// Measured (round 2, one TS-MLP layer 5120 x 5120, profiles/r2_ts_gemm_sweep.txt + tools/jobs/r2_job15.sh): P = 32: 18.7 us (LDS-DMA
// kernel 23.7), P = 64: 23.5 (26.0), P = 128: 35.7 (33.9) - with 8 row blocks the four waves issue 12 DMA pieces each per K-step
// (~150 cycles apiece) in the same instruction stream as their 64 MFMAs, which is what the prefill kernel's loader waves exist
// to avoid; and a short K (layer 0: 5 K-steps) is served better by a 2-way split of 256-column tiles.  Hence M <= 64, K >= 1024.
static bool stream_multiblock(int m, int n, int k, bool w8) {
  // round 3: with 8 waves as 2 x 4 the streaming form also wins at 65 .. 128 rows on the 5120-column shapes (P = 128, one TS-MLP layer:
  // 30.1 us against 33.2 for the LDS-DMA kernel and 35.0 for the 4-wave form, epilogue included - profiles/r3_ts_gemm_sweep.txt)
  const int max_m = opt_get(OPT_GEMM_STREAM_MB, n <= 8192 && opt_get(OPT_GEMM_STREAM_MB_WAVES, 8) == 8 ? 128 : 64);
  return m > 16 && m <= max_m && m <= 128 && !w8 && k % 64 == 0 && k >= 1024 &&
         2 * 8 * (((n + kRingPanel - 1) / kRingPanel + 7) / 8) <= device_cus();
}
static bool use_stream(const ChattsLinearArgs* a) {
  if (!(a->a_hi && a->a_lo) || opt_get(OPT_GEMM_STREAM, 1) == 0) return false;
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
  const int force_sk = opt_get(OPT_GEMM_SK, 0);
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

size_t gemm_workspace(int m, int n, int k) {
  int bm, sk, sk2;
  pick_geometry(m, n, k, bm, sk);
  if (m >= 2 && k % 64 == 0) {                   // the prefill kernel's own split choice (any M it may be handed planes for)
    RingGeom g;
    pick_ring(m, n, k, g);
    if (g.sk > sk) sk = g.sk;
  }
  if ((m <= 16 && k % 64 == 0) || stream_multiblock(m, n, k, false)) {
    sk2 = pick_stream_sk(n, k, false);         // (the fp8 variant's K-steps are twice as long: never more splits)
    if (sk2 > sk) sk = sk2;
  }
  return sk > 1 ? (size_t)sk * m * n * sizeof(float) : 0;
}

int launch_gemm(const ChattsLinearArgs* a_in, hipStream_t s, const RopeFuse* rope, bool* rope_done) {
  int bm, sk;
  ChattsLinearArgs a_copy;
  const ChattsLinearArgs* a = a_in;
  if (a_in->w8 && a_in->w8_format == CHATTS_W8_INT8 && !use_stream(a_in)) {      // int8 copies: the weight-streaming kernel only
    a_copy = *a_in;
    a_copy.w8 = nullptr; a_copy.w8_scale = nullptr;
    a = &a_copy;
  }
  const bool stream = use_stream(a);
  const bool ring = !stream && use_ring(a);
  RingGeom rg{};
  CHATTS_REQUIRE(!a->planes_tiled || ring, CHATTS_E_SHAPE, "linear: tiled planes are the prefill kernel's operand format (M=%d K=%d take another kernel)", a->m, a->k);
  CHATTS_REQUIRE(!ring || !a->w_tiled || (((uintptr_t)a->w_tiled % 16) == 0 && a->ldw == a->k), CHATTS_E_SHAPE, "linear: w_tiled needs 16-byte alignment and ldw == K");
  if (stream) {
    sk = pick_stream_sk(a->n, a->k, a->w8 != nullptr);
    bm = 16;
  } else if (ring) {
    if (const int rc = check_ring_operands(a)) return rc;
    if (a->w8) {                                 // prefill keeps multiplying the bf16 copy (MFMA-bound: nothing to gain from fewer bytes)
      a_copy = *a;
      a_copy.w8 = nullptr; a_copy.w8_scale = nullptr;
      a = &a_copy;
    }
    pick_ring(a->m, a->n, a->k, rg);
    sk = rg.sk;
    bm = 128;
  } else {
    CHATTS_REQUIRE(a->a, CHATTS_E_SHAPE, "linear: a == NULL needs K %% 64 == 0 (K=%d) for the plane path", a->k);
    pick_geometry(a->m, a->n, a->k, bm, sk);
  }
  const int kps = k_per_split(a->k, sk, stream ? stream_bk(a->w8 != nullptr) : (ring ? 64 : 32));
  sk = (a->k + kps - 1) / kps;
  GemmParams p;
  p.a = a->a; p.w = a->w; p.bias = a->bias; p.resid = a->resid;
  p.m = a->m; p.n = a->n; p.k = a->k; p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc;
  p.epilogue = a->epilogue; p.k_per_split = kps; p.direct = sk == 1;
  p.w8 = a->w8; p.w8_scale = a->w8_scale; p.ldw8 = a->ldw8; p.w8_format = a->w8_format;
  p.c_hi = a->c_hi; p.c_lo = a->c_lo; p.ldcp = a->ld_cplanes;
  p.wt = ring ? a->w_tiled : nullptr; p.planes_tiled = a->planes_tiled;
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
    const int stages = opt_get(OPT_GEMM_STREAM_STAGES, 4);
    const bool w8x = opt_get(OPT_GEMM_STREAM_MB_WAVES, 8) == 8;   // 8 waves as 2 x 4 for the multi-block forms (bit-identical to 4)
    if (a->m > 64) rc = w8x ? launch_stream_t<3, false, 8, 8>(p, a, sk, s) : launch_stream_t<3, false, 8>(p, a, sk, s);   // 3 x 48 KB stages: one workgroup per CU
    else if (a->m > 32) rc = w8x ? launch_stream_t<4, false, 4, 8>(p, a, sk, s) : launch_stream_t<4, false, 4>(p, a, sk, s);     // 4 x 32 KB
    else if (a->m > 16) rc = w8x ? launch_stream_t<4, false, 2, 8>(p, a, sk, s) : launch_stream_t<4, false, 2>(p, a, sk, s);     // 4 x 24 KB
    else if (a->w8 && a->w8_format == CHATTS_W8_INT8)
      rc = stages == 3 ? launch_stream_t<3, true, 1, 4, true>(p, a, sk, s) : launch_stream_t<4, true, 1, 4, true>(p, a, sk, s);
    // fp8 W: eight waves per workgroup since round 6 (kernel trace of config 5 per geometry, profiles/r6_cfg5_stream_geometry.txt: gate_up 30.0 -> 29.2,
    // o / down 11.8 -> 11.1, qkv 10.3 -> 9.7, lm_head 153 -> 141 us; bit-identical to four waves)
    else if (a->w8 && opt_get(OPT_GEMM_STREAM_WAVES, 8) == 8)
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
    // GEMM_PRECISION = 1: the "bf16" SPEED mode (one pass over the hi plane; NOT parity grade, never the headline path)
    const int rc = launch_ring(p, a->a_hi, a->a_lo, a->ld_planes, rg, opt_get(OPT_GEMM_PRECISION, 0) == 1, s);
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
  const bool post_norm = a->post_norm_w != nullptr;
  if (sk > 1 && post_norm) {        // epilogue + the consumer's RMSNorm in one row-wise launch
    const int qsplit = opt_get(OPT_EPI_NORM_Q, 8);
    if (a->m <= 16 && qsplit > 1 && qsplit <= 256 && sk <= 8 && a->n <= 8192 && a->n % 4 == 0 &&
        (a->epilogue != CHATTS_EPI_RESID || a->c != a->resid)) {          // few rows: Q workgroups per row (c must not alias resid)
      if (opt_get(OPT_EPI_NORM_Q_GROUPS, 3) == 1)
        hipLaunchKernelGGL((splitk_epilogue_norm_q_kernel<8, 1>), dim3(a->m, qsplit), dim3(256), 0, s,
                           reinterpret_cast<const float*>(a->workspace), sk, a->m, a->n, a->bias, a->resid, a->c, a->ldc, a->epilogue,
                           a->w8 ? a->w8_scale : nullptr, a->post_norm_w, a->post_norm_eps, a->post_hi, a->post_lo, a->ld_post);
      else
        hipLaunchKernelGGL((splitk_epilogue_norm_q_kernel<8, 3>), dim3(a->m, qsplit), dim3(256), 0, s,
                           reinterpret_cast<const float*>(a->workspace), sk, a->m, a->n, a->bias, a->resid, a->c, a->ldc, a->epilogue,
                           a->w8 ? a->w8_scale : nullptr, a->post_norm_w, a->post_norm_eps, a->post_hi, a->post_lo, a->ld_post);
      CHATTS_CHECK_LAUNCH("splitk_epilogue_norm_q");
      return CHATTS_OK;
    }
    const float* ws_f = reinterpret_cast<const float*>(a->workspace);
    const float* sc = a->w8 ? a->w8_scale : nullptr;
    const int its = (a->n + 1023) / 1024;
    if (sk <= 4 && its <= 8 && opt_get(OPT_EPI_NORM_REG, 1) != 0) {      // all round trips of a thread at once (bit-identical; 13.5)
#define CHATTS_EPI_REG(K) hipLaunchKernelGGL((splitk_epilogue_norm_reg_kernel<K>), dim3(a->m), dim3(256), 0, s, ws_f, sk, a->m, a->n, a->bias, \
                                             a->resid, a->c, a->ldc, a->epilogue, sc, a->post_norm_w, a->post_norm_eps, a->post_hi, a->post_lo, a->ld_post)
      switch (its) {
        case 1: CHATTS_EPI_REG(1); break;
        case 2: CHATTS_EPI_REG(2); break;
        case 3: CHATTS_EPI_REG(3); break;
        case 4: CHATTS_EPI_REG(4); break;
        case 5: CHATTS_EPI_REG(5); break;
        case 6: CHATTS_EPI_REG(6); break;
        default: CHATTS_EPI_REG(8); break;
      }
#undef CHATTS_EPI_REG
      CHATTS_CHECK_LAUNCH("splitk_epilogue_norm_reg");
      return CHATTS_OK;
    }
    hipLaunchKernelGGL(splitk_epilogue_norm_kernel, dim3(a->m), dim3(256), 0, s, ws_f, sk,
                       a->m, a->n, a->bias, a->resid, a->c, a->ldc, a->epilogue, sc, a->post_norm_w,
                       a->post_norm_eps, a->post_hi, a->post_lo, a->ld_post);
    CHATTS_CHECK_LAUNCH("splitk_epilogue_norm");
    return CHATTS_OK;
  }
  if (sk > 1 && rope && rope_done && !post_norm && a->epilogue == CHATTS_EPI_NONE && !a->c_hi && !a->w8 &&
      a->n == (rope->n_q + 2 * rope->n_kv) * kHeadDim) {
    const int waves = a->m * (rope->n_q + 2 * rope->n_kv);
    hipLaunchKernelGGL(splitk_epilogue_rope_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, reinterpret_cast<const float*>(a->workspace),
                       sk, a->m, a->n, a->bias, a->c, a->ldc, *rope);
    CHATTS_CHECK_LAUNCH("splitk_epilogue_rope");
    *rope_done = true;
    return CHATTS_OK;
  }
  if (sk > 1 && sk <= 8 && a->epilogue != CHATTS_EPI_SWIGLU && a->n % 4 == 0 && a->ldc % 4 == 0 &&
      ((uintptr_t)a->c % 16) == 0 && ((uintptr_t)a->resid % 16) == 0 && ((uintptr_t)a->bias % 16) == 0 &&
      (!a->w8 || ((uintptr_t)a->w8_scale % 16) == 0) && opt_get(OPT_EPI_V4, 1) != 0) {
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
                       a->ldc, a->epilogue, a->w8 ? a->w8_scale : nullptr, a->c_hi, a->c_lo, a->ld_cplanes);
    CHATTS_CHECK_LAUNCH("splitk_epilogue");
  }
  if (post_norm)                  // no split-K epilogue to fuse into: the contract still holds, as its own launch
    return chatts_rmsnorm_planes(a->c, a->post_norm_w, a->post_hi, a->post_lo, a->ld_post, a->m, a->n, a->post_norm_eps,
                                 reinterpret_cast<chatts_stream_t>(s));
  return CHATTS_OK;
}

}  // namespace chatts
