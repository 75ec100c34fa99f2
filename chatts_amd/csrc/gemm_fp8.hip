// gemm_fp8.hip - SPEED MODE (not parity grade): fp8 x fp8 -> f32 projections on the CDNA4 block-scaled matrix pipe.
//
// BASELINE.json config 5 names "fp8 weights (CDNA4 fp8 MFMA)"; the reference hands quant_config straight to vLLM
// (NetManAIOps/ChatTS chatts/vllm/chatts_vllm.py:475,481), whose fp8 path quantises the activations per token and multiplies fp8 x fp8.
// The parity-grade path of this library cannot do that - the bar is the float32 reference within 1e-3, so activations stay float32
// (two bf16 MFMA passes) and fp8 weights are widened exactly - but the MFMA-bound stages (prefill GEMMs, the TS encoder at thousands
// of patches) pay 4x the matrix work of an fp8 x fp8 product for it.  This file is that product as a LABELLED option
// (precision="fp8"): results are ~1e-2 from the default mode, measured and recorded (profiles/r4_fp8_speed_mode.json), never quoted
// as the parity line.
//   * quantize_rows_fp8_kernel: one workgroup per activation row - optional RMSNorm (Qwen2RMSNorm), amax, scale = amax / 448,
//     OCP e4m3fn codes (v_cvt_pk_fp8_f32, round to nearest even, saturating) + the row's float32 scale: vLLM's dynamic per-token
//     activation quantisation;
//   * gemm_fp8_kernel: C[m, n] = epilogue(sa[m] * sw[n] * sum_k A8[m, k] W8[n, k]) with v_mfma_scale_f32_16x16x128_f8f6f4 (both
//     operand formats e4m3, block scales fixed to 1.0: the instruction is the only fp8 form that runs at twice the bf16 rate) - 256 x
//     256 x 128 or 128 x 256 x 128 tiles, 8 waves, operands staged global -> registers -> LDS (XOR-swizzled 128-byte rows: conflict-free
//     fragment reads), the global loads of the K-step after next in flight under the MFMAs.
//     Weights: the per-row power-of-two-scaled e4m3 copies the fp8 decode path already streams (ChattsLinearArgs.w8).
#include <type_traits>

#include "common.h"

namespace chatts {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// ---- activation rows -> fp8 ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const float* __restrict__ x, int k, int ldx, const float* __restrict__ norm_w,
                                                               float eps, uint8_t* __restrict__ q, int ldq, float* __restrict__ scale) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (size_t)row * ldx;
  float rstd = 1.f;
  if (norm_w) {
    float ss = 0.f;
    for (int i = tid * 4; i < k; i += 1024) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = block_sum<4>(ss, red);
    rstd = rsqrtf(ss / (float)k + eps);
  }
  float amax = 0.f;
  for (int i = tid * 4; i < k; i += 1024) {
    f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
    if (norm_w) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(norm_w + i);
      v.x = g.x * (v.x * rstd); v.y = g.y * (v.y * rstd); v.z = g.z * (v.z * rstd); v.w = g.w * (v.w * rstd);
    }
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  amax = wave_max(amax);
  if (lane == 0) red[wave] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float s = amax > 0.f ? amax / 448.f : 1.f;        // e4m3fn: largest finite value 448
  const float inv = 1.f / s;
  if (tid == 0) scale[row] = s;
  uint8_t* qr = q + (size_t)row * ldq;
  for (int i = tid * 4; i < k; i += 1024) {
    f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
    if (norm_w) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(norm_w + i);
      v.x = g.x * (v.x * rstd); v.y = g.y * (v.y * rstd); v.z = g.z * (v.z * rstd); v.w = g.w * (v.w * rstd);
    }
    int packed = 0;
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(v.x * inv, v.y * inv, packed, false);
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(v.z * inv, v.w * inv, packed, true);
    *reinterpret_cast<int*>(qr + i) = packed;
  }
}

// ---- the GEMM --------------------------------------------------------------------------------------------------------------
struct GemmFp8Params {
  const uint8_t* a8;       // [M, lda] e4m3
  const float* a_scale;    // [M]
  const uint8_t* w8;       // [N, ldw] e4m3
  const float* w_scale;    // [N]
  const float* bias;       // [N] or null (SWIGLU: interleaved like the rows)
  const float* resid;      // [M, ldc] (EPI_RESID)
  float* c;                // [M, ldc]  (SWIGLU: [M, N / 2])
  int m, n, k, lda, ldw, ldc;
  int order;               // 0: N-tile fastest (consecutive workgroups share the A tile); 1: 8 contiguous (panel major, M-tile minor) ranges, one per XCD
};

constexpr int kF8BN = 256, kF8BK = 128;                     // BK in bytes = fp8 values per MFMA step
constexpr int kF8Row = kF8BK;                               // LDS rows of 128 bytes; 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7)
// (the bf16 LDS-DMA kernel's swizzle, gemm.hip).  A lane group g = lane / 16 of a fragment takes chunks g and g + 4 of its row - the
// instruction pairs equal byte positions of equal lane groups in A and B, so WHICH 32 of the 128 K bytes a group holds is free as long
// as both operands agree - and with that choice every 16-lane service group of both ds_read_b128 touches 16 distinct 16-byte bank
// groups.  (First version: contiguous 32 bytes per lane, rows padded to 144 bytes: SQ_LDS_BANK_CONFLICT = 36 % of the LDS cycles.)
__device__ __forceinline__ int f8_lds_off(int r, int c) { return r * kF8BK + ((c ^ ((r >> 1) & 7)) << 4); }
constexpr int kF8Threads = 512;

__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu8_f(float x) { return x / (1.0f + expf(-x)); }

// Tile (64 WM) x 256 x 128, eight waves as WM x (8 / WM), every wave 64 rows x (256 / (8 / WM)) columns: WM = 4 -> 256 x 256 with
// 64 x 128 wave tiles (128 accumulator registers), WM = 2 -> 128 x 256 with 64 x 64 wave tiles.  An fp8 tile-step is a quarter of the
// bf16x2 step's matrix time for the same operand bytes, so the tile has to be this large for the CU's fetch rate (~64 GB/s measured
// with the first 128 x 128 version: 1.95 us per K-step against 0.21 us of MFMA) to keep the pipe fed: 256 flop per fetched byte at
// WM = 4, 171 at WM = 2; the launcher picks per shape by rounds of workgroups.  Operands: global -> registers (TWO K-steps ahead) ->
// LDS (two stages) -> fragments.
template <int EPI, int WM>
__global__ __launch_bounds__(kF8Threads) void gemm_fp8_kernel(GemmFp8Params p) {
  constexpr int WN = 8 / WM, BM = 64 * WM, FN = kF8BN / WN / 16, NA = BM / 64;       // FN column fragments per wave; NA A pieces per thread
  constexpr int A_TILE = BM * kF8Row, STAGE = (BM + kF8BN) * kF8Row;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 stages][A tile | W tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // Tile order.  Workgroup b runs on XCD b % 8 and every XCD has its own L2, so the (W panel major, M-tile minor) sequence is cut into
  // 8 contiguous ranges, one per XCD: the M-tiles of a panel run side by side on ONE XCD and its L2 fetches the panel once (with the
  // N-major order of the first version every M-tile re-streamed all of W: 8 x 141 MB for gate_up at M = 1024 - an HBM-bound GEMM).
  const int mt_count = (p.m + BM - 1) / BM, nt_count = (p.n + kF8BN - 1) / kF8BN, total = mt_count * nt_count;
  const int per_xcd = (total + 7) / 8;
  int idx = (int)blockIdx.x, n0, m0;
  if (p.order) {
    idx = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || idx >= total) return;
    n0 = (idx / mt_count) * kF8BN; m0 = (idx % mt_count) * BM;
  } else {
    if (idx >= total) return;
    n0 = (idx % nt_count) * kF8BN; m0 = (idx / nt_count) * BM;
  }
  // staging: thread t moves 16-byte piece (row t / 8 + 64 j, column block t % 8) of A (j < NA) and of W (j < 4)
  const int srow = tid >> 3, scol = (tid & 7) * 16;
  const uint8_t* ag[NA];
  const uint8_t* wg[4];
  bool aok[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int ar = m0 + srow + 64 * j;
    aok[j] = ar < p.m;
    ag[j] = p.a8 + (size_t)(aok[j] ? ar : 0) * p.lda + scol;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int wr = n0 + srow + 64 * j;
    wg[j] = p.w8 + (size_t)(wr < p.n ? wr : 0) * p.ldw + scol;
  }
  constexpr int DEPTH = WM == 4 ? 1 : 2;        // K-steps of global loads in flight (the 256-row tile has no registers for a second set)
  i32x4 ra[DEPTH][NA], rw[DEPTH][4];
  auto gload = [&](auto setc, int k0) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[S][j] = aok[j] ? *reinterpret_cast<const i32x4*>(ag[j] + k0) : (i32x4){0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) rw[S][j] = *reinterpret_cast<const i32x4*>(wg[j] + k0);
  };
  auto lstore = [&](auto setc, int stage) {
    constexpr int S = decltype(setc)::value;
    char* at = smem + (size_t)stage * STAGE;
    char* wt = at + A_TILE;
#pragma unroll
    for (int j = 0; j < NA; ++j) *reinterpret_cast<i32x4*>(at + f8_lds_off(srow + 64 * j, tid & 7)) = ra[S][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<i32x4*>(wt + f8_lds_off(srow + 64 * j, tid & 7)) = rw[S][j];
  };
  f32x4 acc[4][FN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment addresses: lane l holds row (l % 16) of a 16-row block and the 16-byte chunks g, g + 4 of its K bytes (g = l / 16); row
  // blocks start at multiples of 16, so the swizzle term (r >> 1) & 7 of a lane is the same for every block
  const int frow = lane & 15, fg = lane >> 4, fsw = (frow >> 1) & 7;
  const int fo0 = frow * kF8BK + ((fg ^ fsw) << 4), fo1 = frow * kF8BK + (((fg + 4) ^ fsw) << 4);
  auto compute = [&](int stage) {
    const char* at = smem + (size_t)stage * STAGE + (size_t)(wm * 64) * kF8Row;
    const char* wt = smem + (size_t)stage * STAGE + A_TILE + (size_t)(wn * (kF8BN / WN)) * kF8Row;
    i32x8 af[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const i32x4 lo = *reinterpret_cast<const i32x4*>(at + i * 16 * kF8Row + fo0), hi = *reinterpret_cast<const i32x4*>(at + i * 16 * kF8Row + fo1);
      af[i] = (i32x8){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const i32x4 lo = *reinterpret_cast<const i32x4*>(wt + j * 16 * kF8Row + fo0), hi = *reinterpret_cast<const i32x4*>(wt + j * 16 * kF8Row + fo1);
      const i32x8 bf = (i32x8){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i], bf, acc[i][j], 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 0x7f7f7f7f, 0,
                                                                     0x7f7f7f7f /* E8M0 127 = 2^0 */);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  const int nk = p.k / kF8BK;
  gload(S0{}, 0);
  if constexpr (DEPTH == 1) {
    lstore(S0{}, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) gload(S0{}, (kt + 1) * kF8BK);
      compute(kt & 1);
      if (kt + 1 < nk) lstore(S0{}, (kt & 1) ^ 1);
      __syncthreads();          // (one register set: the store has to wait for this step's loads, it cannot move ahead of the MFMAs)
    }
  } else {
  if (nk > 1) gload(S1{}, kF8BK);
  lstore(S0{}, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    // even step: stage 0 holds kt; register set 0 is free -> kt + 2; set 1 holds kt + 1 -> stage 1
    // (moving the LDS stores of the next stage AHEAD of this step's MFMAs was measured: 367 -> 407 us for the 8192 x 5120 x 5120
    //  layer - the stores then wait for global loads that are only one step old; behind the MFMAs the loads have had two steps)
    if (kt + 2 < nk) gload(S0{}, (kt + 2) * kF8BK);
    compute(0);
    if (kt + 1 < nk) lstore(S1{}, 1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    // odd step: stage 1 holds kt + 1; set 1 is free -> kt + 3; set 0 holds kt + 2 -> stage 0
    if (kt + 3 < nk) gload(S1{}, (kt + 3) * kF8BK);
    compute(1);
    if (kt + 2 < nk) lstore(S0{}, 0);
    __syncthreads();
  }
  }

  // epilogue: acc[i][j][r] = C[m0 + wm 64 + i 16 + (lane / 16) 4 + r][n0 + wn (256 / WN) + j 16 + lane % 16]
  const int crow0 = m0 + wm * 64 + (lane >> 4) * 4, ccol0 = n0 + wn * (kF8BN / WN) + (lane & 15);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = crow0 + i * 16 + r;
      if (row >= p.m) continue;
      const float sa = p.a_scale[row];
      if (EPI == CHATTS_EPI_SWIGLU) {          // column tiles alternate gate / up (W rows interleaved in blocks of 16)
#pragma unroll
        for (int j = 0; j < FN; j += 2) {
          const int cg = ccol0 + j * 16, cu = cg + 16;
          if (cu >= p.n) continue;
          const float ar[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
          const float br[4] = {acc[i][j + 1].x, acc[i][j + 1].y, acc[i][j + 1].z, acc[i][j + 1].w};
          float g = ar[r] * sa * p.w_scale[cg], u = br[r] * sa * p.w_scale[cu];
          if (p.bias) { g += p.bias[cg]; u += p.bias[cu]; }
          p.c[(size_t)row * p.ldc + (cg >> 5) * 16 + (cg & 15)] = silu8_f(g) * u;
        }
      } else {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int col = ccol0 + j * 16;
          if (col >= p.n) continue;
          const float ar[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
          float v = ar[r] * sa * p.w_scale[col];
          if (p.bias) v += p.bias[col];
          if (EPI == CHATTS_EPI_GELU) v = gelu_erf_f(v);
          if (EPI == CHATTS_EPI_RESID) v = p.resid[(size_t)row * p.ldc + col] + v;
          p.c[(size_t)row * p.ldc + col] = v;
        }
      }
    }
  }
}

template <int EPI>
static void launch_fp8(const GemmFp8Params& p, hipStream_t s) {
  // 256- or 128-row tiles: whole rounds of workgroups over the CUs decide (a 256-row tile-step costs ~1.9x a 128-row one)
  const int cus = device_cus() > 0 ? device_cus() : 256;
  const long nt = (p.n + kF8BN - 1) / kF8BN;
  const long t256 = nt * ((p.m + 255) / 256), t128 = nt * ((p.m + 127) / 128);
  // (the 256-row form spills ~40 registers as compiled by ROCm 7.2: priced accordingly until it is measured to win)
  const double c256 = 2.2 * (double)((t256 + cus - 1) / cus), c128 = 1.0 * (double)((t128 + cus - 1) / cus);
  int wm = c256 < c128 ? 4 : 2;
  { const int e = opt_get(OPT_FP8_BM, 0); wm = e == 256 ? 4 : e == 128 ? 2 : wm; }
  static bool attr_done = false;
  if (!attr_done) {          // up to 147 KB of dynamic LDS: above the 64 KB default cap
    const int big = 2 * (256 + kF8BN) * kF8Row, small = 2 * (128 + kF8BN) * kF8Row;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<CHATTS_EPI_NONE, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<CHATTS_EPI_GELU, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<CHATTS_EPI_RESID, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<CHATTS_EPI_SWIGLU, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<CHATTS_EPI_NONE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, small);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<CHATTS_EPI_GELU, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, small);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<CHATTS_EPI_RESID, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, small);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<CHATTS_EPI_SWIGLU, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, small);
    attr_done = true;
  }
  if (wm == 4) {
    const dim3 grid((unsigned)((t256 + 7) / 8 * 8));
    hipLaunchKernelGGL((gemm_fp8_kernel<EPI, 4>), grid, dim3(kF8Threads), (size_t)2 * (256 + kF8BN) * kF8Row, s, p);
  } else {
    const dim3 grid((unsigned)((t128 + 7) / 8 * 8));
    hipLaunchKernelGGL((gemm_fp8_kernel<EPI, 2>), grid, dim3(kF8Threads), (size_t)2 * (128 + kF8BN) * kF8Row, s, p);
  }
}

}  // namespace chatts

using namespace chatts;

extern "C" int chatts_quantize_rows_fp8(const float* x, int m, int k, int ldx, const float* norm_w, float norm_eps, uint8_t* q, int ldq,
                                        float* scale, chatts_stream_t stream) {
  CHATTS_REQUIRE(m >= 0 && k > 0 && k % 4 == 0, CHATTS_E_SHAPE, "quantize_rows_fp8: m=%d k=%d (K must be a multiple of 4)", m, k);
  if (m == 0) return CHATTS_OK;
  CHATTS_REQUIRE(x && q && scale, CHATTS_E_BADARG, "quantize_rows_fp8: null pointer");
  CHATTS_REQUIRE(ldx >= k && ldx % 4 == 0 && ldq >= k && ldq % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)q % 4) == 0 &&
                     (!norm_w || ((uintptr_t)norm_w % 16) == 0), CHATTS_E_SHAPE, "quantize_rows_fp8: leading dimensions / alignment");
  hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3(m), dim3(256), 0, as_stream(stream), x, k, ldx, norm_w, norm_eps, q, ldq, scale);
  CHATTS_CHECK_LAUNCH("quantize_rows_fp8");
  return CHATTS_OK;
}

extern "C" int chatts_linear_fp8(const ChattsLinearFp8Args* a, chatts_stream_t stream) {
  CHATTS_REQUIRE(a, CHATTS_E_BADARG, "linear_fp8: null args");
  CHATTS_REQUIRE(a->m >= 0 && a->n > 0 && a->k > 0, CHATTS_E_BADARG, "linear_fp8: bad sizes m=%d n=%d k=%d", a->m, a->n, a->k);
  if (a->m == 0) return CHATTS_OK;
  CHATTS_REQUIRE(a->a8 && a->a_scale && a->w8 && a->w_scale && a->c, CHATTS_E_BADARG, "linear_fp8: null pointer");
  CHATTS_REQUIRE(a->k % kF8BK == 0, CHATTS_E_SHAPE, "linear_fp8: K=%d must be a multiple of %d (pad the operands with zeros)", a->k, kF8BK);
  CHATTS_REQUIRE(a->n % 16 == 0 && (a->epilogue != CHATTS_EPI_SWIGLU || a->n % 32 == 0), CHATTS_E_SHAPE, "linear_fp8: N=%d", a->n);
  CHATTS_REQUIRE(a->lda8 >= a->k && a->lda8 % 16 == 0 && a->ldw8 >= a->k && a->ldw8 % 16 == 0 && ((uintptr_t)a->a8 % 16) == 0 &&
                     ((uintptr_t)a->w8 % 16) == 0, CHATTS_E_SHAPE, "linear_fp8: operand leading dimensions / 16-byte alignment");
  const int ncols = a->epilogue == CHATTS_EPI_SWIGLU ? a->n / 2 : a->n;
  CHATTS_REQUIRE(a->ldc >= ncols, CHATTS_E_SHAPE, "linear_fp8: ldc=%d < %d", a->ldc, ncols);
  CHATTS_REQUIRE(a->epilogue != CHATTS_EPI_RESID || a->resid, CHATTS_E_BADARG, "linear_fp8: EPI_RESID without resid");
  const int order = opt_get(OPT_FP8_ORDER, 1);
  GemmFp8Params p{a->a8, a->a_scale, a->w8, a->w_scale, a->bias, a->resid, a->c, a->m, a->n, a->k, a->lda8, a->ldw8, a->ldc, order};
  switch (a->epilogue) {
    case CHATTS_EPI_GELU: launch_fp8<CHATTS_EPI_GELU>(p, as_stream(stream)); break;
    case CHATTS_EPI_RESID: launch_fp8<CHATTS_EPI_RESID>(p, as_stream(stream)); break;
    case CHATTS_EPI_SWIGLU: launch_fp8<CHATTS_EPI_SWIGLU>(p, as_stream(stream)); break;
    case CHATTS_EPI_NONE: launch_fp8<CHATTS_EPI_NONE>(p, as_stream(stream)); break;
    default: CHATTS_REQUIRE(false, CHATTS_E_BADARG, "linear_fp8: epilogue %d", a->epilogue);
  }
  CHATTS_CHECK_LAUNCH("gemm_fp8");
  return CHATTS_OK;
}
