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
//     operand formats e4m3, block scales fixed to 1.0: the instruction is the only fp8 form that runs at twice the bf16 rate) - 128 x
//     128 x 128 tiles, 4 waves of 64 x 64, operands staged global -> registers -> LDS (rows padded to 144 bytes: the 32-byte
//     fragment reads of 16 rows spread over the banks), next tile's global loads in flight under the current tile's MFMAs.
//     Weights: the per-row power-of-two-scaled e4m3 copies the fp8 decode path already streams (ChattsLinearArgs.w8).
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
};

constexpr int kF8BM = 128, kF8BN = 128, kF8BK = 128;      // BK in bytes = fp8 values
constexpr int kF8Row = kF8BK + 16;                          // padded LDS row (bytes)
constexpr int kF8Tile = kF8BM * kF8Row;                     // one operand tile in LDS

__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu8_f(float x) { return x / (1.0f + expf(-x)); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_fp8_kernel(GemmFp8Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 stages][A tile | W tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * kF8BN, m0 = blockIdx.y * kF8BM;
  // staging: thread t loads 16-byte piece (row = t / 8 + 32 j, column block t % 8) of both tiles, j < 4
  const int srow = tid >> 3, scol = (tid & 7) * 16;
  const uint8_t* ag[4];
  const uint8_t* wg[4];
  bool aok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ar = m0 + srow + 32 * j, wr = n0 + srow + 32 * j;
    aok[j] = ar < p.m;
    ag[j] = p.a8 + (size_t)(aok[j] ? ar : 0) * p.lda + scol;
    wg[j] = p.w8 + (size_t)(wr < p.n ? wr : 0) * p.ldw + scol;
  }
  i32x4 ra[4], rw[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ra[j] = aok[j] ? *reinterpret_cast<const i32x4*>(ag[j] + k0) : (i32x4){0, 0, 0, 0};
      rw[j] = *reinterpret_cast<const i32x4*>(wg[j] + k0);
    }
  };
  auto lstore = [&](int stage) {
    char* at = smem + (size_t)stage * 2 * kF8Tile;
    char* wt = at + kF8Tile;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<i32x4*>(at + (srow + 32 * j) * kF8Row + scol) = ra[j];
      *reinterpret_cast<i32x4*>(wt + (srow + 32 * j) * kF8Row + scol) = rw[j];
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.k / kF8BK;
  gload(0);
  lstore(0);
  __syncthreads();
  // fragment addresses: lane l holds row (l % 16) of a 16-row block, K bytes 32 (l / 16) .. + 31 (A and W alike: the dot product
  // pairs equal byte positions of equal lane groups)
  const int frow = lane & 15, fcol = (lane >> 4) * 32;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * kF8BK);
    const char* at = smem + (size_t)cur * 2 * kF8Tile + (size_t)(wm * 64 + frow) * kF8Row + fcol;
    const char* wt = smem + (size_t)cur * 2 * kF8Tile + kF8Tile + (size_t)(wn * 64 + frow) * kF8Row + fcol;
    i32x8 af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const i32x4 lo = *reinterpret_cast<const i32x4*>(at + i * 16 * kF8Row), hi = *reinterpret_cast<const i32x4*>(at + i * 16 * kF8Row + 16);
      af[i] = (i32x8){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      const i32x4 l2 = *reinterpret_cast<const i32x4*>(wt + i * 16 * kF8Row), h2 = *reinterpret_cast<const i32x4*>(wt + i * 16 * kF8Row + 16);
      bf[i] = (i32x8){l2.x, l2.y, l2.z, l2.w, h2.x, h2.y, h2.z, h2.w};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i], bf[j], acc[i][j], 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 0x7f7f7f7f,
                                                                     0, 0x7f7f7f7f /* E8M0 127 = 2^0 */);
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }

  // epilogue: acc[i][j][r] = C[m0 + wm 64 + i 16 + (lane / 16) 4 + r][n0 + wn 64 + j 16 + lane % 16]
  const int crow0 = m0 + wm * 64 + (lane >> 4) * 4, ccol0 = n0 + wn * 64 + (lane & 15);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = crow0 + i * 16 + r;
      if (row >= p.m) continue;
      const float sa = p.a_scale[row];
      if (EPI == CHATTS_EPI_SWIGLU) {          // column tiles alternate gate / up (W rows interleaved in blocks of 16)
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
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
        for (int j = 0; j < 4; ++j) {
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
  GemmFp8Params p{a->a8, a->a_scale, a->w8, a->w_scale, a->bias, a->resid, a->c, a->m, a->n, a->k, a->lda8, a->ldw8, a->ldc};
  const dim3 grid((a->n + kF8BN - 1) / kF8BN, (a->m + kF8BM - 1) / kF8BM);
  const size_t lds = (size_t)2 * 2 * kF8Tile;
  static bool attr_done = false;
  if (!attr_done) {          // 73 KB of dynamic LDS: above the 64 KB default cap
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<CHATTS_EPI_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<CHATTS_EPI_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<CHATTS_EPI_RESID>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel<CHATTS_EPI_SWIGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  switch (a->epilogue) {
    case CHATTS_EPI_GELU: hipLaunchKernelGGL(gemm_fp8_kernel<CHATTS_EPI_GELU>, grid, dim3(256), lds, as_stream(stream), p); break;
    case CHATTS_EPI_RESID: hipLaunchKernelGGL(gemm_fp8_kernel<CHATTS_EPI_RESID>, grid, dim3(256), lds, as_stream(stream), p); break;
    case CHATTS_EPI_SWIGLU: hipLaunchKernelGGL(gemm_fp8_kernel<CHATTS_EPI_SWIGLU>, grid, dim3(256), lds, as_stream(stream), p); break;
    case CHATTS_EPI_NONE: hipLaunchKernelGGL(gemm_fp8_kernel<CHATTS_EPI_NONE>, grid, dim3(256), lds, as_stream(stream), p); break;
    default: CHATTS_REQUIRE(false, CHATTS_E_BADARG, "linear_fp8: epilogue %d", a->epilogue);
  }
  CHATTS_CHECK_LAUNCH("gemm_fp8");
  return CHATTS_OK;
}
