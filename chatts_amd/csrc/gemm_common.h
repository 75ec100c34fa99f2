// gemm_common.h - what the MFMA GEMM kernels share: the launch parameters and the small epilogue helpers (gemm.hip, gemm_ring.hip).
#pragma once
#include "common.h"

namespace chatts {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct GemmParams {
  const float* a;
  const uint16_t* w;
  const float* bias;
  const float* resid;
  float* c;          // final output, or split-K partials [sk][M][N]
  int m, n, k, lda, ldw, ldc, epilogue;
  int k_per_split;   // multiple of 32
  int direct;        // 1: apply epilogue here; 0: write raw partials
  const uint8_t* w8;     // optional fp8 (e4m3fn) copy of W: streamed instead of the bf16 copy, widened (exactly) to
  const float* w8_scale; // bf16 while it is staged to LDS; the per-row power-of-two scale is applied in the epilogue
  int ldw8;
  int w8_format;         // CHATTS_W8_FP8 / CHATTS_W8_INT8 (gemm_stream_kernel only; the other kernels never see an int8 copy)
  uint16_t* c_hi;        // optional: the output goes out as bf16 hi / lo planes [M, ldcp] (the next GEMM's operand
  uint16_t* c_lo;        // format) instead of float32 c
  int ldcp;
  const uint16_t* wt;    // optional (gemm_ring_kernel): W in the tiled layout of chatts_tile_bf16 - a 1 KB LDS-DMA piece is 1 KB of memory
  int planes_tiled;      // a_hi / a_lo are in that layout too
};

__device__ __forceinline__ void store_planes(uint16_t* hi, uint16_t* lo, size_t off, float v) {
  // no contraction: when v is a product (SwiGLU) the compiler would otherwise fold it into the subtraction as an FMA and
  // lo would no longer be the split of the ROUNDED float32 value the float32 path stores
#pragma clang fp contract(off)
  const __bf16 h = (__bf16)v;
  const __bf16 l = (__bf16)(v - (float)h);
  hi[off] = __builtin_bit_cast(uint16_t, h);
  lo[off] = __builtin_bit_cast(uint16_t, l);
}

__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu_g(float x) { return x / (1.0f + expf(-x)); }

constexpr int kRingPanel = 256;      // W rows per panel of the prefill kernel
// gemm_ring.hip: the prefill kernel's decomposition (M-tiles per panel, K splits) and launcher
struct RingGeom {
  int T, sk;          // M-tiles per panel, K splits
  int F, P;           // 16-row fragments of M, W panels
  int units;          // T * P * sk, ordered (split, panel, M-tile): the M-tiles of a panel are adjacent
  int wpx;            // workgroups per XCD (grid = 8 * wpx)
};
void ring_pick(int m, int n, int k, int cus, int force_t, int force_sk, RingGeom& g);
int launch_ring(const GemmParams& p, const uint16_t* a_hi, const uint16_t* a_lo, int ldp, const RingGeom& g, bool single, hipStream_t s);

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

}  // namespace chatts
