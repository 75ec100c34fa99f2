// f16q.h - the "f16q" operand format of the 1.5-pass prefill projections (round 6; gemm_f16q.hip and the kernels that PRODUCE its planes).
//
// A float32 activation x is carried as   x = hi + lo   with
//     hi  = f16_rne(clamp(x, +-65504))                              one f16 plane            [M, K]
//     lo  = x - hi   (exact in float32; |lo| <= 2^-11 |x|)          quantised to OCP e4m3fn  [M, K]   q = e4m3_rne(lo * 2^-E)
//     E   = the block's scale exponent, one e8m0 byte (E + 127) per row and 128 consecutive K-values      [M, K / 128]:
//           the smallest power of two 2^E with  max|lo| / 2^E <= 448  over the block (no saturation); an all-zero block stores 127.
// The GEMM multiplies hi with an f16 copy of W on v_mfma_f32_16x16x32_f16 (products exact, float32 accumulate) and q with an e4m3 copy of W
// (per-row power-of-two scale) on v_mfma_scale_f32_16x16x128_f8f6f4 - the CDNA4 block-scaled matrix pipe, half the cycles of a bf16 pass -
// into the SAME accumulator: 1.5 pass-equivalents per product instead of the 2 of bf16 hi + bf16 lo.  What is lost: q and W8 carry 3
// mantissa bits, so the residual pass is 2^-4 accurate on a term that is 2^-11 of the product: ~2^-15 per product against 2^-17 of
// bf16x2.  Full depth (48 layers, emulated in float64: tools/split_emulation.py, profiles/r6_split_emulation.json): 1.3e-4 on the
// first-token logits against 4.1e-5 for bf16x2 and the 1e-3 bar.
// ONE definition of the arithmetic: every producer and tests/f16q_ref.py round the same way, so planes are bit-reproducible.
#pragma once
#include "common.h"

namespace chatts {

typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
constexpr int kF16qBlock = 128;

__device__ __forceinline__ _Float16 f16q_hi(float x) { return (_Float16)fminf(fmaxf(x, -65504.f), 65504.f); }
// the block's scale exponent E from the largest |lo| of the block
__device__ __forceinline__ int f16q_exp(float amax) {
  const uint32_t b = __float_as_uint(amax);
  if (b == 0u) return 0;
  const int E = (int)(b >> 23) - 127 - 8 + ((b & 0x7fffffu) > 0x600000u ? 1 : 0);      // amax = m 2^e: m <= 1.75 -> e - 8, else e - 7
  return E < -127 ? -127 : E;
}
__device__ __forceinline__ float f16q_inv(int E) { return __uint_as_float((uint32_t)(127 - E) << 23); }      // 2^-E, exact
__device__ __forceinline__ uint32_t f16q_byte(int E) { return (uint32_t)(E + 127); }
// four residuals -> four e4m3 codes (one dword); `inv` = 2^-E of their block
__device__ __forceinline__ uint32_t f16q_pack4(float l0, float l1, float l2, float l3, float inv) {
  int p = 0;
  p = __builtin_amdgcn_cvt_pk_fp8_f32(l0 * inv, l1 * inv, p, false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(l2 * inv, l3 * inv, p, true);
  return (uint32_t)p;
}

// plane pointers of one activation matrix in this format
struct F16qPlanes {
  _Float16* hi;      // [M, ld]
  uint8_t* lo8;      // [M, ld]
  uint8_t* sc;       // [M, ldsc]   ldsc >= K / 128
  int ld, ldsc;
  int tiled;         // hi / lo8 in the TILED layout (below) over a matrix of `ld` columns (then ld = K exactly); the scales stay row-major
};

// TILED planes: what an LDS-DMA piece of gemm_f16q_kernel deposits is consecutive memory (profiles/r6_feed_probe.txt: 27-30 B per clock and
// CU against 18 for 64-byte and 10-12 for 32-byte row slices).  Blocks of 16 rows x 32 K-values at index (row / 16) * (K / 32) + k / 32:
//   hi   1 KB per block: 16-byte chunk (8 values) c of row r at position l = 4 r + (c ^ ((r >> 3) << 1)), r = row % 16, c = (k / 8) % 4
//        (chatts_tile_bf16's order: the XOR is the fragment reads' bank swizzle)
//   lo8  512 B per block: bytes 16 h .. 16 h + 15 of row r (h = (k / 16) % 2) at position 16 h + r
// Element offsets of the 4 consecutive values k .. k + 3 (k % 4 == 0) of `row`:
__device__ __forceinline__ size_t f16q_hi_off(const F16qPlanes& o, int row, int k) {
  if (!o.tiled) return (size_t)row * o.ld + k;
  const int r = row & 15, c = (k >> 3) & 3;
  return ((size_t)(row >> 4) * (o.ld >> 5) + (k >> 5)) * 512 + (size_t)(((r << 2) | (c ^ ((r >> 3) << 1))) << 3) + (k & 7);
}
__device__ __forceinline__ size_t f16q_lo_off(const F16qPlanes& o, int row, int k) {
  if (!o.tiled) return (size_t)row * o.ld + k;
  return ((size_t)(row >> 4) * (o.ld >> 5) + (k >> 5)) * 512 + (size_t)(((((k >> 4) & 1) << 4) | (row & 15)) << 4) + (k & 15);
}

}  // namespace chatts
