// gemv_common.h - the arithmetic of the batch-1 weight-streaming GEMV, shared by gemv.hip and decode_mega.hip so that both
// accumulate a row in exactly the same order (bit-identical results).
#pragma once
#include "common.h"

namespace chatts {

__device__ __forceinline__ float dot8(const u32x4 wv, const f32x4 xa, const f32x4 xb, float acc) {
  acc = fmaf(bf16_lo(wv.x), xa.x, acc);
  acc = fmaf(bf16_hi(wv.x), xa.y, acc);
  acc = fmaf(bf16_lo(wv.y), xa.z, acc);
  acc = fmaf(bf16_hi(wv.y), xa.w, acc);
  acc = fmaf(bf16_lo(wv.z), xb.x, acc);
  acc = fmaf(bf16_hi(wv.z), xb.y, acc);
  acc = fmaf(bf16_lo(wv.w), xb.z, acc);
  acc = fmaf(bf16_hi(wv.w), xb.w, acc);
  return acc;
}

// one RMSNorm partial-sum step (Qwen2RMSNorm: sum of squares in float32); ONE definition = one contraction pattern everywhere
__device__ __forceinline__ float sumsq4(float ss, const f32x4 v) {
  ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  return ss;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

template <int ROWS, int EPI>
__device__ __forceinline__ int task_row(int task, int r) {
  if (EPI == CHATTS_EPI_SWIGLU) {   // gate/up interleaved in blocks of 16 rows: unit u -> rows g(u), g(u)+16
    const int unit = task * (ROWS / 2) + (r >> 1);
    return (unit >> 4) * 32 + (unit & 15) + (r & 1) * 16;
  }
  return task * ROWS + r;
}


// host: default waves per workgroup / workgroups per CU of the stand-alone bf16 GEMV (gemv.hip)
void gemv_default_geometry(int n, int k, int epilogue, int cus, int* nw_out, int* occ_out);

}  // namespace chatts
