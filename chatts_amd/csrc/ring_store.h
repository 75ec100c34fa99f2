// ring_store.h - the branch-free epilogue of the prefill kernels (gemm_ring.hip, gemm_f16q.hip): a compute wave holds, per fragment pair
// (i, j), 4 consecutive output columns of one token (MFMA operands swapped, D = W . A^T) - every load / store is 16 bytes.
#pragma once
#include "gemm_common.h"

namespace chatts {

// modes of the branch-free epilogue (selected once per unit)
enum { kStRaw = 0, kStNone, kStGelu, kStResid, kStSwiglu };

__device__ __forceinline__ void ring_wait_vmcnt(int n) {      // wave-uniform n: at most n of this wave's loads outstanding
#define CHATTS_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    CHATTS_VM(0) CHATTS_VM(1) CHATTS_VM(2) CHATTS_VM(3) CHATTS_VM(4) CHATTS_VM(5) CHATTS_VM(6) CHATTS_VM(7) CHATTS_VM(8) CHATTS_VM(9)
    CHATTS_VM(10) CHATTS_VM(11) CHATTS_VM(12) CHATTS_VM(13) CHATTS_VM(14) CHATTS_VM(15) CHATTS_VM(16) CHATTS_VM(17) CHATTS_VM(18)
    default: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;      // (never more than 2 x 9 in flight besides the awaited one)
  }
#undef CHATTS_VM
}

struct RingUnit {
  int panel, split, m0, f, f0, kbeg, nh;      // f0 = fragments of wave row 0 (ceil(f / 2)); nh = half-steps (K / 32) of this unit
};
__device__ __forceinline__ RingUnit ring_unit(const GemmParams& p, const RingGeom& g, int u) {
  RingUnit r;
  const int mt = u % g.T, rest = u / g.T;
  r.panel = rest % g.P;
  r.split = rest / g.P;
  const int base = g.F / g.T, rem = g.F % g.T;
  r.f = base + (mt < rem);
  r.m0 = (mt * base + (mt < rem ? mt : rem)) * 16;
  r.f0 = (r.f + 1) >> 1;
  r.kbeg = r.split * p.k_per_split;
  int kend = r.kbeg + p.k_per_split;
  if (kend > p.k) kend = p.k;
  r.nh = (kend - r.kbeg) >> 5;
  return r;
}

// The epilogue of one wave: acc[i][j] = D[feature = fb .. fb + 3][token] of fragment pair (i, j), fb = n0 + wn * 64 + j * 16 + (lane >> 4) * 4,
// token = m0 + rowbase + i * 16 + (lane & 15).  MODE / PLANES are compile-time: no branches, all loads of a row block in flight at once.
template <int MODE, bool PLANES>
__device__ __forceinline__ void ring_store(const GemmParams& p, const f32x4 (&acc)[5][4], int fml, int tok0, int fb0, int lane, int split) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  const int tl = lane & 15, fq = (lane >> 4) * 4;
  auto put = [&](size_t row_off_c, size_t row_off_p, int col, const float (&v)[4]) __attribute__((always_inline)) {
    if constexpr (PLANES) {
#pragma clang fp contract(off)      // lo is the split of the ROUNDED float32 value (store_planes' arithmetic)
      bf16x4_t hv, lv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const __bf16 h = (__bf16)v[r];
        hv[r] = h;
        lv[r] = (__bf16)(v[r] - (float)h);
      }
      *reinterpret_cast<bf16x4_t*>(p.c_hi + row_off_p + col) = hv;
      *reinterpret_cast<bf16x4_t*>(p.c_lo + row_off_p + col) = lv;
    } else {
      *reinterpret_cast<f32x4*>(p.c + row_off_c + col) = (f32x4){v[0], v[1], v[2], v[3]};
    }
  };
  if constexpr (MODE == kStRaw) {
    float* ws = p.c + (size_t)split * p.m * p.n;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i < fml) {
        const int tok = tok0 + i * 16 + tl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int fb = fb0 + j * 16 + fq;
          if (tok < p.m && fb < p.n) *reinterpret_cast<f32x4*>(ws + (size_t)tok * p.n + fb) = acc[i][j];
        }
      }
    }
  } else if constexpr (MODE == kStSwiglu) {
    f32x4 bg[2], bu[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int fb = fb0 + q * 32 + fq;                 // packed gate rows; the up rows are + 16
      if (fb + 19 >= p.n) fb = 0;
      bg[q] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + fb) : (f32x4){0.f, 0.f, 0.f, 0.f};
      bu[q] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + fb + 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i < fml) {
        const int tok = tok0 + i * 16 + tl;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int fb = fb0 + q * 32 + fq;
          const int ocol = (fb >> 5) * 16 + fq;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = silu_g(acc[i][2 * q][r] + bg[q][r]) * (acc[i][2 * q + 1][r] + bu[q][r]);
          if (tok < p.m && fb + 19 < p.n) put((size_t)tok * p.ldc, (size_t)tok * p.ldcp, ocol, v);
        }
      }
    }
  } else {
    f32x4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int fb = fb0 + j * 16 + fq;
      if (fb >= p.n) fb = 0;
      b4[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + fb) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i < fml) {
        const int tok = tok0 + i * 16 + tl;
        const int tokc = tok < p.m ? tok : p.m - 1;      // clamped: the loads are unconditional, the stores masked
        f32x4 rs[4];
        if constexpr (MODE == kStResid) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            int fb = fb0 + j * 16 + fq;
            if (fb >= p.n) fb = 0;
            rs[j] = *reinterpret_cast<const f32x4*>(p.resid + (size_t)tokc * p.ldc + fb);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int fb = fb0 + j * 16 + fq;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[i][j][r] + b4[j][r];
            if constexpr (MODE == kStGelu) v[r] = gelu_erf_f(v[r]);
            if constexpr (MODE == kStResid) v[r] = rs[j][r] + v[r];
          }
          if (tok < p.m && fb < p.n) put((size_t)tok * p.ldc, (size_t)tok * p.ldcp, fb, v);
        }
      }
    }
  }
}

}  // namespace chatts
