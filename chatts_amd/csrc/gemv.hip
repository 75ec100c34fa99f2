// gemv.hip - batch-1 decode projections: y[N] = epilogue(W[N,K] . x[K]),  W bf16 streamed from HBM.
//
// This kernel sets generated tokens/s: every decode step streams all decoder weights once
// (27.98 GB for ChatTS-14B), so it is HBM-bound by construction.  It follows the guide's
// "GEMV / M<=16 decode weights" rule: weights go straight to VGPRs as non-temporal 16-byte loads,
// many loads in flight per lane, no LDS round trip for the streamed operand.
//
// Geometry: one wave per group of 2 rows over the whole K; the small, re-used operand x (20-55 KB f32) is
// staged once per workgroup in LDS (4-16 waves share it), permuted so that each lane's two 16-byte reads
// per 512-element chunk are lane-linear (conflict-free ds_read_b128); 2 rows x 2 chunks = 4 independent
// 16-byte weight loads in flight per lane, 16-32 waves per CU.  A K-split variant that kept x in registers
// and issued 12-14 loads per lane was measured and lost on every shape (tools/gemv_sweep.py).
// Arithmetic: bf16 -> f32 widening is exact and x is f32, so every product is exact in the f32 FMA;
// only the summation order differs from the CPU oracle (parity budget: 1e-3 relative on logits).
// Fusions: RMSNorm of x in the prologue (Qwen2RMSNorm.forward), bias, residual add, SwiGLU.
#include "common.h"
#include "gemv_common.h"
#include "tp_common.h"

namespace chatts {

// EPI_RESID of a row-parallel projection under tensor parallelism (ChattsLinearArgs.tp_reduce): the sum over the ranks is formed
// inside the launch.  Internal template value, not part of the C-ABI's epilogue enum.
constexpr int kEpiResidTp = 4;

struct GemvParams {
  const uint16_t* w;
  const float* x;
  const float* bias;
  const float* resid;
  float* out;
  const float* norm_w;
  float eps;
  int n;       // weight rows
  int k;
  int ldw;
  int tasks;   // number of row-group tasks
  const uint8_t* w8;      // fp8 (OCP e4m3fn) weights [N, ldw8] or NULL
  const float* w8_scale;  // per-row power-of-two scale: w = scale[row] * float(w8)
  int ldw8;
  int w8_format;          // CHATTS_W8_FP8 / CHATTS_W8_INT8
  // optional 4-bit copy of W (GPTQ codes, row-major: byte j of a row = codes 2j | 2j+1 << 4) with one (scale, scale * zero) pair
  // per `w4_group` weights of a row: w = bf16_rne(q * scale - scale_zero) - exactly the bf16 matrix the other kernels stream
  const uint8_t* w4;
  const float* w4_sz;     // [N, K / w4_group, 2]
  int ldw4, w4_group;
  TpParams tp;            // kEpiResidTp only
  int ks;                 // gemv_ksplit_kernel: waves per row group
};

// ---- the exchange inside the launch (kEpiResidTp) ---------------------------------------------------------------------------
// A wave that has finished the ROWS rows of a task PUSHES them at once - lane l holds (row l % ROWS, destination l / ROWS): ROWS x W
// independent 8-byte granule stores, one per xGMI link and row - and goes on streaming its next task.  After its last task it
// POLLS the rows of each of its tasks from the local exchange buffer - lane l = (row l % ROWS, source rank l / ROWS), all W x ROWS
// polls of a task in flight at once - adds them in RANK ORDER (v_readlane: every rank forms the same sum, bit for bit, so the
// replicated residual stream cannot diverge) and writes out = resid + sum: the arithmetic of tp_allreduce_kernel.  Between a
// wave's push and its poll lie the peers' pushes of the same rows, i.e. one link latency when the ranks run in step; the other waves
// of the CU keep the weight stream going meanwhile.  Every rank launches the same geometry, so the workgroup that waits for a row
// and the peer workgroup that produces it have the same index: no circular wait even if a grid were not fully resident.
template <int ROWS>
__device__ __forceinline__ void tp_push_task(const GemvParams& p, uint32_t epoch, int task, int lane, const float (&acc)[ROWS]) {
  static_assert(ROWS * kMaxWorld <= 64, "one lane per (row, rank)");
  const int r = lane % ROWS, k = lane / ROWS;
  const int row = task * ROWS + r;
  float v = acc[0];
#pragma unroll
  for (int i = 1; i < ROWS; ++i)
    if (r == i) v = acc[i];
  if (k < p.tp.world && row < p.n) {
    const int q = (p.tp.rank + k) % p.tp.world;           // start with myself, then ring order: spreads the links
    put(push_ptr(p.tp, q, epoch) + row, epoch, push_bits(p.tp, q, __float_as_uint(v)));
  }
}
template <int ROWS>
__device__ __forceinline__ void tp_take_task(const GemvParams& p, uint32_t epoch, int task, int lane) {
  const int r = lane % ROWS, src = lane / ROWS;
  const int row = task * ROWS + r;
  uint32_t bits = 0;
  if (src < p.tp.world && row < p.n) (void)take(slot_ptr(p.tp, p.tp.rank, epoch, src) + row, epoch, bits, &p.tp.ctr[2]);
  float sum[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) sum[i] = 0.f;
#pragma unroll
  for (int s = 0; s < kMaxWorld; ++s) {
    if (s < p.tp.world) {                                  // (wave-uniform)
#pragma unroll
      for (int i = 0; i < ROWS; ++i) {
        const float v = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)bits, s * ROWS + i));
        sum[i] = s == 0 ? v : sum[i] + v;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int rw = task * ROWS + i;
    if (lane == i && rw < p.n) p.out[rw] = p.resid[rw] + sum[i];
  }
}

// lane r (< ROWS, or < ROWS/2 for SwiGLU) of the finishing wave writes row r
template <int ROWS, int EPI>
__device__ __forceinline__ void gemv_epilogue(const GemvParams& p, int task, int lane, const float (&acc)[ROWS], uint32_t epoch = 0) {
  if (EPI == kEpiResidTp) {
    tp_push_task<ROWS>(p, epoch, task, lane, acc);
  } else if (EPI == CHATTS_EPI_SWIGLU) {
#pragma unroll
    for (int u = 0; u < ROWS / 2; ++u) {
      const int rg = task_row<ROWS, EPI>(task, 2 * u);
      if (lane == u && rg + 16 < p.n) {
        float g = acc[2 * u], v = acc[2 * u + 1];
        if (p.bias) { g += p.bias[rg]; v += p.bias[rg + 16]; }
        p.out[task * (ROWS / 2) + u] = silu_f(g) * v;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int row = task * ROWS + r;
      if (lane == r && row < p.n) {
        float v = acc[r];
        if (p.bias) v += p.bias[row];
        if (EPI == CHATTS_EPI_RESID) v = p.resid[row] + v;
        p.out[row] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// x in LDS, one wave per row group over the whole K.  blockDim = 64 * NW (NW = 4, 8 or 16 waves share one
// staged copy of x); ROWS x UNR 16-byte weight loads in flight per lane.
// ------------------------------------------------------------------------------------------------
template <int ROWS, int UNR, int EPI, bool NORM>
__global__ __launch_bounds__(1024) void gemv_ldsx_kernel(GemvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* xs4 = reinterpret_cast<f32x4*>(smem);            // permuted x: [chunk][half][lane] float4
  float* red = reinterpret_cast<float*>(smem) + (size_t)((p.k + 511) / 512) * 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x, nw = nthreads >> 6;
  const int K = p.k;
  const int nchunks = (K + 511) >> 9;
  uint32_t epoch = 0;
  if (EPI == kEpiResidTp) epoch = tp_epoch(p.tp);

  float rstd = 1.f;
  if (NORM) {
    float ss = 0.f;
    for (int k4 = tid * 4; k4 < K; k4 += nthreads * 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.x + k4);
      ss = sumsq4(ss, v);
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    rstd = rsqrtf(t / (float)K + p.eps);
  }
  for (int k4 = tid * 4; k4 < nchunks * 512; k4 += nthreads * 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k4 < K) {
      v = *reinterpret_cast<const f32x4*>(p.x + k4);
      if (NORM) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.norm_w + k4);
        v.x = g.x * (v.x * rstd); v.y = g.y * (v.y * rstd); v.z = g.z * (v.z * rstd); v.w = g.w * (v.w * rstd);
      }
    }
    const int chunk = k4 >> 9, within = k4 & 511;
    xs4[chunk * 128 + ((within >> 2) & 1) * 64 + (within >> 3)] = v;
  }
  __syncthreads();

  for (int task = blockIdx.x * nw + wave; task < p.tasks; task += gridDim.x * nw) {
    const uint16_t* wrow[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      int row = task_row<ROWS, EPI>(task, r);
      if (row >= p.n) row = 0;
      wrow[r] = p.w + (size_t)row * p.ldw + lane * 8;
    }
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
    for (int c = 0; c < nchunks; c += UNR) {
      u32x4 wv[UNR][ROWS];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const bool ok = c + u < nchunks && ((c + u) << 9) + lane * 8 < K;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          wv[u][r] = (u32x4){0u, 0u, 0u, 0u};
          if (ok) wv[u][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + ((c + u) << 9)));
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (c + u < nchunks) {
          const f32x4 xa = xs4[(c + u) * 128 + lane], xb = xs4[(c + u) * 128 + 64 + lane];
#pragma unroll
          for (int r = 0; r < ROWS; ++r) acc[r] = dot8(wv[u][r], xa, xb, acc[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]);
    gemv_epilogue<ROWS, EPI>(p, task, lane, acc, epoch);
  }
  if (EPI == kEpiResidTp) {
    // (no counter bump: the launch's epoch is counted by the host, TpParams::idx - a later collective of the step bumps the counter)
    for (int task = blockIdx.x * nw + wave; task < p.tasks; task += gridDim.x * nw) tp_take_task<ROWS>(p, epoch, task, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// K-split form for SHARD-sized projections (a rank of TP = 4 / 8: qkv is 896 x 5120 at TP = 8 - 448 row groups for 256 CUs).  A wave
// that walks the whole K alone is a chain of K / 1024 dependent HBM round trips, and with one or two such waves per CU nothing
// hides them: the launch sits at 9.6 us for 9 MB (profiles/r4_tp8_shard_kernel_trace.txt) where 3.3 us + bytes / 6.7 TB/s says 4.7.
// Here p.ks waves share a row group: wave j of the group takes the chunk pairs j, j + ks, ... (all of its loads in flight at once when
// ks = chunks / 2), the partial sums meet in LDS and wave 0 of the group adds them in wave order (fixed -> reproducible; NOT the
// bits of the whole-K kernel, which stays the one every TP = 1 shape runs) and applies the epilogue.  Same x staging / fused RMSNorm.
// ------------------------------------------------------------------------------------------------
template <int EPI, bool NORM>
__global__ __launch_bounds__(1024) void gemv_ksplit_kernel(GemvParams p) {
  constexpr int ROWS = 2, UNR = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* xs4 = reinterpret_cast<f32x4*>(smem);            // permuted x: [chunk][half][lane] float4
  float* red = reinterpret_cast<float*>(smem) + (size_t)((p.k + 511) / 512) * 512;
  float* part = red + 16;                                 // [nw][ROWS] partial sums of the K-slices
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x, nw = nthreads >> 6;
  const int K = p.k;
  const int nchunks = (K + 511) >> 9;

  float rstd = 1.f;
  if (NORM) {
    float ss = 0.f;
    for (int k4 = tid * 4; k4 < K; k4 += nthreads * 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.x + k4);
      ss = sumsq4(ss, v);
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    rstd = rsqrtf(t / (float)K + p.eps);
  }
  for (int k4 = tid * 4; k4 < nchunks * 512; k4 += nthreads * 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k4 < K) {
      v = *reinterpret_cast<const f32x4*>(p.x + k4);
      if (NORM) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.norm_w + k4);
        v.x = g.x * (v.x * rstd); v.y = g.y * (v.y * rstd); v.z = g.z * (v.z * rstd); v.w = g.w * (v.w * rstd);
      }
    }
    const int chunk = k4 >> 9, within = k4 & 511;
    xs4[chunk * 128 + ((within >> 2) & 1) * 64 + (within >> 3)] = v;
  }
  __syncthreads();

  const int ks = p.ks, ngroups = nw / ks, grp = wave / ks, j = wave - grp * ks;
  for (int t0 = blockIdx.x * ngroups; t0 < p.tasks; t0 += gridDim.x * ngroups) {      // (uniform trip count: barriers inside)
    const int task = t0 + grp;
    const bool valid = grp < ngroups && task < p.tasks;
    float acc[ROWS] = {0.f, 0.f};
    if (valid) {
      const uint16_t* wrow[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        int row = task_row<ROWS, EPI>(task, r);
        if (row >= p.n) row = 0;
        wrow[r] = p.w + (size_t)row * p.ldw + lane * 8;
      }
      for (int c = j * UNR; c < nchunks; c += ks * UNR) {
        u32x4 wv[UNR][ROWS];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const bool ok = c + u < nchunks && ((c + u) << 9) + lane * 8 < K;
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            wv[u][r] = (u32x4){0u, 0u, 0u, 0u};
            if (ok) wv[u][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + ((c + u) << 9)));
          }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          if (c + u < nchunks) {
            const f32x4 xa = xs4[(c + u) * 128 + lane], xb = xs4[(c + u) * 128 + 64 + lane];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[r] = dot8(wv[u][r], xa, xb, acc[r]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]);
      if (lane == 0) { part[wave * ROWS] = acc[0]; part[wave * ROWS + 1] = acc[1]; }
    }
    __syncthreads();
    if (valid && j == 0) {
      float tot[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        tot[r] = part[wave * ROWS + r];
        for (int i = 1; i < ks; ++i) tot[r] += part[(wave + i) * ROWS + r];          // wave order: fixed
      }
      gemv_epilogue<ROWS, EPI>(p, task, lane, tot);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// fp8 weights (OCP e4m3fn, CDNA4 v_cvt_pk_f32_fp8): the same geometry with 16 weights per 16-byte load, so a
// chunk is 1024 elements and x is staged as [chunk][quarter][lane] float4.  Each row carries a power-of-two
// scale, w = scale[row] * float(q): the dequantised weight is exactly representable in bf16, i.e. the fp8 copy
// is a lossless encoding of the bf16 matrix the prefill GEMM streams.  Halves the decode weight traffic.
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot16_fp8(const u32x4 wv, const f32x4 x0, const f32x4 x1, const f32x4 x2, const f32x4 x3,
                                            float acc) {
  f32x2 a;
  a = __builtin_amdgcn_cvt_pk_f32_fp8(wv.x, false); acc = fmaf(a.x, x0.x, acc); acc = fmaf(a.y, x0.y, acc);
  a = __builtin_amdgcn_cvt_pk_f32_fp8(wv.x, true);  acc = fmaf(a.x, x0.z, acc); acc = fmaf(a.y, x0.w, acc);
  a = __builtin_amdgcn_cvt_pk_f32_fp8(wv.y, false); acc = fmaf(a.x, x1.x, acc); acc = fmaf(a.y, x1.y, acc);
  a = __builtin_amdgcn_cvt_pk_f32_fp8(wv.y, true);  acc = fmaf(a.x, x1.z, acc); acc = fmaf(a.y, x1.w, acc);
  a = __builtin_amdgcn_cvt_pk_f32_fp8(wv.z, false); acc = fmaf(a.x, x2.x, acc); acc = fmaf(a.y, x2.y, acc);
  a = __builtin_amdgcn_cvt_pk_f32_fp8(wv.z, true);  acc = fmaf(a.x, x2.z, acc); acc = fmaf(a.y, x2.w, acc);
  a = __builtin_amdgcn_cvt_pk_f32_fp8(wv.w, false); acc = fmaf(a.x, x3.x, acc); acc = fmaf(a.y, x3.y, acc);
  a = __builtin_amdgcn_cvt_pk_f32_fp8(wv.w, true);  acc = fmaf(a.x, x3.z, acc); acc = fmaf(a.y, x3.w, acc);
  return acc;
}

// int8 form of the same 16 weights: four sign-extended bytes per dword (v_cvt_f32_i32 with an SDWA byte select, or a bit-field extract
// before it - the compiler's choice), the same float32 FMAs.  |q| <= 127: the products are exact as in the fp8 form.
__device__ __forceinline__ float dot16_i8(const u32x4 wv, const f32x4 x0, const f32x4 x1, const f32x4 x2, const f32x4 x3, float acc) {
  auto four = [&](uint32_t d, const f32x4 x) {
    acc = fmaf((float)(int)(int8_t)(d & 0xffu), x.x, acc);
    acc = fmaf((float)(int)(int8_t)((d >> 8) & 0xffu), x.y, acc);
    acc = fmaf((float)(int)(int8_t)((d >> 16) & 0xffu), x.z, acc);
    acc = fmaf((float)((int)d >> 24), x.w, acc);
  };
  four(wv.x, x0); four(wv.y, x1); four(wv.z, x2); four(wv.w, x3);
  return acc;
}

template <int ROWS, int UNR, int EPI, bool NORM, bool I8 = false>
__global__ __launch_bounds__(1024) void gemv8_ldsx_kernel(GemvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* xs4 = reinterpret_cast<f32x4*>(smem);            // permuted x: [chunk][quarter][lane] float4
  const int K = p.k;
  const int nchunks = (K + 1023) >> 10;
  float* red = reinterpret_cast<float*>(smem) + (size_t)nchunks * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x, nw = nthreads >> 6;
  uint32_t epoch = 0;
  if (EPI == kEpiResidTp) epoch = tp_epoch(p.tp);

  float rstd = 1.f;
  if (NORM) {
    float ss = 0.f;
    for (int k4 = tid * 4; k4 < K; k4 += nthreads * 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.x + k4);
      ss = sumsq4(ss, v);
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    rstd = rsqrtf(t / (float)K + p.eps);
  }
  for (int k4 = tid * 4; k4 < nchunks * 1024; k4 += nthreads * 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k4 < K) {
      v = *reinterpret_cast<const f32x4*>(p.x + k4);
      if (NORM) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.norm_w + k4);
        v.x = g.x * (v.x * rstd); v.y = g.y * (v.y * rstd); v.z = g.z * (v.z * rstd); v.w = g.w * (v.w * rstd);
      }
    }
    const int chunk = k4 >> 10, within = k4 & 1023;
    xs4[chunk * 256 + ((within >> 2) & 3) * 64 + (within >> 4)] = v;
  }
  __syncthreads();

  for (int task = blockIdx.x * nw + wave; task < p.tasks; task += gridDim.x * nw) {
    const uint8_t* wrow[ROWS];
    float scale[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      int row = task_row<ROWS, EPI>(task, r);
      if (row >= p.n) row = 0;
      wrow[r] = p.w8 + (size_t)row * p.ldw8 + lane * 16;
      scale[r] = p.w8_scale[row];
    }
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
    for (int c = 0; c < nchunks; c += UNR) {
      u32x4 wv[UNR][ROWS];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const bool ok = c + u < nchunks && ((c + u) << 10) + lane * 16 < K;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          wv[u][r] = (u32x4){0u, 0u, 0u, 0u};
          if (ok) wv[u][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + ((c + u) << 10)));
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (c + u < nchunks) {
          const f32x4* xb = xs4 + (c + u) * 256 + lane;
          const f32x4 x0 = xb[0], x1 = xb[64], x2 = xb[128], x3 = xb[192];
#pragma unroll
          for (int r = 0; r < ROWS; ++r) acc[r] = I8 ? dot16_i8(wv[u][r], x0, x1, x2, x3, acc[r]) : dot16_fp8(wv[u][r], x0, x1, x2, x3, acc[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]) * scale[r];     // power-of-two scale: exact
    gemv_epilogue<ROWS, EPI>(p, task, lane, acc, epoch);
  }
  if (EPI == kEpiResidTp) {
    // (no counter bump: the launch's epoch is counted by the host, TpParams::idx - a later collective of the step bumps the counter)
    for (int task = blockIdx.x * nw + wave; task < p.tasks; task += gridDim.x * nw) tp_take_task<ROWS>(p, epoch, task, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// 4-bit weights (GPTQ-Int4 checkpoints: ChatTS-14B-GPTQ-Int4, NetManAIOps/ChatTS README.md:52,262-263).  Same geometry as the
// fp8 kernel - a lane's load is 8 bytes = 16 codes, a chunk is 1024 elements, x staged as [chunk][quarter][lane] float4 -
// with a (scale, scale * zero) pair per group of `w4_group` (a multiple of 16) weights.  The engine's weights are DEFINED as
// bf16_rne(scale * (q - zero)) (chatts_amd/gptq.py: the prefill GEMM streams that bf16 matrix), so the kernel rebuilds exactly
// that value: q * scale - scale * zero is exact in float32 (fp16 scale x 5-bit integer), one v_cvt_pk_bf16_f32 per pair rounds
// it, and the widened result meets x in the same exact-product float32 FMA as everywhere else.  ~5 VALU ops per weight at
// 0.5 B per weight: the VALU and HBM are about equally loaded.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// 16 codes (two dwords) of one row against 16 x values held as packed bf16 pairs (hi and lo planes of the float32 x):
//   per byte b of a dword: codes 2b (low nibble) and 2b+1 (high nibble) -> v_cvt_f32_ubyte{b} of the masked dwords (no shifts),
//   one packed FMA builds both float32 weights, one v_cvt_pk_bf16_f32 rounds them to THE bf16 weights as a pair, and two
//   v_dot2c_f32_bf16 multiply the pair with x_hi and x_lo (exact products, float32 accumulate: the bf16x2 scheme of the
//   MFMA GEMMs).  ~3.4 VALU instructions per weight (the first version, unpacking to float32 FMAs, needed 7.7 and ran at
//   the speed of the bf16 kernel - profiles/r2_pmc_gemv4.txt).
#define CHATTS_CVT_UBYTE(dst, src, n) asm("v_cvt_f32_ubyte" #n " %0, %1" : "=v"(dst) : "v"(src))
__device__ __forceinline__ void dot16_int4(const u32x2 wv, const float sc, const float nsz, const u32x4 xh0, const u32x4 xh1,
                                           const u32x4 xl0, const u32x4 xl1, float& acc_h, float& acc_l) {
  auto eight = [&](uint32_t d, const u32x4 xh, const u32x4 xl) {
    const uint32_t lo = d & 0x0f0f0f0fu, hi = (d >> 4) & 0x0f0f0f0fu;      // bytes = codes of the even / odd elements
    float qe[4], qo[4];
    CHATTS_CVT_UBYTE(qe[0], lo, 0); CHATTS_CVT_UBYTE(qo[0], hi, 0);
    CHATTS_CVT_UBYTE(qe[1], lo, 1); CHATTS_CVT_UBYTE(qo[1], hi, 1);
    CHATTS_CVT_UBYTE(qe[2], lo, 2); CHATTS_CVT_UBYTE(qo[2], hi, 2);
    CHATTS_CVT_UBYTE(qe[3], lo, 3); CHATTS_CVT_UBYTE(qo[3], hi, 3);
    const uint32_t xhs[4] = {xh.x, xh.y, xh.z, xh.w}, xls[4] = {xl.x, xl.y, xl.z, xl.w};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f32x2 v = {fmaf(qe[b], sc, nsz), fmaf(qo[b], sc, nsz)};               // exact: fp16 scale x 5-bit integer
      const bf16x2_t w = __builtin_convertvector(v, bf16x2_t);             // round to nearest even: the bf16 weights (k, k+1)
      acc_h = __builtin_amdgcn_fdot2_f32_bf16(w, __builtin_bit_cast(bf16x2_t, xhs[b]), acc_h, false);
      acc_l = __builtin_amdgcn_fdot2_f32_bf16(w, __builtin_bit_cast(bf16x2_t, xls[b]), acc_l, false);
    }
  };
  eight(wv.x, xh0, xl0);
  eight(wv.y, xh1, xl1);
}

template <int ROWS, int UNR, int EPI, bool NORM>
__global__ __launch_bounds__(1024) void gemv4_ldsx_kernel(GemvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // x as packed bf16 pairs: [chunk][part][lane] 16-byte entries; part 0 / 1 = hi pairs 0-3 / 4-7 of the lane's 16 elements,
  // part 2 / 3 = the lo pairs (x = hi + lo to 2^-17: the bf16x2 split)
  u32x4* xs4 = reinterpret_cast<u32x4*>(smem);
  const int K = p.k;
  const int nchunks = (K + 1023) >> 10;
  float* red = reinterpret_cast<float*>(smem) + (size_t)nchunks * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x, nw = nthreads >> 6;

  float rstd = 1.f;
  if (NORM) {
    float ss = 0.f;
    for (int k4 = tid * 4; k4 < K; k4 += nthreads * 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.x + k4);
      ss = sumsq4(ss, v);
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    rstd = rsqrtf(t / (float)K + p.eps);
  }
  uint32_t* xs1 = reinterpret_cast<uint32_t*>(smem);
  for (int k4 = tid * 4; k4 < nchunks * 1024; k4 += nthreads * 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k4 < K) {
      v = *reinterpret_cast<const f32x4*>(p.x + k4);
      if (NORM) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.norm_w + k4);
        v.x = g.x * (v.x * rstd); v.y = g.y * (v.y * rstd); v.z = g.z * (v.z * rstd); v.w = g.w * (v.w * rstd);
      }
    }
    uint16_t h[4], l[4];
    split_bf16x2(v.x, h[0], l[0]); split_bf16x2(v.y, h[1], l[1]); split_bf16x2(v.z, h[2], l[2]); split_bf16x2(v.w, h[3], l[3]);
    const int chunk = k4 >> 10, within = k4 & 1023, ln = within >> 4, pair = (within & 15) >> 1;      // pair in {0, 2, 4, 6}
    const int base = ((chunk * 4 + (pair >> 2)) * 64 + ln) * 4 + (pair & 3);                            // dword index of the hi pair
    xs1[base] = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
    xs1[base + 1] = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
    xs1[base + 2 * 64 * 4] = (uint32_t)l[0] | ((uint32_t)l[1] << 16);
    xs1[base + 2 * 64 * 4 + 1] = (uint32_t)l[2] | ((uint32_t)l[3] << 16);
  }
  __syncthreads();

  const int ngroups = K / p.w4_group;
  const int gshift = 31 - __builtin_clz(p.w4_group);        // group sizes are powers of two (checked by the launcher)
  for (int task = blockIdx.x * nw + wave; task < p.tasks; task += gridDim.x * nw) {
    const uint8_t* wrow[ROWS];
    const float2* szrow[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      int row = task_row<ROWS, EPI>(task, r);
      if (row >= p.n) row = 0;
      wrow[r] = p.w4 + (size_t)row * p.ldw4 + lane * 8;
      szrow[r] = reinterpret_cast<const float2*>(p.w4_sz) + (size_t)row * ngroups;
    }
    float acc_h[ROWS], acc_l[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) { acc_h[r] = 0.f; acc_l[r] = 0.f; }
    for (int c = 0; c < nchunks; c += UNR) {
      u32x2 wv[UNR][ROWS];
      float2 sz[UNR][ROWS];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int k0 = ((c + u) << 10) + lane * 16;
        const bool ok = c + u < nchunks && k0 < K;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          wv[u][r] = (u32x2){0u, 0u};
          sz[u][r] = make_float2(0.f, 0.f);
          if (ok) {
            wv[u][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wrow[r] + ((c + u) << 9)));
            sz[u][r] = szrow[r][k0 >> gshift];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (c + u < nchunks) {
          const u32x4* xb = xs4 + (c + u) * 256 + lane;
          const u32x4 xh0 = xb[0], xh1 = xb[64], xl0 = xb[128], xl1 = xb[192];
#pragma unroll
          for (int r = 0; r < ROWS; ++r) dot16_int4(wv[u][r], sz[u][r].x, -sz[u][r].y, xh0, xh1, xl0, xl1, acc_h[r], acc_l[r]);
        }
      }
    }
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc_h[r] + acc_l[r]);
    gemv_epilogue<ROWS, EPI>(p, task, lane, acc);
  }
}

// ---- dispatch ------------------------------------------------------------------------------------
template <int ROWS, int UNR, int EPI>
static void launch_ldsx_norm(const GemvParams& p, bool norm, int blocks, int threads, size_t lds, hipStream_t s) {
  if (norm) hipLaunchKernelGGL((gemv_ldsx_kernel<ROWS, UNR, EPI, true>), dim3(blocks), dim3(threads), lds, s, p);
  else hipLaunchKernelGGL((gemv_ldsx_kernel<ROWS, UNR, EPI, false>), dim3(blocks), dim3(threads), lds, s, p);
}
template <int ROWS, int UNR>
static void launch_ldsx(const GemvParams& p, int epi, bool norm, int blocks, int threads, size_t lds, hipStream_t s) {
  switch (epi) {
    case CHATTS_EPI_RESID: launch_ldsx_norm<ROWS, UNR, CHATTS_EPI_RESID>(p, norm, blocks, threads, lds, s); break;
    case kEpiResidTp: hipLaunchKernelGGL((gemv_ldsx_kernel<ROWS, UNR, kEpiResidTp, false>), dim3(blocks), dim3(threads), lds, s, p); break;
    case CHATTS_EPI_SWIGLU: launch_ldsx_norm<ROWS, UNR, CHATTS_EPI_SWIGLU>(p, norm, blocks, threads, lds, s); break;
    default: launch_ldsx_norm<ROWS, UNR, CHATTS_EPI_NONE>(p, norm, blocks, threads, lds, s); break;
  }
}

template <int EPI>
static void launch4_norm(const GemvParams& p, bool norm, int blocks, int threads, size_t lds, hipStream_t s) {
  if (norm) hipLaunchKernelGGL((gemv4_ldsx_kernel<2, 4, EPI, true>), dim3(blocks), dim3(threads), lds, s, p);
  else hipLaunchKernelGGL((gemv4_ldsx_kernel<2, 4, EPI, false>), dim3(blocks), dim3(threads), lds, s, p);
}

template <int ROWS, int UNR, int EPI>
static void launch8_ru(const GemvParams& p, bool norm, int blocks, int threads, size_t lds, hipStream_t s) {
  if (p.w8_format == CHATTS_W8_INT8) {
    if (norm) hipLaunchKernelGGL((gemv8_ldsx_kernel<ROWS, UNR, EPI, true, true>), dim3(blocks), dim3(threads), lds, s, p);
    else hipLaunchKernelGGL((gemv8_ldsx_kernel<ROWS, UNR, EPI, false, true>), dim3(blocks), dim3(threads), lds, s, p);
    return;
  }
  if (norm) hipLaunchKernelGGL((gemv8_ldsx_kernel<ROWS, UNR, EPI, true>), dim3(blocks), dim3(threads), lds, s, p);
  else hipLaunchKernelGGL((gemv8_ldsx_kernel<ROWS, UNR, EPI, false>), dim3(blocks), dim3(threads), lds, s, p);
}
template <int EPI>
static void launch8_norm(const GemvParams& p, bool norm, int rows, int unr, int blocks, int threads, size_t lds, hipStream_t s) {
  if (rows == 4 && unr == 2) launch8_ru<4, 2, EPI>(p, norm, blocks, threads, lds, s);
  else if (rows == 4) launch8_ru<4, 1, EPI>(p, norm, blocks, threads, lds, s);
  else if (unr == 4) launch8_ru<2, 4, EPI>(p, norm, blocks, threads, lds, s);
  else launch8_ru<2, 2, EPI>(p, norm, blocks, threads, lds, s);
}

// Waves per workgroup (and workgroups per CU, 0 = whatever fits) of the bf16 GEMV for a shape: the sweep results of
// tools/gemv_sweep.py.  Also read by decode_mega.hip, whose fused RMSNorm reproduces this kernel's summation order.
void gemv_default_geometry(int n, int k, int epilogue, int cus, int* nw_out, int* occ_out) {
  const int swiglu = epilogue == CHATTS_EPI_SWIGLU;
  const int units = swiglu ? n / 2 : n;
  int nw_auto, occ_auto = 0;
  if (k > 8192) nw_auto = 16;                       // down_proj: 55 KB of x per workgroup -> amortise over 16 waves
  else if (units >= 65536) { nw_auto = 4; occ_auto = 2; }   // lm_head
  else if (n >= 16384) { nw_auto = 8; occ_auto = 2; }    // gate_up
  else if (n > 6000) nw_auto = 16;                  // qkv
  else nw_auto = 4;                                    // o_proj
  // Balanced layout (profiles/r2_gemv_sweep_balanced.txt): when the CUs divide the row-group tasks, one workgroup per CU whose
  // wave count divides the tasks per CU gives every wave the same number of tasks (no ragged last round): o_proj 11.5 ->
  // 11.2 us, gate_up 44.5 -> 44.1, down_proj 24.9 -> 24.5; lm_head (297 tasks per CU) keeps the 2-workgroup layout.
  const int tasks_r2 = (units + (swiglu ? 0 : 1)) / (swiglu ? 1 : 2);
  int balanced_nw = 0;
  if (units < 65536 && tasks_r2 % cus == 0) {
    const int per_cu = tasks_r2 / cus;
    for (int w = 16; w >= 5 && !balanced_nw; --w)
      if (per_cu % w == 0) balanced_nw = w;
  }
  if (balanced_nw) { nw_auto = balanced_nw; occ_auto = 1; }
  *nw_out = nw_auto;
  *occ_out = occ_auto;
}

// Waves per row group of the bf16 GEMV (1 = the whole-K kernel): a pure function of the shape, so that a host-side test can pin which
// shapes take which kernel (tests/test_host_logic.py: every TP = 1 projection of the supported models resolves to 1 - the K-split form
// sums a row in another order - and the TP = 8 shard shapes to > 1).
//   tasks = row groups (2 rows, or one gate / up pair), nchunks = K / 512, w_bytes = N K 2.
int gemv_pick_ksplit(int tasks, int nchunks, bool norm, double w_bytes, int cus) {
  if (nchunks < 4) return 1;
  int ks = 1;
  const double per_cu = (double)tasks / cus;
  const int rounds = (nchunks + 1) / 2;                   // chunk pairs of a row: a wave should keep at least one
  if (per_cu < 6.0) {
    // the row groups alone leave the chip nearly empty - fewer than ~6 waves per CU
    ks = (int)(12.0 / (per_cu > 0.25 ? per_cu : 0.25) + 0.999);
    if (ks > rounds) ks = rounds;
    if (ks > 8) ks = 8;
  } else if (norm && per_cu < 16.0 && w_bytes < 48e6) {
    // a SHARD-sized matrix whose row groups alone half-fill the chip (gate_up of a TP = 8 rank: 35 MB, 13.5 waves per CU): every wave
    // still walks K / 1024 dependent round trips; one chunk pair per wave measured 1.2 us per launch better (profiles/r4_tp_shard_step*).
    // The 48 MB bound keeps every TP = 1 shape (o_proj: 52 MB, 10 waves per CU; the 8B qkv: 50.3 MB) on the whole-K kernel; `norm`
    // restricts the rule to the column-parallel projections (qkv, gate_up: RMSNorm in the prologue) - the row-parallel ones (o_proj,
    // down_proj) must sum a row in the same order whether or not they carry the exchange (tp_reduce), which the K-split form does not do.
    ks = rounds > 8 ? 8 : rounds;
  }
  return ks;
}

int launch_gemv(const ChattsLinearArgs* a, hipStream_t s) {
  GemvParams p;
  p.tp = TpParams{};
  p.w = a->w; p.x = a->a; p.bias = a->bias; p.resid = a->resid; p.out = a->c;
  p.norm_w = a->norm_w; p.eps = a->norm_eps; p.n = a->n; p.k = a->k; p.ldw = a->ldw;
  p.w8 = a->w8; p.w8_scale = a->w8_scale; p.ldw8 = a->ldw8; p.w8_format = a->w8_format;
  p.w4 = a->w4; p.w4_sz = a->w4_sz; p.ldw4 = a->ldw4; p.w4_group = a->w4_group;
  const bool norm = a->norm_w != nullptr;
  const int cus = device_cus();
  int epilogue = a->epilogue;
  if (a->tp_reduce) {       // row-parallel projection: the sum over the ranks is formed inside this launch (kEpiResidTp)
    CHATTS_REQUIRE(a->epilogue == CHATTS_EPI_RESID && !a->bias && !a->norm_w && !a->w4, CHATTS_E_BADARG,
                   "gemv: tp_reduce needs EPI_RESID without bias / fused norm / 4-bit codes");
    CHATTS_REQUIRE(a->n <= tp_capacity(a->tp_reduce), CHATTS_E_SHAPE, "gemv: tp_reduce of %d rows exceeds the exchange buffer (%lld)", a->n,
                   (long long)tp_capacity(a->tp_reduce));
    epilogue = kEpiResidTp;
  }
  const int swiglu = a->epilogue == CHATTS_EPI_SWIGLU;
  const int units = swiglu ? a->n / 2 : a->n;
  const int nchunks = (a->k + 511) / 512;
  const size_t lds = (size_t)nchunks * 512 * 4 + 64 * 4;
  CHATTS_REQUIRE(lds <= 160 * 1024, CHATTS_E_SHAPE, "gemv: K=%d too large for the LDS-resident x", a->k);
  // Geometry from the sweep in tools/gemv_sweep.py (profiles/gemv_sweep_r1.log): 2 rows x 2 chunks in flight per lane
  // wins on every decode shape; what varies is how many waves share one staged copy of x and how many
  // workgroups a CU should hold.
  int rows = opt_get(OPT_GEMV_ROWS, 2);
  if (rows != 4) rows = 2;
  int unr = opt_get(OPT_GEMV_UNR, 2);
  if (unr != 4) unr = 2;
  int nw_auto, occ_auto;
  gemv_default_geometry(a->n, a->k, a->epilogue, cus, &nw_auto, &occ_auto);
  int nw = opt_get(OPT_GEMV_NW, nw_auto);
  if (nw < 1 || nw > 16) nw = 4;
  int occ = (int)((150 * 1024) / lds);          // workgroups per CU that fit in LDS
  const int wave_cap = 32 / nw;                  // 32 waves per CU
  if (occ > wave_cap) occ = wave_cap;
  if (occ_auto && occ > occ_auto) occ = occ_auto;
  if (occ < 1) occ = 1;
  occ = opt_get(OPT_GEMV_OCC, occ);
  const int upt = swiglu ? rows / 2 : rows;
  p.tasks = (units + upt - 1) / upt;
  int blocks = (p.tasks + nw - 1) / nw;
  if (blocks > cus * occ) blocks = cus * occ;
  blocks = opt_get(OPT_GEMV_BLOCKS, blocks);
  // several emulated ranks on ONE device (tests, tools/jobs): their exchange-carrying launches wait for each other, so all of them
  // must be resident at once - cap the grid (results do not depend on it: a row is always summed by one wave in the same order)
  if (a->tp_reduce) { const int cap = opt_get(OPT_TP_FUSE_BLOCKS, 0); if (cap > 0 && blocks > cap) blocks = cap; }
  if (blocks < 1) blocks = 1;
  const int threads = nw * 64;
  // CHATTS_GEMV_LDSPAD = p: declare 1/p of a CU's LDS, so that exactly p workgroups fit per CU and a grid of
  // cus * p workgroups is necessarily spread evenly (every CU streams the same number of bytes)
  size_t lds_launch = lds;
  const int pad = opt_get(OPT_GEMV_LDSPAD, 0);
  if (pad >= 1 && (size_t)(160 * 1024 / pad) / 16 * 16 > lds) lds_launch = (size_t)(160 * 1024 / pad) / 16 * 16;
  if (lds_launch > 64 * 1024 && lds <= 64 * 1024) lds_launch = 64 * 1024;      // default dynamic-LDS cap (still 2 per CU)
  if (a->w4 != nullptr) {                       // 4-bit codes: 2 rows x 4 chunks of 1024 elements (8-byte loads) in flight
    const int nchunks4 = (a->k + 1023) / 1024;
    const size_t lds4 = (size_t)nchunks4 * 1024 * 4 + 64 * 4;
    p.tasks = (units + (swiglu ? 1 : 2) - 1) / (swiglu ? 1 : 2);
    int blocks4 = (p.tasks + nw - 1) / nw;
    // ~5 VALU ops per weight make this kernel ALU-latency bound: it wants every wave slot the registers allow (89 VGPRs -> 5 per
    // SIMD = 20 per CU), i.e. two workgroups per CU, not the one of the bf16 layouts
    const int occ4 = opt_get(OPT_GEMV_OCC, 20 / nw >= 1 ? 20 / nw : 1);
    if (blocks4 > cus * occ4) blocks4 = cus * occ4;
    if (blocks4 < 1) blocks4 = 1;
    switch (a->epilogue) {
      case CHATTS_EPI_RESID: launch4_norm<CHATTS_EPI_RESID>(p, norm, blocks4, threads, lds4, s); break;
      case CHATTS_EPI_SWIGLU: launch4_norm<CHATTS_EPI_SWIGLU>(p, norm, blocks4, threads, lds4, s); break;
      default: launch4_norm<CHATTS_EPI_NONE>(p, norm, blocks4, threads, lds4, s); break;
    }
    CHATTS_CHECK_LAUNCH("gemv4_ldsx");
    return CHATTS_OK;
  }
  if (a->w8 != nullptr) {                       // 8-bit weights: rows x chunks of 1024 elements in flight per lane (GEMV8_ROWS x GEMV8_UNR)
    const int nchunks8 = (a->k + 1023) / 1024;
    const size_t lds8 = (size_t)nchunks8 * 1024 * 4 + 64 * 4;
    // per shape (kernel trace of the fp8 bench under each geometry, profiles/r6_gemv8_geometry.txt): the vocabulary projection wants 4 rows x 2
    // chunks in flight (778 MB: 144.8 -> 125.0 us), gate_up 4 rows x 1 (27.8 -> 26.9 us); the shorter launches keep 2 x 2 (4 rows halve
    // their task count: o / down 11.8 -> 14.5 us, qkv 10.3 -> 13.1)
    const int rows_auto = (units >= 65536 || (swiglu && a->n >= 16384)) ? 4 : 2, unr_auto = swiglu && a->n >= 16384 ? 1 : 2;
    int rows8 = opt_get(OPT_GEMV8_ROWS, rows_auto), unr8 = opt_get(OPT_GEMV8_UNR, unr_auto);
    if (rows8 != 4 || a->tp_reduce) rows8 = 2;          // (the exchange-carrying form pushes 2 rows per task: tp_push_task's lane map)
    if (rows8 == 4) unr8 = unr8 == 2 ? 2 : 1;
    else if (unr8 != 4) unr8 = 2;
    const int upt8 = swiglu ? rows8 / 2 : rows8;
    p.tasks = (units + upt8 - 1) / upt8;
    const int nw8 = opt_get(OPT_GEMV8_NW, nw);
    int occ8 = (int)((150 * 1024) / lds8);
    if (occ8 > 32 / nw8) occ8 = 32 / nw8;
    if (occ_auto && occ8 > occ_auto) occ8 = occ_auto;
    if (occ8 < 1) occ8 = 1;
    occ8 = opt_get(OPT_GEMV8_OCC, opt_get(OPT_GEMV_OCC, occ8));
    int blocks8 = (p.tasks + nw8 - 1) / nw8;
    if (blocks8 > cus * occ8) blocks8 = cus * occ8;
    if (a->tp_reduce) { const int cap = opt_get(OPT_TP_FUSE_BLOCKS, 0); if (cap > 0 && blocks8 > cap) blocks8 = cap; }
    if (blocks8 < 1) blocks8 = 1;
    const int threads8 = (nw8 >= 1 && nw8 <= 16 ? nw8 : nw) * 64;
    if (a->tp_reduce) p.tp = tp_issue(a->tp_reduce, false);      // (after the last check that can refuse the call: one issue per launch)
    switch (epilogue) {
      case kEpiResidTp: launch8_norm<kEpiResidTp>(p, false, rows8, unr8, blocks8, threads8, lds8, s); break;
      case CHATTS_EPI_RESID: launch8_norm<CHATTS_EPI_RESID>(p, norm, rows8, unr8, blocks8, threads8, lds8, s); break;
      case CHATTS_EPI_SWIGLU: launch8_norm<CHATTS_EPI_SWIGLU>(p, norm, rows8, unr8, blocks8, threads8, lds8, s); break;
      default: launch8_norm<CHATTS_EPI_NONE>(p, norm, rows8, unr8, blocks8, threads8, lds8, s); break;
    }
    CHATTS_CHECK_LAUNCH("gemv8_ldsx");
    return CHATTS_OK;
  }
  // K-split form (see gemv_ksplit_kernel): when the row groups alone leave the chip nearly empty - fewer than ~6 waves per CU - and a
  // row is long enough to split.  Every TP = 1 shape of the supported models has >= 10 waves per CU and keeps the whole-K kernel.
  p.ks = 1;
  if (!a->tp_reduce && rows == 2 && unr == 2 && nchunks >= 4) {
    int ks = gemv_pick_ksplit(p.tasks, nchunks, norm, (double)a->n * a->k * 2.0, cus);
    ks = opt_get(OPT_GEMV_KS, ks);
    if (ks > 16) ks = 16;
    if (ks > 1) {
      int groups = 16 / ks;
      if (groups < 1) groups = 1;
      const int nwk = groups * ks;
      int blocksk = (p.tasks + groups - 1) / groups;
      const int occk = (int)((150 * 1024) / lds) < 32 / nwk ? (int)((150 * 1024) / lds) : 32 / nwk;
      if (blocksk > cus * (occk < 1 ? 1 : occk)) blocksk = cus * (occk < 1 ? 1 : occk);
      p.ks = ks;
      const dim3 g(blocksk), b(nwk * 64);
      switch (epilogue) {
        case CHATTS_EPI_RESID:
          if (norm) hipLaunchKernelGGL((gemv_ksplit_kernel<CHATTS_EPI_RESID, true>), g, b, lds, s, p);
          else hipLaunchKernelGGL((gemv_ksplit_kernel<CHATTS_EPI_RESID, false>), g, b, lds, s, p);
          break;
        case CHATTS_EPI_SWIGLU:
          if (norm) hipLaunchKernelGGL((gemv_ksplit_kernel<CHATTS_EPI_SWIGLU, true>), g, b, lds, s, p);
          else hipLaunchKernelGGL((gemv_ksplit_kernel<CHATTS_EPI_SWIGLU, false>), g, b, lds, s, p);
          break;
        default:
          if (norm) hipLaunchKernelGGL((gemv_ksplit_kernel<CHATTS_EPI_NONE, true>), g, b, lds, s, p);
          else hipLaunchKernelGGL((gemv_ksplit_kernel<CHATTS_EPI_NONE, false>), g, b, lds, s, p);
          break;
      }
      CHATTS_CHECK_LAUNCH("gemv_ksplit");
      return CHATTS_OK;
    }
  }
  if (a->tp_reduce) p.tp = tp_issue(a->tp_reduce, false);
  if (rows == 4 && unr == 2) launch_ldsx<4, 2>(p, epilogue, norm, blocks, threads, lds_launch, s);
  else if (rows == 4) launch_ldsx<4, 4>(p, epilogue, norm, blocks, threads, lds_launch, s);
  else if (unr == 2) launch_ldsx<2, 2>(p, epilogue, norm, blocks, threads, lds_launch, s);
  else launch_ldsx<2, 4>(p, epilogue, norm, blocks, threads, lds_launch, s);
  CHATTS_CHECK_LAUNCH("gemv_ldsx");
  return CHATTS_OK;
}

}  // namespace chatts

// host-only query (no device touched): the K-split factor the bf16 decode GEMV picks for a shape on this device - see gemv_pick_ksplit
extern "C" int chatts_gemv_ksplit(int n, int k, int epilogue, int has_norm) {
  if (n <= 0 || k <= 0) return -1;
  const int swiglu = epilogue == CHATTS_EPI_SWIGLU;
  const int units = swiglu ? n / 2 : n, upt = swiglu ? 1 : 2;
  const int cus = chatts::device_cus() > 0 ? chatts::device_cus() : 256;
  return chatts::gemv_pick_ksplit((units + upt - 1) / upt, (k + 511) / 512, has_norm != 0, (double)n * k * 2.0, cus);
}

