// gemv.hip - batch-1 decode projections: y[N] = epilogue(W[N,K] . x[K]),  W bf16 streamed from HBM.
//
// This kernel sets generated tokens/s: every decode step streams all decoder weights once
// (27.98 GB for ChatTS-14B), so it is HBM-bound by construction; the design follows the guide's
// "GEMV / M<=16 decode weights" rule: weights go straight to VGPRs as non-temporal 16-byte loads
// with several loads in flight per lane, no LDS round trip for the streamed operand.  The small,
// re-used operand x (20-55 KB f32) is staged once per workgroup in LDS, permuted so that each lane's
// two 16-byte reads per chunk are lane-linear (conflict-free ds_read_b128).
// Arithmetic: bf16 -> f32 widening is exact and x is f32, so every product is exact in f32 FMA;
// only the summation order differs from the CPU oracle (parity budget: 1e-3 relative on logits).
//
// Fusions: RMSNorm of x in the prologue (Qwen2RMSNorm.forward), bias, residual add, SwiGLU.
#include "common.h"

namespace chatts {

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
};

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

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// ROWS weight rows per task (SWIGLU: ROWS/2 gate/up pairs); WK = 1: each wave owns a task and the
// whole K; WK = 4: the workgroup's 4 waves share one task and take K-chunks round-robin.
template <int ROWS, int WK, int EPI, bool NORM>
__global__ __launch_bounds__(256) void gemv_kernel(GemvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* xs4 = reinterpret_cast<f32x4*>(smem);            // permuted x: [chunk][half][lane] float4
  float* red = reinterpret_cast<float*>(smem) + (size_t)((p.k + 511) / 512) * 512;  // 4*ROWS + 8 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.k;
  const int nchunks = (K + 511) >> 9;

  // ---- stage x (optionally RMS-normalised) into LDS, permuted ------------------------------------
  float ss = 0.f;
  if (NORM) {
    for (int k4 = tid * 4; k4 < K; k4 += 1024) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.x + k4);
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = block_sum<4>(ss, red);
  }
  const float rstd = NORM ? rsqrtf(ss / (float)K + p.eps) : 1.f;
  for (int k4 = tid * 4; k4 < nchunks * 512; k4 += 1024) {
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

  const int task_stride = WK == 1 ? gridDim.x * 4 : gridDim.x;
  for (int task = WK == 1 ? blockIdx.x * 4 + wave : blockIdx.x; task < p.tasks; task += task_stride) {
    const uint16_t* wrow[ROWS];
    bool valid[ROWS];
    int rowid[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      int row;
      if (EPI == CHATTS_EPI_SWIGLU) {
        const int unit = task * (ROWS / 2) + (r >> 1);
        row = (unit >> 4) * 32 + (unit & 15) + (r & 1) * 16;
      } else {
        row = task * ROWS + r;
      }
      rowid[r] = row;
      valid[r] = row < p.n;
      wrow[r] = p.w + (size_t)(valid[r] ? row : 0) * p.ldw;
    }
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;

    // two chunks (2 x ROWS 16-byte loads) in flight per lane per iteration
    const int cstep = WK == 1 ? 1 : 4;
    for (int c = WK == 1 ? 0 : wave; c < nchunks; c += 2 * cstep) {
      const int c1 = c + cstep;
      const int k0 = (c << 9) + lane * 8, k1 = (c1 << 9) + lane * 8;
      const bool ok0 = k0 < K, ok1 = c1 < nchunks && k1 < K;
      u32x4 w0[ROWS], w1[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        w0[r] = (u32x4){0u, 0u, 0u, 0u};
        w1[r] = (u32x4){0u, 0u, 0u, 0u};
        if (ok0) w0[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + k0));
      }
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
        if (ok1) w1[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow[r] + k1));
      const f32x4 xa0 = xs4[c * 128 + lane], xb0 = xs4[c * 128 + 64 + lane];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) acc[r] = dot8(w0[r], xa0, xb0, acc[r]);
      if (c1 < nchunks) {
        const f32x4 xa1 = xs4[c1 * 128 + lane], xb1 = xs4[c1 * 128 + 64 + lane];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[r] = dot8(w1[r], xa1, xb1, acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]);

    if (WK == 4) {   // combine the 4 K-slices in a fixed order (deterministic)
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) red[8 + wave * ROWS + r] = acc[r];
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
        acc[r] = (red[8 + r] + red[8 + ROWS + r]) + (red[8 + 2 * ROWS + r] + red[8 + 3 * ROWS + r]);
      __syncthreads();
      if (wave != 0) continue;
    }

    // ---- epilogue: lane r finishes row r ------------------------------------------------------
    if (EPI == CHATTS_EPI_SWIGLU) {
#pragma unroll
      for (int u = 0; u < ROWS / 2; ++u) {
        if (lane == u && valid[2 * u]) {
          float g = acc[2 * u], v = acc[2 * u + 1];
          if (p.bias) { g += p.bias[rowid[2 * u]]; v += p.bias[rowid[2 * u + 1]]; }
          const int unit = task * (ROWS / 2) + u;
          p.out[unit] = silu_f(g) * v;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        if (lane == r && valid[r]) {
          float v = acc[r];
          if (p.bias) v += p.bias[rowid[r]];
          if (EPI == CHATTS_EPI_RESID) v = p.resid[rowid[r]] + v;
          p.out[rowid[r]] = v;
        }
      }
    }
  }
}

template <int ROWS, int WK, int EPI>
static int launch_norm(const GemvParams& p, bool norm, int blocks, size_t lds, hipStream_t s) {
  if (norm)
    hipLaunchKernelGGL((gemv_kernel<ROWS, WK, EPI, true>), dim3(blocks), dim3(256), lds, s, p);
  else
    hipLaunchKernelGGL((gemv_kernel<ROWS, WK, EPI, false>), dim3(blocks), dim3(256), lds, s, p);
  return 0;
}

template <int ROWS, int WK>
static int launch_epi(const GemvParams& p, int epi, bool norm, int blocks, size_t lds, hipStream_t s) {
  switch (epi) {
    case CHATTS_EPI_NONE: return launch_norm<ROWS, WK, CHATTS_EPI_NONE>(p, norm, blocks, lds, s);
    case CHATTS_EPI_RESID: return launch_norm<ROWS, WK, CHATTS_EPI_RESID>(p, norm, blocks, lds, s);
    case CHATTS_EPI_SWIGLU: return launch_norm<ROWS, WK, CHATTS_EPI_SWIGLU>(p, norm, blocks, lds, s);
  }
  return -1;
}

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// Geometry: keep >= ~8 waves per CU streaming.  Large N: one task (4 rows, whole K) per wave, grid-stride.
// Small N (o_proj/down_proj, N = H): the workgroup's waves split K so that N/2 workgroup tasks exist.
int launch_gemv(const ChattsLinearArgs* a, hipStream_t s) {
  GemvParams p;
  p.w = a->w; p.x = a->a; p.bias = a->bias; p.resid = a->resid; p.out = a->c;
  p.norm_w = a->norm_w; p.eps = a->norm_eps; p.n = a->n; p.k = a->k; p.ldw = a->ldw;
  const bool norm = a->norm_w != nullptr;
  const int cus = device_cus();
  const size_t lds = (size_t)((a->k + 511) / 512) * 512 * 4 + 64 * 4;
  int occ = (int)((150 * 1024) / lds);
  if (occ > 8) occ = 8;
  if (occ < 1) occ = 1;
  occ = env_int("CHATTS_GEMV_OCC", occ);
  const int units = a->epilogue == CHATTS_EPI_SWIGLU ? a->n / 2 : a->n;
  // split K inside the workgroup when there are too few rows to give every resident wave a 4-row task
  int wk = (units / 4) < cus * occ * 4 ? 4 : 1;
  wk = env_int("CHATTS_GEMV_WK", wk);
  int rows = wk == 4 ? 2 : 4;
  rows = env_int("CHATTS_GEMV_ROWS", rows);
  const int upt = a->epilogue == CHATTS_EPI_SWIGLU ? rows / 2 : rows;   // units per task
  p.tasks = (units + upt - 1) / upt;
  int blocks = wk == 1 ? (p.tasks + 3) / 4 : p.tasks;
  if (blocks > cus * occ) blocks = cus * occ;
  if (blocks < 1) blocks = 1;
  int rc;
  if (rows == 4 && wk == 1) rc = launch_epi<4, 1>(p, a->epilogue, norm, blocks, lds, s);
  else if (rows == 2 && wk == 1) rc = launch_epi<2, 1>(p, a->epilogue, norm, blocks, lds, s);
  else if (rows == 4 && wk == 4) rc = launch_epi<4, 4>(p, a->epilogue, norm, blocks, lds, s);
  else rc = launch_epi<2, 4>(p, a->epilogue, norm, blocks, lds, s);
  if (rc != 0) {
    set_error("gemv: unsupported epilogue %d", a->epilogue);
    return CHATTS_E_BADARG;
  }
  CHATTS_CHECK_LAUNCH("gemv");
  return CHATTS_OK;
}

}  // namespace chatts
