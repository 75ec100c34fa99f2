// gemv.hip - batch-1 decode projections: y[N] = epilogue(W[N,K] . x[K]),  W bf16 streamed from HBM.
//
// This kernel sets generated tokens/s: every decode step streams all decoder weights once
// (27.98 GB for ChatTS-14B), so it is HBM-bound by construction.  It follows the guide's
// "GEMV / M<=16 decode weights" rule: weights go straight to VGPRs as non-temporal 16-byte loads,
// many loads in flight per lane, no LDS round trip for the streamed operand.
//
// Two geometries:
//  * gemv_regx_kernel (N up to a few 10k rows): the workgroup's 4 waves split K in 512-element chunks
//    (chunk c -> wave c & 3) and each wave keeps ITS slice of x in registers for the whole launch
//    (<= 7 chunks x 8 floats), so there is no LDS staging, no prologue barrier for the plain variants,
//    and all NCH x ROWS 16-byte weight loads of a row group are issued back to back (12-14 per lane).
//    The next row group's loads are issued before the cross-wave reduction of the current one.
//    Partial sums meet in LDS (one barrier per row group, double-buffered slots) and are added in a
//    fixed order -> run-to-run deterministic.
//  * gemv_ldsx_kernel (lm_head, N = 152k): one wave per 4-row group over the whole K, x staged once per
//    workgroup in LDS, permuted so each lane's two 16-byte reads per chunk are lane-linear.
// Arithmetic: bf16 -> f32 widening is exact and x is f32, so every product is exact in the f32 FMA;
// only the summation order differs from the CPU oracle (parity budget: 1e-3 relative on logits).
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

template <int ROWS, int EPI>
__device__ __forceinline__ int task_row(int task, int r) {
  if (EPI == CHATTS_EPI_SWIGLU) {   // gate/up interleaved in blocks of 16 rows: unit u -> rows g(u), g(u)+16
    const int unit = task * (ROWS / 2) + (r >> 1);
    return (unit >> 4) * 32 + (unit & 15) + (r & 1) * 16;
  }
  return task * ROWS + r;
}

// lane r (< ROWS, or < ROWS/2 for SwiGLU) of the finishing wave writes row r
template <int ROWS, int EPI>
__device__ __forceinline__ void gemv_epilogue(const GemvParams& p, int task, int lane, const float (&acc)[ROWS]) {
  if (EPI == CHATTS_EPI_SWIGLU) {
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
// x in registers, K split over the 4 waves.  NCH = max chunks per wave (K <= 2048 * NCH).
// ------------------------------------------------------------------------------------------------
template <int NCH, int ROWS, int EPI, bool NORM>
__global__ __launch_bounds__(256) void gemv_regx_kernel(GemvParams p) {
  __shared__ float red[2][4][ROWS];
  __shared__ float ssq[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.k;

  // this lane's x elements: chunk c = wave + 4 i, k = c*512 + lane*8 .. +7
  f32x4 xa[NCH], xb[NCH];
  bool okc[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int k0 = ((wave + 4 * i) << 9) + lane * 8;
    okc[i] = k0 < K;
    xa[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    xb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (okc[i]) {
      xa[i] = *reinterpret_cast<const f32x4*>(p.x + k0);
      xb[i] = *reinterpret_cast<const f32x4*>(p.x + k0 + 4);
    }
  }

  u32x4 wr[NCH][ROWS];
  auto issue = [&](int task) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      int row = task_row<ROWS, EPI>(task, r);
      if (row >= p.n) row = 0;
      const uint16_t* wrow = p.w + (size_t)row * p.ldw + lane * 8;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        wr[i][r] = (u32x4){0u, 0u, 0u, 0u};
        if (okc[i]) wr[i][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wrow + ((wave + 4 * i) << 9)));
      }
    }
  };

  int task = blockIdx.x;
  if (task < p.tasks) issue(task);   // weights of the first row group fly while x is normalised

  if (NORM) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      ss += xa[i].x * xa[i].x + xa[i].y * xa[i].y + xa[i].z * xa[i].z + xa[i].w * xa[i].w +
            xb[i].x * xb[i].x + xb[i].y * xb[i].y + xb[i].z * xb[i].z + xb[i].w * xb[i].w;
    ss = wave_sum(ss);
    if (lane == 0) ssq[wave] = ss;
    __syncthreads();
    const float rstd = rsqrtf(((ssq[0] + ssq[1]) + (ssq[2] + ssq[3])) / (float)K + p.eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (okc[i]) {
        const int k0 = ((wave + 4 * i) << 9) + lane * 8;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(p.norm_w + k0);
        const f32x4 gb = *reinterpret_cast<const f32x4*>(p.norm_w + k0 + 4);
        xa[i].x = ga.x * (xa[i].x * rstd); xa[i].y = ga.y * (xa[i].y * rstd);
        xa[i].z = ga.z * (xa[i].z * rstd); xa[i].w = ga.w * (xa[i].w * rstd);
        xb[i].x = gb.x * (xb[i].x * rstd); xb[i].y = gb.y * (xb[i].y * rstd);
        xb[i].z = gb.z * (xb[i].z * rstd); xb[i].w = gb.w * (xb[i].w * rstd);
      }
    }
  }

  int parity = 0;
  for (; task < p.tasks; task += gridDim.x) {
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) a = dot8(wr[i][r], xa[i], xb[i], a);
      acc[r] = a;
    }
    const int next = task + gridDim.x;
    if (next < p.tasks) issue(next);            // next row group's loads overlap the reduction below
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) red[parity][wave][r] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < ROWS; ++r)
        acc[r] = (red[parity][0][r] + red[parity][1][r]) + (red[parity][2][r] + red[parity][3][r]);
      gemv_epilogue<ROWS, EPI>(p, task, lane, acc);
    }
    parity ^= 1;
  }
}

// ------------------------------------------------------------------------------------------------
// x in LDS, one wave per row group over the whole K (large N).
// ------------------------------------------------------------------------------------------------
template <int ROWS, int EPI, bool NORM>
__global__ __launch_bounds__(256) void gemv_ldsx_kernel(GemvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* xs4 = reinterpret_cast<f32x4*>(smem);            // permuted x: [chunk][half][lane] float4
  float* red = reinterpret_cast<float*>(smem) + (size_t)((p.k + 511) / 512) * 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.k;
  const int nchunks = (K + 511) >> 9;

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

  for (int task = blockIdx.x * 4 + wave; task < p.tasks; task += gridDim.x * 4) {
    const uint16_t* wrow[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      int row = task_row<ROWS, EPI>(task, r);
      if (row >= p.n) row = 0;
      wrow[r] = p.w + (size_t)row * p.ldw;
    }
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
    for (int c = 0; c < nchunks; c += 2) {     // two chunks (2 x ROWS 16-byte loads) in flight per lane
      const int c1 = c + 1;
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
    gemv_epilogue<ROWS, EPI>(p, task, lane, acc);
  }
}

// ---- dispatch ------------------------------------------------------------------------------------
template <int NCH, int ROWS, int EPI>
static void launch_regx_norm(const GemvParams& p, bool norm, int blocks, hipStream_t s) {
  if (norm) hipLaunchKernelGGL((gemv_regx_kernel<NCH, ROWS, EPI, true>), dim3(blocks), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemv_regx_kernel<NCH, ROWS, EPI, false>), dim3(blocks), dim3(256), 0, s, p);
}
template <int NCH, int ROWS>
static void launch_regx(const GemvParams& p, int epi, bool norm, int blocks, hipStream_t s) {
  switch (epi) {
    case CHATTS_EPI_RESID: launch_regx_norm<NCH, ROWS, CHATTS_EPI_RESID>(p, norm, blocks, s); break;
    case CHATTS_EPI_SWIGLU: launch_regx_norm<NCH, ROWS, CHATTS_EPI_SWIGLU>(p, norm, blocks, s); break;
    default: launch_regx_norm<NCH, ROWS, CHATTS_EPI_NONE>(p, norm, blocks, s); break;
  }
}
template <int EPI>
static void launch_ldsx_norm(const GemvParams& p, bool norm, int blocks, size_t lds, hipStream_t s) {
  if (norm) hipLaunchKernelGGL((gemv_ldsx_kernel<4, EPI, true>), dim3(blocks), dim3(256), lds, s, p);
  else hipLaunchKernelGGL((gemv_ldsx_kernel<4, EPI, false>), dim3(blocks), dim3(256), lds, s, p);
}

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

int launch_gemv(const ChattsLinearArgs* a, hipStream_t s) {
  GemvParams p;
  p.w = a->w; p.x = a->a; p.bias = a->bias; p.resid = a->resid; p.out = a->c;
  p.norm_w = a->norm_w; p.eps = a->norm_eps; p.n = a->n; p.k = a->k; p.ldw = a->ldw;
  const bool norm = a->norm_w != nullptr;
  const int cus = device_cus();
  const int swiglu = a->epilogue == CHATTS_EPI_SWIGLU;
  const int units = swiglu ? a->n / 2 : a->n;
  const int nchunks = (a->k + 511) / 512;
  const int nch = (nchunks + 3) / 4;                  // chunks per wave in the K-split geometry
  // 0 = auto, 1 = force x-in-LDS, 2 = force x-in-registers (tuning / tests)
  const int force = env_int("CHATTS_GEMV_GEOM", 0);
  const bool regx = force == 2 || (force == 0 && nch <= 7 && units < 65536);
  if (regx && nch <= 7) {
    const int rows = nch <= 4 ? 4 : 2;
    const int upt = swiglu ? rows / 2 : rows;
    p.tasks = (units + upt - 1) / upt;
    // resident workgroups per CU (VGPR-bound: ~88 regs for NCH<=4, ~130 for NCH=7)
    int occ = env_int("CHATTS_GEMV_OCC", nch <= 4 ? 5 : 3);
    const int maxb = cus * occ;
    const int iters = (p.tasks + maxb - 1) / maxb;
    int blocks = (p.tasks + iters - 1) / iters;       // balanced: every workgroup runs `iters` (or iters-1) tasks
    if (blocks < 1) blocks = 1;
    if (nch <= 1) launch_regx<1, 4>(p, a->epilogue, norm, blocks, s);
    else if (nch == 2) launch_regx<2, 4>(p, a->epilogue, norm, blocks, s);
    else if (nch == 3) launch_regx<3, 4>(p, a->epilogue, norm, blocks, s);
    else if (nch == 4) launch_regx<4, 4>(p, a->epilogue, norm, blocks, s);
    else launch_regx<7, 2>(p, a->epilogue, norm, blocks, s);
    CHATTS_CHECK_LAUNCH("gemv_regx");
    return CHATTS_OK;
  }
  const size_t lds = (size_t)nchunks * 512 * 4 + 64 * 4;
  CHATTS_REQUIRE(lds <= 160 * 1024, CHATTS_E_SHAPE, "gemv: K=%d too large for the LDS-resident x", a->k);
  int occ = (int)((150 * 1024) / lds);
  if (occ > 8) occ = 8;
  if (occ < 1) occ = 1;
  occ = env_int("CHATTS_GEMV_OCC", occ);
  const int upt = swiglu ? 2 : 4;
  p.tasks = (units + upt - 1) / upt;
  int blocks = (p.tasks + 3) / 4;
  if (blocks > cus * occ) blocks = cus * occ;
  if (blocks < 1) blocks = 1;
  switch (a->epilogue) {
    case CHATTS_EPI_RESID: launch_ldsx_norm<CHATTS_EPI_RESID>(p, norm, blocks, lds, s); break;
    case CHATTS_EPI_SWIGLU: launch_ldsx_norm<CHATTS_EPI_SWIGLU>(p, norm, blocks, lds, s); break;
    default: launch_ldsx_norm<CHATTS_EPI_NONE>(p, norm, blocks, lds, s); break;
  }
  CHATTS_CHECK_LAUNCH("gemv_ldsx");
  return CHATTS_OK;
}

}  // namespace chatts
