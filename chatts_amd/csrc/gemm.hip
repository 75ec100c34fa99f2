// gemm.hip - prefill / TS-encoder projections:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
// A float32 activations, W bfloat16 weights (HF nn.Linear layout, K contiguous for both operands).
//
// Precision scheme "bf16x2": the reference path we must match is float32 (1e-3 relative on logits,
// identical greedy tokens).  Weights are bf16 by definition, so only A needs care: each f32 element
// is split while it is staged into LDS,  a = hi + lo,  hi = bf16(a), lo = bf16(a - hi), and the
// wave issues two v_mfma_f32_16x16x32_bf16 per fragment pair (hi.W, lo.W) into ONE f32 accumulator.
// The W tile is staged and read once; products are exact (8b x 8b mantissas), accumulation is f32.
//
// Geometry: 256 threads = 4 waves (WM x WN), tile BM x 128, BK = 32, double-buffered LDS,
// register-staged global loads issued one K-step ahead, one barrier per K-step.
// LDS rows are 64 B (32 bf16); 16-byte chunk c of row r lives at chunk c ^ (((r>>3)&1)<<1), which
// makes every ds_read_b128 fragment read conflict-free (checked exhaustively over the 4 lane groups).
// Split-K (grid.z) fills the 256 CUs when M is small; partials go to a workspace and a second kernel
// applies the epilogue.
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
};

__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu_g(float x) { return x / (1.0f + expf(-x)); }

__device__ __forceinline__ int lds_off(int r, int c) {   // byte offset of 16-byte chunk c of row r
  return r * 64 + ((c ^ (((r >> 3) & 1) << 1)) << 4);
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

  // ---- epilogue.  C/D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg -----------
  const int ccol = lane & 15, crow0 = (lane >> 4) * 4;
  if (!p.direct) {
    float* ws = p.c + (size_t)blockIdx.z * p.m * p.n;
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
            if (row < p.m) p.c[(size_t)row * p.ldc + ocol] = silu_g(acc[i][j][r] * sg + bg) * (acc[i][j + 1][r] * su + bu);
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
        p.c[(size_t)row * p.ldc + col] = v;
      }
    }
}

// Sum split-K partials in a fixed order and apply the epilogue.  One thread per output element.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ ws, int sk, int m, int n,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ resid, float* __restrict__ c,
                                                             int ldc, int epilogue, const float* __restrict__ scale) {
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
    c[(size_t)row * ldc + col] = silu_g(g) * u;
    return;
  }
  float v = 0.f;
  for (int s = 0; s < sk; ++s) v += ws[s * plane + (size_t)row * n + col];
  if (scale) v *= scale[col];
  if (bias) v += bias[col];
  if (epilogue == CHATTS_EPI_GELU) v = gelu_erf_f(v);
  if (epilogue == CHATTS_EPI_RESID) v = resid[(size_t)row * ldc + col] + v;
  c[(size_t)row * ldc + col] = v;
}

static int gemm_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

static void pick_geometry(int m, int n, int k, int& bm, int& sk) {
  bm = m > 64 ? 128 : (m > 32 ? 64 : (m > 16 ? 32 : 16));
  const int force_bm = gemm_env_int("CHATTS_GEMM_BM", 0);          // tuning / tests only
  if (force_bm == 16 || force_bm == 32 || force_bm == 64 || force_bm == 128) bm = force_bm;
  // short K (no split-K possible, e.g. the first TS-MLP layer, K = 288) and few tiles: smaller M-tiles fill more CUs
  if (!force_bm && k < 512 && ((m + bm - 1) / bm) * ((n + 127) / 128) * 4 <= device_cus() && bm > 32) bm = 32;
  const int tiles = ((m + bm - 1) / bm) * ((n + 127) / 128);
  // Split-K so that the workgroups fill whole "rounds" of the resident slots (3 workgroups of 48 KB LDS per CU):
  // efficiency of a launch = blocks / (slots * ceil(blocks / slots)); split-K costs a partials round trip + an
  // epilogue launch, hence the small penalty.  (tools/gemm_sweep.py: qkv @ M=798 wants 3, o/down 2, gate_up 1.)
  const int slots = 3 * device_cus();
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
  const int force_sk = gemm_env_int("CHATTS_GEMM_SK", 0);
  if (force_sk > 0 && force_sk <= 16 && k / force_sk >= 32) sk = force_sk;
}

static int k_per_split(int k, int sk) {
  int kps = (k + sk - 1) / sk;
  return ((kps + 31) / 32) * 32;
}

size_t gemm_workspace(int m, int n, int k) {
  int bm, sk;
  pick_geometry(m, n, k, bm, sk);
  return sk > 1 ? (size_t)sk * m * n * sizeof(float) : 0;
}

int launch_gemm(const ChattsLinearArgs* a, hipStream_t s) {
  int bm, sk;
  pick_geometry(a->m, a->n, a->k, bm, sk);
  const int kps = k_per_split(a->k, sk);
  sk = (a->k + kps - 1) / kps;
  GemmParams p;
  p.a = a->a; p.w = a->w; p.bias = a->bias; p.resid = a->resid;
  p.m = a->m; p.n = a->n; p.k = a->k; p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc;
  p.epilogue = a->epilogue; p.k_per_split = kps; p.direct = sk == 1;
  p.w8 = a->w8; p.w8_scale = a->w8_scale; p.ldw8 = a->ldw8;
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
  if (a->w8) {
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
  if (sk > 1) {
    const int ncols = a->epilogue == CHATTS_EPI_SWIGLU ? a->n / 2 : a->n;
    const size_t total = (size_t)a->m * ncols;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float*>(a->workspace), sk, a->m, a->n, a->bias, a->resid, a->c,
                       a->ldc, a->epilogue, a->w8 ? a->w8_scale : nullptr);
    CHATTS_CHECK_LAUNCH("splitk_epilogue");
  }
  return CHATTS_OK;
}

}  // namespace chatts
