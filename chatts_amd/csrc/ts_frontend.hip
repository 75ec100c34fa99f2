// ts_frontend.hip - sp-masked patchify of the padded [N, 2*Lmax] (value, mask) series tensor.
// Replaces the per-series Python loop of TimeSeriesEmbedding.forward
// (NetManAIOps/ChatTS chatts/vllm/chatts_vllm.py:93-183): there, 2 .item() host syncs and ~10 tiny
// torch ops per series; here one wave per series for the mask reduction and one wave per patch for
// the feature row, no host sync.
#include "common.h"
#include "gemm_common.h"

namespace chatts {

// One wave per series: valid_len = sum(long(mask)) (chatts_vllm.py:98-99), patch_cnt = ceil(vl/ps).
// The (value, mask) pairs are read as float2 -> each lane streams 8 B, 512 B per wave instruction.
__global__ __launch_bounds__(256) void ts_patch_cnt_kernel(const float* __restrict__ series, int n_series,
                                                          int lmax, int patch, int32_t* __restrict__ vl_out,
                                                          int64_t* __restrict__ pc_out) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n_series) return;
  const float2* row = reinterpret_cast<const float2*>(series) + (size_t)wave * lmax;
  int cnt = 0;
  for (int t = lane; t < lmax; t += 64) cnt += (int)(long long)row[t].y;  // .long(): truncate toward zero
  cnt = wave_sum_i(cnt);
  if (lane == 0) {
    if (vl_out) vl_out[wave] = cnt;
    if (pc_out) pc_out[wave] = (int64_t)((cnt + patch - 1) / patch);
  }
}

// One wave per output patch row.  Lanes 0..ps-1 own the ps values of the patch; in mode 1 every lane
// then copies position-embedding elements (ps*emb contiguous floats in the output row).
// The series owning patch row `p` is found by a binary search over row_off (N+1 ints, L2/L1 resident).
__global__ __launch_bounds__(256) void ts_patchify_kernel(ChattsPatchifyArgs a) {
  const int p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (p >= a.total_patches) return;
  int lo = 0, hi = a.n_series;  // largest s with row_off[s] <= p
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.row_off[mid] <= p) lo = mid; else hi = mid;
  }
  const int s = lo;
  const int vl = a.valid_len[s];
  const int pp = p - a.row_off[s];          // patch index inside the series
  const int ps = a.patch_size;
  const float2* row = reinterpret_cast<const float2*>(a.series) + (size_t)s * a.lmax;
  // the row goes out as float32, or (out_hi / out_lo set) directly as the bf16 hi / lo planes of that float32 value - the
  // operand format of the first MLP GEMM (same split as chatts_split_bf16x2: hi = bf16(v), lo = bf16(v - hi))
  const size_t obase = (size_t)p * a.ld_out;
  auto put = [&](int col, float v) {
    if (a.out_hi) {
      uint16_t h, l;
      split_bf16x2(v, h, l);
      a.out_hi[obase + col] = h;
      a.out_lo[obase + col] = l;
    } else {
      a.out[obase + col] = v;
    }
  };
  const int t0 = pp * ps;
  const float last = row[vl - 1].x;          // last VALID value (chatts_vllm.py:122); vl >= 1 here
  int feat;
  if (a.mode == 1) {
    feat = ps + ps * a.emb_dim;
    if (lane < ps) {
      const int t = t0 + lane;
      put(lane, t < vl ? row[t].x : last);
    }
    // out[ps + j*emb + e] = pos_table[idx_j][e], idx_j = t0+j if valid else padding_idx (= max_seq_len)
    for (int q = lane; q < ps * a.emb_dim; q += 64) {
      const int j = q / a.emb_dim, e = q - j * a.emb_dim;
      const int t = t0 + j;
      const int idx = t < vl ? t : a.max_seq_len;
      put(ps + q, a.pos_table[(size_t)idx * a.emb_dim + e]);
    }
  } else if (a.mode == 2) {
    feat = 2 * ps;
    if (lane < ps) {
      const int t = t0 + lane;
      const int den = a.max_valid_len - 1 > 1 ? a.max_valid_len - 1 : 1;   // max(1, max_vl-1), :147
      put(2 * lane, t < vl ? row[t].x : last);
      put(2 * lane + 1, t < vl ? (float)t / (float)den : -1.0f);
    }
  } else {
    feat = ps;
    if (lane < ps) {
      const int t = t0 + lane;
      put(lane, t < vl ? row[t].x : last);
    }
  }
  for (int q = feat + lane; q < a.ld_out; q += 64) put(q, 0.f);   // K padding for the MFMA GEMM
}

// Value-preserved normalisation on the device (sp_encoding, chatts/utils/encoding_utils.py:23-37), one workgroup per series:
// mean, centred values, scale = max|x - mean| / 3 when some |x - mean| >= 3, and the (value, 1.0)-interleaved float32 rows
// of the padded encoder input - plus the statistics the prompt prefix prints.  Everything is float64 like the reference's
// numpy; the sum runs in a FIXED order (256 strided partial sums, then a binary tree), so it is reproducible, but it is not
// numpy's order: the mean can differ from np.mean in the last bit (max / min / ends are exact).
__global__ __launch_bounds__(256) void ts_normalise_kernel(const double* __restrict__ raw, const int32_t* __restrict__ len, int lmax,
                                                          float* __restrict__ enc, double* __restrict__ stats) {
  __shared__ double red[256];
  __shared__ double bc[2];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int L = len[s];
  const double* x = raw + (size_t)s * lmax;
  float2* out = reinterpret_cast<float2*>(enc) + (size_t)s * lmax;
  double* st = stats + (size_t)s * 6;          // mean, factor, max, min, left, right
  if (L <= 0) {
    for (int t = tid; t < lmax; t += 256) out[t] = make_float2(0.f, 0.f);
    if (tid < 6) st[tid] = tid == 1 ? 1.0 : 0.0;
    return;
  }
  auto tree = [&](double v, int op) {            // op 0 sum, 1 max, 2 min; every thread returns the result
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        const double a = red[tid], b = red[tid + o];
        red[tid] = op == 0 ? a + b : (op == 1 ? (a > b ? a : b) : (a < b ? a : b));
      }
      __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
  };
  double sum = 0.0, mx = -INFINITY, mn = INFINITY;
  for (int t = tid; t < L; t += 256) { const double v = x[t]; sum += v; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
  const double mean = tree(sum, 0) / (double)L;
  mx = tree(mx, 1);
  mn = tree(mn, 2);
  double dev = 0.0;
  for (int t = tid; t < L; t += 256) { const double d = fabs(x[t] - mean); dev = d > dev ? d : dev; }
  dev = tree(dev, 1);
  const double factor = dev >= 3.0 ? dev / 3.0 : 1.0;
  for (int t = tid; t < lmax; t += 256) {
    if (t < L) {
      double v = x[t] - mean;
      if (dev >= 3.0) v = v / factor;            // the reference divides in place (scaled /= factor), float64
      out[t] = make_float2((float)v, 1.0f);
    } else {
      out[t] = make_float2(0.f, 0.f);            // zero padding: mask 0 (encoding_utils.py:78-84)
    }
  }
  if (tid == 0) { st[0] = mean; st[1] = factor; st[2] = mx; st[3] = mn; st[4] = x[0]; st[5] = x[L - 1]; }
  (void)bc;
}

}  // namespace chatts

// ---- patchify + the first MLP layer in ONE launch (round 6; VERDICT r5 next #3a) -------------------------------------------------------
// out0[p, n] = GELU(feat[p, :] . W0[n, :] + b0[n]) written as bf16 hi / lo planes - the operand of layer 1.  The separate form cost a
// patchify launch (5.5 us) plus a 20-workgroup run of the prefill kernel for 3 MB of weights (13 us: K = 320 is ten half-steps of
// pipeline fill).  Here a one-wave workgroup owns 16 patch rows x 64 output columns; it builds the activation fragments of its
// 16-row block IN REGISTERS, straight from the series / position table (the feature row never exists in memory), loads its W0
// fragments straight from L2 (3 MB, shared by everybody) and issues the same MFMAs in the same order as the prefill kernel does
// (per 32 K-values: lo pass, then hi pass; operands swapped, D = W . A^T): bit-identical planes.  No LDS, no barrier.
namespace chatts {
__device__ __forceinline__ float ts_feature(const ChattsPatchifyArgs& a, const float2* row, int vl, float last, int t0, int k) {
  const int ps = a.patch_size;
  if (a.mode == 1) {
    if (k < ps) { const int t = t0 + k; return t < vl ? row[t].x : last; }
    const int q = k - ps;
    if (q >= ps * a.emb_dim) return 0.f;
    const int j = q / a.emb_dim, e = q - j * a.emb_dim;
    const int t = t0 + j;
    return a.pos_table[(size_t)(t < vl ? t : a.max_seq_len) * a.emb_dim + e];
  }
  if (a.mode == 2) {
    if (k >= 2 * ps) return 0.f;
    const int t = t0 + (k >> 1);
    if (!(k & 1)) return t < vl ? row[t].x : last;
    const int den = a.max_valid_len - 1 > 1 ? a.max_valid_len - 1 : 1;
    return t < vl ? (float)t / (float)den : -1.0f;
  }
  if (k >= ps) return 0.f;
  const int t = t0 + k;
  return t < vl ? row[t].x : last;
}

// the 8 consecutive features k0 .. k0 + 7 of one patch row (k0 % 8 == 0), LOADS ONLY and branch-free per lane: the tail value "last valid
// value" is row[vl - 1], i.e. the load index is clamped instead of selecting afterwards; out-of-range groups load a valid address and
// are zeroed.  MODE is compile time (the generic per-element form of ts_feature would put every mode's code, with its divisions, into each
// of the 18 groups of a lane).  Mode 1 needs patch size and embedding width multiples of 8 (every shipped config: 16 / 16): a group then
// lies wholly inside the value part or inside ONE position's embedding row - two 16-byte loads, one division per group.
template <int MODE>
__device__ __forceinline__ void ts_feature8(const ChattsPatchifyArgs& a, const float2* row, int vl, int t0, int k0, float (&v)[8]) {
  const int ps = a.patch_size;
  if constexpr (MODE == 1) {
    if (k0 < ps) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int t = t0 + k0 + q; v[q] = row[t < vl ? t : vl - 1].x; }
    } else {
      const int q0 = k0 - ps;
      const bool ok = q0 < ps * a.emb_dim;
      const int j = ok ? q0 / a.emb_dim : 0, e0 = ok ? q0 - j * a.emb_dim : 0;
      const int t = t0 + j;
      const float* src = a.pos_table + (size_t)(t < vl ? t : a.max_seq_len) * a.emb_dim + e0;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
      const float m = ok ? 1.f : 0.f;
      v[0] = x0.x * m; v[1] = x0.y * m; v[2] = x0.z * m; v[3] = x0.w * m; v[4] = x1.x * m; v[5] = x1.y * m; v[6] = x1.z * m; v[7] = x1.w * m;
    }
  } else if constexpr (MODE == 2) {      // (value, position) pairs: k even = value of t0 + k / 2, k odd = t / max(1, max_vl - 1) or -1
    const bool ok = k0 < 2 * ps;
    const int den = a.max_valid_len - 1 > 1 ? a.max_valid_len - 1 : 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = t0 + (ok ? (k0 >> 1) + q : 0);
      const float x = row[t < vl ? t : vl - 1].x;
      v[2 * q] = ok ? x : 0.f;
      v[2 * q + 1] = ok ? (t < vl ? (float)t / (float)den : -1.0f) : 0.f;
    }
  } else {
    const bool ok = k0 < ps;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int t = t0 + (ok ? k0 + q : 0); const float x = row[t < vl ? t : vl - 1].x; v[q] = ok ? x : 0.f; }
  }
}

template <int KSTEPS, int MODE, int NJ = 2>      // NJ: 16-column fragments per wave;  K-steps of 32 (feature count rounded up) and the feature mode: compiled in, so that every load is in flight at once
__global__ __launch_bounds__(64) void ts_layer0_kernel(ChattsPatchifyArgs a, const uint16_t* __restrict__ w0, int ldw, const float* __restrict__ b0,
                                                         int hidden, uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo, int ld_out) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  // workgroup = ONE wave = 16 patch rows x 64 output columns (no LDS, no barrier: nothing to share): it builds (and splits) one activation
  // fragment per K-step and re-uses it for four W fragments; P = 128, H = 5120: 640 independent waves over 256 CUs
  const int lane = threadIdx.x & 63;
  const int n0 = blockIdx.x * (16 * NJ), r0 = blockIdx.y * 16;
  const int frow = lane & 15, kc = lane >> 4;                  // this lane's row of a 16-row block, and its 8 K-values of a 32-deep step
  if (r0 >= a.total_patches) return;                           // (wave-uniform: a block wholly beyond the last patch row)
  // W0 fragments first: rows n0 + 16 j + frow, K-values 32 s + 8 kc ..; all KSTEPS x 4 loads are independent of everything else
  bf16x8_t wf[KSTEPS][NJ];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      int wr = n0 + j * 16 + frow;
      if (wr > hidden - 1) wr = hidden - 1;
      wf[s][j] = *reinterpret_cast<const bf16x8_t*>(w0 + (size_t)wr * ldw + s * 32 + kc * 8);
    }
  // the patch row this lane feeds, clamped to the last real row (stores are masked).  Its series: up to 63 series - lane i holds
  // row_off[i] / valid_len[i] (ONE round trip), the search is a wave-uniform walk over lanes and two ds_bpermute; more series - the
  // binary search of ts_patchify_kernel (log2 N dependent round trips)
  const int nser = a.n_series;
  const int my_off = a.row_off[lane <= nser ? lane : nser];
  const int my_vl = a.valid_len[lane < nser ? lane : nser - 1];
  int p = r0 + frow;
  if (p > a.total_patches - 1) p = a.total_patches - 1;
  int sidx = 0, off = 0, vl = 0;
  if (nser <= 63) {
    for (int i = 1; i < nser; ++i) sidx += __builtin_amdgcn_readlane(my_off, i) <= p ? 1 : 0;
    off = __shfl(my_off, sidx, 64);
    vl = __shfl(my_vl, sidx, 64);
  } else {
    int lo = 0, hi = nser;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (a.row_off[mid] <= p) lo = mid; else hi = mid;
    }
    sidx = lo;
    off = a.row_off[lo];
    vl = a.valid_len[lo];
  }
  const int t0 = (p - off) * a.patch_size;
  const float2* rowp = reinterpret_cast<const float2*>(a.series) + (size_t)sidx * a.lmax;
  // every feature load of the block before the first use
  float raw[KSTEPS][8];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) ts_feature8<MODE>(a, rowp, vl, t0, s * 32 + kc * 8, raw[s]);
  f32x4 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    bf16x8_t ahi, alo;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      uint16_t h, l;
      split_bf16x2(raw[s][q], h, l);
      ahi[q] = __builtin_bit_cast(__bf16, h);
      alo[q] = __builtin_bit_cast(__bf16, l);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][j], alo, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][j], ahi, acc[j], 0, 0, 0);
  }
  // lane holds D[feature n0 + 16 j + 4 kc + r][token r0 + frow]: bias, exact-erf GELU, split, 8-byte stores (ring_store's arithmetic)
  const int tok = r0 + frow;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int fb = n0 + j * 16 + kc * 4;
    const int fbc = fb < hidden ? fb : 0;
    const f32x4 bias = b0 ? *reinterpret_cast<const f32x4*>(b0 + fbc) : (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x4_t hv, lv;
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = gelu_erf_f(acc[j][r] + bias[r]);
        const __bf16 h = (__bf16)v;
        hv[r] = h;
        lv[r] = (__bf16)(v - (float)h);
      }
    }
    if (tok < a.total_patches && fb < hidden) {
      *reinterpret_cast<bf16x4_t*>(out_hi + (size_t)tok * ld_out + fb) = hv;
      *reinterpret_cast<bf16x4_t*>(out_lo + (size_t)tok * ld_out + fb) = lv;
    }
  }
}
}  // namespace chatts

using namespace chatts;

extern "C" int chatts_ts_normalise(const double* raw, const int32_t* lengths, int n_series, int lmax, float* enc, double* stats,
                                   chatts_stream_t stream) {
  CHATTS_REQUIRE(n_series >= 0 && lmax >= 0, CHATTS_E_BADARG, "ts_normalise: bad sizes n=%d lmax=%d", n_series, lmax);
  if (n_series == 0 || lmax == 0) return CHATTS_OK;
  CHATTS_REQUIRE(raw && lengths && enc && stats, CHATTS_E_BADARG, "ts_normalise: null pointer");
  hipLaunchKernelGGL(ts_normalise_kernel, dim3(n_series), dim3(256), 0, as_stream(stream), raw, lengths, lmax, enc, stats);
  CHATTS_CHECK_LAUNCH("ts_normalise");
  return CHATTS_OK;
}

extern "C" int chatts_ts_patch_cnt(const float* series, int n_series, int lmax, int patch_size,
                                   int32_t* valid_len, int64_t* patch_cnt, chatts_stream_t stream) {
  CHATTS_REQUIRE(n_series >= 0 && lmax >= 0 && patch_size > 0, CHATTS_E_BADARG,
                 "ts_patch_cnt: bad sizes n=%d lmax=%d patch=%d", n_series, lmax, patch_size);
  if (n_series == 0) return CHATTS_OK;
  CHATTS_REQUIRE(series != nullptr || lmax == 0, CHATTS_E_BADARG, "ts_patch_cnt: null series");
  CHATTS_REQUIRE(valid_len || patch_cnt, CHATTS_E_BADARG, "ts_patch_cnt: no output");
  const int blocks = (n_series + 3) / 4;
  hipLaunchKernelGGL(ts_patch_cnt_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), series, n_series,
                     lmax, patch_size, valid_len, patch_cnt);
  CHATTS_CHECK_LAUNCH("ts_patch_cnt");
  return CHATTS_OK;
}

extern "C" int chatts_ts_patchify(const ChattsPatchifyArgs* a, chatts_stream_t stream) {
  CHATTS_REQUIRE(a != nullptr, CHATTS_E_BADARG, "ts_patchify: null args");
  CHATTS_REQUIRE(a->total_patches >= 0 && a->n_series >= 0, CHATTS_E_BADARG, "ts_patchify: negative size");
  if (a->total_patches == 0) return CHATTS_OK;
  CHATTS_REQUIRE(a->series && a->row_off && a->valid_len && (a->out || (a->out_hi && a->out_lo)), CHATTS_E_BADARG,
                 "ts_patchify: null pointer");
  CHATTS_REQUIRE(a->patch_size > 0 && a->patch_size <= 64, CHATTS_E_SHAPE, "ts_patchify: patch_size %d not in 1..64",
                 a->patch_size);
  CHATTS_REQUIRE(a->mode >= 0 && a->mode <= 2, CHATTS_E_BADARG, "ts_patchify: mode %d", a->mode);
  const int feat = a->mode == 1 ? a->patch_size * (1 + a->emb_dim) : (a->mode == 2 ? 2 * a->patch_size : a->patch_size);
  CHATTS_REQUIRE(a->ld_out >= feat, CHATTS_E_SHAPE, "ts_patchify: ld_out %d < features %d", a->ld_out, feat);
  if (a->mode == 1)
    CHATTS_REQUIRE(a->pos_table && a->emb_dim > 0 && a->max_seq_len > 0, CHATTS_E_BADARG,
                   "ts_patchify: position table missing");
  const int blocks = (a->total_patches + 3) / 4;
  hipLaunchKernelGGL(ts_patchify_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), *a);
  CHATTS_CHECK_LAUNCH("ts_patchify");
  return CHATTS_OK;
}

// Whole encoder in one call (the C-ABI SURVEY.md section 8b proposes for a3 + a4): patchify, then the MLP
// (Linear + exact-erf GELU) x (n-1) + Linear, ping-ponging between two caller-owned [P, H] buffers.
extern "C" int chatts_ts_encode(const float* series, const int32_t* row_off, const int32_t* valid_len, int n_series, int lmax,
                                int max_valid_len, int total_patches, const ChattsTsWeights* w, float* feat, float* h0,
                                float* h1, float* out, void* workspace, size_t workspace_bytes, chatts_stream_t stream) {
  CHATTS_REQUIRE(w != nullptr && w->num_layers >= 1 && w->num_layers <= 8, CHATTS_E_BADARG, "ts_encode: bad weights");
  if (total_patches == 0) return CHATTS_OK;
  StageRange stage("chatts.ts_encode");
  CHATTS_REQUIRE(feat && out && (w->num_layers < 2 || h0) && (w->num_layers < 3 || h1), CHATTS_E_BADARG,
                 "ts_encode: null buffer");
  ChattsPatchifyArgs pa{};
  pa.series = series; pa.row_off = row_off; pa.valid_len = valid_len; pa.pos_table = w->pos_table; pa.out = feat;
  pa.n_series = n_series; pa.lmax = lmax; pa.patch_size = w->patch_size; pa.mode = w->mode; pa.emb_dim = w->emb_dim;
  pa.max_seq_len = w->max_seq_len; pa.max_valid_len = max_valid_len; pa.total_patches = total_patches;
  pa.ld_out = w->in_features_pad;
  // Plane path (P > 1 and every K a multiple of 64): each activation matrix lives as bf16 hi / lo planes inside the caller's
  // float32 scratch of the same byte size (4 B per element = 2 + 2) - patchify writes the planes of layer 0's operand, every
  // GELU epilogue writes the planes of the next layer's, and the GEMMs stage all operands with whole-line LDS-DMA
  // (gemm_stream_kernel for P <= 16: the 213 MB of MLP weights are streamed once; gemm_dma_kernel above).  The values are
  // the splits of exactly the float32 numbers the float32 path would hold, so both paths give the same result.
  const int P = total_patches, H = w->hidden;
  const bool planes = P > 1 && w->in_features_pad % 64 == 0 && H % 64 == 0 && opt_get(OPT_TS_F32_PATH, 0) == 0;
  auto hi_of = [&](float* buf) { return reinterpret_cast<chatts_bf16*>(buf); };
  auto lo_of = [&](float* buf, int k) { return reinterpret_cast<chatts_bf16*>(buf) + (size_t)P * k; };
  if (planes) { pa.out = nullptr; pa.out_hi = hi_of(feat); pa.out_lo = lo_of(feat, w->in_features_pad); }
  int rc;
  float* cur = feat;
  int k = w->in_features_pad;
  int l0 = 0;
  // patchify + layer 0 as one launch (ts_layer0_kernel) when layer 0 is a hidden layer of the plane path: its GELU output goes out as the
  // planes layer 1 reads; the feature matrix is never written.  TS_L0_FUSED=0 keeps the two-launch form (bit-identical).
  const int feat_n = w->mode == 1 ? w->patch_size * (1 + w->emb_dim) : (w->mode == 2 ? 2 * w->patch_size : w->patch_size);
  const int ksteps = (feat_n + 31) / 32;
  const bool l0_shape = w->mode == 1 ? (w->patch_size % 8 == 0 && w->emb_dim % 8 == 0 && ksteps <= 10) : (w->patch_size % 8 == 0 && ksteps <= 2);
  if (planes && w->num_layers >= 2 && H % 16 == 0 && ksteps >= 1 && l0_shape && ksteps * 32 <= w->in_features_pad && h0 &&
      ((uintptr_t)w->w[0] % 16) == 0 && ((uintptr_t)w->b[0] % 16) == 0 && ((uintptr_t)w->pos_table % 16) == 0 && opt_get(OPT_TS_L0_FUSED, 1) != 0) {
    CHATTS_REQUIRE(pa.series && pa.row_off && pa.valid_len && (w->mode != 1 || (pa.pos_table && w->emb_dim > 0 && w->max_seq_len > 0)), CHATTS_E_BADARG,
                   "ts_encode: null series / offsets / position table");
    const dim3 grid((H + 31) / 32, (P + 15) / 16), block(64);      // (NJ = 2: 32 columns per wave)
    uint16_t* oh = hi_of(h0);
    uint16_t* ol = lo_of(h0, H);
#define CHATTS_TS_L0(KS, MD) hipLaunchKernelGGL((ts_layer0_kernel<KS, MD>), grid, block, 0, as_stream(stream), pa, w->w[0], w->in_features_pad, w->b[0], H, oh, ol, H)
    if (w->mode == 1) {
      switch (ksteps) {
        case 1: CHATTS_TS_L0(1, 1); break;
        case 2: CHATTS_TS_L0(2, 1); break;
        case 3: CHATTS_TS_L0(3, 1); break;
        case 4: CHATTS_TS_L0(4, 1); break;
        case 5: CHATTS_TS_L0(5, 1); break;
        case 6: CHATTS_TS_L0(6, 1); break;
        case 7: CHATTS_TS_L0(7, 1); break;
        case 8: CHATTS_TS_L0(8, 1); break;
        case 9: CHATTS_TS_L0(9, 1); break;
        default: CHATTS_TS_L0(10, 1); break;
      }
    } else if (w->mode == 2) {
      if (ksteps == 1) CHATTS_TS_L0(1, 2); else CHATTS_TS_L0(2, 2);
    } else {
      if (ksteps == 1) CHATTS_TS_L0(1, 0); else CHATTS_TS_L0(2, 0);
    }
#undef CHATTS_TS_L0
    CHATTS_CHECK_LAUNCH("ts_layer0");
    cur = h0;
    k = H;
    l0 = 1;
  } else {
    if ((rc = chatts_ts_patchify(&pa, stream)) != 0) return rc;
  }
  for (int l = l0; l < w->num_layers; ++l) {
    const bool last = l == w->num_layers - 1;
    float* dst = last ? out : ((l & 1) ? h1 : h0);
    ChattsLinearArgs la{};
    la.w = w->w[l]; la.bias = w->b[l]; la.m = P; la.n = H; la.k = k;
    la.lda = k; la.ldw = k; la.ldc = H; la.epilogue = last ? CHATTS_EPI_NONE : CHATTS_EPI_GELU;
    la.workspace = workspace; la.workspace_bytes = workspace_bytes;
    if (planes) {
      la.a_hi = hi_of(cur); la.a_lo = lo_of(cur, k); la.ld_planes = k;
      if (last) la.c = dst;
      else { la.c_hi = hi_of(dst); la.c_lo = lo_of(dst, H); la.ld_cplanes = H; }
    } else {
      la.a = cur; la.c = dst;
    }
    if ((rc = chatts_linear(&la, stream)) != 0) return rc;
    cur = dst;
    k = H;
  }
  return CHATTS_OK;
}
