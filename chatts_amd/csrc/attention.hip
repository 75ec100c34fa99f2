// attention.hip - causal grouped-query attention over the float32 KV cache (head_dim 128).
// Math per transformers eager_attention_forward (oracle/qwen_decoder.py): scores = q.k^T * d^-1/2,
// float32 softmax, out = P.V; GQA: the n_q/n_kv query heads of a group share one K/V head (repeat_kv).
//
// Kernels:
//  * attn_decode_kernel (T = 1, the per-token path): ONE WAVE per (kv head, 16-key tile).  A wave has no
//    barriers to wait for, issues all of its K (8 x 16 B) and V (16 x 8 B) loads at once and finishes in about
//    one memory round trip; hundreds of such waves spread the cache read over all CUs.  It also fuses what
//    used to be a separate launch: per-head q/k RMSNorm (Qwen3), RoPE of q and of the new k, and the write of
//    the new K/V row into the cache (done by the single wave whose tile contains `pos`, which then uses the row
//    from LDS).  Softmax statistics are wave shuffles; P is broadcast lane->wave with v_readlane.
//    Each wave emits an (m, l, o[G][128]) partial; attn_decode_combine_kernel merges the partials of a head.
//  * attn_rows_kernel (short prefill chunks, T < 16, and the last-row path): one workgroup per (kv head, query row), 64-key
//    tiles, online softmax, float32 VALU.  Longer chunks run the flash kernels further down: attn_prefill_bf16x3_kernel (bf16
//    matrix pipe on hi / lo planes, the default) and attn_prefill_mfma_kernel (exact-f32 MFMA).
// The KV cache is float32 on purpose: the parity target is the float32 reference path and the cache
// is ~1% of decode traffic at the benchmark context.
#include <math.h>

#include "common.h"

namespace chatts {

}  // namespace chatts
#include "attn_decode.h"
namespace chatts {

// ---------------------------------------------------------------------------------------------------
// decode: grid (n_kv, n_slots), 64 threads.  Slot s walks tiles s, s + n_slots, ... of 16 keys.
// ---------------------------------------------------------------------------------------------------
template <int GMAX, bool EXACT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void attn_decode_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) float q_s[GMAX * kHeadDim];
  __shared__ __attribute__((aligned(16))) float knew_s[kHeadDim];
  __shared__ __attribute__((aligned(16))) float vnew_s[kHeadDim];
  attn_decode_wave<GMAX, true, EXACT>(p, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x, q_s, knew_s, vnew_s, p.part_o, p.part_ml,
                                      (size_t)blockIdx.z * p.n_q + blockIdx.x * (EXACT ? GMAX : p.n_q / p.n_kv));
}

// out[h] = sum_s e^{m_s - M} o_s / sum_s e^{m_s - M} l_s over the slots that saw keys (<= 64 slots): attn_combine_wave
// (attn_decode.h), one workgroup (2 waves) per head.
__global__ __launch_bounds__(128) void attn_decode_combine_kernel(AttnParams p) {
  attn_combine_wave(p, blockIdx.x, blockIdx.y, threadIdx.x, threadIdx.x & 63, p.part_o, p.part_ml,
                    ((size_t)blockIdx.y * p.n_q + blockIdx.x) * p.n_splits);
}

// ---------------------------------------------------------------------------------------------------
// prefill: grid (n_kv, T, n_splits), 256 threads, 64-key tiles.
//   phase 1  scores: 4 lanes per key (32 dims each), xor-shuffle reduce
//   phase 2  one wave per head: tile max / rescale factor / p = exp(s - m) written back to LDS
//   phase 3  P.V: thread = (dim, key half), V rows read fully coalesced, 8 loads in flight per thread
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_rows_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) float q_s[kMaxGroup * kHeadDim];
  __shared__ float s_s[kMaxGroup * kTile];
  __shared__ float alpha_s[kMaxGroup];
  __shared__ float o_s[kMaxGroup * kHeadDim];

  const int hk = blockIdx.x, row = blockIdx.y, split = blockIdx.z;
  const int G = p.n_q / p.n_kv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pos = (p.pos0_dev ? *p.pos0_dev : p.pos0) + row;
  const int heads = p.n_q + 2 * p.n_kv;
  const float scale = 0.08838834764831845f;
  const int ntiles = pos / kTile + 1;

  for (int i = tid; i < G * kHeadDim; i += 256)
    q_s[i] = p.qkv[((size_t)row * heads + hk * G) * kHeadDim + i];
  __syncthreads();

  const KvLayout kvl{p.table, p.n_kv, p.max_ctx, p.log_block};
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};   // heads wave, wave + 4
  float acc[kMaxGroup];
#pragma unroll
  for (int g = 0; g < kMaxGroup; ++g) acc[g] = 0.f;
  const int key_l = tid >> 2, quarter = tid & 3;
  const int d_o = tid & 127, half = tid >> 7;

  for (int tile = split; tile < ntiles; tile += p.n_splits) {
    const int j0 = tile * kTile;
    const size_t toff = kv_tile_off(kvl, hk, j0);     // first row of the tile (one block-table lookup when paged)
    {
      const int j = j0 + key_l;
      const int jc = j <= pos ? j : pos;
      const float* kr = p.kc + toff + (size_t)(jc - j0) * kHeadDim + quarter * 4;
      f32x4 kv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) kv[i] = *reinterpret_cast<const f32x4*>(kr + i * 16);
      float dot[kMaxGroup];
#pragma unroll
      for (int g = 0; g < kMaxGroup; ++g) dot[g] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int g = 0; g < kMaxGroup; ++g) {
          if (g < G) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(q_s + g * kHeadDim + quarter * 4 + i * 16);
            dot[g] = fmaf(kv[i].x, qv.x, dot[g]);
            dot[g] = fmaf(kv[i].y, qv.y, dot[g]);
            dot[g] = fmaf(kv[i].z, qv.z, dot[g]);
            dot[g] = fmaf(kv[i].w, qv.w, dot[g]);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < kMaxGroup; ++g) {
        if (g < G) {
          float v = dot[g];
          v += lane_xor1(v);
          v += lane_xor2(v);
          if (quarter == 0) s_s[g * kTile + key_l] = j <= pos ? v * scale : -INFINITY;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int g = wave + gi * 4;
      if (g < G) {
        const float s = s_s[g * kTile + lane];
        const float m_new = fmaxf(m_run[gi], wave_max(s));
        const float pr = expf(s - m_new);
        const float a = expf(m_run[gi] - m_new);
        l_run[gi] = l_run[gi] * a + wave_sum(pr);
        m_run[gi] = m_new;
        s_s[g * kTile + lane] = pr;
        if (lane == 0) alpha_s[g] = a;
      }
    }
    __syncthreads();
    {
#pragma unroll
      for (int g = 0; g < kMaxGroup; ++g)
        if (g < G) acc[g] *= alpha_s[g];
      const int jb = j0 + half * 32;
#pragma unroll
      for (int b8 = 0; b8 < 32; b8 += 8) {
        float vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = jb + b8 + u;
          vv[u] = p.vc[toff + (size_t)((j <= pos ? j : pos) - j0) * kHeadDim + d_o];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
          for (int g = 0; g < kMaxGroup; ++g)
            if (g < G) acc[g] = fmaf(s_s[g * kTile + half * 32 + b8 + u], vv[u], acc[g]);   // p = 0 for masked keys
        }
      }
    }
    __syncthreads();
  }

  if (half == 1) {
#pragma unroll
    for (int g = 0; g < kMaxGroup; ++g)
      if (g < G) o_s[g * kHeadDim + d_o] = acc[g];
  }
  if (lane == 0) {
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int g = wave + gi * 4;
      if (g < G) { s_s[g * 2] = m_run[gi]; s_s[g * 2 + 1] = l_run[gi]; }
    }
  }
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int g = 0; g < kMaxGroup; ++g) {
      if (g < G) {
        const float o = acc[g] + o_s[g * kHeadDim + d_o];
        const int hq = hk * G + g;
        if (p.n_splits == 1) {
          p.out[((size_t)row * p.n_q + hq) * kHeadDim + d_o] = o / s_s[g * 2 + 1];
        } else {
          const size_t pi = ((size_t)row * p.n_q + hq) * p.n_splits + split;
          p.part_o[pi * kHeadDim + d_o] = o;
          if (d_o == 0) { p.part_ml[pi * 2] = s_s[g * 2]; p.part_ml[pi * 2 + 1] = s_s[g * 2 + 1]; }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// prefill, flash style on the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate, bitwise an
// fmaf chain - the parity target is the float32 reference, and attention is ~3 % of prefill FLOPs, so the f32 MFMA
// rate (155 TF) is enough).  grid (ceil(T/64), n_q), 256 threads: wave w owns query rows q0+16w .. +15 of ONE q head.
//   Q fragment  A[i = lane&15][K-slot group lane>>4 = head dims 32*(lane>>4) ..+31]: 32 floats per lane, loaded once
//   K tile      [KT][136] f32 in LDS: B fragments are ds_read_b128 (4 MFMAs per read), conflict-free with the rotation kq & 1
//   S = Q.K^T   KT/16 accumulators in C layout (col = lane&15 = key, row = 4*(lane>>4) + reg)
//   softmax     online, per C-layout row: 16-lane xor shuffles for max and sum
//   P           C layout -> wave-private LDS [16][KT+2] -> A layout (row = lane&15, k = lane>>4 + 4j)
//   O += P.V    V tile [KT][132] f32 in LDS, 8 accumulators with permuted columns (lane n, acc db -> head dim 8n + db)
// Heaviest query tiles (most key tiles under the causal mask) are scheduled first (pairing heavy + light tiles in one
// workgroup was measured: slower, the chip then holds one wave per SIMD and nothing hides the per-tile overheads).
// ---------------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256, 3) void attn_prefill_mfma_kernel(AttnParams p) {
  constexpr int KS = 136, VS = 132, PS = KT + 2, NB = KT / 16;
  __shared__ __attribute__((aligned(16))) float v_s[KT * VS];
  __shared__ __attribute__((aligned(16))) float k_s[KT * KS];
  __shared__ float p_all[4 * 16 * PS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hq = blockIdx.y;
  const int G = p.n_q / p.n_kv, hk = hq / G;
  const int heads = p.n_q + 2 * p.n_kv;
  const int T = p.t, pos0 = p.pos0_dev ? *p.pos0_dev : p.pos0;
  const int qt = gridDim.x - 1 - blockIdx.x;                // heaviest query tiles (most key tiles) first
  const int q0 = qt * 64;
  const int arow = lane & 15, kq = lane >> 4;              // A/B fragment coordinates
  const int ccol = lane & 15, crow0 = (lane >> 4) * 4;     // C layout
  const float scale2 = 0.08838834764831845f * 1.4426950408889634f;   // d^-1/2 * log2(e): the softmax runs on exp2
  float* p_s = p_all + wave * 16 * PS;

  float qreg[32];
  {
    int qrow = q0 + wave * 16 + arow;
    if (qrow > T - 1) qrow = T - 1;                         // padded rows compute garbage that is never stored
    // K-slot assignment: the MFMA's K index is free as long as both operands agree, so lane group kq takes head dims
    // kq*32 .. kq*32+31 (contiguous: the K fragments below are ds_read_b128, four MFMAs per read), visiting its eight
    // 4-float chunks rotated by kq & 1 - the rotation that makes the b128 reads of the [KT][136] tile conflict-free.
    const float* qp = p.qkv + ((size_t)qrow * heads + hq) * kHeadDim + kq * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + 4 * ((i + (kq & 1)) & 7));
      qreg[4 * i] = qv.x; qreg[4 * i + 1] = qv.y; qreg[4 * i + 2] = qv.z; qreg[4 * i + 3] = qv.w;
    }
  }
  int qpos[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) qpos[r] = pos0 + q0 + wave * 16 + crow0 + r;
  float m_run[4], l_run[4];
  f32x4 o_acc[8];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
#pragma unroll
  for (int db = 0; db < 8; ++db) o_acc[db] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int last_q = q0 + 63 < T - 1 ? q0 + 63 : T - 1;
  const int kmax = pos0 + last_q;                           // last key any row of this workgroup may see
  const int nkt = kmax / KT + 1;
  const KvLayout kvl{p.table, p.n_kv, p.max_ctx, p.log_block};

  // K/V tiles go global -> registers -> LDS; the loads of tile kt+1 are issued BEFORE the MFMAs of tile kt, so their L2
  // latency hides under the tile's 128 MFMAs per wave instead of sitting between two barriers.
  constexpr int SLOTS = KT * 32 / 256;                      // float4 slots per thread and operand
  f32x4 kreg[SLOTS], vreg[SLOTS];
  auto load_tile = [&](int kt) {
    const size_t toff = kv_tile_off(kvl, hk, kt * KT);      // first row of the tile (one block-table lookup when paged)
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int idx = tid + i * 256, key = idx >> 5, c4 = idx & 31;
      const int kr = kt * KT + key <= kmax ? key : kmax - kt * KT;    // clamped row inside the tile; masked by key index below
      kreg[i] = *reinterpret_cast<const f32x4*>(p.kc + toff + (size_t)kr * kHeadDim + c4 * 4);
      vreg[i] = *reinterpret_cast<const f32x4*>(p.vc + toff + (size_t)kr * kHeadDim + c4 * 4);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int idx = tid + i * 256, key = idx >> 5, c4 = idx & 31;
      *reinterpret_cast<f32x4*>(k_s + key * KS + c4 * 4) = kreg[i];
      *reinterpret_cast<f32x4*>(v_s + key * VS + c4 * 4) = vreg[i];
    }
  };
  // Key split (flash-decoding style): workgroup z takes key tiles z, z + NS, ...  T/16 x heads waves is all the parallelism one
  // sequence has (~2 waves per SIMD at T = 798, and the causal mask makes them unequal); splitting the keys doubles it.
  const int NS = p.n_splits, split = blockIdx.z;
  if (split < nkt) load_tile(split);
  for (int kt = split; kt < nkt; kt += NS) {
    const int k0 = kt * KT;
    __syncthreads();                                        // the previous tile has been consumed by every wave
    store_tile();
    __syncthreads();
    if (kt + NS < nkt) load_tile(kt + NS);

    f32x4 s_acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) s_acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x4 kf[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        kf[nb] = *reinterpret_cast<const f32x4*>(k_s + (nb * 16 + arow) * KS + kq * 32 + 4 * ((i + (kq & 1)) & 7));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          s_acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(qreg[4 * i + e], kf[nb][e], s_acc[nb], 0, 0, 0);
    }
    // Online softmax in the log2 domain (scores are pre-scaled by d^-1/2 * log2 e, so every exponential is one v_exp_f32),
    // written stage by stage over the four rows a lane holds: the DPP reductions are dependent chains with wait states,
    // and four independent chains in flight are what hides them (measured with s_memtime: this block was 3.5k cycles per
    // tile as a per-row loop with expf, against 2k cycles for the 64 MFMAs of Q.K^T).
    float mx[4], ps[4], alpha[4], m_ref[4];
    const bool diag = k0 + KT - 1 > pos0 + q0 + wave * 16;  // wave-uniform: only such a tile holds keys some row of this wave cannot see
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mx[r] = -INFINITY;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float sc = s_acc[nb][r] * scale2;
        if (diag && k0 + nb * 16 + ccol > qpos[r]) sc = -INFINITY;  // causal mask
        s_acc[nb][r] = sc;
        mx[r] = fmaxf(mx[r], sc);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], lane_xor1(mx[r]));
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], lane_xor2(mx[r]));
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], row_ror4(mx[r]));
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], row_ror8(mx[r]));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m_new = fmaxf(m_run[r], mx[r]);
      // a key split other than 0 may start on a tile this row cannot see at all: keep m = -inf, contribute nothing
      m_ref[r] = m_new == -INFINITY ? 0.f : m_new;
      alpha[r] = __builtin_amdgcn_exp2f(m_run[r] - m_ref[r]);
      m_run[r] = m_new;
      ps[r] = 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float e = __builtin_amdgcn_exp2f(s_acc[nb][r] - m_ref[r]);
        s_acc[nb][r] = e;
        ps[r] += e;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) ps[r] += lane_xor1(ps[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) ps[r] += lane_xor2(ps[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) ps[r] += row_ror4(ps[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) ps[r] += row_ror8(ps[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      l_run[r] = l_run[r] * alpha[r] + ps[r];
#pragma unroll
      for (int db = 0; db < 8; ++db) o_acc[db][r] *= alpha[r];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) p_s[(crow0 + r) * PS + nb * 16 + ccol] = s_acc[nb][r];
    }
    __builtin_amdgcn_wave_barrier();                        // P is wave private: LDS ops of one wave stay in order
    // O += P.V with the output columns permuted: accumulator db of lane n holds head dim n*8 + db, so one key row gives a
    // lane its eight V values as two ds_read_b128 (conflict-free for the [KT][132] tile) and the final store is 32 B per lane
#pragma unroll
    for (int j = 0; j < KT / 4; ++j) {                      // unrolled: the reads of step j+1 are issued under the MFMAs of step j
      const float pa = p_s[arow * PS + kq + 4 * j];
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(v_s + (kq + 4 * j) * VS + arow * 8);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(v_s + (kq + 4 * j) * VS + arow * 8 + 4);
#pragma unroll
      for (int db = 0; db < 4; ++db) o_acc[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, v0[db], o_acc[db], 0, 0, 0);
#pragma unroll
      for (int db = 0; db < 4; ++db) o_acc[4 + db] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, v1[db], o_acc[4 + db], 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qrow = q0 + wave * 16 + crow0 + r;
    if (qrow < T && NS > 1) {                               // partial (unnormalised) result of this key split
      const size_t pi = ((size_t)qrow * p.n_q + hq) * NS + split;
      float* dst = p.part_o + pi * kHeadDim + ccol * 8;
      *reinterpret_cast<f32x4*>(dst) = (f32x4){o_acc[0][r], o_acc[1][r], o_acc[2][r], o_acc[3][r]};
      *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){o_acc[4][r], o_acc[5][r], o_acc[6][r], o_acc[7][r]};
      if (ccol == 0) { p.part_ml[pi * 2] = m_run[r] * 0.6931471805599453f; p.part_ml[pi * 2 + 1] = l_run[r]; }   // m back to nats
    } else if (qrow < T) {
      const float inv = 1.0f / l_run[r];
      const size_t oi = ((size_t)qrow * p.n_q + hq) * kHeadDim + ccol * 8;
      if (p.out_hi) {
        typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
        bf16x8v hv, lv;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
#pragma clang fp contract(off)   // lo = split of the ROUNDED product, as chatts_split_bf16x2 of the float32 output would give
          const float v = o_acc[db][r] * inv;
          const __bf16 h = (__bf16)v;
          hv[db] = h;
          lv[db] = (__bf16)(v - (float)h);
        }
        *reinterpret_cast<bf16x8v*>(p.out_hi + oi) = hv;
        *reinterpret_cast<bf16x8v*>(p.out_lo + oi) = lv;
      } else {
        float* dst = p.out + oi;
        *reinterpret_cast<f32x4*>(dst) = (f32x4){o_acc[0][r] * inv, o_acc[1][r] * inv, o_acc[2][r] * inv, o_acc[3][r] * inv};
        *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){o_acc[4][r] * inv, o_acc[5][r] * inv, o_acc[6][r] * inv, o_acc[7][r] * inv};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// prefill, bf16x3 form: the same flash loop as attn_prefill_mfma_kernel, but Q.K^T and P.V run on the bf16 matrix pipe
// (v_mfma_f32_16x16x32_bf16, 16x the rate of the float32 MFMA) with BOTH operands split x = hi + lo into bf16 planes and three
// passes per product (hi.hi + hi.lo + lo.hi; the dropped lo.lo term is 2^-16 of the product - the accuracy class of the
// bf16x2 GEMMs, where only one operand needs the split).  Per 32-key tile and wave: 24 + 24 MFMAs of ~17 cycles instead of
// 64 + 64 of 32.  K / V tiles are split once, by all 256 threads, while they are staged: K as [key][d] planes (272-byte rows:
// the B fragments of Q.K^T are conflict-free ds_read_b128), V TRANSPOSED as [d][key] planes (80-byte rows) because the B
// operand of P.V needs 8 consecutive KEYS of one head dim per lane - the staging threads of V therefore vary over keys
// within a wave (contiguous 2-byte LDS writes) and pay with 16-byte global reads at a 512-byte stride (L2-resident rows).
// ---------------------------------------------------------------------------------------------------
typedef __bf16 abf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float (&x)[8], abf16x8_t& hi, abf16x8_t& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)x[j];
    hi[j] = h;
    lo[j] = (__bf16)(x[j] - (float)h);
  }
}

__global__ __launch_bounds__(256, 2) void attn_prefill_bf16x3_kernel(AttnParams p) {
  constexpr int KT = 32, NB = 2, KROW = 272, VROW = 80, PS = 36;
  __shared__ __attribute__((aligned(16))) char k_hi[KT * KROW];
  __shared__ __attribute__((aligned(16))) char k_lo[KT * KROW];
  __shared__ __attribute__((aligned(16))) char vt_hi[kHeadDim * VROW];
  __shared__ __attribute__((aligned(16))) char vt_lo[kHeadDim * VROW];
  __shared__ __attribute__((aligned(16))) float p_all[4 * 16 * PS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hq = blockIdx.y;
  const int G = p.n_q / p.n_kv, hk = hq / G;
  const int heads = p.n_q + 2 * p.n_kv;
  const int T = p.t, pos0 = p.pos0_dev ? *p.pos0_dev : p.pos0;
  const int qt = gridDim.x - 1 - blockIdx.x;                // heaviest query tiles (most key tiles) first
  const int q0 = qt * 64;
  const int arow = lane & 15, kq = lane >> 4;              // A/B fragment coordinates: row / column, group of 8 K-values
  const int ccol = lane & 15, crow0 = (lane >> 4) * 4;     // C layout
  const float scale2 = 0.08838834764831845f * 1.4426950408889634f;   // d^-1/2 * log2(e): the softmax runs on exp2
  float* p_s = p_all + wave * 16 * PS;

  abf16x8_t q_hi[4], q_lo[4];                                // A fragments of this wave's 16 query rows, 4 K-steps of 32 dims
  {
    int qrow = q0 + wave * 16 + arow;
    if (qrow > T - 1) qrow = T - 1;                         // padded rows compute garbage that is never stored
    const float* qp = p.qkv + ((size_t)qrow * heads + hq) * kHeadDim + kq * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qp + ks * 32), b = *reinterpret_cast<const f32x4*>(qp + ks * 32 + 4);
      const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      split8(x, q_hi[ks], q_lo[ks]);
    }
  }
  int qpos[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) qpos[r] = pos0 + q0 + wave * 16 + crow0 + r;
  float m_run[4], l_run[4];
  f32x4 o_acc[8];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
#pragma unroll
  for (int db = 0; db < 8; ++db) o_acc[db] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int last_q = q0 + 63 < T - 1 ? q0 + 63 : T - 1;
  const int kmax = pos0 + last_q;                           // last key any row of this workgroup may see
  const int nkt = kmax / KT + 1;
  const KvLayout kvl{p.table, p.n_kv, p.max_ctx, p.log_block};

  // tile kt+1 is loaded into registers before the MFMAs of tile kt (as in the float32 kernel).
  // K: thread (key = idx >> 5, 4 dims idx & 31): a wave reads two whole key rows.  V: a lane takes the SAME 4 dims of the key
  // pair (2j, 2j+1), j = lane & 15, so that the transposed store is one 4-byte write per dim - the two keys' bf16 values side
  // by side, 16 lanes = 16 consecutive dwords (the first version wrote 2 bytes per lane: 35 % of all LDS cycles were conflicts).
  f32x4 kreg[4], vreg[4];
  const int vj = lane & 15;
  auto load_tile = [&](int kt) {
    const size_t toff = kv_tile_off(kvl, hk, kt * KT);      // first row of the tile (one block-table lookup when paged)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int kkey = idx >> 5, kc4 = idx & 31;
      const int kr = kt * KT + kkey <= kmax ? kkey : kmax - kt * KT;    // clamped row inside the tile; masked by key index below
      kreg[i] = *reinterpret_cast<const f32x4*>(p.kc + toff + (size_t)kr * kHeadDim + kc4 * 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int vc4 = (tid >> 4) + 16 * i;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int vkey = 2 * vj + h;
        const int vr = kt * KT + vkey <= kmax ? vkey : kmax - kt * KT;
        vreg[2 * i + h] = *reinterpret_cast<const f32x4*>(p.vc + toff + (size_t)vr * kHeadDim + vc4 * 4);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int kkey = idx >> 5, kc4 = idx & 31;
      const float kx[4] = {kreg[i].x, kreg[i].y, kreg[i].z, kreg[i].w};
      abf16x4_t kh, kl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)kx[e];
        kh[e] = h;
        kl[e] = (__bf16)(kx[e] - (float)h);
      }
      *reinterpret_cast<abf16x4_t*>(k_hi + kkey * KROW + kc4 * 8) = kh;
      *reinterpret_cast<abf16x4_t*>(k_lo + kkey * KROW + kc4 * 8) = kl;
    }
    typedef __bf16 abf16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int vc4 = (tid >> 4) + 16 * i;
      const float v0[4] = {vreg[2 * i].x, vreg[2 * i].y, vreg[2 * i].z, vreg[2 * i].w};
      const float v1[4] = {vreg[2 * i + 1].x, vreg[2 * i + 1].y, vreg[2 * i + 1].z, vreg[2 * i + 1].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        abf16x2_t h, l;
        h[0] = (__bf16)v0[e]; h[1] = (__bf16)v1[e];
        l[0] = (__bf16)(v0[e] - (float)h[0]); l[1] = (__bf16)(v1[e] - (float)h[1]);
        *reinterpret_cast<abf16x2_t*>(vt_hi + (vc4 * 4 + e) * VROW + vj * 4) = h;
        *reinterpret_cast<abf16x2_t*>(vt_lo + (vc4 * 4 + e) * VROW + vj * 4) = l;
      }
    }
  };
  const int NS = p.n_splits, split = blockIdx.z;
  if (split < nkt) load_tile(split);
  for (int kt = split; kt < nkt; kt += NS) {
    const int k0 = kt * KT;
    __syncthreads();                                        // the previous tile has been consumed by every wave
    store_tile();
    __syncthreads();
    if (kt + NS < nkt) load_tile(kt + NS);
    // the workgroup walks the keys its LAST query row sees; a wave whose 16 rows all lie before this tile has nothing to add
    // (every score masked): wave-uniform skip, the barriers above keep the cadence
    if (k0 > pos0 + q0 + wave * 16 + 15) continue;

    f32x4 s_acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) s_acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int off = (nb * 16 + arow) * KROW + ks * 64 + kq * 16;
        const abf16x8_t bh = *reinterpret_cast<const abf16x8_t*>(k_hi + off);
        const abf16x8_t bl = *reinterpret_cast<const abf16x8_t*>(k_lo + off);
        s_acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q_lo[ks], bh, s_acc[nb], 0, 0, 0);
        s_acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q_hi[ks], bl, s_acc[nb], 0, 0, 0);
        s_acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q_hi[ks], bh, s_acc[nb], 0, 0, 0);
      }
    }
    // online softmax in the log2 domain: identical to attn_prefill_mfma_kernel (C layout: column = key, 4 rows per lane)
    float mx[4], ps[4], alpha[4], m_ref[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mx[r] = -INFINITY;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float sc = s_acc[nb][r] * scale2;
        if (k0 + nb * 16 + ccol > qpos[r]) sc = -INFINITY;  // causal mask
        s_acc[nb][r] = sc;
        mx[r] = fmaxf(mx[r], sc);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], lane_xor1(mx[r]));
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], lane_xor2(mx[r]));
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], row_ror4(mx[r]));
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], row_ror8(mx[r]));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m_new = fmaxf(m_run[r], mx[r]);
      m_ref[r] = m_new == -INFINITY ? 0.f : m_new;           // a key split may start on a tile this row cannot see at all
      alpha[r] = __builtin_amdgcn_exp2f(m_run[r] - m_ref[r]);
      m_run[r] = m_new;
      ps[r] = 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float e = __builtin_amdgcn_exp2f(s_acc[nb][r] - m_ref[r]);
        s_acc[nb][r] = e;
        ps[r] += e;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) ps[r] += lane_xor1(ps[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) ps[r] += lane_xor2(ps[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) ps[r] += row_ror4(ps[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) ps[r] += row_ror8(ps[r]);
    // no running maximum of this wave's rows moved (every alpha == 1.0f exactly): the 32 rescaling multiplies would change nothing
    const bool moved = __builtin_amdgcn_ballot_w64(alpha[0] != 1.f || alpha[1] != 1.f || alpha[2] != 1.f || alpha[3] != 1.f) != 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      l_run[r] = l_run[r] * alpha[r] + ps[r];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) p_s[(crow0 + r) * PS + nb * 16 + ccol] = s_acc[nb][r];
    }
    if (moved) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int db = 0; db < 8; ++db) o_acc[db][r] *= alpha[r];
      }
    }
    __builtin_amdgcn_wave_barrier();                        // P is wave private: LDS ops of one wave stay in order
    // P as the A operand of P.V: query row arow, keys kq*8 .. kq*8+7 (two conflict-free ds_read_b128 of the [16][36] tile)
    abf16x8_t p_hi, p_lo;
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p_s + arow * PS + kq * 8), b = *reinterpret_cast<const f32x4*>(p_s + arow * PS + kq * 8 + 4);
      const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      split8(x, p_hi, p_lo);
    }
#pragma unroll
    for (int db = 0; db < 8; ++db) {                        // head dims db*16 .. db*16+15: B fragment = 8 keys of dim db*16 + arow
      const int off = (db * 16 + arow) * VROW + kq * 16;
      const abf16x8_t vh = *reinterpret_cast<const abf16x8_t*>(vt_hi + off);
      const abf16x8_t vl = *reinterpret_cast<const abf16x8_t*>(vt_lo + off);
      o_acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p_lo, vh, o_acc[db], 0, 0, 0);
      o_acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p_hi, vl, o_acc[db], 0, 0, 0);
      o_acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p_hi, vh, o_acc[db], 0, 0, 0);
    }
  }
  // C layout of the output: lane holds head dims db*16 + ccol (db = 0..7) of query rows crow0 + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qrow = q0 + wave * 16 + crow0 + r;
    if (qrow >= T) continue;
    if (NS > 1) {                                           // partial (unnormalised) result of this key split
      const size_t pi = ((size_t)qrow * p.n_q + hq) * NS + split;
#pragma unroll
      for (int db = 0; db < 8; ++db) p.part_o[pi * kHeadDim + db * 16 + ccol] = o_acc[db][r];
      if (ccol == 0) { p.part_ml[pi * 2] = m_run[r] * 0.6931471805599453f; p.part_ml[pi * 2 + 1] = l_run[r]; }   // m back to nats
    } else {
      const float inv = 1.0f / l_run[r];
      const size_t oi = ((size_t)qrow * p.n_q + hq) * kHeadDim;
#pragma unroll
      for (int db = 0; db < 8; ++db) {
#pragma clang fp contract(off)   // lo = split of the ROUNDED product, as chatts_split_bf16x2 of the float32 output would give
        const float v = o_acc[db][r] * inv;
        if (p.out_hi) {
          const __bf16 h = (__bf16)v;
          p.out_hi[oi + db * 16 + ccol] = __builtin_bit_cast(uint16_t, h);
          p.out_lo[oi + db * 16 + ccol] = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)h));
        } else {
          p.out[oi + db * 16 + ccol] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// prefill with K / V planes: the flash loop on TRANSPOSED tiles.  kv_planes_kernel splits the sequence's K / V rows into bf16 hi / lo
// planes ONCE per layer and chunk (attn_prefill_bf16x3_kernel does it once per (query tile, query head): 65 times per tile at T = 798);
// attn_prefill_planes_kernel then computes S^T = K.Q^T and O^T = V^T.P^T, which puts "query = lane & 15" on every operand:
//   * S^T in the C layout gives a lane 8 keys of ONE query: the row maximum is an in-lane maximum + two lane exchanges (xor 16, 32)
//     instead of 4 rows x 4 DPP steps, the row sum stays a per-lane partial until the very end, one alpha per lane;
//   * those 8 probabilities ARE the B fragment of V^T.P^T - P never goes through LDS.  The MFMA's K index is a free choice as long as
//     A and B agree: slot 8 g + j of lane group g means key 4 g + j (j < 4) or 16 + 4 g + (j - 4), and kv_planes_kernel writes the
//     V^T rows in that slot order, so the A fragment is still one 16-byte LDS read;
//   * O^T in the C layout gives a lane 4 consecutive head dims of its query: alpha applies as is, the output is 16-byte stores.
// Staging is 8 16-byte copies per thread.  Same three-pass bf16 split products as the kernel above; the sums run in another order, so
// the two agree to rounding (tests: both against the float64 reference).  Needs the host to know pos0 (plane count) and a free
// workspace; otherwise the kernel above runs.
// Plane layout, per (kv head, 32-key tile) 32 KB: K_hi [32 keys][128], K_lo, V^T_hi [128 dims][32 slots], V^T_lo; rows past the last
// key repeat it (finite values behind the causal mask).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kv_planes_kernel(AttnParams p, int n_keys, uint16_t* planes) {
  constexpr int KT = 32;
  const int tid = threadIdx.x, kt = blockIdx.x, hk = blockIdx.y;
  const KvLayout kvl{p.table, p.n_kv, p.max_ctx, p.log_block};
  const size_t toff = kv_tile_off(kvl, hk, kt * KT);
  uint16_t* dst = planes + ((size_t)hk * gridDim.x + kt) * (4 * KT * kHeadDim);
  const int last = n_keys - 1 - kt * KT;                    // last valid row inside this tile (>= 0)
#pragma unroll
  for (int i = 0; i < 4; ++i) {                             // K: thread = (key, 4 dims), rows stay rows
    const int idx = tid + i * 256;
    const int key = idx >> 5, c4 = idx & 31;
    const int kr = key <= last ? key : last;
    const f32x4 v = *reinterpret_cast<const f32x4*>(p.kc + toff + (size_t)kr * kHeadDim + c4 * 4);
    const float x[4] = {v.x, v.y, v.z, v.w};
    abf16x4_t h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 hh = (__bf16)x[e];
      h[e] = hh;
      l[e] = (__bf16)(x[e] - (float)hh);
    }
    *reinterpret_cast<abf16x4_t*>(dst + key * kHeadDim + c4 * 4) = h;
    *reinterpret_cast<abf16x4_t*>(dst + KT * kHeadDim + key * kHeadDim + c4 * 4) = l;
  }
  // V transposed: thread = (head dim, 16 keys kg*16 ..); a wave reads 64 consecutive dims of one key (256 contiguous bytes) per load.
  // Keys kg*16 + 4 g + j (j = 0..3) go to slots 8 g + 4 kg + j.
  const int d = tid & 127, kg = tid >> 7;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    abf16x4_t h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = kg * 16 + 4 * g + j;
      const int vr = key <= last ? key : last;
      const float x = p.vc[toff + (size_t)vr * kHeadDim + d];
      const __bf16 hh = (__bf16)x;
      h[j] = hh;
      l[j] = (__bf16)(x - (float)hh);
    }
    *reinterpret_cast<abf16x4_t*>(dst + 2 * KT * kHeadDim + d * KT + 8 * g + 4 * kg) = h;
    *reinterpret_cast<abf16x4_t*>(dst + 3 * KT * kHeadDim + d * KT + 8 * g + 4 * kg) = l;
  }
}

__global__ __launch_bounds__(256, 2) void attn_prefill_planes_kernel(AttnParams p) {
  // LDS rows WITHOUT padding, 16-byte chunks XOR-swizzled for ds_read_b128's lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...:
  // every group holds each qi once, with g = g0 for eight of them and g0 + 1 for the other eight): K chunk c of key k sits at
  // c ^ (k & 15), V^T chunk c of dim d at c ^ ((d >> 2) & 2) - each group then covers all 64 banks exactly once (272- / 80-byte padded
  // rows, right for 16 consecutive lanes, gave every group a 2-way conflict: 44 % of the LDS cycles).
  constexpr int KT = 32, KROW = 256, VROW = 64;
  __shared__ __attribute__((aligned(16))) char k_hi[KT * KROW];
  __shared__ __attribute__((aligned(16))) char k_lo[KT * KROW];
  __shared__ __attribute__((aligned(16))) char vt_hi[kHeadDim * VROW];
  __shared__ __attribute__((aligned(16))) char vt_lo[kHeadDim * VROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = p.n_q / p.n_kv;
  // XCD-aware order.  Workgroups go to the 8 XCDs round robin by linear index, each XCD has its own 4 MB L2, and every workgroup
  // streams its kv head's planes (0.8 MB at 798 keys; 233 MB of tile reads per launch in total).  With (query tile, query head) as
  // the grid every XCD saw all kv heads - 6.4 MB through a 4 MB L2.  Here workgroup L works for kv head (L % 8) % n_kv: an XCD keeps
  // ONE kv head's planes (8 / n_kv XCDs share a head), and the items of a head - its G query heads x query tiles, heaviest first - are
  // dealt to those XCDs in turn.  (Measured at 798 keys: the locality itself is worth nothing yet - 43.0 against 42.6 us for a plain
  // head-fastest order - what counts is that the heaviest tiles of ALL heads start first: 47.8 us with (query tile, head) as a 2-D grid.)
  int hk, item;
  if (p.xcd_share > 0) {
    const int c = blockIdx.x & 7;
    hk = c % p.n_kv;
    item = (blockIdx.x >> 3) * p.xcd_share + c / p.n_kv;
    if (item >= G * p.units) return;
  } else {
    hk = (blockIdx.x % p.n_q) / G;
    item = (blockIdx.x / p.n_q) * G + (blockIdx.x % p.n_q) % G;
  }
  const int xi = item / G, hq = hk * G + item % G;
  const int heads = p.n_q + 2 * p.n_kv;
  const int T = p.t, pos0 = p.pos0;
  const int qt = p.units - 1 - xi;                          // heaviest query tiles (most key tiles) first, over ALL heads
  const int q0 = qt * 64;
  const int qi = lane & 15, g = lane >> 4;                  // this lane's query (of the wave's 16) and its group of 8 K-values / 4 C rows
  const float scale2 = 0.08838834764831845f * 1.4426950408889634f;   // d^-1/2 * log2(e): the softmax runs on exp2

  abf16x8_t q_hi[4], q_lo[4];                                // B fragments of Q^T: query qi, dims ks*32 + g*8 .. +7
  {
    int qrow = q0 + wave * 16 + qi;
    if (qrow > T - 1) qrow = T - 1;                         // padded rows compute garbage that is never stored
    const float* qp = p.qkv + ((size_t)qrow * heads + hq) * kHeadDim + g * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(qp + ks * 32), b = *reinterpret_cast<const f32x4*>(qp + ks * 32 + 4);
      const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      split8(x, q_hi[ks], q_lo[ks]);
    }
  }
  const int qpos = pos0 + q0 + wave * 16 + qi;
  float m_run = -INFINITY, l_part = 0.f;                    // l_part: this lane's 8 slots only; the four groups meet after the loop
  f32x4 o_acc[8];                                           // O^T: head dims db*16 + 4 g + r of query qi
#pragma unroll
  for (int db = 0; db < 8; ++db) o_acc[db] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int last_q = q0 + 63 < T - 1 ? q0 + 63 : T - 1;
  const int nkt = (pos0 + last_q) / KT + 1;                 // tiles up to the last key any row of this workgroup may see
  // 512 16-byte chunks per 8 KB plane: thread takes chunks tid and tid + 256.  K chunk j: key j >> 4, column j & 15; V^T chunk j: dim
  // j >> 2, column j & 3.  Loads run TWO tiles ahead in two register sets: a tile's matrix work (~1.4k cycles) does not cover an L2
  // round trip.
  f32x4 kreg[2][4], vreg[2][4];
  const uint16_t* src0 = p.kv_planes + (size_t)hk * p.kv_plane_tiles * (4 * KT * kHeadDim) + tid * 8;
  auto load_tile = [&](int kt, f32x4 (&kr)[4], f32x4 (&vr)[4]) {
    if (kt >= nkt) return;
    const uint16_t* src = src0 + (size_t)kt * (4 * KT * kHeadDim);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      kr[i] = *reinterpret_cast<const f32x4*>(src + i * 2048);
      kr[2 + i] = *reinterpret_cast<const f32x4*>(src + KT * kHeadDim + i * 2048);
      vr[i] = *reinterpret_cast<const f32x4*>(src + 2 * KT * kHeadDim + i * 2048);
      vr[2 + i] = *reinterpret_cast<const f32x4*>(src + 3 * KT * kHeadDim + i * 2048);
    }
  };
  auto store_tile = [&](int kt, const f32x4 (&kr)[4], const f32x4 (&vr)[4]) {
    if (kt >= nkt) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = tid + i * 256;
      const int ko = (j >> 4) * KROW + (((j & 15) ^ ((j >> 4) & 15)) << 4);
      const int vo = (j >> 2) * VROW + (((j & 3) ^ ((j >> 4) & 2)) << 4);         // (d >> 2) & 2 with d = j >> 2
      *reinterpret_cast<f32x4*>(k_hi + ko) = kr[i];
      *reinterpret_cast<f32x4*>(k_lo + ko) = kr[2 + i];
      *reinterpret_cast<f32x4*>(vt_hi + vo) = vr[i];
      *reinterpret_cast<f32x4*>(vt_lo + vo) = vr[2 + i];
    }
  };
  auto tile = [&](int kt) {
    const int k0 = kt * KT;
    if (kt >= nkt || k0 > pos0 + q0 + wave * 16 + 15) return;      // no tile / every score of this wave masked: wave-uniform skip

    f32x4 s_acc[2];                                         // S^T: keys k0 + nb*16 + 4 g + r, query qi
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) s_acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int off = (nb * 16 + qi) * KROW + (((ks * 4 + g) ^ qi) << 4);
        const abf16x8_t ah = *reinterpret_cast<const abf16x8_t*>(k_hi + off);
        const abf16x8_t al = *reinterpret_cast<const abf16x8_t*>(k_lo + off);
        s_acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, q_lo[ks], s_acc[nb], 0, 0, 0);
        s_acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, q_hi[ks], s_acc[nb], 0, 0, 0);
        s_acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, q_hi[ks], s_acc[nb], 0, 0, 0);
      }
    }
    const bool diag = k0 + KT - 1 > pos0 + q0 + wave * 16;  // wave-uniform: only such a tile holds keys some query of this wave cannot see
    float e[8];
    float mx = -INFINITY;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sc = s_acc[nb][r] * scale2;
        if (diag && k0 + nb * 16 + 4 * g + r > qpos) sc = -INFINITY;   // causal mask
        e[nb * 4 + r] = sc;
        mx = fmaxf(mx, sc);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float m_ref = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_ref);
    m_run = m_new;
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      e[j] = __builtin_amdgcn_exp2f(e[j] - m_ref);
      ps += e[j];
    }
    l_part = l_part * alpha + ps;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {    // some running maximum of this wave moved
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        o_acc[db].x *= alpha; o_acc[db].y *= alpha; o_acc[db].z *= alpha; o_acc[db].w *= alpha;
      }
    }
    abf16x8_t p_hi, p_lo;                                    // B fragment of P^T: slots 8 g .. 8 g + 7 of query qi
    split8(e, p_hi, p_lo);
#pragma unroll
    for (int db = 0; db < 8; ++db) {                        // head dims db*16 .. db*16+15: A fragment = 8 slots of dim db*16 + qi
      const int off = (db * 16 + qi) * VROW + ((g ^ ((qi >> 2) & 2)) << 4);
      const abf16x8_t vh = *reinterpret_cast<const abf16x8_t*>(vt_hi + off);
      const abf16x8_t vl = *reinterpret_cast<const abf16x8_t*>(vt_lo + off);
      o_acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, p_lo, o_acc[db], 0, 0, 0);
      o_acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, p_hi, o_acc[db], 0, 0, 0);
      o_acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, p_hi, o_acc[db], 0, 0, 0);
    }
  };
  load_tile(0, kreg[0], vreg[0]);
  load_tile(1, kreg[1], vreg[1]);
  for (int kt = 0; kt < nkt; kt += 2) {
    __syncthreads();                                        // the previous tile has been consumed by every wave
    store_tile(kt, kreg[0], vreg[0]);
    __syncthreads();
    load_tile(kt + 2, kreg[0], vreg[0]);
    tile(kt);
    if (kt + 1 >= nkt) break;
    __syncthreads();
    store_tile(kt + 1, kreg[1], vreg[1]);
    __syncthreads();
    load_tile(kt + 3, kreg[1], vreg[1]);
    tile(kt + 1);
  }
  float l = l_part + __shfl_xor(l_part, 16);
  l += __shfl_xor(l, 32);
  const int qrow = q0 + wave * 16 + qi;
  if (qrow >= T) return;
  const float inv = 1.0f / l;
  const size_t oi = ((size_t)qrow * p.n_q + hq) * kHeadDim + 4 * g;
#pragma unroll
  for (int db = 0; db < 8; ++db) {
#pragma clang fp contract(off)   // lo = split of the ROUNDED product, as chatts_split_bf16x2 of the float32 output would give
    const float v[4] = {o_acc[db].x * inv, o_acc[db].y * inv, o_acc[db].z * inv, o_acc[db].w * inv};
    if (p.out_hi) {
      abf16x4_t h, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16 hh = (__bf16)v[j];
        h[j] = hh;
        lo[j] = (__bf16)(v[j] - (float)hh);
      }
      *reinterpret_cast<abf16x4_t*>(p.out_hi + oi + db * 16) = h;
      *reinterpret_cast<abf16x4_t*>(p.out_lo + oi + db * 16) = lo;
    } else {
      *reinterpret_cast<f32x4*>(p.out + oi + db * 16) = (f32x4){v[0], v[1], v[2], v[3]};
    }
  }
}

// merge the split partials of attn_rows_kernel (splits that saw no tile hold m = -inf, l = 0)
__global__ __launch_bounds__(128) void attn_combine_kernel(AttnParams p) {
  const int hq = blockIdx.x, row = blockIdx.y, d = threadIdx.x;
  const size_t base = ((size_t)row * p.n_q + hq) * p.n_splits;
  float M = -INFINITY;
  for (int s = 0; s < p.n_splits; ++s) M = fmaxf(M, p.part_ml[(base + s) * 2]);
  float num = 0.f, den = 0.f;
  for (int s = 0; s < p.n_splits; ++s) {
    const float m = p.part_ml[(base + s) * 2];
    if (m == -INFINITY) continue;
    const float w = expf(m - M);
    num = fmaf(w, p.part_o[(base + s) * kHeadDim + d], num);
    den = fmaf(w, p.part_ml[(base + s) * 2 + 1], den);
  }
  const size_t oi = ((size_t)row * p.n_q + hq) * kHeadDim + d;
  const float v = num / den;
  if (p.out_hi) {                                 // the o_proj operand format, written where the value is made
    const __bf16 h = (__bf16)v;
    p.out_hi[oi] = __builtin_bit_cast(uint16_t, h);
    p.out_lo[oi] = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)h));
  } else {
    p.out[oi] = v;
  }
}

// Combine of the key-split MFMA prefill path (n_splits <= 4): one workgroup per query row, every thread merges float4s of
// the row's n_q x 128 outputs (the (head, row)-per-workgroup kernel above costs 16 us at T = 798 just in launch count).
__global__ __launch_bounds__(256) void attn_combine_rows_kernel(AttnParams p) {
  const int row = blockIdx.x, NS = p.n_splits;
  const int per_row = p.n_q * (kHeadDim / 4);
  for (int q = threadIdx.x; q < per_row; q += 256) {
    const int hq = q / (kHeadDim / 4), d4 = (q % (kHeadDim / 4)) * 4;
    const size_t base = ((size_t)row * p.n_q + hq) * NS;
    float M = -INFINITY;
    for (int s = 0; s < NS; ++s) M = fmaxf(M, p.part_ml[(base + s) * 2]);
    f32x4 num = {0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    for (int s = 0; s < NS; ++s) {
      const float m = p.part_ml[(base + s) * 2];
      if (m == -INFINITY) continue;
      const float w = expf(m - M);
      const f32x4 o = *reinterpret_cast<const f32x4*>(p.part_o + (base + s) * kHeadDim + d4);
      num.x = fmaf(w, o.x, num.x); num.y = fmaf(w, o.y, num.y); num.z = fmaf(w, o.z, num.z); num.w = fmaf(w, o.w, num.w);
      den = fmaf(w, p.part_ml[(base + s) * 2 + 1], den);
    }
    const size_t oi = ((size_t)row * p.n_q + hq) * kHeadDim + d4;
    const float v[4] = {num.x / den, num.y / den, num.z / den, num.w / den};
    if (p.out_hi) {
      typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
      bf16x4v hv, lv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const __bf16 h = (__bf16)v[j];
        hv[j] = h;
        lv[j] = (__bf16)(v[j] - (float)h);
      }
      *reinterpret_cast<bf16x4v*>(p.out_hi + oi) = hv;
      *reinterpret_cast<bf16x4v*>(p.out_lo + oi) = lv;
    } else {
      *reinterpret_cast<f32x4*>(p.out + oi) = (f32x4){v[0], v[1], v[2], v[3]};
    }
  }
}

}  // namespace chatts

using namespace chatts;

extern "C" size_t chatts_attn_workspace(int t, int n_q, int n_splits) {
  if (n_splits < 1) n_splits = 1;      // the decode kernel always emits partials, even with one slot
  return (size_t)t * n_q * n_splits * (kHeadDim + 2) * sizeof(float);
}

static int bind_workspace(AttnParams& p, void* workspace, size_t workspace_bytes) {
  const size_t need = chatts_attn_workspace(p.t, p.n_q, p.n_splits);
  CHATTS_REQUIRE(workspace && workspace_bytes >= need, CHATTS_E_WORKSPACE,
                 "attention: needs %zu workspace bytes, got %zu", need, workspace_bytes);
  p.part_o = reinterpret_cast<float*>(workspace);
  p.part_ml = p.part_o + (size_t)p.t * p.n_q * p.n_splits * kHeadDim;
  return CHATTS_OK;
}

// cache -> kernel parameters; the block-paged form is validated here (every entry point goes through this)
static int bind_cache(AttnParams& p, const ChattsKvCache* cache) {
  p.kc = cache->k; p.vc = cache->v; p.max_ctx = cache->max_ctx;
  p.table = cache->block_table; p.log_block = 0; p.table_stride = cache->table_stride;
  if (cache->block_table) {
    p.log_block = kv_log_block(cache->block_size);
    CHATTS_REQUIRE(p.log_block > 0, CHATTS_E_BADARG, "attention: block_size %d is not a power of two in 64..32768", cache->block_size);
    CHATTS_REQUIRE(cache->max_ctx % cache->block_size == 0, CHATTS_E_BADARG,
                   "attention: a paged cache of %d positions needs whole blocks of %d", cache->max_ctx, cache->block_size);
  }
  return CHATTS_OK;
}

namespace chatts {
int attention_impl(const float* qkv, int t, int n_q, int n_kv, int pos0, const int32_t* pos0_dev, const ChattsKvCache* cache, float* out,
                   uint16_t* out_hi, uint16_t* out_lo, int n_splits, void* workspace, size_t workspace_bytes, chatts_stream_t stream);
}  // namespace chatts

extern "C" int chatts_attention(const float* qkv, int t, int n_q, int n_kv, int pos0, const int32_t* pos0_dev,
                                const ChattsKvCache* cache, float* out, int n_splits, void* workspace,
                                size_t workspace_bytes, chatts_stream_t stream) {
  return chatts::attention_impl(qkv, t, n_q, n_kv, pos0, pos0_dev, cache, out, nullptr, nullptr, n_splits, workspace,
                                workspace_bytes, stream);
}

// chatts_attention with an optional plane output (internal: the decoder's prefill feeds o_proj's LDS-DMA GEMM directly).
// Planes are written by the MFMA kernel / its combine only (t >= 16, n_splits <= 4): the caller checks that.
int chatts::attention_impl(const float* qkv, int t, int n_q, int n_kv, int pos0, const int32_t* pos0_dev,
                           const ChattsKvCache* cache, float* out, uint16_t* out_hi, uint16_t* out_lo, int n_splits,
                           void* workspace, size_t workspace_bytes, chatts_stream_t stream) {
  CHATTS_REQUIRE(t >= 0 && n_q > 0 && n_kv > 0 && n_splits >= 1 && n_splits <= kMaxSlots, CHATTS_E_BADARG,
                 "attention: bad sizes");
  if (t == 0) return CHATTS_OK;
  CHATTS_REQUIRE(qkv && (out || (out_hi && out_lo)) && cache && cache->k && cache->v, CHATTS_E_BADARG, "attention: null pointer");
  CHATTS_REQUIRE(!out_hi || (t >= 16 && n_splits <= 4), CHATTS_E_BADARG, "attention: plane output needs the MFMA path");
  CHATTS_REQUIRE(n_q % n_kv == 0 && n_q / n_kv <= kMaxGroup, CHATTS_E_SHAPE,
                 "attention: GQA group %d/%d unsupported (max %d)", n_q, n_kv, kMaxGroup);
  if (!pos0_dev)
    CHATTS_REQUIRE(pos0 >= 0 && pos0 + t <= cache->max_ctx, CHATTS_E_SHAPE, "attention: positions exceed the cache");
  AttnParams p{};
  p.qkv = qkv; p.out = out; p.pos0_dev = pos0_dev; p.pos0 = pos0;
  p.t = t; p.n_q = n_q; p.n_kv = n_kv; p.n_splits = n_splits;
  if (const int rc = bind_cache(p, cache)) return rc;
  p.out_hi = out_hi; p.out_lo = out_lo;
  if (n_splits > 1) {
    const int rc = bind_workspace(p, workspace, workspace_bytes);
    if (rc) return rc;
  }
  const bool force_rows = opt_get(OPT_ATTN_ROWS, 0) != 0;   // debugging aid: VALU kernel for every T
  if (t >= 16 && n_splits <= 4 && (!force_rows || out_hi)) {
    // (64-key tiles were measured too: 155 us vs 148 us at T = 798 - the per-tile work is not what is slow)
    const int bf16x3 = opt_get(OPT_ATTN_BF16X3, 1);      // 0: the float32-MFMA kernel
    // K / V split into bf16 planes once (kv_planes_kernel) when the caller's workspace is free (one key split) and holds them
    const bool planes_on = opt_get(OPT_ATTN_PLANES, 1) != 0;
    const int n_keys = pos0 + t, tiles = (n_keys + 31) / 32;
    const size_t plane_bytes = (size_t)tiles * n_kv * 4 * 32 * kHeadDim * sizeof(uint16_t);
    const bool planes = bf16x3 && planes_on && !pos0_dev && n_splits == 1 && t >= 64 && workspace && workspace_bytes >= plane_bytes &&
                        ((uintptr_t)workspace % 16) == 0;
    if (planes) {
      uint16_t* pl = static_cast<uint16_t*>(workspace);
      const int nqt = (t + 63) / 64;
      hipLaunchKernelGGL(kv_planes_kernel, dim3(tiles, n_kv), dim3(256), 0, as_stream(stream), p, n_keys, pl);
      CHATTS_CHECK_LAUNCH("kv_planes");
      p.kv_planes = pl; p.kv_plane_tiles = tiles;
      const bool xcd_on = opt_get(OPT_ATTN_XCD, 1) != 0;
      p.units = nqt;
      p.xcd_share = xcd_on && n_kv <= 8 && 8 % n_kv == 0 ? 8 / n_kv : 0;
      const int G = n_q / n_kv;
      const int grid = p.xcd_share ? 8 * ((G * p.units + p.xcd_share - 1) / p.xcd_share) : n_q * p.units;
      hipLaunchKernelGGL(attn_prefill_planes_kernel, dim3(grid), dim3(256), 0, as_stream(stream), p);
    } else if (bf16x3) hipLaunchKernelGGL(attn_prefill_bf16x3_kernel, dim3((t + 63) / 64, n_q, n_splits), dim3(256), 0, as_stream(stream), p);
    else hipLaunchKernelGGL(attn_prefill_mfma_kernel<32>, dim3((t + 63) / 64, n_q, n_splits), dim3(256), 0, as_stream(stream), p);
    CHATTS_CHECK_LAUNCH("attn_prefill_mfma");
    if (n_splits > 1) {
      hipLaunchKernelGGL(attn_combine_rows_kernel, dim3(t), dim3(256), 0, as_stream(stream), p);
      CHATTS_CHECK_LAUNCH("attn_combine_rows");
    }
    return CHATTS_OK;
  }
  hipLaunchKernelGGL(attn_rows_kernel, dim3(n_kv, t, n_splits), dim3(256), 0, as_stream(stream), p);
  CHATTS_CHECK_LAUNCH("attn_rows");
  if (n_splits > 1) {
    hipLaunchKernelGGL(attn_combine_kernel, dim3(n_q, t), dim3(128), 0, as_stream(stream), p);
    CHATTS_CHECK_LAUNCH("attn_combine");
  }
  return CHATTS_OK;
}

namespace chatts {
// chatts_attention_decode_batched with an optional plane output (internal: the decoder's batched step feeds o_proj's
// weight-streaming kernel directly from the combine)
int attention_decode_batched_impl(const float* qkv_raw, int batch, int n_q, int n_kv, const float* q_norm_w,
                                  const float* k_norm_w, float norm_eps, const float* cos_tab, const float* sin_tab, int pos,
                                  const int32_t* pos_dev, const ChattsKvCache* cache, size_t seq_stride, float* out,
                                  uint16_t* out_hi, uint16_t* out_lo, int n_splits, void* workspace, size_t workspace_bytes,
                                  chatts_stream_t stream) {
  CHATTS_REQUIRE(batch >= 1 && n_q > 0 && n_kv > 0 && n_splits >= 1 && n_splits <= kMaxSlots, CHATTS_E_BADARG,
                 "attention_decode: bad sizes (batch >= 1, 1 <= n_splits <= %d)", kMaxSlots);
  CHATTS_REQUIRE(qkv_raw && (out || (out_hi && out_lo)) && cos_tab && sin_tab && cache && cache->k && cache->v, CHATTS_E_BADARG,
                 "attention_decode: null pointer");
  CHATTS_REQUIRE((q_norm_w == nullptr) == (k_norm_w == nullptr), CHATTS_E_BADARG,
                 "attention_decode: q_norm and k_norm must both be set or both be null");
  CHATTS_REQUIRE(n_q % n_kv == 0 && n_q / n_kv <= kMaxGroup, CHATTS_E_SHAPE,
                 "attention: GQA group %d/%d unsupported (max %d)", n_q, n_kv, kMaxGroup);
  CHATTS_REQUIRE(batch == 1 || pos_dev, CHATTS_E_BADARG, "attention_decode: batch > 1 needs per-sequence positions on the device");
  if (!pos_dev)
    CHATTS_REQUIRE(pos >= 0 && pos < cache->max_ctx, CHATTS_E_SHAPE, "attention_decode: position exceeds the cache");
  AttnParams p{};
  p.qkv = qkv_raw; p.out = out; p.pos0_dev = pos_dev; p.pos0 = pos;
  p.t = batch; p.n_q = n_q; p.n_kv = n_kv; p.n_splits = n_splits;
  if (const int rc = bind_cache(p, cache)) return rc;
  p.q_norm_w = q_norm_w; p.k_norm_w = k_norm_w; p.cos_tab = cos_tab; p.sin_tab = sin_tab; p.eps = norm_eps;
  p.seq_stride = seq_stride; p.out_hi = out_hi; p.out_lo = out_lo; p.kv_round = kv_round_mode();
  const int rc = bind_workspace(p, workspace, workspace_bytes);
  if (rc) return rc;
  const dim3 grid(n_kv, n_splits, batch);
  // registers sized for the group (every head's arithmetic is the same in all instantiations); the groups of the shipped models
  // (Qwen2.5-14B: 5, Qwen3-8B: 4) get the form with the head count known to the compiler (a group of 8 spills in that form: not
  // built); option ATTN_EXACT=0: the run-time form
  const bool exact = opt_get(OPT_ATTN_EXACT, 1) != 0;
  const dim3 wave(64);
  switch (n_q / n_kv) {
    case 4: if (exact) { hipLaunchKernelGGL((attn_decode_kernel<4, true>), grid, wave, 0, as_stream(stream), p); break; }
    case 1: case 2: case 3: hipLaunchKernelGGL((attn_decode_kernel<4, false>), grid, wave, 0, as_stream(stream), p); break;
    case 5: if (exact) hipLaunchKernelGGL((attn_decode_kernel<5, true>), grid, wave, 0, as_stream(stream), p);
            else hipLaunchKernelGGL((attn_decode_kernel<5, false>), grid, wave, 0, as_stream(stream), p);
            break;
    case 6: hipLaunchKernelGGL((attn_decode_kernel<6, false>), grid, wave, 0, as_stream(stream), p); break;
    default: hipLaunchKernelGGL((attn_decode_kernel<kMaxGroup, false>), grid, wave, 0, as_stream(stream), p); break;
  }
  CHATTS_CHECK_LAUNCH("attn_decode");
  hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(n_q, batch), dim3(128), 0, as_stream(stream), p);
  CHATTS_CHECK_LAUNCH("attn_decode_combine");
  return CHATTS_OK;
}
}  // namespace chatts

extern "C" int chatts_attention_decode_batched(const float* qkv_raw, int batch, int n_q, int n_kv, const float* q_norm_w,
                                               const float* k_norm_w, float norm_eps, const float* cos_tab,
                                               const float* sin_tab, int pos, const int32_t* pos_dev,
                                               const ChattsKvCache* cache, size_t seq_stride, float* out, int n_splits,
                                               void* workspace, size_t workspace_bytes, chatts_stream_t stream) {
  return chatts::attention_decode_batched_impl(qkv_raw, batch, n_q, n_kv, q_norm_w, k_norm_w, norm_eps, cos_tab, sin_tab, pos,
                                               pos_dev, cache, seq_stride, out, nullptr, nullptr, n_splits, workspace,
                                               workspace_bytes, stream);
}

extern "C" int chatts_attention_decode_fused(const float* qkv_raw, int n_q, int n_kv, const float* q_norm_w,
                                             const float* k_norm_w, float norm_eps, const float* cos_tab,
                                             const float* sin_tab, int pos, const int32_t* pos_dev,
                                             const ChattsKvCache* cache, float* out, int n_splits, void* workspace,
                                             size_t workspace_bytes, chatts_stream_t stream) {
  return chatts_attention_decode_batched(qkv_raw, 1, n_q, n_kv, q_norm_w, k_norm_w, norm_eps, cos_tab, sin_tab, pos, pos_dev,
                                         cache, 0, out, n_splits, workspace, workspace_bytes, stream);
}

