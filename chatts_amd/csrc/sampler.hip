// sampler.hip - temperature / top-k / top-p sampling of the next token on the GPU.
// The reference's evaluation drivers do not decode greedily: SamplingParams(temperature=0.2)
// (NetManAIOps/ChatTS chatts/utils/inference_tsmllm_vllm.py:43-46, inference_tsmllm_deepspeed.py:95-100) and
// SamplingParams(temperature=0.5, top_p=0.95) (chatts/utils/llm_utils.py:94,153).  The sampler itself lives in vLLM / HF
// (NOT IN REFERENCE); the rule restated here (and in oracle/sampler.py) is vLLM's: logits / temperature, keep the top_k
// largest, then the smallest set of most probable tokens whose mass reaches top_p, renormalise, draw.
//
// One workgroup of 1024 threads per sequence; the logits (152 k floats, L2-resident right after lm_head) are re-read per
// pass instead of being sorted:
//   max -> [top-k: 3-pass radix select of the k-th largest logit by COUNT] -> Z = sum e_i, e_i = exp((l_i - max)/T)
//       -> [top-p: 3-pass radix select of the cut e* by MASS: largest e* with mass{e_i >= e*} >= top_p * Z]
//       -> draw: first token (ascending id) whose running mass exceeds u * Z_kept.
// Masses are accumulated as 64-bit fixed point (e_i * 2^40, e_i <= 1): integer sums are order-independent, so the kept
// set and the drawn token do not depend on atomic ordering or reduction trees - same seed, same logits => same token.
// Ties at either cut are all kept (a sort keeps an arbitrary subset of them).
// u comes from the counter-based hash used for the synthetic weights: (seed, sequence, draw counter) -> 24 bits.
#include "common.h"

namespace chatts {

struct SampleParams {
  const float* logits;
  int64_t stride, vocab, vocab_offset;
  float inv_temp, top_p;
  int top_k;
  uint32_t seed;
  int64_t* token;
  float* token_logit;
  int64_t* out_tokens;
  int64_t out_stride;
  int32_t* step;
  int32_t* pos;
  int pos_limit;
  int32_t* n_kept;
  float* kept_mass;
  // per-row mode (ChattsSamplingArgs.temperature_rows != NULL): row b draws with (temperature, top_k, top_p, seed)[b], read on the
  // device at run time; temperature 0 = that row decodes greedily
  const float* temp_rows;
  const int32_t* topk_rows;
  const float* topp_rows;
  const uint32_t* seed_rows;
};

__device__ __forceinline__ uint32_t smix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// ascending uint32 order == ascending float order (finite values and infinities)
__device__ __forceinline__ uint32_t order_key(float f) {
  const uint32_t b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ unsigned long long mass_q(float e) { return (unsigned long long)(e * 1099511627776.0f); }   // e * 2^40

constexpr int kBins = 2048;

// Largest bin b with S(b) = sum_{j >= b} hist[j] >= target (target >= 1, <= total); *above = S(b) - hist[b].
// Thread t owns bins 2047-2t and 2046-2t; inclusive scan over threads = suffix sums from the top bin downwards.
__device__ void suffix_select(const unsigned long long* hist, unsigned long long target, unsigned long long* wave_tot,
                              int* sel_bin, unsigned long long* sel_above) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int j0 = kBins - 1 - 2 * t, j1 = j0 - 1;
  const unsigned long long h0 = hist[j0], h1 = hist[j1], local = h0 + h1;
  unsigned long long incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  unsigned long long base = 0;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  incl += base;
  const unsigned long long excl = incl - local;
  if (incl >= target && excl < target) {            // exactly one thread
    if (excl + h0 >= target) { *sel_bin = j0; *sel_above = excl; }
    else { *sel_bin = j1; *sel_above = excl + h0; }
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void sample_kernel(SampleParams p) {
  __shared__ unsigned long long hist[kBins];
  __shared__ unsigned long long wave_tot[16];
  __shared__ float fred[16];
  __shared__ int sel_bin;
  __shared__ unsigned long long sel_above;
  __shared__ int cnt_red[16];
  const int seq = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* logits = p.logits + (size_t)seq * p.stride;
  const int64_t V = p.vocab;
  const int32_t step = p.step ? p.step[seq] : 0;     // draw counter; read before the first barrier, rewritten at the very end
  float inv_temp = p.inv_temp, top_p = p.top_p;
  int top_k = p.top_k;
  uint32_t seed = p.seed, seq_salt = (uint32_t)seq;
  if (p.temp_rows) {
    const float tr = p.temp_rows[seq];
    if (!(tr > 0.f)) {                               // a greedy row among sampling ones: torch.argmax's token, same side effects
      __shared__ float gsv[16];
      __shared__ int64_t gsi[16];
      float best;
      int64_t bi;
      block_argmax_first(logits, V, gsv, gsi, best, bi);
      if (t == 0) {
        const int64_t tok = bi + p.vocab_offset;
        if (p.token) p.token[seq] = tok;
        if (p.token_logit) p.token_logit[seq] = best;
        if (p.out_tokens && p.step && (p.out_stride == 0 || step < p.out_stride)) p.out_tokens[(size_t)seq * p.out_stride + step] = tok;
        if (p.step) p.step[seq] = step + 1;
        if (p.pos && p.pos[seq] >= 0 && (p.pos_limit <= 0 || p.pos[seq] < p.pos_limit)) p.pos[seq] += 1;
        if (p.n_kept) p.n_kept[seq] = 1;
        if (p.kept_mass) p.kept_mass[seq] = 0.f;
      }
      return;
    }
    inv_temp = 1.0f / tr;
    top_p = p.topp_rows ? p.topp_rows[seq] : 1.0f;
    if (!(top_p > 0.f) || top_p > 1.0f) top_p = 1.0f;
    top_k = p.topk_rows ? p.topk_rows[seq] : 0;
    seed = p.seed_rows ? p.seed_rows[seq] : 0u;
    seq_salt = 0u;                                   // the row's own seed identifies the request: its tokens do not depend on the slot
  }

  // ---- max
  float m = -INFINITY;
  for (int64_t i = t; i < V; i += 1024) m = fmaxf(m, logits[i]);
  m = wave_max(m);
  if (lane == 0) fred[wave] = m;
  __syncthreads();
  m = fred[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) m = fmaxf(m, fred[w]);
  __syncthreads();

  // ---- top-k: the k-th largest logit by count (key of the cut; everything >= it is kept)
  uint32_t k_cut = 0;                                   // order_key >= 0 keeps everything
  if (top_k > 0 && (int64_t)top_k < V) {
    unsigned long long want = (unsigned long long)top_k;
    uint32_t prefix = 0;
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
      for (int b = t; b < kBins; b += 1024) hist[b] = 0;
      __syncthreads();
      const int sh = shifts[pass];
      const uint32_t dmask = (1u << widths[pass]) - 1u;
      const uint32_t hi_mask = pass == 0 ? 0u : ~((1u << (sh + widths[pass])) - 1u);
      for (int64_t i = t; i < V; i += 1024) {
        const uint32_t key = order_key(logits[i]);
        if ((key & hi_mask) == prefix) atomicAdd(&hist[(key >> sh) & dmask], 1ull);
      }
      __syncthreads();
      suffix_select(hist, want, wave_tot, &sel_bin, &sel_above);
      prefix |= (uint32_t)sel_bin << sh;
      want -= sel_above;
      __syncthreads();
    }
    k_cut = prefix;
  }

  // ---- Z over the top-k set, as fixed point
  const float it = inv_temp;
  unsigned long long zq = 0;
  for (int64_t i = t; i < V; i += 1024) {
    const float l = logits[i];
    if (order_key(l) >= k_cut) zq += mass_q(expf((l - m) * it));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) zq += __shfl_xor(zq, o, 64);
  if (lane == 0) wave_tot[wave] = zq;
  __syncthreads();
  zq = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) zq += wave_tot[w];
  __syncthreads();

  // ---- top-p: the cut e* by mass
  uint32_t e_cut = 0;                                   // float bits of e* (e >= 0: bit order == value order)
  if (top_p < 1.0f && zq > 0) {
    unsigned long long want = (unsigned long long)((double)top_p * (double)zq);
    if (want < 1) want = 1;
    if (want > zq) want = zq;
    uint32_t prefix = 0;
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
      for (int b = t; b < kBins; b += 1024) hist[b] = 0;
      __syncthreads();
      const int sh = shifts[pass];
      const uint32_t dmask = (1u << widths[pass]) - 1u;
      const uint32_t hi_mask = pass == 0 ? 0u : ~((1u << (sh + widths[pass])) - 1u);
      for (int64_t i = t; i < V; i += 1024) {
        const float l = logits[i];
        if (order_key(l) < k_cut) continue;
        const float e = expf((l - m) * it);
        const uint32_t key = __float_as_uint(e);
        const unsigned long long q = mass_q(e);
        if (q != 0 && (key & hi_mask) == prefix) atomicAdd(&hist[(key >> sh) & dmask], q);
      }
      __syncthreads();
      suffix_select(hist, want, wave_tot, &sel_bin, &sel_above);
      prefix |= (uint32_t)sel_bin << sh;
      want -= sel_above;
      __syncthreads();
    }
    e_cut = prefix;
  }

  // ---- draw.  Wave w owns the contiguous id range [w * seg, (w + 1) * seg); pass A: its kept mass and count.
  const int64_t seg = (V + 15) / 16;
  const int64_t s0 = wave * seg, s1 = s0 + seg < V ? s0 + seg : V;
  unsigned long long wq = 0;
  int wc = 0;
  for (int64_t i = s0 + lane; i < s1; i += 64) {
    const float l = logits[i];
    if (order_key(l) < k_cut) continue;
    const float e = expf((l - m) * it);
    if (__float_as_uint(e) < e_cut) continue;
    const unsigned long long q = mass_q(e);
    wq += q;
    wc += q != 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { wq += __shfl_xor(wq, o, 64); wc += __shfl_xor(wc, o, 64); }
  if (lane == 0) { wave_tot[wave] = wq; cnt_red[wave] = wc; }
  __syncthreads();
  unsigned long long total = 0, before = 0;
  int kept = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) { total += wave_tot[w]; kept += cnt_red[w]; }
  const uint32_t r = smix32(seed ^ smix32(seq_salt * 0x9e3779b9u + (uint32_t)step * 0x85ebca6bu + 0x68bc21ebu)) >> 8;
  // target in [0, total): (total * r) >> 24
  const unsigned long long target = __umul64hi(total, (unsigned long long)r << 40);
  int my_wave = -1;
  for (int w = 0; w < 16; ++w) {
    if (my_wave < 0 && before + wave_tot[w] > target) my_wave = w;
    if (my_wave < 0) before += wave_tot[w];
  }
  // pass B: the owning wave walks its range, 64 ids at a time, with a wave-level inclusive scan
  if (wave == my_wave) {
    unsigned long long run = before;
    int64_t chosen = -1;
    float chosen_logit = 0.f;
    for (int64_t base = s0; base < s1 && chosen < 0; base += 64) {
      const int64_t i = base + lane;
      unsigned long long q = 0;
      float l = 0.f;
      if (i < s1) {
        l = logits[i];
        if (order_key(l) >= k_cut) {
          const float e = expf((l - m) * it);
          if (__float_as_uint(e) >= e_cut) q = mass_q(e);
        }
      }
      unsigned long long incl = q;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      const bool hit = q != 0 && run + incl > target && run + incl - q <= target;
      const unsigned long long ball = __ballot(hit);
      if (ball) {
        const int src = __ffsll((long long)ball) - 1;
        chosen = __shfl(i, src, 64);
        chosen_logit = __shfl(l, src, 64);
      }
      run += __shfl(incl, 63, 64);
    }
    if (lane == 0) {
      if (chosen < 0) chosen = s0;                       // unreachable when total > 0; keeps the id in range otherwise
      const int64_t tok = chosen + p.vocab_offset;
      if (p.token) p.token[seq] = tok;
      if (p.token_logit) p.token_logit[seq] = chosen_logit;
      if (p.out_tokens && p.step && (p.out_stride == 0 || step < p.out_stride)) p.out_tokens[(size_t)seq * p.out_stride + step] = tok;
      if (p.step) p.step[seq] = step + 1;
      if (p.pos && p.pos[seq] >= 0 && (p.pos_limit <= 0 || p.pos[seq] < p.pos_limit)) p.pos[seq] += 1;   // parked slots (pos < 0) stay parked
      if (p.n_kept) p.n_kept[seq] = kept;
      if (p.kept_mass) p.kept_mass[seq] = zq ? (float)((double)total / (double)zq) : 0.f;
    }
  }
}

}  // namespace chatts

using namespace chatts;

extern "C" int chatts_sample_batched(const float* logits, int batch, int64_t logits_stride, int64_t vocab, int64_t vocab_offset,
                                     const ChattsSamplingArgs* sa, int64_t* token, float* token_logit, int64_t* out_tokens,
                                     int64_t out_stride, int32_t* step_dev, int32_t* pos_dev, int pos_limit,
                                     chatts_stream_t stream) {
  CHATTS_REQUIRE(logits && sa && vocab > 0 && batch >= 1 && logits_stride >= vocab, CHATTS_E_BADARG, "sample: bad arguments");
  CHATTS_REQUIRE(sa->temperature_rows || (sa->temperature > 0.f && sa->top_p > 0.f), CHATTS_E_BADARG,
                 "sample: temperature %g and top_p %g must be positive (temperature 0 = greedy: call chatts_argmax)",
                 (double)sa->temperature, (double)sa->top_p);
  CHATTS_REQUIRE(batch == 1 || out_tokens == nullptr || out_stride > 0, CHATTS_E_BADARG, "sample: out_stride missing");
  SampleParams p;
  p.logits = logits; p.stride = logits_stride; p.vocab = vocab; p.vocab_offset = vocab_offset;
  p.inv_temp = sa->temperature > 0.f ? 1.0f / sa->temperature : 1.0f; p.top_p = sa->top_p; p.top_k = sa->top_k; p.seed = sa->seed;
  p.temp_rows = sa->temperature_rows; p.topk_rows = sa->top_k_rows; p.topp_rows = sa->top_p_rows; p.seed_rows = sa->seed_rows;
  p.token = token; p.token_logit = token_logit; p.out_tokens = out_tokens; p.out_stride = out_stride;
  p.step = step_dev; p.pos = pos_dev; p.pos_limit = pos_limit; p.n_kept = sa->n_kept; p.kept_mass = sa->kept_mass;
  hipLaunchKernelGGL(sample_kernel, dim3(batch), dim3(1024), 0, as_stream(stream), p);
  CHATTS_CHECK_LAUNCH("sample");
  return CHATTS_OK;
}
