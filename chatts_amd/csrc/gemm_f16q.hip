// gemm_f16q.hip - the prefill projection on the 1.5-pass operand split (round 6):  C = epilogue(A . W^T) with A in the f16q format (f16q.h):
//     acc += W16 . hi^T    v_mfma_f32_16x16x32_f16            every 32 K-values   (f16 x f16: exact products, the bf16 weights are exact in f16)
//     acc += W8 . q^T      v_mfma_scale_f32_16x16x128_f8f6f4  every 128 K-values  (e4m3 residual x e4m3 weights, e8m0 scales: the CDNA4 block-scaled pipe)
// Per (16 x 16 tile, 128 K-values) 4 x 16 + 32 = 96 matrix-pipe cycles against 8 x 16 = 128 of the bf16 hi + bf16 lo kernel (gemm_ring.hip);
// on GEMM-like operand data the pipe alone runs the mix 24 % faster at the same power-bound clock (profiles/r6_mfma_mix_probe.txt).
//
// Structure: the round-5 ring (four 32-deep half-stage slots requested three half-steps ahead with COUNTED vmcnt, persistent workgroups over
// units = (K split, 256-row W panel, M-tile of f <= 9 fragments), operands swapped so that a lane holds 4 consecutive output columns)
// with two changes the new arithmetic forces:
//  * EIGHT waves, all of them computing, each issuing its share (<= 7) of a half-stage's LDS-DMA pieces behind the half-step barrier.  The
//    e4m3 fragments of a 128-deep group have to be collected over four half-steps (8 registers per fragment, 72 per wave) and the f16
//    pass has no lo sweep left to cover its fragment reads: the wave needs ~230 registers, which three waves per SIMD (four loaders beside
//    eight compute waves, 168 registers each) cannot have.
//  * The K index of the block-scaled MFMA is permuted: lane group g (lane / 16) of BOTH operands takes bytes 8 g .. 8 g + 7 of every
//    quarter-stage's 32-byte row - the instruction pairs equal (lane group, byte) positions of A and B, so any assignment of K-values
//    to them is the same dot product - and the four quarter-stages of a group land in register pairs 0-1 .. 6-7 as they arrive: one
//    ds_read_b64 per fragment and half-step, no e4m3 data has to stay in LDS beyond its half-step.  All four lane groups of a row then
//    hold values of all 128 K: the activation scale is one e8m0 byte per (row, 128 K-values), the weight scale one byte per W row.
// Slot (38 KB): A_hi [9][16 rows][64 B] | W16 [16][16][64 B] (16-byte chunks XOR-swizzled on the source side as in gemm_ring.hip) |
// A8 [10][2 halves][16 rows][16 B] (five fragment pairs) | W8 [16][2][16][16 B] (a lane's 8 bytes: half = g / 2, offset 8 (g % 2): conflict-free ds_read_b64).
// Behind the ring: scale bytes of the activation rows [2 groups][256] and of the panel's W rows [2 units][256] (1-byte LDS-DMA pieces,
// counted with their half-stage), and the epilogue's scratch for the block maximum of the SwiGLU output.
#include <type_traits>

#include "f16q.h"
#include "gemm_common.h"
#include "ring_store.h"

namespace chatts {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int kQMaxF = 9, kQSlots = 4;
constexpr int kQAhi = 0, kQW16 = kQMaxF * 1024, kQA8 = kQW16 + 16 * 1024, kQW8 = kQA8 + 5 * 1024, kQSlot = kQW8 + 16 * 512;      // 38 KB
constexpr int kQRing = kQSlots * kQSlot;                     // 155648
// behind the ring: scale bytes as the 1-byte LDS-DMA leaves them - ONE DWORD PER LANE (the byte in its low 8 bits) - for the activation
// rows of a tile [2 groups][3 pieces x 64 lanes] and the W rows of a panel [2 units][4 x 64]; then the SwiGLU epilogue's scratch
constexpr int kQAScBuf = 3 * 256, kQWScBuf = 4 * 256;
constexpr int kQASc = kQRing, kQWSc = kQASc + 2 * kQAScBuf, kQScr = kQWSc + 2 * kQWScBuf;
constexpr int kQLds = kQScr + 8 * 80 * 4;                    // 161792 of the 163840 bytes
constexpr int kQThreads = 512;

struct F16qGemm {
  GemmParams g;            // m n k ldc epilogue k_per_split direct bias resid c (a / w unused)
  const _Float16* a_hi;
  const uint8_t* a_lo8;
  const uint8_t* a_sc;
  int lda, ldsc;
  int a_tiled;             // a_hi / a_lo8 in the tiled plane layout of f16q.h (lda == K)
  const _Float16* w16;
  const uint8_t* w8;
  const uint8_t* w8e;      // e8m0 byte per W row
  int ldw;
  int w_tiled;             // w16 in the layout of chatts_tile_bf16 (16-bit elements), w8 in that of chatts_tile_e4m3: a piece = 1 KB of memory
  _Float16* c_hi;          // SwiGLU output in the f16q format (down_proj's operand), or null
  uint8_t* c_lo8;
  uint8_t* c_sc;
  int ldcp, ldcsc;
  int c_tiled;             // the SwiGLU plane output goes out tiled (over ldcp = N / 2 columns)
};

// Both MFMAs as inline asm with the accumulator TIED (D = C in place).  Through the builtins hipcc (ROCm 7.2) gave most products a
// destination different from their C operand - the accumulators migrated through the register file, the 4-fragment form needed 256
// registers + 28 spilled where its live values are ~190 (profiles/r6_f16q_build_notes.txt).  What hipcc does not do for an asm statement
// (guide 5.7): the wait states between the last MFMA and a VALU read of its result - f16q_mfma_drain() before the epilogue.
__device__ __forceinline__ void mfma_h(f32x4& c, const f16x8_t& w, const f16x8_t& a) {      // D = W . A^T (operands swapped)
  asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(a));
}
template <int J>
__device__ __forceinline__ void mfma_q(f32x4& c, const i32x8& w, const i32x8& a, int sw, int sa) {
  // srcA = the e4m3 W fragment (scale: byte J of sw), srcB = the e4m3 residual fragment (scale: byte 0 of sa); formats default to e4m3
  if constexpr (J == 0) asm("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(w), "v"(a), "v"(sw), "v"(sa));
  if constexpr (J == 1) asm("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,0,0] op_sel_hi:[0,0,0]" : "+v"(c) : "v"(w), "v"(a), "v"(sw), "v"(sa));
  if constexpr (J == 2) asm("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[1,0,0]" : "+v"(c) : "v"(w), "v"(a), "v"(sw), "v"(sa));
  if constexpr (J == 3) asm("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(c) : "v"(w), "v"(a), "v"(sw), "v"(sa));
}
__device__ __forceinline__ void f16q_mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// SwiGLU output of one unit as f16q planes: the unit's 128 output columns are ONE scale block per token, spread over the four waves of a
// wave row - block maximum through LDS (one barrier, executed by all eight waves)
__device__ __forceinline__ void f16q_store_swiglu(const F16qGemm& q, const f32x4 (&acc)[5][4], int fml, int tok0, int fb0, int lane, int wave,
                                                  int panel, char* smem) {
  const GemmParams& p = q.g;
  const int tl = lane & 15, fq = (lane >> 4) * 4;
  const F16qPlanes cp{q.c_hi, q.c_lo8, q.c_sc, q.ldcp, q.ldcsc, q.c_tiled};
  float* scr = reinterpret_cast<float*>(smem + kQScr);
  f32x4 bg[2], bu[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int fb = fb0 + h * 32 + fq;
    if (fb + 19 >= p.n) fb = 0;
    bg[h] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + fb) : (f32x4){0.f, 0.f, 0.f, 0.f};
    bu[h] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + fb + 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float lo[5][2][4];
  f16x4_t hv[5][2];
  {
#pragma clang fp contract(off)      // lo is the residual of the ROUNDED float32 value
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i >= fml) continue;                                    // (wave-uniform)
      float amax = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = silu_g(acc[i][2 * h][r] + bg[h][r]) * (acc[i][2 * h + 1][r] + bu[h][r]);
          const _Float16 hh = f16q_hi(v);
          hv[i][h][r] = hh;
          lo[i][h][r] = v - (float)hh;
          amax = fmaxf(amax, fabsf(lo[i][h][r]));
        }
      amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      if (lane < 16) scr[wave * 80 + i * 16 + tl] = amax;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int wrow = (wave >> 2) * 4;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    if (i < fml) {
      const int tok = tok0 + i * 16 + tl;
      const float* s = scr + i * 16 + tl;
      const float amax = fmaxf(fmaxf(s[wrow * 80], s[(wrow + 1) * 80]), fmaxf(s[(wrow + 2) * 80], s[(wrow + 3) * 80]));
      const int E = f16q_exp(amax);
      const float inv = f16q_inv(E);
      if (tok < p.m) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int fb = fb0 + h * 32 + fq;
          const int ocol = (fb >> 5) * 16 + fq;
          if (fb + 19 < p.n) {
            *reinterpret_cast<f16x4_t*>(q.c_hi + f16q_hi_off(cp, tok, ocol)) = hv[i][h];
            *reinterpret_cast<uint32_t*>(q.c_lo8 + f16q_lo_off(cp, tok, ocol)) = f16q_pack4(lo[i][h][0], lo[i][h][1], lo[i][h][2], lo[i][h][3], inv);
          }
        }
        if ((wave & 3) == 0 && lane < 16) q.c_sc[(size_t)tok * q.ldcsc + panel] = (uint8_t)f16q_byte(E);
      }
    }
  }
}

// MAXFML: the largest per-wave fragment count this instantiation carries (tiles of <= 2 MAXFML fragments).  The 5-fragment form needs ~40
// registers more (accumulators + e4m3 fragments); tiles of <= 8 fragments run the leaner instantiation.
template <int MAXFML>
__global__ __launch_bounds__(kQThreads) void gemm_f16q_kernel(F16qGemm q, RingGeom g) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const GemmParams& p = q.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // units of this workgroup: as gemm_ring_kernel (XCD x owns a contiguous eighth of the unit sequence, its workgroups interleave)
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int share = g.units >> 3, rem = g.units & 7;
  const int ubeg = xcd * share + (xcd < rem ? xcd : rem), ucnt = share + (xcd < rem);
  if (local >= ucnt) return;
  const int uend = ubeg + ucnt, u0 = ubeg + local, ustep = g.wpx;

  // ---- this wave's share of the LDS-DMA pieces (1 KB per wave instruction) of a half-stage ------------------------------------------
  // f16 pieces: 16 rows x 64 B - W16 fragments {wave, wave + 8}, A_hi fragments {wave, wave + 8} < f
  // e4m3 pieces: 2 fragments x [2 halves][16 rows][16 B] - W8 fragment pair `wave`, A8 fragment pair 7 - wave (< ceil(f / 2))
  // scale pieces (64 x 1 byte): activation rows 64 pc .. of the tile with a group's first quarter (pc = 0, 1, 2 on waves 2, 5, 6),
  //                             W rows 64 wave .. of the panel with the unit's first half-stage (waves 0 .. 3)
  const int prow = lane >> 2, gchunk = (lane & 3) ^ ((lane >> 5) << 1);
  const int f8frag = lane >> 5, f8half = (lane >> 4) & 1, f8row = lane & 15;
  const char* const b_w16 = reinterpret_cast<const char*>(q.w16);
  const char* const b_ahi = reinterpret_cast<const char*>(q.a_hi);
  const char* const b_w8 = reinterpret_cast<const char*>(q.w8);
  const char* const b_a8 = reinterpret_cast<const char*>(q.a_lo8);
  const int nkt = p.k >> 5;                                      // half-stages of the whole K (tiled operands)
  const int pcA = wave == 2 ? 0 : (wave == 5 ? 1 : (wave == 6 ? 2 : -1));
  uint32_t o_w16[2], o_ahi[2], o_w8 = 0, o_a8 = 0, o_asc = 0, o_wsc = 0;
  int n_ahi = 0, n_a8 = 0, n_asc = 0, nh_cur = 0, hh = 0, ucur = u0;
  int gq = 0, uq = 0;                                            // running group / unit count of the DMA side (scale buffer parity)
  auto setup = [&](int u) {
    const RingUnit r = ring_unit(p, g, u);
    nh_cur = r.nh;
    const int n0 = r.panel * kRingPanel;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (q.w_tiled) {
        int rb = (n0 >> 4) + wave + 8 * h;
        const int rbmax = ((p.n + 15) >> 4) - 1;
        if (rb > rbmax) rb = rbmax;
        o_w16[h] = (uint32_t)(((size_t)rb * nkt + (r.kbeg >> 5)) * 1024 + lane * 16);
      } else {
        int wr = n0 + (wave + 8 * h) * 16 + prow;
        if (wr > p.n - 1) wr = p.n - 1;
        o_w16[h] = (uint32_t)(((size_t)wr * q.ldw + r.kbeg) * 2 + gchunk * 16);
      }
      const int fr = wave + 8 * h;
      if (q.a_tiled) {
        const int fb = (r.m0 >> 4) + (fr < r.f ? fr : r.f - 1);
        o_ahi[h] = (uint32_t)(((size_t)fb * nkt + (r.kbeg >> 5)) * 1024 + lane * 16);
      } else {
        int am = r.m0 + fr * 16 + prow;
        if (am > p.m - 1) am = p.m - 1;
        o_ahi[h] = (uint32_t)(((size_t)am * q.lda + r.kbeg) * 2 + gchunk * 16);
      }
    }
    n_ahi = r.f > wave + 8 ? 2 : (r.f > wave ? 1 : 0);
    {
      if (q.w_tiled) {
        int pb = (n0 >> 5) + wave;
        const int pbmax = ((p.n + 31) >> 5) - 1;
        if (pb > pbmax) pb = pbmax;
        o_w8 = (uint32_t)(((size_t)pb * nkt + (r.kbeg >> 5)) * 1024 + lane * 16);
      } else {
        int wr = n0 + (2 * wave + f8frag) * 16 + f8row;
        if (wr > p.n - 1) wr = p.n - 1;
        o_w8 = (uint32_t)((size_t)wr * q.ldw + r.kbeg + f8half * 16);
      }
      const int pr = 7 - wave;
      if (q.a_tiled) {
        int fr = 2 * pr + f8frag;
        if (fr > r.f - 1) fr = r.f - 1;
        o_a8 = (uint32_t)(((size_t)((r.m0 >> 4) + fr) * nkt + (r.kbeg >> 5)) * 512 + (lane & 31) * 16);
      } else {
        int am = r.m0 + (2 * pr + f8frag) * 16 + f8row;
        if (am > p.m - 1) am = p.m - 1;
        o_a8 = (uint32_t)((size_t)am * q.lda + r.kbeg + f8half * 16);
      }
      n_a8 = 2 * pr < r.f ? 1 : 0;
    }
    n_asc = (pcA >= 0 && pcA * 64 < r.f * 16) ? 1 : 0;
    if (n_asc) {
      int am = r.m0 + pcA * 64 + lane;
      if (am > p.m - 1) am = p.m - 1;
      o_asc = (uint32_t)((size_t)am * q.ldsc + (r.kbeg >> 7));
    }
    if (wave < 4) {
      int wr = n0 + wave * 64 + lane;
      if (wr > p.n - 1) wr = p.n - 1;
      o_wsc = (uint32_t)wr;
    }
  };
  int gi = 0;                                                    // global index of the next half-stage to request
  // Requesting the next half-stage is spread over the half-step: issue_begin (roll over into the next unit if need be; FAST: the caller
  // guarantees the half-stage belongs to the current unit), then the wave's pieces ONE AT A TIME between the rows of its MFMA sweeps
  // (an LDS-DMA instruction holds the wave's issue port for ~60-150 cycles: behind four queued MFMAs the matrix pipe keeps running, and
  // the requests reach the memory pipeline as a steady stream instead of 40 KB behind every barrier), then issue_end -> the piece count.
  // Pieces: 0, 1 = W16 fragments wave, wave + 8; 2 = W8 pair; 3, 4 = A_hi fragments wave, wave + 8; 5 = A8 pair; 6 = activation scale
  // bytes (a group's first quarter); 7 = W scale bytes (a unit's first half-stage).
  bool live = true;
  char* ibase = smem;
  uint32_t ik16 = 0, ik8 = 0, ikw16 = 0, ikw8 = 0;
  const uint32_t wstep16 = q.w_tiled ? 1024u : 64u, wstep8 = q.w_tiled ? 1024u : 32u, astep16 = q.a_tiled ? 1024u : 64u, astep8 = q.a_tiled ? 512u : 32u;
  auto issue_begin = [&](auto fast_c) {
    constexpr bool FAST = decltype(fast_c)::value;
    if constexpr (!FAST) {
      if (hh == nh_cur) {
        ucur += ustep;
        hh = 0;
        if (ucur < uend) setup(ucur);
      }
      live = ucur < uend;
    }
    ibase = smem + (gi & (kQSlots - 1)) * kQSlot;
    ik16 = (uint32_t)hh * astep16;
    ik8 = (uint32_t)hh * astep8;
    ikw16 = (uint32_t)hh * wstep16;
    ikw8 = (uint32_t)hh * wstep8;
  };
  auto piece = [&](auto p_c, auto sq_c) {
    constexpr int P = decltype(p_c)::value, SQ = decltype(sq_c)::value;
    if (!live) return;
    if constexpr (P == 0) __builtin_amdgcn_global_load_lds((gptr_t)(b_w16 + o_w16[0] + ikw16), (lptr_t)(ibase + kQW16 + wave * 1024), 16, 0, 0);
    if constexpr (P == 1) __builtin_amdgcn_global_load_lds((gptr_t)(b_w16 + o_w16[1] + ikw16), (lptr_t)(ibase + kQW16 + (wave + 8) * 1024), 16, 0, 0);
    if constexpr (P == 2) __builtin_amdgcn_global_load_lds((gptr_t)(b_w8 + o_w8 + ikw8), (lptr_t)(ibase + kQW8 + wave * 1024), 16, 0, 0);
    if constexpr (P == 3) { if (n_ahi > 0) __builtin_amdgcn_global_load_lds((gptr_t)(b_ahi + o_ahi[0] + ik16), (lptr_t)(ibase + kQAhi + wave * 1024), 16, 0, 0); }
    if constexpr (P == 4) { if (n_ahi > 1) __builtin_amdgcn_global_load_lds((gptr_t)(b_ahi + o_ahi[1] + ik16), (lptr_t)(ibase + kQAhi + (wave + 8) * 1024), 16, 0, 0); }
    if constexpr (P == 5) { if (n_a8) __builtin_amdgcn_global_load_lds((gptr_t)(b_a8 + o_a8 + ik8), (lptr_t)(ibase + kQA8 + (7 - wave) * 1024), 16, 0, 0); }
    if constexpr (P == 6) {
      if (SQ == 0 || (SQ < 0 && (hh & 3) == 0)) {
        if (n_asc) __builtin_amdgcn_global_load_lds((gptr_t)(q.a_sc + o_asc + (hh >> 2)), (lptr_t)(smem + kQASc + (gq & 1) * kQAScBuf + pcA * 256), 1, 0, 0);
      }
    }
    if constexpr (P == 7) {
      if (hh == 0 && wave < 4) __builtin_amdgcn_global_load_lds((gptr_t)(q.w8e + o_wsc), (lptr_t)(smem + kQWSc + (uq & 1) * kQWScBuf + wave * 256), 1, 0, 0);
    }
  };
  auto issue_end = [&](auto sq_c) -> int {
    constexpr int SQ = decltype(sq_c)::value;
    if (!live) return 0;
    int n = 3 + n_ahi + n_a8;
    if (SQ == 0 || (SQ < 0 && (hh & 3) == 0)) {
      n += n_asc;
      ++gq;
      if (hh == 0) {
        n += wave < 4 ? 1 : 0;
        ++uq;
      }
    }
    ++gi;
    ++hh;
    return n;
  };
  using SlowAny = std::integral_constant<int, -1>;
  auto issue_all = [&]() -> int {                                // a whole half-stage at once (prologue)
    issue_begin(std::false_type{});
    piece(std::integral_constant<int, 0>{}, SlowAny{}); piece(std::integral_constant<int, 1>{}, SlowAny{});
    piece(std::integral_constant<int, 2>{}, SlowAny{}); piece(std::integral_constant<int, 3>{}, SlowAny{});
    piece(std::integral_constant<int, 4>{}, SlowAny{}); piece(std::integral_constant<int, 5>{}, SlowAny{});
    piece(std::integral_constant<int, 6>{}, SlowAny{}); piece(std::integral_constant<int, 7>{}, SlowAny{});
    return issue_end(SlowAny{});
  };

  // ---- compute side: wave (wm, wn) holds f0 (wm = 0) or f - f0 fragments of the tile x 64 columns wn * 64 .. ------------------------------
  const int wm = wave >> 2, wn = wave & 3;
  const int lane_off = (lane & 15) * 64 + ((((lane >> 4) ^ (((lane >> 3) & 1) << 1))) << 4);      // this lane's 16-byte chunk in a 16-row f16 block
  const int f8_off = ((lane >> 5) & 1) * 256 + (lane & 15) * 16 + ((lane >> 4) & 1) * 8;           // this lane's 8 bytes in an e4m3 fragment
  setup(u0);
  (void)issue_all();                                             // half-stage 0
  int q1 = issue_all(), q2 = issue_all(), q3;
  ring_wait_vmcnt(q1 + q2);                                      // half-stage 0 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();                                  // ... and everybody's
  q3 = issue_all();                                              // half-stage 3 (slot 3 has never been read)
  // now outstanding: half-stages 1 (q1), 2 (q2), 3 (q3)
  int gh = 0, gr = 0, ur = 0;                                    // half-steps done; groups / units consumed (scale buffer parity)
  for (int u = u0; u < uend; u += ustep) {
    const RingUnit r = ring_unit(p, g, u);
    const int fml = wm ? r.f - r.f0 : r.f0;
    const int rowblk = wm ? r.f0 : 0;
    f32x4 acc[5][4];      // (rows >= MAXFML are never touched: the compiler drops them)
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_nop 4" ::: "memory");      // (VALU write -> MFMA C read)

    auto k_loop = [&](auto fml_c) {
      constexpr int FML = decltype(fml_c)::value;
      constexpr int FA = FML ? FML : 1;
      f16x8_t wf[4], ahi[FA];
      i32x8 a8[FA], w8[4];
      const int a_off = kQAhi + rowblk * 1024 + lane_off, w_off = kQW16 + wn * 4096 + lane_off;
      const int a8_off = kQA8 + rowblk * 512 + f8_off, w8_off = kQW8 + wn * 2048 + f8_off;
      auto slot = [&](int h) { return smem + ((gh + h) & (kQSlots - 1)) * kQSlot; };
      auto read_q = [&](const char* base, auto sq_c) {           // quarter-stage sq of the group -> register pair sq of every e4m3 fragment
        constexpr int SQ = decltype(sq_c)::value;
        if constexpr (FML > 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const i32x2 t = *reinterpret_cast<const i32x2*>(base + w8_off + j * 512);
            w8[j][2 * SQ] = t.x; w8[j][2 * SQ + 1] = t.y;
          }
#pragma unroll
          for (int i = 0; i < FML; ++i) {
            const i32x2 t = *reinterpret_cast<const i32x2*>(base + a8_off + i * 512);
            a8[i][2 * SQ] = t.x; a8[i][2 * SQ + 1] = t.y;
          }
        }
      };
      // the panel's W row scales of this wave's four column fragments, one byte each (byte j = fragment j)
      int sw = 0;
      if constexpr (FML > 0) {
        const uint8_t* ws = reinterpret_cast<const uint8_t*>(smem + kQWSc + (ur & 1) * kQWScBuf + (wn * 64 + (lane & 15)) * 4);
        sw = (int)ws[0] | ((int)ws[64] << 8) | ((int)ws[128] << 16) | ((int)ws[192] << 24);
      }
      // half-step h of the unit (S = h % 4): its f16 fragments are in registers; every fragment is re-read from half-stage h + 1 right
      // behind its last use in the sweep (A fragment i after row i, W fragment j after its MFMA of the last row): no second register set
      auto half = [&](int h, auto s_c, auto more_c, auto fast_c) {
        constexpr int S = decltype(s_c)::value;
        constexpr bool more = decltype(more_c)::value;
        ring_wait_vmcnt(q2 + q3);                                // this wave's pieces of half-stage gh + h + 1 have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every fragment read of half-stage gh + h has returned
        __builtin_amdgcn_s_barrier();                            // h + 1 is published; the slot of h is free
        issue_begin(fast_c);                                     // half-stage h + 4 -> the slot of h, piece by piece below
        const char* nb = slot(h + 1);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int kRows = FML * (S == 3 ? 2 : 1);            // MFMA rows of this half-step = interleave points
        auto piece_at = [&](auto row_c) {                        // the piece that rides behind MFMA row `row` (8 pieces over kRows rows, rest at the end)
          constexpr int ROW = decltype(row_c)::value;
          if constexpr (ROW < 8) piece(std::integral_constant<int, ROW>{}, s_c);
        };
        if constexpr (FML > 0) {
#pragma unroll
          for (int i = 0; i < FML; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              mfma_h(acc[i][j], wf[j], ahi[i]);
              if constexpr (more) if (i == FML - 1) wf[j] = *reinterpret_cast<const f16x8_t*>(nb + w_off + j * 1024);
            }
            if constexpr (more) ahi[i] = *reinterpret_cast<const f16x8_t*>(nb + a_off + i * 1024);
            if (i == 0) piece_at(std::integral_constant<int, 0>{});
            if (i == 1) piece_at(std::integral_constant<int, 1>{});
            if (i == 2) piece_at(std::integral_constant<int, 2>{});
            if (i == 3) piece_at(std::integral_constant<int, 3>{});
            if (i == 4) piece_at(std::integral_constant<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (S == 3) {                                // the group's residual pass
            const uint8_t* as = reinterpret_cast<const uint8_t*>(smem + kQASc + (gr & 1) * kQAScBuf + (rowblk * 16 + (lane & 15)) * 4);
            int sa[FML];
#pragma unroll
            for (int i = 0; i < FML; ++i) sa[i] = (int)as[i * 64];
#pragma unroll
            for (int i = 0; i < FML; ++i) {
              mfma_q<0>(acc[i][0], w8[0], a8[i], sw, sa[i]);
              mfma_q<1>(acc[i][1], w8[1], a8[i], sw, sa[i]);
              mfma_q<2>(acc[i][2], w8[2], a8[i], sw, sa[i]);
              mfma_q<3>(acc[i][3], w8[3], a8[i], sw, sa[i]);
              if (i == 0) piece_at(std::integral_constant<int, FML + 0>{});
              if (i == 1) piece_at(std::integral_constant<int, FML + 1>{});
              if (i == 2) piece_at(std::integral_constant<int, FML + 2>{});
              if (i == 3) piece_at(std::integral_constant<int, FML + 3>{});
              if (i == 4) piece_at(std::integral_constant<int, FML + 4>{});
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if constexpr (S == 3) ++gr;
        // the pieces that found no row
        if constexpr (kRows <= 0) piece(std::integral_constant<int, 0>{}, s_c);
        if constexpr (kRows <= 1) piece(std::integral_constant<int, 1>{}, s_c);
        if constexpr (kRows <= 2) piece(std::integral_constant<int, 2>{}, s_c);
        if constexpr (kRows <= 3) piece(std::integral_constant<int, 3>{}, s_c);
        if constexpr (kRows <= 4) piece(std::integral_constant<int, 4>{}, s_c);
        if constexpr (kRows <= 5) piece(std::integral_constant<int, 5>{}, s_c);
        if constexpr (kRows <= 6) piece(std::integral_constant<int, 6>{}, s_c);
        if constexpr (kRows <= 7) piece(std::integral_constant<int, 7>{}, s_c);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) read_q(nb, std::integral_constant<int, (S + 1) & 3>{});
        const int q4 = issue_end(s_c);
        q1 = q2; q2 = q3; q3 = q4;
      };
      using S0 = std::integral_constant<int, 0>;
      using S1 = std::integral_constant<int, 1>;
      using S2 = std::integral_constant<int, 2>;
      using S3 = std::integral_constant<int, 3>;
      {
        const char* b0 = slot(0);
        if constexpr (FML > 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const f16x8_t*>(b0 + w_off + j * 1024);
#pragma unroll
          for (int i = 0; i < FML; ++i) ahi[i] = *reinterpret_cast<const f16x8_t*>(b0 + a_off + i * 1024);
        }
        read_q(b0, S0{});
      }
      // (nh is a multiple of 4: K splits are whole 128-deep groups; the last group is peeled: its last half-step reads nothing ahead and
      //  the loop body carries no run-time test; the half-stages requested inside the loop - h + 4 - all belong to this unit)
      using Yes = std::true_type;
      using No = std::false_type;
      int h = 0;
      for (; h + 4 < r.nh; h += 4) {
        half(h, S0{}, Yes{}, Yes{});
        half(h + 1, S1{}, Yes{}, Yes{});
        half(h + 2, S2{}, Yes{}, Yes{});
        half(h + 3, S3{}, Yes{}, Yes{});
      }
      half(h, S0{}, Yes{}, No{});
      half(h + 1, S1{}, Yes{}, No{});
      half(h + 2, S2{}, Yes{}, No{});
      half(h + 3, S3{}, No{}, No{});
    };
    switch (fml) {
      case 0: k_loop(std::integral_constant<int, 0>{}); break;
      case 1: k_loop(std::integral_constant<int, 1>{}); break;
      case 2: k_loop(std::integral_constant<int, 2>{}); break;
      case 3: k_loop(std::integral_constant<int, 3>{}); break;
      case 4: k_loop(std::integral_constant<int, 4>{}); break;
      default:
        if constexpr (MAXFML >= 5) k_loop(std::integral_constant<int, 5>{});
        break;
    }
    gh += r.nh;
    ++ur;
    f16q_mfma_drain();

    const int tok0 = r.m0 + rowblk * 16, fb0 = r.panel * kRingPanel + wn * 64;
    if (!p.direct) ring_store<kStRaw, false>(p, acc, fml, tok0, fb0, lane, r.split);
    else if (p.epilogue == CHATTS_EPI_SWIGLU) {
      if (q.c_hi) f16q_store_swiglu(q, acc, fml, tok0, fb0, lane, wave, r.panel, smem);
      else ring_store<kStSwiglu, false>(p, acc, fml, tok0, fb0, lane, 0);
    } else if (p.epilogue == CHATTS_EPI_RESID) ring_store<kStResid, false>(p, acc, fml, tok0, fb0, lane, 0);
    else if (p.epilogue == CHATTS_EPI_GELU) ring_store<kStGelu, false>(p, acc, fml, tok0, fb0, lane, 0);
    else ring_store<kStNone, false>(p, acc, fml, tok0, fb0, lane, 0);
  }
}

// ---- the format's stand-alone producers ------------------------------------------------------------------------------------------------------
// x float32 [M, K] -> f16q planes (one workgroup per row; a 128-value block = 32 consecutive threads)
__device__ __forceinline__ float half_wave_max(float v) {      // max over the 32 lanes of this lane's half-wave
  v = row16_max(v);
  return fmaxf(v, __shfl_xor(v, 16, 64));
}
__global__ __launch_bounds__(256) void split_f16q_kernel(const float* __restrict__ x, int k, int ldx, F16qPlanes o) {
  const int row = blockIdx.x;
  const float* xr = x + (size_t)row * ldx;
  for (int c0 = 0; c0 < k; c0 += 1024) {
    const int col = c0 + threadIdx.x * 4;
    const bool live = col < k;                                   // (K is a multiple of 128: a half-wave is live or idle as a whole)
    const f32x4 v = live ? *reinterpret_cast<const f32x4*>(xr + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x4_t hv;
    float lo[4];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hv[j] = f16q_hi(v[j]);
      lo[j] = v[j] - (float)hv[j];
      amax = fmaxf(amax, fabsf(lo[j]));
    }
    amax = half_wave_max(amax);
    const int E = f16q_exp(amax);
    if (live) {
      *reinterpret_cast<f16x4_t*>(o.hi + f16q_hi_off(o, row, col)) = hv;
      *reinterpret_cast<uint32_t*>(o.lo8 + f16q_lo_off(o, row, col)) = f16q_pack4(lo[0], lo[1], lo[2], lo[3], f16q_inv(E));
      if ((threadIdx.x & 31) == 0) o.sc[(size_t)row * o.ldsc + (col >> 7)] = (uint8_t)f16q_byte(E);
    }
  }
}

// bf16 W [N, K] -> f16 copy (exact for |w| in [2^-24 x 2^7, 65504]) + e4m3 copy with ONE power-of-two scale per row (byte E + 127)
__global__ __launch_bounds__(256) void weights_f16q_kernel(const uint16_t* __restrict__ w, int k, int ldw, _Float16* __restrict__ w16,
                                                          uint8_t* __restrict__ w8, uint8_t* __restrict__ w8e, int ldo) {
  __shared__ float red[4];
  const int row = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint16_t* wr = w + (size_t)row * ldw;
  float amax = 0.f;
  for (int col = threadIdx.x * 4; col < k; col += 1024) {
    const u32x2 t = *reinterpret_cast<const u32x2*>(wr + col);
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(bf16_lo(t.x)), fabsf(bf16_hi(t.x))), fmaxf(fabsf(bf16_lo(t.y)), fabsf(bf16_hi(t.y)))));
  }
  amax = wave_max(amax);
  if (lane == 0) red[wave] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int E = f16q_exp(amax);
  const float inv = f16q_inv(E);
  if (threadIdx.x == 0) w8e[row] = (uint8_t)f16q_byte(E);
  for (int col = threadIdx.x * 4; col < k; col += 1024) {
    const u32x2 t = *reinterpret_cast<const u32x2*>(wr + col);
    const float v[4] = {bf16_lo(t.x), bf16_hi(t.x), bf16_lo(t.y), bf16_hi(t.y)};
    f16x4_t hv;
#pragma unroll
    for (int j = 0; j < 4; ++j) hv[j] = (_Float16)v[j];
    *reinterpret_cast<f16x4_t*>(w16 + (size_t)row * ldo + col) = hv;
    *reinterpret_cast<uint32_t*>(w8 + (size_t)row * ldo + col) = f16q_pack4(v[0], v[1], v[2], v[3], inv);
  }
}


// e4m3 [rows, k] row-major -> the tiled layout of the f16q kernel's e4m3 pieces: block (b = row / 32, t = k / 32) is the 1 KB at
// ((b * k / 32) + t) * 1 KB; inside it the 16-byte chunk at position l (0 .. 63) holds row 32 b + (l >> 5) * 16 + (l & 15), bytes
// 32 t + ((l >> 4) & 1) * 16 .. + 15 - the order in which the lanes of a fragment-pair piece deposit it.  Rows beyond the matrix repeat the last.
__global__ __launch_bounds__(256) void tile_e4m3_kernel(const uint8_t* __restrict__ src, int rows, int k, int ld, uint8_t* __restrict__ dst) {
  const int nkt = k >> 5;
  const size_t chunks = (size_t)((rows + 31) >> 5) * nkt * 64;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (size_t)gridDim.x * 256) {
    const int l = (int)(i & 63);
    const size_t blk = i >> 6;
    const int t = (int)(blk % nkt);
    int row = (int)(blk / nkt) * 32 + (l >> 5) * 16 + (l & 15);
    if (row > rows - 1) row = rows - 1;
    *reinterpret_cast<u32x4*>(dst + i * 16) = *reinterpret_cast<const u32x4*>(src + (size_t)row * ld + t * 32 + ((l >> 4) & 1) * 16);
  }
}

// one row's 4 consecutive values of this thread -> the row's f16q planes (a 128-value block = this thread's half-wave; every lane of the
// half-wave must call, `live` says whether its columns exist)
__device__ __forceinline__ void f16q_put4(const F16qPlanes& o, int row, int col, const float (&v)[4], bool live) {
#pragma clang fp contract(off)      // lo is the residual of the ROUNDED float32 value (v may be a product)
  f16x4_t hv;
  float lo[4];
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hv[j] = f16q_hi(v[j]);
    lo[j] = v[j] - (float)hv[j];
    amax = fmaxf(amax, fabsf(lo[j]));
  }
  amax = half_wave_max(live ? amax : 0.f);
  const int E = f16q_exp(amax);
  if (live) {
    *reinterpret_cast<f16x4_t*>(o.hi + f16q_hi_off(o, row, col)) = hv;
    *reinterpret_cast<uint32_t*>(o.lo8 + f16q_lo_off(o, row, col)) = f16q_pack4(lo[0], lo[1], lo[2], lo[3], f16q_inv(E));
    if ((threadIdx.x & 31) == 0) o.sc[(size_t)row * o.ldsc + (col >> 7)] = (uint8_t)f16q_byte(E);
  }
}

// RMSNorm with the row written as f16q planes (rmsnorm_kernel<true>'s arithmetic; one workgroup per row)
__global__ __launch_bounds__(256) void rmsnorm_f16q_kernel(const float* __restrict__ x, const float* __restrict__ w, int hidden, float eps, F16qPlanes o) {
  __shared__ float red[4];
  const float* xr = x + (size_t)blockIdx.x * hidden;
  float ss = 0.f;
  for (int k = threadIdx.x * 4; k < hidden; k += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + k);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = block_sum<4>(ss, red);
  const float rstd = rsqrtf(ss / (float)hidden + eps);
  for (int k0 = 0; k0 < hidden; k0 += 1024) {
    const int k = k0 + threadIdx.x * 4;
    const bool live = k < hidden;
    const f32x4 v = live ? *reinterpret_cast<const f32x4*>(xr + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 g = live ? *reinterpret_cast<const f32x4*>(w + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const float ov[4] = {g.x * (v.x * rstd), g.y * (v.y * rstd), g.z * (v.z * rstd), g.w * (v.w * rstd)};
    f16q_put4(o, blockIdx.x, k, ov, live);
  }
}

// Split-K epilogue of the residual projections (splitk_epilogue_norm_reg_kernel's arithmetic: slabs in split order, bias, residual, the
// row's sum of squares in column-group order) with the RMSNorm that follows written as f16q planes; norm_w == null: no planes
template <int kIt>
__global__ __launch_bounds__(256) void splitk_epilogue_f16q_kernel(const float* __restrict__ ws, int sk, int m, int n, const float* __restrict__ bias,
                                                                  const float* __restrict__ resid, float* __restrict__ c, int ldc, int epilogue,
                                                                  const float* __restrict__ norm_w, float eps, F16qPlanes o) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const size_t plane = (size_t)m * n;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 t[kIt][4], rs[kIt], gw[kIt], b4[kIt];
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int col = threadIdx.x * 4 + it * 1024;
    const bool live = col < n;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      t[it][s] = live ? *reinterpret_cast<const f32x4*>(ws + (s < sk ? s : sk - 1) * plane + (size_t)row * n + col) : zero;
    rs[it] = (live && epilogue == CHATTS_EPI_RESID) ? *reinterpret_cast<const f32x4*>(resid + (size_t)row * ldc + col) : zero;
    gw[it] = (live && norm_w) ? *reinterpret_cast<const f32x4*>(norm_w + col) : zero;
    b4[it] = (live && bias) ? *reinterpret_cast<const f32x4*>(bias + col) : zero;
  }
  f32x4 keep[kIt];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int col = threadIdx.x * 4 + it * 1024;
    f32x4 v = zero;
    if (col < n) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s < sk) { v.x += t[it][s].x; v.y += t[it][s].y; v.z += t[it][s].z; v.w += t[it][s].w; }
      if (bias) { v.x += b4[it].x; v.y += b4[it].y; v.z += b4[it].z; v.w += b4[it].w; }
      if (epilogue == CHATTS_EPI_RESID) { v.x = rs[it].x + v.x; v.y = rs[it].y + v.y; v.z = rs[it].z + v.z; v.w = rs[it].w + v.w; }
      *reinterpret_cast<f32x4*>(c + (size_t)row * ldc + col) = v;
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    keep[it] = v;
  }
  if (!norm_w) return;
  ss = block_sum<4>(ss, red);
  const float rstd = rsqrtf(ss / (float)n + eps);
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int col = threadIdx.x * 4 + it * 1024;
    const f32x4 v = keep[it], g = gw[it];
    const float ov[4] = {g.x * (v.x * rstd), g.y * (v.y * rstd), g.z * (v.z * rstd), g.w * (v.w * rstd)};
    f16q_put4(o, row, col, ov, col < n);
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------------------
// (T, sk) per shape: gemm_ring's cost model with this kernel's step (a 128-deep group of a tile with f fragments: f16 sweeps 4 x 16 f
// cycles per SIMD pair + the residual sweep 32 f, beside (24 + 1.5 f) KB of LDS-DMA), K splits in whole groups
void f16q_pick(int m, int n, int k, int cus, int force_t, int force_sk, RingGeom& g) {
  const int F = (m + 15) / 16, P = (n + kRingPanel - 1) / kRingPanel, nk = k / 128;
  g.F = F; g.P = P;
  const int slots = cus >= 8 ? (cus / 8) * 8 : 8;
  double best = 1e30;
  const int tmin = (F + kQMaxF - 1) / kQMaxF;
  int bt = tmin, bs = 1;
  int tmax = (F + 2) / 3;
  if (tmax < tmin) tmax = tmin;
  for (int T = tmin; T <= tmax; ++T) {
    const int fmax = (F + T - 1) / T;
    const double step = 2.0 * (180.0 * fmax + 900.0);            // cycles per 128-deep group (first guess; refitted from the sweep)
    for (int sk = 1; sk <= 4; ++sk) {                            // (the row-wise epilogue sums <= 4 slabs)
      if (sk > 1 && nk / sk < 2) break;
      const long long units = (long long)T * P * sk;
      const long long rounds = (units + slots - 1) / slots;
      const int steps = (nk + sk - 1) / sk;
      double t = (double)rounds * (steps * step + 9000.0);
      if (sk > 1) t += 12000.0 + (double)(sk + 2) * m * n * 4.0 / 2500.0;
      if (t < best) { best = t; bt = T; bs = sk; }
    }
  }
  if (force_t >= tmin && force_t <= F) bt = force_t;
  if (force_sk >= 1 && force_sk <= 4 && nk / force_sk >= 1) bs = force_sk;
  g.T = bt; g.sk = bs;
  g.units = bt * P * bs;
  int wpx = (g.units + 7) / 8;
  if (wpx > slots / 8) wpx = slots / 8;
  g.wpx = wpx;
}

int launch_f16q(const F16qGemm& q, const RingGeom& g, hipStream_t s) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16q_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, kQLds);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16q_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, kQLds);
    CHATTS_REQUIRE(e == hipSuccess, CHATTS_E_LAUNCH, "gemm_f16q: cannot reserve %d bytes of LDS: %s", kQLds, hipGetErrorString(e));
    configured = true;
  }
  const int fmax = (g.F + g.T - 1) / g.T;
  if (fmax <= 8) hipLaunchKernelGGL(gemm_f16q_kernel<4>, dim3(8 * g.wpx), dim3(kQThreads), kQLds, s, q, g);
  else hipLaunchKernelGGL(gemm_f16q_kernel<5>, dim3(8 * g.wpx), dim3(kQThreads), kQLds, s, q, g);
  return CHATTS_OK;
}

}  // namespace chatts

using namespace chatts;

extern "C" int chatts_split_f16q(const float* x, int m, int k, int ldx, chatts_f16* hi, uint8_t* lo8, uint8_t* scale, int ld_planes, int ld_scale,
                                 int tiled, chatts_stream_t stream) {
  CHATTS_REQUIRE(!tiled || ld_planes == k, CHATTS_E_SHAPE, "split_f16q: tiled planes need ld_planes == K");
  CHATTS_REQUIRE(m >= 0 && k > 0 && k % kF16qBlock == 0, CHATTS_E_SHAPE, "split_f16q: m=%d k=%d (K must be a multiple of 128)", m, k);
  if (m == 0) return CHATTS_OK;
  CHATTS_REQUIRE(x && hi && lo8 && scale, CHATTS_E_BADARG, "split_f16q: null pointer");
  CHATTS_REQUIRE(ldx >= k && ldx % 4 == 0 && ld_planes >= k && ld_planes % 16 == 0 && ld_scale >= k / kF16qBlock && ((uintptr_t)x % 16) == 0 &&
                     ((uintptr_t)hi % 16) == 0 && ((uintptr_t)lo8 % 16) == 0, CHATTS_E_SHAPE, "split_f16q: leading dimensions / alignment");
  F16qPlanes o{reinterpret_cast<_Float16*>(hi), lo8, scale, ld_planes, ld_scale, tiled};
  hipLaunchKernelGGL(split_f16q_kernel, dim3(m), dim3(256), 0, as_stream(stream), x, k, ldx, o);
  CHATTS_CHECK_LAUNCH("split_f16q");
  return CHATTS_OK;
}

extern "C" int chatts_weights_f16q(const chatts_bf16* w, int n, int k, int ldw, chatts_f16* w16, uint8_t* w8, uint8_t* w8_exp, int ld_out,
                                   chatts_stream_t stream) {
  CHATTS_REQUIRE(n >= 0 && k > 0 && k % 4 == 0, CHATTS_E_SHAPE, "weights_f16q: n=%d k=%d", n, k);
  if (n == 0) return CHATTS_OK;
  CHATTS_REQUIRE(w && w16 && w8 && w8_exp, CHATTS_E_BADARG, "weights_f16q: null pointer");
  CHATTS_REQUIRE(ldw >= k && ldw % 4 == 0 && ld_out >= k && ld_out % 16 == 0 && ((uintptr_t)w % 8) == 0 && ((uintptr_t)w16 % 16) == 0 &&
                     ((uintptr_t)w8 % 16) == 0, CHATTS_E_SHAPE, "weights_f16q: leading dimensions / alignment");
  hipLaunchKernelGGL(weights_f16q_kernel, dim3(n), dim3(256), 0, as_stream(stream), w, k, ldw, reinterpret_cast<_Float16*>(w16), w8, w8_exp, ld_out);
  CHATTS_CHECK_LAUNCH("weights_f16q");
  return CHATTS_OK;
}


static int f16q_splitk_epilogue(const ChattsLinearF16qArgs* a, int sk, hipStream_t s) {
  CHATTS_REQUIRE(sk <= 4 && a->n <= 8192 && a->n % 4 == 0 && a->epilogue != CHATTS_EPI_SWIGLU && a->epilogue != CHATTS_EPI_GELU, CHATTS_E_SHAPE,
                 "linear_f16q: split-K epilogue handles <= 4 slabs of <= 8192 columns (EPI_NONE / EPI_RESID); got sk=%d n=%d", sk, a->n);
  F16qPlanes o{reinterpret_cast<_Float16*>(a->post_hi), a->post_lo8, a->post_scale, a->ld_post, a->ld_pscale, a->planes_tiled};
  if (a->post_norm_w)
    CHATTS_REQUIRE(o.hi && o.lo8 && o.sc && a->n % kF16qBlock == 0 && a->ld_post >= a->n && a->ld_post % 4 == 0 && a->ld_pscale >= a->n / kF16qBlock,
                   CHATTS_E_SHAPE, "linear_f16q: post-norm planes need hi / lo8 / scale, N %% 128 == 0, ld_post >= N");
  const float* ws = reinterpret_cast<const float*>(a->workspace);
#define CHATTS_EPI_Q(K) hipLaunchKernelGGL((splitk_epilogue_f16q_kernel<K>), dim3(a->m), dim3(256), 0, s, ws, sk, a->m, a->n, a->bias, a->resid, a->c, \
                                           a->ldc, a->epilogue, a->post_norm_w, a->post_norm_eps, o)
  switch ((a->n + 1023) / 1024) {
    case 1: CHATTS_EPI_Q(1); break;
    case 2: CHATTS_EPI_Q(2); break;
    case 3: CHATTS_EPI_Q(3); break;
    case 4: CHATTS_EPI_Q(4); break;
    case 5: CHATTS_EPI_Q(5); break;
    case 6: CHATTS_EPI_Q(6); break;
    default: CHATTS_EPI_Q(8); break;
  }
#undef CHATTS_EPI_Q
  CHATTS_CHECK_LAUNCH("splitk_epilogue_f16q");
  return CHATTS_OK;
}

extern "C" size_t chatts_tile_e4m3_bytes(int rows, int k) { return rows > 0 && k > 0 ? (size_t)((rows + 31) / 32) * 32 * k : 0; }

extern "C" int chatts_tile_e4m3(const uint8_t* src, int rows, int k, int ld, uint8_t* dst, chatts_stream_t stream) {
  CHATTS_REQUIRE(rows >= 0 && k > 0 && k % 32 == 0 && ld >= k && ld % 16 == 0, CHATTS_E_SHAPE, "tile_e4m3: rows=%d k=%d ld=%d (K %% 32 == 0, ld %% 16 == 0)", rows, k, ld);
  if (rows == 0) return CHATTS_OK;
  CHATTS_REQUIRE(src && dst && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, CHATTS_E_BADARG, "tile_e4m3: null / unaligned pointer");
  const size_t chunks = chatts_tile_e4m3_bytes(rows, k) / 16;
  const unsigned blocks = (unsigned)((chunks + 255) / 256 < 16384 ? (chunks + 255) / 256 : 16384);
  hipLaunchKernelGGL(tile_e4m3_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, rows, k, ld, dst);
  CHATTS_CHECK_LAUNCH("tile_e4m3");
  return CHATTS_OK;
}

extern "C" int chatts_rmsnorm_f16q(const float* x, const float* w, chatts_f16* hi, uint8_t* lo8, uint8_t* scale, int ld_planes, int ld_scale, int t,
                                   int hidden, float eps, int tiled, chatts_stream_t stream) {
  CHATTS_REQUIRE(!tiled || ld_planes == hidden, CHATTS_E_SHAPE, "rmsnorm_f16q: tiled planes need ld_planes == hidden");
  CHATTS_REQUIRE(t >= 0 && hidden > 0 && hidden % kF16qBlock == 0, CHATTS_E_SHAPE, "rmsnorm_f16q: t=%d hidden=%d (a multiple of 128)", t, hidden);
  if (t == 0) return CHATTS_OK;
  CHATTS_REQUIRE(x && w && hi && lo8 && scale, CHATTS_E_BADARG, "rmsnorm_f16q: null pointer");
  CHATTS_REQUIRE(ld_planes >= hidden && ld_planes % 4 == 0 && ld_scale >= hidden / kF16qBlock, CHATTS_E_SHAPE, "rmsnorm_f16q: leading dimensions");
  F16qPlanes o{reinterpret_cast<_Float16*>(hi), lo8, scale, ld_planes, ld_scale, tiled};
  hipLaunchKernelGGL(rmsnorm_f16q_kernel, dim3(t), dim3(256), 0, as_stream(stream), x, w, hidden, eps, o);
  CHATTS_CHECK_LAUNCH("rmsnorm_f16q");
  return CHATTS_OK;
}

extern "C" size_t chatts_linear_f16q_workspace(int m, int n, int k) {
  if (m < 1 || n < 1 || k < 128 || k % 128) return 0;
  RingGeom g;
  f16q_pick(m, n, k, device_cus(), opt_get(OPT_GEMM_T, 0), opt_get(OPT_GEMM_SK, 0), g);
  return g.sk > 1 ? (size_t)g.sk * m * n * sizeof(float) : 0;
}

extern "C" int chatts_linear_f16q(const ChattsLinearF16qArgs* a, chatts_stream_t stream) {
  CHATTS_REQUIRE(a, CHATTS_E_BADARG, "linear_f16q: null args");
  CHATTS_REQUIRE(a->m >= 0 && a->n > 0 && a->k > 0, CHATTS_E_BADARG, "linear_f16q: bad sizes m=%d n=%d k=%d", a->m, a->n, a->k);
  if (a->m == 0) return CHATTS_OK;
  CHATTS_REQUIRE(a->a_hi && a->a_lo8 && a->a_scale && a->w16 && a->w8 && a->w8_exp, CHATTS_E_BADARG, "linear_f16q: null operand");
  CHATTS_REQUIRE(a->k % kF16qBlock == 0 && a->n % 16 == 0 && (a->epilogue != CHATTS_EPI_SWIGLU || a->n % 256 == 0), CHATTS_E_SHAPE,
                 "linear_f16q: K=%d must be a multiple of 128, N=%d of 16 (SwiGLU: 256)", a->k, a->n);
  CHATTS_REQUIRE(!a->w_tiled || a->ldw == a->k, CHATTS_E_SHAPE, "linear_f16q: tiled weights need ldw == K");
  CHATTS_REQUIRE(!a->planes_tiled || (a->ld_a == a->k && (!a->c_hi || a->ld_cplanes == (a->epilogue == CHATTS_EPI_SWIGLU ? a->n / 2 : a->n)) &&
                                      (!a->post_norm_w || a->ld_post == a->n)), CHATTS_E_SHAPE,
                 "linear_f16q: tiled planes need ld_a == K, ld_cplanes == the output width, ld_post == N");
  CHATTS_REQUIRE(a->ld_a >= a->k && a->ld_a % 16 == 0 && a->ldw >= a->k && a->ldw % 16 == 0 && a->ld_scale >= a->k / kF16qBlock &&
                     ((uintptr_t)a->a_hi % 16) == 0 && ((uintptr_t)a->a_lo8 % 16) == 0 && ((uintptr_t)a->w16 % 16) == 0 && ((uintptr_t)a->w8 % 16) == 0,
                 CHATTS_E_SHAPE, "linear_f16q: operand leading dimensions (multiples of 16) / 16-byte alignment");
  const bool cplanes = a->c_hi != nullptr;
  const int ncols = a->epilogue == CHATTS_EPI_SWIGLU ? a->n / 2 : a->n;
  if (cplanes)
    CHATTS_REQUIRE(a->epilogue == CHATTS_EPI_SWIGLU && a->c_lo8 && a->c_scale && a->ld_cplanes >= ncols && a->ld_cplanes % 4 == 0 &&
                       a->ld_cscale >= ncols / kF16qBlock && ((uintptr_t)a->c_hi % 8) == 0 && ((uintptr_t)a->c_lo8 % 4) == 0,
                   CHATTS_E_SHAPE, "linear_f16q: plane output is the SwiGLU epilogue's (hi, lo8, scale; ld_cplanes >= N / 2)");
  else
    CHATTS_REQUIRE(a->c && a->ldc >= ncols && a->ldc % 4 == 0 && ((uintptr_t)a->c % 16) == 0, CHATTS_E_SHAPE, "linear_f16q: c / ldc=%d", a->ldc);
  CHATTS_REQUIRE(((uintptr_t)a->bias % 16) == 0 && ((uintptr_t)a->resid % 16) == 0, CHATTS_E_SHAPE, "linear_f16q: 16-byte aligned bias / resid");
  CHATTS_REQUIRE(a->epilogue != CHATTS_EPI_RESID || a->resid, CHATTS_E_BADARG, "linear_f16q: EPI_RESID without resid");
  RingGeom g;
  f16q_pick(a->m, a->n, a->k, device_cus(), opt_get(OPT_GEMM_T, 0), opt_get(OPT_GEMM_SK, 0), g);
  int sk = g.sk;
  if (cplanes || a->epilogue == CHATTS_EPI_SWIGLU) sk = 1;          // (the SwiGLU output is large: never split)
  int kps = ((a->k / kF16qBlock + sk - 1) / sk) * kF16qBlock;
  sk = (a->k + kps - 1) / kps;
  g.sk = sk;
  g.units = g.T * g.P * sk;
  int wpx = (g.units + 7) / 8;
  if (wpx > device_cus() / 8) wpx = device_cus() / 8;
  g.wpx = wpx < 1 ? 1 : wpx;
  F16qGemm q{};
  q.g.bias = a->bias; q.g.resid = a->resid; q.g.m = a->m; q.g.n = a->n; q.g.k = a->k; q.g.ldc = a->ldc; q.g.epilogue = a->epilogue;
  q.g.k_per_split = kps; q.g.direct = sk == 1;
  if (sk > 1) {
    const size_t need = (size_t)sk * a->m * a->n * sizeof(float);
    CHATTS_REQUIRE(a->workspace && a->workspace_bytes >= need, CHATTS_E_WORKSPACE, "linear_f16q: split-K needs %zu workspace bytes, got %zu", need,
                   a->workspace_bytes);
    q.g.c = reinterpret_cast<float*>(a->workspace);
  } else {
    q.g.c = a->c;
  }
  q.a_hi = reinterpret_cast<const _Float16*>(a->a_hi); q.a_lo8 = a->a_lo8; q.a_sc = a->a_scale; q.lda = a->ld_a; q.ldsc = a->ld_scale;
  q.w16 = reinterpret_cast<const _Float16*>(a->w16); q.w8 = a->w8; q.w8e = a->w8_exp; q.ldw = a->ldw; q.w_tiled = a->w_tiled; q.a_tiled = a->planes_tiled; q.c_tiled = a->planes_tiled;
  q.c_hi = reinterpret_cast<_Float16*>(a->c_hi); q.c_lo8 = a->c_lo8; q.c_sc = a->c_scale; q.ldcp = a->ld_cplanes; q.ldcsc = a->ld_cscale;
  if (const int rc = launch_f16q(q, g, as_stream(stream))) return rc;
  CHATTS_CHECK_LAUNCH("gemm_f16q");
  if (sk > 1) return f16q_splitk_epilogue(a, sk, as_stream(stream));
  if (a->post_norm_w)
    return chatts_rmsnorm_f16q(a->c, a->post_norm_w, a->post_hi, a->post_lo8, a->post_scale, a->ld_post, a->ld_pscale, a->m, a->n, a->post_norm_eps, a->planes_tiled, stream);
  return CHATTS_OK;
}
