// mfma_mix_probe.hip - round-6 gate (ii): what the matrix pipe of MI355X sustains, at its power-bound clock, on the instruction MIXES a
// prefill projection could issue per (16 x 16 output tile, 128 K-values) - diagnostic, not part of the library:
//   bf16x2         8 x v_mfma_f32_16x16x32_bf16  (hi + lo plane of the activations x bf16 weights: today's parity mode)
//   f16            4 x v_mfma_f32_16x16x32_f16   (one pass)
//   mx8            1 x v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, e8m0 block scales)
//   f16 + mx8      4 x f16 + 1 x mx8             (the 1.5-pass split: f16 hi plane x f16 weights + e4m3 residual x e4m3 weights)
//   bf16 + mx8     4 x bf16 + 1 x mx8
// Operand DATA as in the GEMM: weights ~ N(0, 0.02), activations ~ N(0, 1) split hi / residual, block scales chosen per 32 values.
// Every CU runs 8 waves (2 per SIMD) of nothing but these MFMAs on register operands.  Reports time per tile-group, the equivalent
// USEFUL TFLOP/s (2 x 16 x 16 x 128 per tile-group, whatever was issued for it) and the effective shader clock.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_mix_probe mfma_mix_probe.hip && ./mfma_mix_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

struct Ops {
  const bf16x8_t *wb, *hb, *lb;      // bf16 weights, hi plane, lo plane        [512 threads][4 frags]
  const f16x8_t *wh, *hh;            // f16 weights, f16 hi plane
  const i32x8 *w8, *l8;              // e4m3 weights, e4m3 residual             [512][4]
  const int *sw, *sl;                // e8m0 scales, one byte per fragment packed into a dword [512]
};

template <int KIND>
__global__ __launch_bounds__(512) void probe(Ops o, int iters, float* out, unsigned long long* clk) {
  const int t = threadIdx.x;
  bf16x8_t wb[4], hb[4], lb[4];
  f16x8_t wh[4], hh[4];
  i32x8 w8[4], l8[4];
  for (int i = 0; i < 4; ++i) {
    wb[i] = o.wb[t * 4 + i]; hb[i] = o.hb[t * 4 + i]; lb[i] = o.lb[t * 4 + i];
    wh[i] = o.wh[t * 4 + i]; hh[i] = o.hh[t * 4 + i];
    w8[i] = o.w8[t * 4 + i]; l8[i] = o.l8[t * 4 + i];
  }
  const int sw = o.sw[t], sl = o.sl[t];
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {          // one iteration = one 128-deep K group of the wave's 4 x 4 tiles
    if (KIND == 0 || KIND == 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (KIND == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[(j + q) & 3], lb[(i + q) & 3], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[(j + q) & 3], hb[(i + q) & 3], acc[i][j], 0, 0, 0);
      }
    }
    if (KIND == 1 || KIND == 3) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(j + q) & 3], hh[(i + q) & 3], acc[i][j], 0, 0, 0);
    }
    if (KIND == 2 || KIND == 3 || KIND == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8[j], l8[i], acc[i][j], 0, 0, 0, sw, 0, sl);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j].x + acc[i][j].w;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
  if (sum == 12345.678f) out[0] = sum;
}

static unsigned short bf16_bits(float f) {
  unsigned u; std::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_val(unsigned short b) { unsigned u = (unsigned)b << 16; float f; std::memcpy(&f, &u, 4); return f; }
static float gauss() {
  const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
  return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
}
// OCP e4m3fn code of y (|y| <= 448), round to nearest even
static unsigned char e4m3_bits(float y) {
  const unsigned char sgn = y < 0 ? 0x80 : 0;
  float a = fabsf(y);
  if (a > 448.f) a = 448.f;
  if (a == 0.f) return sgn;
  int ex; frexpf(a, &ex);
  int e = ex - 1; if (e < -6) e = -6;
  const float step = ldexpf(1.f, e - 3);
  float q = nearbyintf(a / step) * step;
  if (q > 448.f) q = 448.f;
  int qe; const float qm = frexpf(q, &qe);      // q = qm 2^qe, qm in [0.5, 1)
  int E = qe - 1;
  if (E < -6) return sgn | (unsigned char)nearbyintf(q / ldexpf(1.f, -9));      // subnormal: mantissa only
  const int mant = (int)nearbyintf((qm * 2.f - 1.f) * 8.f);
  return sgn | (unsigned char)(((E + 7) << 3) | mant);
}
// block of 32 values -> e4m3 codes + the e8m0 scale byte (smallest power of two with amax / s <= 448)
static unsigned char mx_block(const float* x, unsigned char* q) {
  float amax = 0.f;
  for (int i = 0; i < 32; ++i) amax = fmaxf(amax, fabsf(x[i]));
  int E = 0;
  if (amax > 0.f) { int ex; const float m = frexpf(amax / 448.f, &ex); E = ex - (m == 0.5f); }
  const float s = ldexpf(1.f, E);
  for (int i = 0; i < 32; ++i) q[i] = e4m3_bits(x[i] / s);
  return (unsigned char)(E + 127);
}

int main() {
  const int n = 512 * 4 * 8;              // 16-bit operands: 8 per thread and fragment
  std::vector<unsigned short> wb(n), hb(n), lb(n), wh(n), hh(n);
  const int n8 = 512 * 4 * 32;            // 8-bit operands: 32 per thread and fragment (one scale block)
  std::vector<unsigned char> w8(n8), l8(n8);
  std::vector<int> sw(512), sl(512);
  srand(1234);
  for (int i = 0; i < n; ++i) {
    const float w = 0.02f * gauss(), a = gauss();
    wb[i] = bf16_bits(w);
    const _Float16 w16 = (_Float16)bf16_val(wb[i]);
    std::memcpy(&wh[i], &w16, 2);
    hb[i] = bf16_bits(a);
    lb[i] = bf16_bits(a - bf16_val(hb[i]));
    const _Float16 a16 = (_Float16)a;
    std::memcpy(&hh[i], &a16, 2);
  }
  for (int t = 0; t < 512; ++t) {
    int swp = 0, slp = 0;
    for (int f = 0; f < 4; ++f) {
      float xw[32], xl[32];
      for (int i = 0; i < 32; ++i) {
        xw[i] = bf16_val(bf16_bits(0.02f * gauss()));
        const float a = gauss();
        xl[i] = a - (float)(_Float16)a;
      }
      const unsigned char ew = mx_block(xw, &w8[(t * 4 + f) * 32]), el = mx_block(xl, &l8[(t * 4 + f) * 32]);
      if (f == 0) { swp = ew * 0x01010101; slp = el * 0x01010101; }      // (one scale per thread: the data decides the power, not which byte)
    }
    sw[t] = swp; sl[t] = slp;
  }
  auto up = [&](const void* v, size_t bytes) { void* d; (void)hipMalloc(&d, bytes); (void)hipMemcpy(d, v, bytes, hipMemcpyHostToDevice); return d; };
  Ops o;
  o.wb = (const bf16x8_t*)up(wb.data(), n * 2); o.hb = (const bf16x8_t*)up(hb.data(), n * 2); o.lb = (const bf16x8_t*)up(lb.data(), n * 2);
  o.wh = (const f16x8_t*)up(wh.data(), n * 2); o.hh = (const f16x8_t*)up(hh.data(), n * 2);
  o.w8 = (const i32x8*)up(w8.data(), n8); o.l8 = (const i32x8*)up(l8.data(), n8);
  o.sw = (const int*)up(sw.data(), 512 * 4); o.sl = (const int*)up(sl.data(), 512 * 4);
  float* out; (void)hipMalloc(&out, 64);
  unsigned long long* clk; (void)hipMalloc(&clk, 64);
  const char* names[] = {"bf16x2: 8 x 16x16x32 bf16 (hi + lo planes)", "f16: 4 x 16x16x32 f16 (one pass)", "mx8: 1 x 16x16x128 f8f6f4 (e4m3 residual x e4m3 W)",
                         "f16 + mx8: 4 x f16 + 1 x f8f6f4 (1.5-pass split)", "bf16 + mx8: 4 x bf16 + 1 x f8f6f4"};
  printf("%-58s %10s %12s %8s\n", "mix per (16 x 16 tile, 128 K-values)", "ns/group", "useful TF/s", "clock");
  for (int kind = 0; kind < 5; ++kind) {
    const int iters = kind == 2 ? 120000 : 30000;      // tens of ms: long enough for the clock to settle at the power budget
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      switch (kind) {
        case 0: hipLaunchKernelGGL((probe<0>), dim3(256), dim3(512), 0, 0, o, iters, out, clk); break;
        case 1: hipLaunchKernelGGL((probe<1>), dim3(256), dim3(512), 0, 0, o, iters, out, clk); break;
        case 2: hipLaunchKernelGGL((probe<2>), dim3(256), dim3(512), 0, 0, o, iters, out, clk); break;
        case 3: hipLaunchKernelGGL((probe<3>), dim3(256), dim3(512), 0, 0, o, iters, out, clk); break;
        default: hipLaunchKernelGGL((probe<4>), dim3(256), dim3(512), 0, 0, o, iters, out, clk); break;
      }
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long hc[2]; (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    // per iteration a wave finishes 16 tile-groups of 2 x 16 x 16 x 128 useful flops; 2 waves per SIMD share a pipe
    const double groups = 256.0 * 8 * (double)iters * 16.0;
    const double useful = groups * 2.0 * 16 * 16 * 128;
    printf("%-58s %10.2f %12.1f %5.2f GHz\n", names[kind], ms * 1e6 / ((double)iters * 16.0 * 2.0), useful / ms / 1e9,
           (double)hc[0] / ((double)hc[1] * 10.0));
  }
  return 0;
}
