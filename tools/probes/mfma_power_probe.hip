// mfma_power_probe.hip - what the matrix pipe of MI355X sustains when every CU issues nothing but MFMAs, as a function of the operand
// DATA (diagnostic, not part of the library).  The chip clocks to its power budget: the round-5 timeline probe saw the prefill
// gate_up GEMM run at 1.57 GHz.  This probe separates "kernel does not keep the pipe busy" from "the pipe at this data is power-bound":
// 16 x 16 x 32 and 32 x 32 x 16 bf16 MFMAs, two waves per SIMD, back to back, operands held in registers:
//   zeros | small integers | weights ~ N(0, 0.02) x activations ~ N(0, 1) (the hi pass) | the same weights x bf16(a - bf16(a)) (the lo pass) |
//   alternating hi / lo sweeps (what the bf16x2 kernels issue)
// Reports TFLOP/s and the effective shader clock (s_memtime ticks per 100 MHz s_memrealtime tick).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power_probe mfma_power_probe.hip && ./mfma_power_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// operands: w [512 threads][4 frags] as B-like data, a_hi / a_lo [512][4]
template <int KIND>
__global__ __launch_bounds__(512) void probe(const bf16x8_t* __restrict__ w, const bf16x8_t* __restrict__ ahi, const bf16x8_t* __restrict__ alo,
                                             int iters, int mode, float* out, unsigned long long* clk) {
  bf16x8_t b[4], h[4], l[4];
  for (int i = 0; i < 4; ++i) {
    b[i] = w[threadIdx.x * 4 + i];
    h[i] = ahi[threadIdx.x * 4 + i];
    l[i] = mode == 0 ? h[i] : alo[threadIdx.x * 4 + i];      // mode 0: both sweeps on the same plane; 1: hi + lo sweeps
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  if (KIND == 0) {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], l[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], h[i], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j].x + acc[i][j].w;
  } else {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j + 2 * rep], l[i + 2 * rep], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j + 2 * rep], h[i + 2 * rep], acc[i][j], 0, 0, 0);
      }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) sum += acc[i][j][0] + acc[i][j][15];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
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

int main() {
  const int n = 512 * 4 * 8;
  std::vector<unsigned short> w(n), hi(n), lo(n), zero(n, 0), ints(n), ints2(n);
  srand(1234);
  for (int i = 0; i < n; ++i) {
    w[i] = bf16_bits(0.02f * gauss());
    const float a = gauss();
    hi[i] = bf16_bits(a);
    lo[i] = bf16_bits(a - bf16_val(hi[i]));
    ints[i] = bf16_bits((float)(i % 7));
    ints2[i] = bf16_bits((float)(i % 5 - 2));
  }
  auto up = [&](const std::vector<unsigned short>& v) { void* d; (void)hipMalloc(&d, n * 2); (void)hipMemcpy(d, v.data(), n * 2, hipMemcpyHostToDevice); return (const bf16x8_t*)d; };
  const bf16x8_t *dw = up(w), *dhi = up(hi), *dlo = up(lo), *dz = up(zero), *di = up(ints), *di2 = up(ints2);
  float* out; (void)hipMalloc(&out, 64);
  unsigned long long* clk; (void)hipMalloc(&clk, 64);
  struct Case { const char* name; const bf16x8_t *w, *h, *l; int mode; };
  const Case cases[] = {{"zeros", dz, dz, dz, 0}, {"small integers", di, di2, di2, 0}, {"weights x hi plane (both sweeps)", dw, dhi, dhi, 0},
                        {"weights x lo plane (both sweeps)", dw, dlo, dlo, 0}, {"weights x (hi sweep + lo sweep) = bf16x2", dw, dhi, dlo, 1}};
  for (int kind = 0; kind < 2; ++kind) {
    for (const Case& c : cases) {
      const int iters = 60000;      // ~30-40 ms: long enough for the clock to settle at the power budget
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL((probe<0>), dim3(256), dim3(512), 0, 0, c.w, c.h, c.l, iters, c.mode, out, clk);
        else hipLaunchKernelGGL((probe<1>), dim3(256), dim3(512), 0, 0, c.w, c.h, c.l, iters, c.mode, out, clk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
      }
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hc[2]; (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
      const double flops = 256.0 * 8 * iters * 32 * 16384.0;      // per iteration and wave: 32 x (16x16x32) = 16 x (32x32x16)
      printf("%-26s %-42s %7.1f TFLOP/s   clock %.2f GHz   %.1f %% of the clock's peak\n", kind == 0 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_32x32x16_bf16", c.name,
             flops / ms / 1e9, (double)hc[0] / ((double)hc[1] * 10.0), 100.0 * (flops / ms / 1e9) / (2500.0 * ((double)hc[0] / ((double)hc[1] * 10.0)) / 2.4));
    }
  }
  return 0;
}
