// mfma_probe.hip - sustained MFMA issue rate per SIMD on gfx950 (diagnostic, not part of the library)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
// Each wave runs the accumulator pattern of the GEMM's K-step: 16 independent 16x16 accumulators swept twice
// (v_mfma_f32_16x16x32_bf16), or 4 independent 32x32 accumulators (v_mfma_f32_32x32x16_bf16) - the same flops.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  bf16x8_t a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(float)(threadIdx.x + i + j); b[i][j] = (__bf16)(float)(threadIdx.x * 3 + i - j); }
  float sum = 0.f;
  if (KIND == 0) {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j].x + acc[i][j].w;
  } else {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * (rep & 1)], b[j + 2 * (rep >> 1)], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) sum += acc[i][j][0] + acc[i][j][15];
  }
  if (sum == 12345.678f) out[0] = sum;
}

template <int KIND>
static void run(const char* name, int threads, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(threads), 0, 0, out, 100);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((probe<KIND>), dim3(256), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * (threads / 64) * iters * 32 * 16384.0;   // per iteration and wave: 32 x (16x16x32) or 16 x (32x32x16)
  const double mfma_per_simd = (double)(threads / 256) * iters * (KIND == 0 ? 32 : 16);
  printf("%-28s %d waves/SIMD : %7.1f TFLOP/s   %.1f ns per MFMA per SIMD\n", name, threads / 256, flops / ms / 1e9, ms * 1e6 / mfma_per_simd);
}

int main() {
  float* out; (void)hipMalloc(&out, 64);
  for (int threads : {256, 512}) {
    if (threads == 256) { run<0>("v_mfma_f32_16x16x32_bf16", 256, out); run<1>("v_mfma_f32_32x32x16_bf16", 256, out); }
    else { run<0>("v_mfma_f32_16x16x32_bf16", 512, out); run<1>("v_mfma_f32_32x32x16_bf16", 512, out); }
  }
  return 0;
}
