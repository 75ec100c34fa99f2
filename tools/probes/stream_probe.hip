// probe: read bandwidth of a buffer with plain vs non-temporal 16-byte loads (cold = after a 1.2 GB flush, warm = re-read)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void rd(const u32x4* p, size_t n, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256 * 4) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      size_t j = i + (size_t)u * gridDim.x * 256;
      if (j < n) v[u] = NT ? __builtin_nontemporal_load(p + j) : p[j]; else v[u] = (u32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const size_t big = 1200ull << 20;
  char *flush, *buf; unsigned* out;
  hipMalloc(&flush, big); hipMalloc(&buf, 300ull << 20); hipMalloc(&out, 4);
  hipMemset(flush, 1, big); hipMemset(buf, 2, 300ull << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t mb : {26, 52, 142, 283}) {
    size_t n = (mb << 20) / 16;
    for (int nt = 0; nt < 2; ++nt) {
      float cold = 1e9, warm = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        rd<false><<<2048, 256>>>((const u32x4*)flush, big / 16, out);   // flush caches
        hipDeviceSynchronize();
        float ms;
        hipEventRecord(e0);
        if (nt) rd<true><<<2048, 256>>>((const u32x4*)buf, n, out); else rd<false><<<2048, 256>>>((const u32x4*)buf, n, out);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (ms < cold) cold = ms;
        hipEventRecord(e0);
        if (nt) rd<true><<<2048, 256>>>((const u32x4*)buf, n, out); else rd<false><<<2048, 256>>>((const u32x4*)buf, n, out);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (ms < warm) warm = ms;
      }
      printf("%4zu MB %s  cold %7.1f us (%5.0f GB/s)   warm re-read %7.1f us (%5.0f GB/s)\n", mb, nt ? "nt   " : "plain",
             cold * 1e3, (mb << 20) / cold / 1e6, warm * 1e3, (mb << 20) / warm / 1e6);
    }
  }
  // warm read with nt after plain prefetch and vice versa
  size_t n = (52ull << 20) / 16;
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      rd<false><<<2048, 256>>>((const u32x4*)flush, big / 16, out); hipDeviceSynchronize();
      if (mode == 0) rd<false><<<2048, 256>>>((const u32x4*)buf, n, out); else rd<true><<<2048, 256>>>((const u32x4*)buf, n, out);
      hipDeviceSynchronize();
      float ms; hipEventRecord(e0);
      if (mode == 0) rd<true><<<2048, 256>>>((const u32x4*)buf, n, out); else rd<false><<<2048, 256>>>((const u32x4*)buf, n, out);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("52 MB: %s prefetch then %s read: %7.1f us (%5.0f GB/s)\n", mode == 0 ? "plain" : "nt", mode == 0 ? "nt" : "plain",
           best * 1e3, (52ull << 20) / best / 1e6);
  }
  return 0;
}
