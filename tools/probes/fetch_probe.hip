// fetch_probe.hip - what is the L2 -> CU fill rate on gfx950, per path?  (diagnostic, not part of the library)
//   hipcc --offload-arch=gfx950 -O3 -o fetch_probe fetch_probe.hip && ./fetch_probe
// Every workgroup re-reads a small L2-resident buffer; per wave-instruction either 8 rows x 128 B (row stride
// `stride` bytes, the GEMM tile pattern) through (a) global_load_dwordx4 into VGPRs or (b) global_load_lds_dwordx4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int UNR>
__global__ __launch_bounds__(256) void probe(const char* buf, unsigned rows, int stride, int iters, u32x4* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 4 + wave;
  u32x4 acc = {0, 0, 0, 0};
  unsigned row = (gw * 8u * UNR) & (rows - 1);   // rows is a power of two
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      u32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const unsigned r = (row + u * 8 + (lane >> 3)) & (rows - 1);
        v[u] = *reinterpret_cast<const u32x4*>(buf + (size_t)r * stride + (lane & 7) * 16);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc ^= v[u];
    } else {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const unsigned r = (row + u * 8 + (lane >> 3)) & (rows - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(buf + (size_t)r * stride + (lane & 7) * 16), (lptr_t)(smem + (wave * UNR + u) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    row = (row + 8u * UNR * 1021u) & (rows - 1);
    buf += 128;                       // next K-step: the neighbouring line of every row
    if ((it & 31) == 31) buf -= 32 * 128;
  }
  if (MODE == 1) acc = *reinterpret_cast<u32x4*>(smem + threadIdx.x * 16);
  if (acc.x == 0x12345678) sink[0] = acc;
}

template <int MODE, int UNR>
static void run(const char* name, const char* buf, unsigned rows, int stride, u32x4* sink, int blocks_per_cu) {
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  const int lds = MODE == 1 ? 4 * UNR * 1024 : 0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, UNR>), dim3(blocks), dim3(256), lds, 0, buf, rows, stride, iters, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, UNR>), dim3(blocks), dim3(256), lds, 0, buf, rows, stride, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * 4 * iters * UNR * 1024;
  printf("%-34s blocks/CU %d  unroll %d  stride %6d : %7.2f TB/s  (%5.1f B/clk/CU at 2.1 GHz)\n", name, blocks_per_cu, UNR, stride,
         bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.1e9);
}

int main() {
  const size_t bytes = 2u << 20;    // 2 MB: resident in every XCD's 4 MB L2 (both strides give a power-of-two row count)
  char* buf; u32x4* sink;
  hipMalloc(&buf, bytes * 64); hipMemset(buf, 1, bytes * 64); hipMalloc(&sink, 64);
  for (int stride : {128, 8192, 10240, 10368, 27648, 27776}) {
    const unsigned rows = stride == 128 ? (unsigned)(bytes / 128) : 4096u;   // strided: 4096 rows x 128 B lines = 512 KB touched, L2 resident
    for (int bpc : {1, 2, 4}) {
      run<0, 4>("global_load_dwordx4 -> VGPR", buf, rows, stride, sink, bpc);
      run<0, 8>("global_load_dwordx4 -> VGPR", buf, rows, stride, sink, bpc);
      run<1, 4>("global_load_lds_dwordx4", buf, rows, stride, sink, bpc);
      run<1, 8>("global_load_lds_dwordx4", buf, rows, stride, sink, bpc);
    }
  }
  return 0;
}
