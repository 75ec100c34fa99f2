// row_stream_probe.hip - which per-instruction SHAPE lets a wave stream rows of a weight matrix at the HBM rate?  (diagnostic)
//   hipcc --offload-arch=gfx950 -O3 -o row_stream_probe row_stream_probe.hip && ./row_stream_probe
// A matrix of N rows x KB bytes (fp8 gate_up of ChatTS-14B: 27648 x 5120 B = 142 MB; 4 copies in rotation) is read once, load-only,
// non-temporal, 8 waves per workgroup, UNR 16-byte loads in flight per lane:
//   MODE 0  GEMV shape        : a wave owns 2 rows; one instruction = 1 row x 1024 B
//   MODE 1  MFMA-operand shape: a wave owns 16 rows; one instruction = 16 rows x 64 B (lane -> row lane & 15, 16-byte chunk lane >> 4),
//                               consecutive instructions walk the SAME 16 rows (1 KB contiguous per row after 16 instructions)
//   MODE 2  whole lines       : a wave owns 16 rows; one instruction = 8 rows x 128 B, two instructions per 128-byte column block
//   MODE 3  MODE 1 with the K walk interleaved over 4 waves of a workgroup (wave w takes 64-byte column 4 i + w of the same 16 rows):
//           what a K-split inside a workgroup would do
//   MODE 5  MODE 2's walk, but every instruction is an LDS-DMA (global_load_lds_dwordx4) into a per-wave ring: no VGPR destination
//   MODE 6  MODE 5 + one s_barrier per 128-byte K-step (what a shared A stage costs the W stream)
//   MODE 7  MODE 2 + two L2-resident 16-byte loads per K-step (the A planes of a batched-decode GEMM)
//   MODE 8  MODE 5 with the rows of a K-step spread like gemm_stream_kernel's pieces (wave w: 8-row groups w, w+8 of a 128-row tile)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int UNR>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ w, int n, int kb, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  if (MODE == 0) {
    const int tasks = n / 2, steps = kb / 1024;
    for (int t = blockIdx.x * nw + wave; t < tasks; t += gridDim.x * nw) {
      const char* r0 = w + (size_t)(2 * t) * kb + lane * 16;
      for (int c = 0; c < steps; c += UNR / 2) {
        u32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR / 2; ++u) {
          const int cc = c + u < steps ? c + u : steps - 1;
          v[2 * u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(r0 + (size_t)cc * 1024));
          v[2 * u + 1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(r0 + kb + (size_t)cc * 1024));
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc ^= v[u];
      }
    }
  } else if (MODE == 1 || MODE == 2) {
    const int tasks = n / 16, steps = kb / 64;     // 64 B of K per row and instruction (MODE 2: 128 B per row, half the rows)
    for (int t = blockIdx.x * nw + wave; t < tasks; t += gridDim.x * nw) {
      const char* base = w + (size_t)(16 * t) * kb;
      for (int c = 0; c < steps; c += UNR) {
        u32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int i = c + u < steps ? c + u : steps - 1;
          size_t off;
          if (MODE == 1) off = (size_t)(lane & 15) * kb + (size_t)i * 64 + (lane >> 4) * 16;
          else off = (size_t)((i & 1) * 8 + (lane >> 3)) * kb + (size_t)(i >> 1) * 128 + (lane & 7) * 16;
          v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + off));
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc ^= v[u];
      }
    }
  } else if (MODE == 5 || MODE == 6 || MODE == 8) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* ring = smem + wave * (UNR * 1024);         // UNR pieces of 1 KB per wave
    const int tasks = n / 16, steps = kb / 64;
    for (int t = blockIdx.x * nw + wave; t < tasks; t += gridDim.x * nw) {
      // MODE 8: the workgroup's 8 tasks form one 128-row tile; wave w takes 8-row groups w and w + 8 of it
      const int t0 = (t / nw) * nw;
      const char* base = MODE == 8 ? w + (size_t)(16 * t0) * kb : w + (size_t)(16 * t) * kb;
      for (int c = 0; c < steps; c += UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int i = c + u < steps ? c + u : steps - 1;
          const int rgrp = MODE == 8 ? (wave + 8 * (i & 1)) * 8 : (i & 1) * 8;
          const size_t off = (size_t)(rgrp + (lane >> 3)) * kb + (size_t)(i >> 1) * 128 + (lane & 7) * 16;
          __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(ring + u * 1024), 16, 0, 2);
          if (MODE == 6 && (u & 1)) __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc.x ^= *reinterpret_cast<const unsigned*>(ring + lane * 4);
      }
    }
  } else if (MODE == 7) {
    const int tasks = n / 16, steps = kb / 64;
    const char* a = w + (size_t)n * kb - 65536;      // a 64 KB region every wave re-reads (stays in L2)
    for (int t = blockIdx.x * nw + wave; t < tasks; t += gridDim.x * nw) {
      const char* base = w + (size_t)(16 * t) * kb;
      for (int c = 0; c < steps; c += UNR) {
        u32x4 v[UNR], av[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int i = c + u < steps ? c + u : steps - 1;
          const size_t off = (size_t)((i & 1) * 8 + (lane >> 3)) * kb + (size_t)(i >> 1) * 128 + (lane & 7) * 16;
          v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + off));
          av[u] = *reinterpret_cast<const u32x4*>(a + ((size_t)(i * 1024 + lane * 16) & 65535));
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc ^= v[u] ^ av[u];
      }
    }
  } else {
    const int tasks = n / 16, steps = kb / 64 / 4;   // a group of 4 waves shares 16 rows
    const int grp = wave >> 2, wq = wave & 3, ngrp = nw >> 2;
    for (int t = blockIdx.x * ngrp + grp; t < tasks; t += gridDim.x * ngrp) {
      const char* base = w + (size_t)(16 * t) * kb;
      for (int c = 0; c < steps; c += UNR) {
        u32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int i = c + u < steps ? c + u : steps - 1;
          v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (size_t)(lane & 15) * kb + (size_t)(4 * i + wq) * 64 + (lane >> 4) * 16));
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc ^= v[u];
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

template <int MODE, int UNR>
static void run(const char* name, char** bufs, int n, int kb, unsigned* sink, int blocks, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int b = 0; b < 4; ++b) probe<MODE, UNR><<<blocks, threads, (MODE == 5 || MODE == 6 || MODE == 8) ? (threads / 64) * UNR * 1024 : 0>>>(bufs[b], n, kb, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms / 4 < best) best = ms / 4;
  }
  printf("%-44s UNR=%2d blocks=%4d x %3d  %7.1f us  %5.2f TB/s\n", name, UNR, blocks, threads, best * 1e3, (double)n * kb / best / 1e9);
}

int main() {
  const int n = 27648, kb = 5120;
  char* bufs[4]; unsigned* sink;
  for (int b = 0; b < 4; ++b) { hipMalloc(&bufs[b], (size_t)n * kb); hipMemset(bufs[b], b + 1, (size_t)n * kb); }
  hipMalloc(&sink, 4);
  hipDeviceSynchronize();
  for (int blocks : {256, 512}) {
    run<0, 4>("GEMV shape (1 row x 1 KB)", bufs, n, kb, sink, blocks, 512);
    run<0, 8>("GEMV shape (1 row x 1 KB)", bufs, n, kb, sink, blocks, 512);
    run<1, 8>("16 rows x 64 B, same rows back to back", bufs, n, kb, sink, blocks, 512);
    run<1, 16>("16 rows x 64 B, same rows back to back", bufs, n, kb, sink, blocks, 512);
    run<2, 8>("8 rows x 128 B (whole lines)", bufs, n, kb, sink, blocks, 512);
    run<2, 16>("8 rows x 128 B (whole lines)", bufs, n, kb, sink, blocks, 512);
    run<3, 10>("16 rows x 64 B, K interleaved over 4 waves", bufs, n, kb, sink, blocks, 512);
    run<5, 8>("whole lines by LDS-DMA, wave walks 16 rows", bufs, n, kb, sink, blocks, 512);
    run<5, 16>("whole lines by LDS-DMA, wave walks 16 rows", bufs, n, kb, sink, blocks, 512);
    run<6, 8>("... + s_barrier per 128-byte K-step", bufs, n, kb, sink, blocks, 512);
    run<8, 8>("LDS-DMA, rows spread like gemm_stream pieces", bufs, n, kb, sink, blocks, 512);
    run<7, 8>("whole lines to VGPRs + 1 L2 load per W load", bufs, n, kb, sink, blocks, 512);
    run<5, 8>("whole lines by LDS-DMA, 4 waves per workgroup", bufs, n, kb, sink, blocks, 256);
    run<2, 8>("whole lines to VGPRs, 4 waves per workgroup", bufs, n, kb, sink, blocks, 256);
  }
  return 0;
}
