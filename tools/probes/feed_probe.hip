// feed_probe.hip - round 6 diagnostic (not part of the library): what a CU can PULL from L2 per clock with the request shapes a prefill
// projection's operand feed uses.  The round-5 ablation of gemm_ring_kernel (profiles/r5_ring_ablate.txt: LDS-DMA only, no MFMA, no
// fragment reads: 1504 cycles per 34 KB half-stage = 23 B/clk/CU - as slow as the whole kernel) says the feed, not the matrix pipe, sets
// that kernel's time.  Hypothesis: a 1 KB piece made of 16 rows x 64 B touches 16 cache lines of 128 B and uses half of each; the other
// half is requested again one half-stage later.  This probe streams a W-like matrix [27648 rows][10240 B] panel by panel (256 rows, six
// workgroups per panel, as the GEMM's M-tiles do) with pieces of   16 rows x 64 B | 8 x 128 | 4 x 256 | 2 x 512 | 1 KB contiguous
// (pre-tiled operand) | 32 rows x 32 B (the e4m3 planes of gemm_f16q),   through LDS-DMA or through plain register loads, from 4 / 8 / 12
// waves per CU, and reports bytes per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 -o feed_probe feed_probe.hip && ./feed_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowBytes = 10240, kPanelRows = 256, kPanels = 108, kTilesPerPanel = 6;
constexpr int kPieces = kPanelRows * kRowBytes / 1024;      // 2560 per panel stream

// LOG2R: log2 of the rows per piece (4 -> 16 rows x 64 B ... 0 -> one row of 1 KB ... ), -1: pre-tiled contiguous, 5: 32 rows x 32 B
template <int LOG2R, bool DMA, int DEPTH>
__global__ __launch_bounds__(768) void feed(const char* __restrict__ w, int nwaves, int units, unsigned long long* clk, u32x4* sink, int wrap) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nwaves) return;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int share = units >> 3;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  u32x4 acc = {0, 0, 0, 0};
  for (int u = xcd * share + local; u < (xcd + 1) * share; u += wpx) {
    const int panel = u / kTilesPerPanel;
    const char* base = w + (size_t)panel * kPanelRows * kRowBytes;
    constexpr int R = LOG2R < 0 ? 1 : (1 << LOG2R), C = 1024 / R, RB = kPanelRows / R, LPR = C / 16 > 0 ? C / 16 : 1;
    auto offset = [&](int p) -> size_t {
      if (wrap) p %= wrap;
      if (LOG2R < 0) return (size_t)p * 1024 + lane * 16;
      if (LOG2R == 5) {      // 32 rows x 32 B: lane -> (row-half of 16, 16-byte half, row)
        const int kc = p / (kPanelRows / 32), rb = p % (kPanelRows / 32);
        return (size_t)(rb * 32 + (lane >> 5) * 16 + (lane & 15)) * kRowBytes + kc * 32 + ((lane >> 4) & 1) * 16;
      }
      const int kc = p / RB, rb = p % RB;
      return (size_t)(rb * R + lane / LPR) * kRowBytes + kc * C + (lane % LPR) * 16;
    };
    if (DMA) {
      int n = 0;
      for (int p = wave; p < kPieces; p += nwaves, ++n) {
        __builtin_amdgcn_global_load_lds((gptr_t)(base + offset(p)), (lptr_t)(smem + (wave * DEPTH + (n % DEPTH)) * 1024), 16, 0, 0);
        if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
      }
    } else {
      for (int p = wave; p < kPieces; p += nwaves * 8) {      // eight loads in flight per wave
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int pj = p + j * nwaves < kPieces ? p + j * nwaves : p;
          v[j] = *reinterpret_cast<const u32x4*>(base + offset(pj));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
  if (!DMA && acc.x == 0x12345u) sink[blockIdx.x * 768 + threadIdx.x] = acc;
}

template <int LOG2R, bool DMA, int DEPTH>
static void run(const char* name, const char* w, int nwaves, unsigned long long* clk, u32x4* sink, int wrap = 0, int grid = 256) {
  const int units = wrap ? grid * 2 : kPanels * kTilesPerPanel;      // 648
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds = 12 * DEPTH * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&feed<LOG2R, DMA, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((feed<LOG2R, DMA, DEPTH>), dim3(grid), dim3(768), lds, 0, w, nwaves, units, clk, sink, wrap);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), clk, 256 * 8, hipMemcpyDeviceToHost);
  double cmax = 0;
  for (auto c : h) cmax = c > cmax ? c : cmax;
  const double bytes = (double)units * kPanelRows * kRowBytes;
  // s_memtime ticks at 100 MHz on this part; bytes per shader clock are derived from the wall time and an assumed 2.1 GHz
  printf("%-34s waves %2d  in flight/wave %2d  grid %3d  %s  %8.1f us  %6.2f TB/s  %5.1f B/clk/CU @2.1GHz\n", name, nwaves, DEPTH, grid, wrap ? "L2-resident" : "streaming  ",
         best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / (double)grid / (best * 1e-3 * 2.1e9));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("  error: %s\n", hipGetErrorString(e));
}

int main() {
  const size_t bytes = (size_t)kPanels * kPanelRows * kRowBytes;
  char* w;
  unsigned long long* clk;
  u32x4* sink;
  hipMalloc(&w, bytes);
  hipMemset(w, 1, bytes);
  hipMalloc(&clk, 256 * 8);
  hipMalloc(&sink, 256 * 768 * 16);
  printf("## feed_probe (MI355X): 648 panel streams of 2.6 MB (108 panels x 6 workgroups), 256 workgroups, 1 KB per wave instruction\n");
  for (int nw : {4, 8, 12}) {
    run<4, true, 8>("LDS-DMA 16 rows x 64 B", w, nw, clk, sink);
    run<3, true, 8>("LDS-DMA  8 rows x 128 B", w, nw, clk, sink);
    run<2, true, 8>("LDS-DMA  4 rows x 256 B", w, nw, clk, sink);
    run<1, true, 8>("LDS-DMA  2 rows x 512 B", w, nw, clk, sink);
    run<-1, true, 8>("LDS-DMA  1 KB contiguous (tiled)", w, nw, clk, sink);
    run<5, true, 8>("LDS-DMA 32 rows x 32 B (e4m3)", w, nw, clk, sink);
  }
  for (int nw : {4, 8}) {
    run<4, true, 4>("LDS-DMA 16 rows x 64 B", w, nw, clk, sink);
    run<-1, true, 4>("LDS-DMA  1 KB contiguous (tiled)", w, nw, clk, sink);
  }
  // the same request stream against a footprint that stays in L2 (each workgroup re-reads 256 KB of its panel), and with fewer CUs busy:
  // is the ceiling the CU's or the chip's?
  for (int grid : {256, 128, 64, 32, 8}) {
    run<-1, true, 8>("LDS-DMA  1 KB contiguous (tiled)", w, 8, clk, sink, 256, grid);
    run<4, true, 8>("LDS-DMA 16 rows x 64 B", w, 8, clk, sink, 2560, grid);
    run<-1, true, 8>("LDS-DMA  1 KB contiguous (tiled)", w, 8, clk, sink, 0, grid);
    run<-1, false, 8>("registers  1 KB contiguous", w, 8, clk, sink, 256, grid);
  }
  for (int nw : {4, 8, 12}) {
    run<4, false, 8>("registers 16 rows x 64 B", w, nw, clk, sink);
    run<3, false, 8>("registers  8 rows x 128 B", w, nw, clk, sink);
    run<-1, false, 8>("registers  1 KB contiguous", w, nw, clk, sink);
  }
  return 0;
}
