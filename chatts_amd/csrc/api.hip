// api.hip - error string, device query, synthetic-weight fill.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace chatts {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int device_cus() {
  static thread_local int cus = 0;
  if (cus > 0) return cus;
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    (void)hipGetLastError();
    cus = 256;   // MI355X; keeps host-only size queries usable without a device
    return cus;
  }
  cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  return cus;
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// One thread per 4 consecutive columns; rows on blockIdx.y (grid-stride).
__global__ __launch_bounds__(256) void fill_hash_kernel(void* __restrict__ dst, int out_f32, uint32_t key,
                                                       float base, float scale, int64_t rows,
                                                       int64_t cols, int64_t ld, int64_t row0,
                                                       int64_t col0, int64_t full_cols) {
  const int64_t c4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c4 >= cols) return;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t c = c4 + j;
      if (c >= cols) break;
      const uint64_t i = (uint64_t)(row0 + r) * (uint64_t)full_cols + (uint64_t)(col0 + c);
      const uint32_t lo = (uint32_t)i, hi = (uint32_t)(i >> 32);
      const uint32_t h = mix32(lo ^ mix32(key + hi * 0x85ebca6bu));
      const int n = (int)(h & 255u) + (int)((h >> 8) & 255u) + (int)((h >> 16) & 255u) + (int)(h >> 24) - 510;
      const float v = base + (float)n * scale;
      const uint16_t b = f32_to_bf16_rne(v);
      if (out_f32)
        reinterpret_cast<float*>(dst)[r * ld + c] = bf16_to_f32(b);
      else
        reinterpret_cast<uint16_t*>(dst)[r * ld + c] = b;
    }
  }
}

}  // namespace chatts

using namespace chatts;

extern "C" const char* chatts_last_error(void) { return g_err; }
extern "C" int chatts_abi_version(void) { return CHATTS_ABI_VERSION; }
extern "C" int chatts_device_cus(void) { return device_cus(); }

extern "C" int chatts_fill_hash(void* dst, int out_f32, uint32_t key, float base, int shift, int64_t rows,
                                int64_t cols, int64_t ld, int64_t row0, int64_t col0, int64_t full_cols,
                                chatts_stream_t stream) {
  CHATTS_REQUIRE(dst != nullptr, CHATTS_E_BADARG, "fill_hash: null dst");
  CHATTS_REQUIRE(rows >= 0 && cols >= 0 && ld >= cols && shift >= 0 && shift < 64, CHATTS_E_BADARG,
                 "fill_hash: bad sizes rows=%lld cols=%lld ld=%lld shift=%d", (long long)rows,
                 (long long)cols, (long long)ld, shift);
  if (rows == 0 || cols == 0) return CHATTS_OK;
  const float scale = ldexpf(1.0f, -shift);
  dim3 block(256);
  dim3 grid((unsigned)((cols + 1023) / 1024), (unsigned)(rows < 65535 ? rows : 65535));
  hipLaunchKernelGGL(fill_hash_kernel, grid, block, 0, as_stream(stream), dst, out_f32, key, base, scale,
                     rows, cols, ld, row0, col0, full_cols);
  CHATTS_CHECK_LAUNCH("fill_hash");
  return CHATTS_OK;
}
