// api.hip - error string, device query, synthetic-weight fill.
#include <dlfcn.h>
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

// tuning options: see common.h.  Plain ints behind relaxed atomics: a host thread may set one while another enqueues (a torn
// read is impossible; which value a concurrent call sees is the caller's business).
// Encoding: 0 = unset (the zero-initialised state: no lazy initialisation, nothing for two first callers to race on - ADVICE r5), otherwise
// bit 32 set and the value in the low 32 bits.
static long long g_opts[OPT_COUNT];
static inline long long opt_pack(int v) { return (1LL << 32) | (long long)(unsigned)v; }
static const char* const g_opt_names[OPT_COUNT] = {
#define CHATTS_OPT_NAME(name) #name,
    CHATTS_OPTIONS(CHATTS_OPT_NAME)
#undef CHATTS_OPT_NAME
};
int opt_get(ChattsOpt o, int dflt) {
  const long long v = __atomic_load_n(&g_opts[o], __ATOMIC_RELAXED);
  return v == 0 ? dflt : (int)(unsigned)(v & 0xffffffffLL);
}

// roctx ranges (see common.h): resolved once; a missing library leaves both pointers null
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
static roctx_push_fn g_roctx_push = nullptr;
static roctx_pop_fn g_roctx_pop = nullptr;
static int g_roctx_state = 0;      // 0 = not tried, 1 = resolved, 2 = unavailable
static void roctx_resolve() {
  if (__atomic_load_n(&g_roctx_state, __ATOMIC_ACQUIRE) != 0) return;
  void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_GLOBAL);
  if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_GLOBAL);
  roctx_push_fn pu = h ? reinterpret_cast<roctx_push_fn>(dlsym(h, "roctxRangePushA")) : nullptr;
  roctx_pop_fn po = h ? reinterpret_cast<roctx_pop_fn>(dlsym(h, "roctxRangePop")) : nullptr;
  if (pu && po) { g_roctx_push = pu; g_roctx_pop = po; __atomic_store_n(&g_roctx_state, 1, __ATOMIC_RELEASE); }
  else __atomic_store_n(&g_roctx_state, 2, __ATOMIC_RELEASE);
}
void stage_push(const char* name) {
  roctx_resolve();
  if (__atomic_load_n(&g_roctx_state, __ATOMIC_ACQUIRE) == 1) (void)g_roctx_push(name);
}
void stage_pop() {
  if (__atomic_load_n(&g_roctx_state, __ATOMIC_ACQUIRE) == 1) (void)g_roctx_pop();
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

static int opt_index(const char* name) {
  if (!name) return -1;
  if (strncmp(name, "CHATTS_", 7) == 0) name += 7;
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, g_opt_names[i]) == 0) return i;
  return -1;
}
extern "C" int chatts_set_option(const char* name, int value) {
  const int i = opt_index(name);
  CHATTS_REQUIRE(i >= 0, CHATTS_E_BADARG, "set_option: unknown option '%s'", name ? name : "(null)");
  __atomic_store_n(&g_opts[i], opt_pack(value), __ATOMIC_RELAXED);
  return CHATTS_OK;
}
extern "C" int chatts_unset_option(const char* name) {
  if (name == nullptr) {      // all of them
      for (int i = 0; i < OPT_COUNT; ++i) __atomic_store_n(&g_opts[i], 0LL, __ATOMIC_RELAXED);
    return CHATTS_OK;
  }
  const int i = opt_index(name);
  CHATTS_REQUIRE(i >= 0, CHATTS_E_BADARG, "unset_option: unknown option '%s'", name);
  __atomic_store_n(&g_opts[i], 0LL, __ATOMIC_RELAXED);
  return CHATTS_OK;
}
extern "C" int chatts_get_option(const char* name, int* value, int* is_set) {
  const int i = opt_index(name);
  CHATTS_REQUIRE(i >= 0 && value, CHATTS_E_BADARG, "get_option: unknown option '%s'", name ? name : "(null)");
  const long long v = __atomic_load_n(&g_opts[i], __ATOMIC_RELAXED);
  *value = v == 0 ? 0 : (int)(unsigned)(v & 0xffffffffLL);
  if (is_set) *is_set = v != 0;
  return CHATTS_OK;
}
extern "C" const char* chatts_option_name(int index) { return index >= 0 && index < OPT_COUNT ? g_opt_names[index] : nullptr; }

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
