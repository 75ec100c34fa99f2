// tp.hip - tensor-parallel exchange for the decoder: one-shot peer-to-peer collectives over xGMI.
//
// The reference gets tensor parallelism from vLLM (tensor_parallel_size=k: NetManAIOps/ChatTS demo/demo_vllm.py:30,
// chatts/utils/llm_utils.py:154), i.e. NCCL all-reduces after o_proj and down_proj.  A decode step exchanges 96 vectors of
// H float32 (20 KB at batch 1): pure latency.  RCCL's ring / tree launch costs ~25 us per call and cannot be captured
// together with our kernels into one hipGraph, so the small exchanges are hand-written here (RCCL keeps the prefill-sized
// messages, chatts_amd/tp.py):
//
//   * every rank owns ONE exchange buffer (fine-grained device memory, exported with hipIpcGetMemHandle); every rank maps
//     all peers' buffers (hipIpcOpenMemHandle) -> a table of W device pointers.  xGMI is point to point: a rank PUSHES
//     its vector to all W-1 peers in parallel over its W-1 links (posted writes, no read round trip);
//   * the payload travels as 8-byte {tag, value} granules written by ONE system-scope store each: the data is its own
//     flag (the low-latency protocol: no fence, no separate flag round trip).  The receiver polls its LOCAL buffer until
//     the tag equals the call's epoch, then adds the W vectors in RANK ORDER - every rank computes bit-identical sums,
//     so the replicated residual stream never diverges between ranks;
//   * the epoch is a device-resident call counter bumped by the last workgroup of a collective, never a kernel argument: a
//     captured decode step replays with fresh epochs.  (Round 4: collectives that ride inside another kernel - the decode
//     GEMVs with ChattsLinearArgs.tp_reduce - do not bump it; the host counts them and the next bumping collective's epoch
//     accounts for them: TpParams::idx, tp_common.h.)  Slots alternate by epoch parity; a rank can run at
//     most one collective ahead of the slowest peer (it needs that peer's contribution to finish), so two slots suffice;
//   * every spin is bounded (~2 s of the 100 MHz wall clock); a timeout sets a status word instead of hanging the GPU.
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "common.h"
#include "tp_common.h"

namespace chatts {

// out[i] = (resid ? resid[i] : 0) + sum_r in_r[i], r = 0..W-1 in rank order.  out may alias resid (x += all-reduced delta).
// E elements per thread.  The exchange is pure latency, so nothing is serialised: a thread first pushes its E values to all W
// ranks (E * W independent stores), then keeps ALL of its E * W polls in flight at once and re-issues only the granules that
// have not arrived (tools/tp_exchange_bench.py: a one-at-a-time poll loop cost 15 us for 5120 values, this form ~1/2 of that).
template <int E>
__global__ __launch_bounds__(1024) void tp_allreduce_kernel(TpParams p, const float* __restrict__ in, const float* resid,
                                                           float* out, int64_t n) {
  const uint32_t epoch = tp_epoch(p);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t base = i0; base < n; base += stride * E) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int64_t i = base + (int64_t)e * stride;
      if (i < n) {
        const uint32_t bits = __float_as_uint(in[i]);
        for (int k = 0; k < p.world; ++k) {                   // start with myself, then ring order: spreads the links
          const int q = (p.rank + k) % p.world;
          put(push_ptr(p, q, epoch) + i, epoch, push_bits(p, q, bits));
        }
      }
    }
    uint32_t val[E][kMaxWorld];
    uint32_t pending = 0;                                     // bit e * 8 + src
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (base + (int64_t)e * stride < n)
        for (int src = 0; src < p.world; ++src) pending |= 1u << (e * kMaxWorld + src);
    uint64_t t0 = 0;
    while (pending) {
      uint64_t g[E][kMaxWorld];
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int src = 0; src < kMaxWorld; ++src)
          if (pending & (1u << (e * kMaxWorld + src)))
            g[e][src] = __hip_atomic_load(slot_ptr(p, p.rank, epoch, src) + base + (int64_t)e * stride, __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int src = 0; src < kMaxWorld; ++src)
          if ((pending & (1u << (e * kMaxWorld + src))) && (uint32_t)(g[e][src] >> 32) == epoch) {
            val[e][src] = (uint32_t)g[e][src];
            pending &= ~(1u << (e * kMaxWorld + src));
          }
      if (pending) {
        if (t0 == 0) t0 = wall_clock64();
        else if (wall_clock64() - t0 > spin_limit(&p.ctr[2])) { atomicOr(&p.ctr[2], 1u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int64_t i = base + (int64_t)e * stride;
      if (i < n) {
        float sum = 0.f;
#pragma unroll
        for (int src = 0; src < kMaxWorld; ++src)
          if (src < p.world) sum = src == 0 ? __uint_as_float(val[e][0]) : sum + __uint_as_float(val[e][src]);
        out[i] = resid ? resid[i] + sum : sum;
      }
    }
  }
  finish_call(p, epoch);
}

// in_r: [rows, row_len] of every rank -> out [rows, W * row_len]: row b = concatenation of the ranks' rows b in rank order
// (vocab-parallel logits -> full vocabulary, one row per sequence)
__global__ __launch_bounds__(1024) void tp_allgather_kernel(TpParams p, const float* __restrict__ in, float* out, int64_t rows,
                                                           int64_t row_len) {
  const uint32_t epoch = tp_epoch(p);
  const int64_t n_local = rows * row_len;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = i0; i < n_local; i += stride) {
    const uint32_t bits = __float_as_uint(in[i]);
    for (int k = 0; k < p.world; ++k) {
      const int q = (p.rank + k) % p.world;
      put(push_ptr(p, q, epoch) + i, epoch, push_bits(p, q, bits));
    }
  }
  bool ok = true;
  for (int src = 0; src < p.world && ok; ++src)
    for (int64_t i = i0; i < n_local && ok; i += stride) {
      uint32_t bits;
      ok = take(slot_ptr(p, p.rank, epoch, src) + i, epoch, bits, &p.ctr[2]);
      const int64_t b = i / row_len, v = i - b * row_len;
      out[(b * p.world + src) * row_len + v] = __uint_as_float(bits);
    }
  finish_call(p, epoch);
}

// Greedy token under a vocab-parallel lm_head: every rank contributes its local (max logit, global token id) per sequence;
// all ranks select the same winner (largest logit, ties -> lowest token id = torch.argmax over the full vocabulary) and
// apply the side effects of chatts_argmax_batched (token, logit, out_tokens[step], ++step, ++pos).  One wave per sequence
// (blockIdx.x = sequence b); lane l < W talks to rank l.
__global__ __launch_bounds__(64) void tp_argmax_kernel(TpParams p, const float* __restrict__ logit, const int64_t* __restrict__ token_in,
                                                      int64_t* token, float* token_logit, int64_t* out_tokens, int64_t out_stride,
                                                      int32_t* step_dev, int32_t* pos_dev, int pos_limit) {
  const uint32_t epoch = tp_epoch(p);
  const int lane = threadIdx.x, b = blockIdx.x;
  if (lane < p.world) {
    uint64_t* g = push_ptr(p, lane, epoch) + 2 * b;
    // (loop-back: the absent peers contribute -inf, so this rank's own pair wins)
    put(g, epoch, (p.loopback && lane != p.rank) ? 0xff800000u : __float_as_uint(logit[b]));
    put(g + 1, epoch, (uint32_t)token_in[b]);
  }
  float best = -INFINITY;
  uint32_t best_tok = 0xffffffffu;
  if (lane < p.world) {
    uint64_t* g = slot_ptr(p, p.rank, epoch, lane) + 2 * b;
    uint32_t lb = 0, tb = 0;
    if (take(g, epoch, lb, &p.ctr[2]) && take(g + 1, epoch, tb, &p.ctr[2])) { best = __uint_as_float(lb); best_tok = tb; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const uint32_t ot = __shfl_xor(best_tok, o, 64);
    if (ob > best || (ob == best && ot < best_tok)) { best = ob; best_tok = ot; }
  }
  if (lane == 0) {
    token[b] = (int64_t)best_tok;
    if (token_logit) token_logit[b] = best;
    if (out_tokens && step_dev) out_tokens[(int64_t)b * out_stride + step_dev[b]] = (int64_t)best_tok;
    if (step_dev) step_dev[b] += 1;
    if (pos_dev && pos_dev[b] >= 0) { const int np = pos_dev[b] + 1; pos_dev[b] = (pos_limit > 0 && np > pos_limit) ? pos_limit : np; }   // parked slots (pos < 0) stay parked
  }
  finish_call(p, epoch);
}


// ---- prefill-sized sums: two-shot all-reduce (direct reduce-scatter, then direct all-gather) over the same mapped buffers ---------
// A prefill chunk sums [T, H] float32 partials twice per layer (16 MB at T = 798): as 8-byte granules that would be 2 x the bytes to
// all W - 1 peers each (230 MB per rank and sum); RCCL does it well but costs a host-enqueued launch between every two layer
// halves (96 per chunk, nothing capturable).  Here ONE kernel per sum, enqueued by the same C call as the layer halves
// (chatts_decoder_prefill under tensor parallelism):
//   1. scatter: rank r stores slice s of its partial into rank s's buffer (area "from r"), 16 bytes per lane, all W - 1 links busy;
//      a system-scope fence, then one flag per (source, workgroup) carrying the epoch;
//   2. rank s waits for its W flags, adds the W areas of its slice IN RANK ORDER (one rank forms each element's sum: every rank
//      receives the same bits) and stores the reduced slice into every rank's gather area; fence; flag per (slice, workgroup);
//   3. every rank waits for the W slices and adds them to its residual stream: x += sum.
// (W - 1) / W of the vector crosses each rank's links twice - 2 x 14 MB over 7 links at T = 798, W = 8.  Workgroup b handles the same
// sub-range of every slice on every rank, so a workgroup only ever waits for workgroups of the SAME index on other ranks.  Areas
// and flags alternate by epoch parity like the granule slots; flags carry the epoch, nothing is ever reset.  Plain 16-byte stores
// into the peers' uncached memory + __threadfence_system() before the flag; the reader fences after its poll.
struct BulkView {
  uint32_t* flags;     // [2 phases][W][kBulkMaxBlocks] of this slot
  float* scatter;      // [W sources][slice_cap]
  float* gather;       // [W slices][slice_cap]
};
__device__ __forceinline__ BulkView bulk_view(const TpParams& p, int owner, uint32_t epoch) {
  char* base = reinterpret_cast<char*>(p.peer[owner]) + p.bulk_off;
  const int slot = (int)(epoch & 1u);
  BulkView v;
  v.flags = reinterpret_cast<uint32_t*>(base) + (size_t)slot * 2 * kMaxWorld * kBulkMaxBlocks;
  float* data = reinterpret_cast<float*>(base + kBulkFlagBytes) + (size_t)slot * 2 * p.world * p.slice_cap;
  v.scatter = data;
  v.gather = data + (size_t)p.world * p.slice_cap;
  return v;
}
// wait until flag == epoch (one thread); false on timeout
__device__ __forceinline__ bool wait_flag(uint32_t* f, uint32_t epoch, uint32_t* status) {
  if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == epoch) return true;
  const uint64_t t0 = wall_clock64();
  while (true) {
    __builtin_amdgcn_s_sleep(2);
    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == epoch) return true;
    if (wall_clock64() - t0 > spin_limit(status)) { atomicOr(status, 1u); return false; }
  }
}

// What a workgroup does between its stores into the peers' areas and the flag that publishes them.  The areas are UNCACHED memory
// (chatts_tp_buffer_alloc: hipDeviceMallocUncached, mapped the same way by every peer), so the stores themselves never sit in a cache:
// they are complete - at the remote memory - when `s_waitcnt vmcnt(0)` returns.  __threadfence_system() adds a write-back of EVERY dirty
// line of this XCD's L2 (buffer_wbl2 sc0 sc1), among them the 16 MB of partial sums the projection in front of this kernel has just
// left there: TP_BULK_FENCE=1 selects it (the round-4 form), the default drains the stores only.
__device__ __forceinline__ void bulk_release(int light) {
  if (light) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else __threadfence_system();
}
// (kBulkThreads threads per workgroup: 256 is the measured choice, see chatts_allreduce_bulk)
template <int kBulkThreads>
__global__ __launch_bounds__(kBulkThreads) void tp_allreduce_bulk_kernel(TpParams p, const float* __restrict__ in, float* x, int64_t n, int light_fence) {
  const uint32_t epoch = tp_epoch(p);
  const int W = p.world, b = blockIdx.x, tid = threadIdx.x;
  // slice s = elements [s * slice, (s + 1) * slice) of the vector (multiples of 4 floats); workgroup b owns [lo, hi) of every slice
  const int64_t slice = ((n + W - 1) / W + 3) / 4 * 4;
  const int64_t per = ((slice + gridDim.x - 1) / gridDim.x + 3) / 4 * 4;
  const int64_t lo = (int64_t)b * per, hi = lo + per < slice ? lo + per : slice;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // 1. scatter my partial: slice s goes to rank s (loop-back: into my own "from q" area, zeros for the absent peers).  Every pass of
  //    this kernel is a chain of round trips to uncached memory, so a thread requests ALL of its positions at once - kU positions x W
  //    loads in flight per lane - before the first store (round 5: one position at a time was 4 dependent trips per pass at [798, 5120]).
  constexpr int kU = kBulkThreads >= 1024 ? 1 : (kBulkThreads >= 512 ? 2 : 4);
  for (int64_t i0 = lo + tid * 4; i0 < hi; i0 += (int64_t)kU * kBulkThreads * 4) {
    f32x4 v[kU][kMaxWorld];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + (int64_t)u * kBulkThreads * 4;
#pragma unroll
      for (int k = 0; k < kMaxWorld; ++k) {
        const int s = (p.rank + k) % W;
        const int64_t g = (int64_t)(p.loopback ? p.rank : s) * slice + i;
        v[u][k] = zero;
        if (i < hi && k < W && !(p.loopback && s != p.rank)) {
          if (g + 3 < n) v[u][k] = *reinterpret_cast<const f32x4*>(in + g);
          else { float t[4] = {0.f, 0.f, 0.f, 0.f}; for (int e = 0; e < 4; ++e) if (g + e < n) t[e] = in[g + e]; v[u][k] = (f32x4){t[0], t[1], t[2], t[3]}; }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + (int64_t)u * kBulkThreads * 4;
      if (i >= hi) continue;
#pragma unroll
      for (int k = 0; k < kMaxWorld; ++k) {
        if (k < W) {
          const int s = (p.rank + k) % W;
          const BulkView dst = bulk_view(p, p.loopback ? p.rank : s, epoch);
          *reinterpret_cast<f32x4*>(dst.scatter + (size_t)(p.loopback ? s : p.rank) * p.slice_cap + i) = v[u][k];
        }
      }
    }
  }
  bulk_release(light_fence);
  __syncthreads();
  if (tid < W) {
    const int s = (p.rank + tid) % W;
    const BulkView dst = bulk_view(p, p.loopback ? p.rank : s, epoch);
    __hip_atomic_store(dst.flags + (size_t)(p.loopback ? s : p.rank) * kBulkMaxBlocks + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 2. reduce my slice (this workgroup's range) in rank order and hand it to everybody
  const BulkView mine = bulk_view(p, p.rank, epoch);
  if (tid < W) (void)wait_flag(mine.flags + (size_t)tid * kBulkMaxBlocks + b, epoch, &p.ctr[2]);     // (a timeout raises the status bit)
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  for (int64_t i0 = lo + tid * 4; i0 < hi; i0 += (int64_t)kU * kBulkThreads * 4) {
    f32x4 c[kU][kMaxWorld];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + (int64_t)u * kBulkThreads * 4;
#pragma unroll
      for (int src = 0; src < kMaxWorld; ++src)
        c[u][src] = (src < W && i < hi) ? *reinterpret_cast<const f32x4*>(mine.scatter + (size_t)src * p.slice_cap + i) : zero;      // kU W loads in flight
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + (int64_t)u * kBulkThreads * 4;
      if (i >= hi) continue;
      f32x4 sum = c[u][0];
#pragma unroll
      for (int src = 1; src < kMaxWorld; ++src)
        if (src < W) { sum.x += c[u][src].x; sum.y += c[u][src].y; sum.z += c[u][src].z; sum.w += c[u][src].w; }                  // rank order
#pragma unroll
      for (int k = 0; k < kMaxWorld; ++k) {
        if (k < W) {
          const int q = (p.rank + k) % W;
          const BulkView dst = bulk_view(p, p.loopback ? p.rank : q, epoch);
          *reinterpret_cast<f32x4*>(dst.gather + (size_t)(p.loopback ? q : p.rank) * p.slice_cap + i) = (p.loopback && q != p.rank) ? zero : sum;
        }
      }
    }
  }
  bulk_release(light_fence);
  __syncthreads();
  if (tid < W) {
    const int q = (p.rank + tid) % W;
    const BulkView dst = bulk_view(p, p.loopback ? p.rank : q, epoch);
    __hip_atomic_store(dst.flags + (size_t)(kMaxWorld + (p.loopback ? q : p.rank)) * kBulkMaxBlocks + b, epoch, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 3. x += the reduced vector (all W slices, this workgroup's range of each)
  if (tid < W) (void)wait_flag(mine.flags + (size_t)(kMaxWorld + tid) * kBulkMaxBlocks + b, epoch, &p.ctr[2]);
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  constexpr int kU3 = kU > 1 ? kU / 2 : 1;                                          // (two positions x 2 W loads: the gathered slices and x)
  for (int64_t i0 = lo + tid * 4; i0 < hi; i0 += (int64_t)kU3 * kBulkThreads * 4) {
    f32x4 r[kU3][kMaxWorld], xv[kU3][kMaxWorld];
#pragma unroll
    for (int u = 0; u < kU3; ++u) {
      const int64_t i = i0 + (int64_t)u * kBulkThreads * 4;
#pragma unroll
      for (int s = 0; s < kMaxWorld; ++s) {
        const int64_t g = (int64_t)s * slice + i;
        r[u][s] = (s < W && i < hi) ? *reinterpret_cast<const f32x4*>(mine.gather + (size_t)s * p.slice_cap + i) : zero;
        xv[u][s] = (s < W && i < hi && g + 3 < n) ? *reinterpret_cast<const f32x4*>(x + g) : zero;
      }
    }
#pragma unroll
    for (int u = 0; u < kU3; ++u) {
      const int64_t i = i0 + (int64_t)u * kBulkThreads * 4;
      if (i >= hi) continue;
#pragma unroll
      for (int s = 0; s < kMaxWorld; ++s) {
        const int64_t g = (int64_t)s * slice + i;
        if (s >= W || g >= n) continue;
        if (g + 3 < n) {
          f32x4 v = xv[u][s];
          v.x += r[u][s].x; v.y += r[u][s].y; v.z += r[u][s].z; v.w += r[u][s].w;
          *reinterpret_cast<f32x4*>(x + g) = v;
        } else {
          const float t[4] = {r[u][s].x, r[u][s].y, r[u][s].z, r[u][s].w};
          for (int e = 0; e < 4; ++e) if (g + e < n) x[g + e] += t[e];
        }
      }
    }
  }
  finish_call(p, epoch);
}

__global__ void tp_reset_kernel(uint64_t* buf, int64_t granules, uint32_t* ctr, uint32_t* bulk_flags) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < granules; i += stride) buf[i] = 0;
  if (bulk_flags)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)(kBulkFlagBytes / 4); i += stride) bulk_flags[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x < 4) ctr[threadIdx.x] = 0;
}

}  // namespace chatts

using namespace chatts;

struct ChattsTpComm {
  TpParams p{};
  void* opened[kMaxWorld] = {};      // IPC mappings to close
  bool ipc = false;
  size_t bytes = 0;
  uint32_t pending = 0;              // collectives issued since the last counter bump (TpParams::idx of the next one)
  // Release form of the bulk sums (bulk_release()).  The light form - s_waitcnt vmcnt(0) before the flags - has only ever run with every
  // rank on ONE device; whether a store to a peer's uncached buffer is visible THERE when vmcnt returns is unproven across xGMI / PCIe.
  // So: peers on other devices (or unknown) => the system-scope fence, until the host has validated the light form on these very
  // links and says so (chatts_tp_set_bulk_release(c, 0): chatts_amd/tp.py's first-contact test).  The TP_BULK_FENCE option, when set,
  // overrides both ways (A/B runs).
  int cross_device = 0;              // some peer buffer lives on another device than ours (or its device could not be determined)
  int shared_device = 0;             // several ranks' buffers on THIS device, not loop-back: their waiting grids must be resident together
  int release_mode = -1;             // -1 = by cross_device, 0 = light, 1 = fence
};
static int ptr_device(const void* p) {
  hipPointerAttribute_t a;
  if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return a.device;
}
static void tp_classify_peers(ChattsTpComm* c) {
  // over ALL pairs of ranks, not only the ones involving this rank: every rank must arrive at the same two flags (the bulk sum's grid
  // follows from them and its flags are per workgroup) - also when the ranks are spread unevenly over the devices
  int dev[kMaxWorld];
  for (int r = 0; r < c->p.world; ++r) dev[r] = ptr_device(c->p.peer[r]);
  for (int i = 0; i < c->p.world; ++i)
    for (int j = i + 1; j < c->p.world; ++j) {
      if (dev[i] < 0 || dev[j] < 0 || dev[i] != dev[j]) c->cross_device = 1;
      else c->shared_device = 1;
    }
}

extern "C" size_t chatts_tp_buffer_bytes(int world, int64_t max_elems) {
  if (world < 1 || world > kMaxWorld || max_elems < 1) return 0;
  const int64_t per = (max_elems + 1) / 2 * 2;
  return (size_t)2 * world * per * sizeof(uint64_t) + 256;        // + the counter words at the end
}

static int64_t bulk_slice_cap(int world, int64_t bulk_elems) { return ((bulk_elems + world - 1) / world + 3) / 4 * 4 + 4; }
extern "C" size_t chatts_tp_buffer_bytes_bulk(int world, int64_t max_elems, int64_t bulk_elems) {
  const size_t g = chatts_tp_buffer_bytes(world, max_elems);
  if (g == 0 || bulk_elems <= 0) return g;
  // flags, then [2 slots][scatter W areas + gather W slices] of slice_cap floats
  return g + kBulkFlagBytes + (size_t)2 * 2 * world * bulk_slice_cap(world, bulk_elems) * sizeof(float);
}

// Exchange buffers are NEVER handed back to the driver while the process lives: a freed buffer goes to a free list and serves a
// later chatts_tp_buffer_alloc.  Measured (round 2): after hipFree of a hipDeviceMallocUncached allocation, memory the driver
// then hands to OTHER allocations (torch's next hipMalloc segment) misbehaved - kernels reading what earlier kernels of the same
// stream had just written saw stale data until the next hipDeviceSynchronize (first model built after an exchange was closed:
// logits off by 20 %; the same sequence with plain hipMalloc buffers, or with this free list: exact).  A long-lived server
// allocates its buffer once; tests and notebooks that create and drop TP models repeatedly are what this protects.
extern "C" int chatts_tp_buffer_free(void* dev_ptr);
struct TpBufRec { void* ptr; size_t bytes; bool in_use; };
static std::mutex g_buf_mu;
static std::vector<TpBufRec> g_bufs;

extern "C" int chatts_tp_buffer_alloc(size_t bytes, void** dev_ptr, uint8_t* handle /* [CHATTS_TP_HANDLE_BYTES] */) {
  CHATTS_REQUIRE(dev_ptr && bytes >= 512, CHATTS_E_BADARG, "tp_buffer_alloc: bad arguments");
  void* ptr = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_buf_mu);
    TpBufRec* best = nullptr;
    for (auto& r : g_bufs)
      if (!r.in_use && r.bytes >= bytes && (!best || r.bytes < best->bytes)) best = &r;
    if (best) { best->in_use = true; ptr = best->ptr; }
  }
  hipError_t e = hipSuccess;
  if (!ptr) {
    // fine-grained (uncached) device memory: remote stores and local polls must not sit in a non-coherent L2
    e = hipExtMallocWithFlags(&ptr, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&ptr, bytes, hipDeviceMallocFinegrained); }
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&ptr, bytes); }
    CHATTS_REQUIRE(e == hipSuccess, CHATTS_E_LAUNCH, "tp_buffer_alloc: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lk(g_buf_mu);
    g_bufs.push_back(TpBufRec{ptr, bytes, true});
  }
  e = hipMemset(ptr, 0, bytes);           // tags 0: no epoch ever matches what an earlier owner left behind
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)chatts_tp_buffer_free(ptr);
    set_error("tp_buffer_alloc: memset: %s", hipGetErrorString(e));
    return CHATTS_E_LAUNCH;
  }
  if (handle) {
    static_assert(sizeof(hipIpcMemHandle_t) <= CHATTS_TP_HANDLE_BYTES, "IPC handle does not fit");
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, ptr);
    if (e != hipSuccess) {
      (void)chatts_tp_buffer_free(ptr);
      set_error("tp_buffer_alloc: hipIpcGetMemHandle: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)", hipGetErrorString(e));
      return CHATTS_E_LAUNCH;
    }
    memset(handle, 0, CHATTS_TP_HANDLE_BYTES);
    memcpy(handle, &h, sizeof(h));
  }
  *dev_ptr = ptr;
  return CHATTS_OK;
}

extern "C" int chatts_tp_buffer_free(void* dev_ptr) {
  if (!dev_ptr) return CHATTS_OK;
  std::lock_guard<std::mutex> lk(g_buf_mu);
  for (auto& r : g_bufs)
    if (r.ptr == dev_ptr) {
      CHATTS_REQUIRE(r.in_use, CHATTS_E_BADARG, "tp_buffer_free: buffer %p freed twice", dev_ptr);
      r.in_use = false;                   // parked for the next chatts_tp_buffer_alloc; the driver gets it back at process exit
      return CHATTS_OK;
    }
  set_error("tp_buffer_free: %p was not allocated by chatts_tp_buffer_alloc", dev_ptr);
  return CHATTS_E_BADARG;
}

static ChattsTpComm* tp_make(int rank, int world, int64_t max_elems, size_t bytes) {
  if (rank < 0 || world < 1 || world > kMaxWorld || rank >= world || max_elems < 2) {
    set_error("tp_init: bad rank %d / world %d / max_elems %lld", rank, world, (long long)max_elems);
    return nullptr;
  }
  if (bytes < chatts_tp_buffer_bytes(world, max_elems)) {
    set_error("tp_init: exchange buffer of %zu bytes < %zu", bytes, chatts_tp_buffer_bytes(world, max_elems));
    return nullptr;
  }
  ChattsTpComm* c = new (std::nothrow) ChattsTpComm();
  if (!c) { set_error("tp_init: out of host memory"); return nullptr; }
  c->p.rank = rank; c->p.world = world; c->p.max_elems = (max_elems + 1) / 2 * 2; c->bytes = bytes;
  // whatever lies behind the granule slots + counter words is the bulk region (chatts_tp_buffer_bytes_bulk): same layout on every rank
  const size_t gbytes = chatts_tp_buffer_bytes(world, max_elems);
  if (bytes > gbytes + kBulkFlagBytes + (size_t)64 * world) {
    c->p.bulk_off = (int64_t)gbytes;
    c->p.slice_cap = (int64_t)((bytes - gbytes - kBulkFlagBytes) / ((size_t)2 * 2 * world * sizeof(float))) / 4 * 4;
  }
  return c;
}

static void tp_bind_local(ChattsTpComm* c, void* local) {
  c->p.peer[c->p.rank] = reinterpret_cast<uint64_t*>(local);
  c->p.ctr = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(local) + (size_t)2 * c->p.world * c->p.max_elems * sizeof(uint64_t));
}

extern "C" ChattsTpComm* chatts_tp_init(int rank, int world, void* local_buf, const uint8_t* handles, size_t bytes, int64_t max_elems) {
  ChattsTpComm* c = tp_make(rank, world, max_elems, bytes);
  if (!c) return nullptr;
  if (!local_buf || (world > 1 && !handles)) { set_error("tp_init: null buffer / handles"); delete c; return nullptr; }
  tp_bind_local(c, local_buf);
  c->ipc = true;
  for (int r = 0; r < world; ++r) {
    if (r == rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)r * CHATTS_TP_HANDLE_BYTES, sizeof(h));
    void* ptr = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      set_error("tp_init: hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
      (void)hipGetLastError();
      chatts_tp_destroy(c);
      return nullptr;
    }
    c->opened[r] = ptr;
    c->p.peer[r] = reinterpret_cast<uint64_t*>(ptr);
  }
  tp_classify_peers(c);
  return c;
}

extern "C" ChattsTpComm* chatts_tp_init_local(int rank, int world, void* const* bufs, size_t bytes, int64_t max_elems) {
  ChattsTpComm* c = tp_make(rank, world, max_elems, bytes);
  if (!c) return nullptr;
  if (!bufs) { set_error("tp_init_local: null buffer table"); delete c; return nullptr; }
  for (int r = 0; r < world; ++r) {
    if (!bufs[r]) { set_error("tp_init_local: null buffer of rank %d", r); delete c; return nullptr; }
    c->p.peer[r] = reinterpret_cast<uint64_t*>(bufs[r]);
  }
  tp_bind_local(c, bufs[rank]);
  tp_classify_peers(c);
  return c;
}

// ONE rank of a `world`-rank group alone on its device: every peer pointer is the local buffer and the kernels run in loop-back
// mode (tp_common.h) - the same stores and polls per element as a real step, no partner needed.  For measuring a rank's step.
extern "C" ChattsTpComm* chatts_tp_init_loopback(int rank, int world, void* local_buf, size_t bytes, int64_t max_elems) {
  ChattsTpComm* c = tp_make(rank, world, max_elems, bytes);
  if (!c) return nullptr;
  if (!local_buf) { set_error("tp_init_loopback: null buffer"); delete c; return nullptr; }
  for (int r = 0; r < world; ++r) c->p.peer[r] = reinterpret_cast<uint64_t*>(local_buf);
  tp_bind_local(c, local_buf);
  c->p.loopback = 1;
  return c;
}

namespace chatts {
TpParams tp_issue(ChattsTpComm* c, bool bumps) {
  TpParams p = c->p;
  p.idx = c->pending;
  c->pending = bumps ? 0u : c->pending + 1u;
  return p;
}
int64_t tp_capacity(const ChattsTpComm* c) { return c ? c->p.max_elems : 0; }
}  // namespace chatts

extern "C" void chatts_tp_destroy(ChattsTpComm* c) {
  if (!c) return;
  if (c->ipc)
    for (int r = 0; r < kMaxWorld; ++r)
      if (c->opened[r] && hipIpcCloseMemHandle(c->opened[r]) != hipSuccess) (void)hipGetLastError();
  delete c;
}

extern "C" int chatts_tp_cross_device(const ChattsTpComm* c) { return c ? c->cross_device : -1; }
extern "C" int chatts_tp_set_cross_device(ChattsTpComm* c, int cross) {
  CHATTS_REQUIRE(c, CHATTS_E_BADARG, "tp_set_cross_device: null comm");
  c->cross_device = cross ? 1 : 0;
  return CHATTS_OK;
}
extern "C" int chatts_tp_set_bulk_release(ChattsTpComm* c, int mode) {
  CHATTS_REQUIRE(c && mode >= -1 && mode <= 1, CHATTS_E_BADARG, "tp_set_bulk_release: comm / mode (-1 = by device, 0 = light, 1 = fence)");
  c->release_mode = mode;
  return CHATTS_OK;
}
// the form the next bulk sum uses: 1 = system-scope fence, 0 = light (drain only)
extern "C" int chatts_tp_bulk_release(const ChattsTpComm* c) {
  if (!c) return -1;
  const int opt = opt_get(OPT_TP_BULK_FENCE, -1);
  if (opt >= 0) return opt != 0;
  if (c->release_mode >= 0) return c->release_mode;
  return c->cross_device && !c->p.loopback;
}
extern "C" int chatts_tp_rank(const ChattsTpComm* c) { return c ? c->p.rank : -1; }
extern "C" int chatts_tp_world(const ChattsTpComm* c) { return c ? c->p.world : -1; }
extern "C" int64_t chatts_tp_max_elems(const ChattsTpComm* c) { return c ? c->p.max_elems : 0; }
// diagnostic, NOT for the hot path: blocks until the device is idle on the null stream's terms (hipMemcpy)
extern "C" int chatts_tp_status(ChattsTpComm* c) {
  CHATTS_REQUIRE(c, CHATTS_E_BADARG, "tp_status: null comm");
  uint32_t v = 0;
  const hipError_t e = hipMemcpy(&v, c->p.ctr + 2, sizeof(v), hipMemcpyDeviceToHost);
  CHATTS_REQUIRE(e == hipSuccess, CHATTS_E_LAUNCH, "tp_status: %s", hipGetErrorString(e));
  return (int)(v & 0x7fffffffu);
}

namespace chatts {
__global__ void tp_bump_kernel(uint32_t* ctr, uint32_t by) { ctr[0] += by; }
}  // namespace chatts
extern "C" int chatts_tp_pending(const ChattsTpComm* c) { return c ? (int)c->pending : 0; }
extern "C" int chatts_tp_flush_epochs(ChattsTpComm* c, chatts_stream_t stream) {
  CHATTS_REQUIRE(c, CHATTS_E_BADARG, "tp_flush_epochs: null comm");
  if (c->pending == 0) return CHATTS_OK;
  hipLaunchKernelGGL(tp_bump_kernel, dim3(1), dim3(1), 0, as_stream(stream), c->p.ctr, c->pending);
  CHATTS_CHECK_LAUNCH("tp_flush_epochs");
  c->pending = 0;
  return CHATTS_OK;
}

extern "C" int chatts_tp_reset(ChattsTpComm* c, chatts_stream_t stream) {
  CHATTS_REQUIRE(c, CHATTS_E_BADARG, "tp_reset: null comm");
  const int64_t granules = (int64_t)2 * c->p.world * c->p.max_elems;
  hipLaunchKernelGGL(tp_reset_kernel, dim3(256), dim3(256), 0, as_stream(stream), c->p.peer[c->p.rank], granules, c->p.ctr,
                     c->p.bulk_off ? reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(c->p.peer[c->p.rank]) + c->p.bulk_off) : nullptr);
  c->pending = 0;
  CHATTS_CHECK_LAUNCH("tp_reset");
  return CHATTS_OK;
}

static int tp_blocks(int64_t n) {      // gather / argmax: one 1024-thread workgroup up to 8 k elements, then 4 per thread
  if (n <= 8192) return 1;
  const int64_t b = (n + 4095) / 4096;
  return (int)(b > 64 ? 64 : b);
}

extern "C" int chatts_allreduce(ChattsTpComm* c, const float* in, float* out, const float* resid, int64_t n, chatts_stream_t stream) {
  CHATTS_REQUIRE(c && in && out, CHATTS_E_BADARG, "allreduce: null argument");
  CHATTS_REQUIRE(n >= 0 && n <= c->p.max_elems, CHATTS_E_SHAPE, "allreduce: %lld elements exceed the exchange buffer (%lld)",
                 (long long)n, (long long)c->p.max_elems);
  if (n == 0) return CHATTS_OK;
  // One value per thread with every poll of the vector in flight at once, as long as the grid stays small: up to 128 workgroups
  // (131072 values: the [16, H] sums of a 16-wide decode step - 19.0 us with the 4-values-per-thread form on 20 workgroups, the
  // largest item of a TP = 8 rank's batched step, profiles/r4_tp8_cfg5_shard_kernel_trace.txt).  CHATTS_TP_AR_BLOCKS lowers the bound
  // for several ranks emulated on ONE device (their waiting grids must be resident together).
  const int ar_cap = opt_get(OPT_TP_AR_BLOCKS, 128);
  if ((n + 1023) / 1024 <= (ar_cap > 16 ? ar_cap : 16)) {
    hipLaunchKernelGGL(tp_allreduce_kernel<1>, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, as_stream(stream), tp_issue(c, true), in, resid, out, n);
  } else {
    const int64_t b = (n + 4095) / 4096;
    hipLaunchKernelGGL(tp_allreduce_kernel<4>, dim3((unsigned)(b > 128 ? 128 : b)), dim3(1024), 0, as_stream(stream), tp_issue(c, true), in, resid, out, n);
  }
  CHATTS_CHECK_LAUNCH("tp_allreduce");
  return CHATTS_OK;
}

extern "C" int64_t chatts_tp_bulk_elems(const ChattsTpComm* c) {
  return (c && c->p.bulk_off) ? (c->p.slice_cap - 4) * c->p.world : 0;
}

extern "C" int chatts_allreduce_bulk(ChattsTpComm* c, const float* in, float* x, int64_t n, chatts_stream_t stream) {
  CHATTS_REQUIRE(c && in && x, CHATTS_E_BADARG, "allreduce_bulk: null argument");
  CHATTS_REQUIRE(n >= 0 && n <= chatts_tp_bulk_elems(c), CHATTS_E_SHAPE, "allreduce_bulk: %lld elements exceed the bulk region (%lld)",
                 (long long)n, (long long)chatts_tp_bulk_elems(c));
  CHATTS_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)x % 16) == 0, CHATTS_E_SHAPE, "allreduce_bulk: pointers must be 16-byte aligned");
  if (n == 0) return CHATTS_OK;
  // one workgroup per ~8 KB of a slice, capped below (every rank computes the same grid from n: the flags are per workgroup).  Round 5
  // tried 256 workgroups x 512 threads (one trip per thread and pass instead of four): 102 us against 51 us per [798, 5120] sum on a
  // loop-back TP = 8 rank - the kernel is paced by its per-workgroup fences and flag round trips (two __threadfence_system() and 2 W
  // system-scope flags each), not by the three passes; profiles/r5_tp_bulk_sweep.txt has the (workgroups, threads) grid.
  // TP_BULK_BLOCKS also lowers the bound for several ranks emulated on ONE device (their waiting grids must be resident together).
  const int64_t slice = ((n + c->p.world - 1) / c->p.world + 3) / 4 * 4;
  int64_t blocks = (slice + 2047) / 2048;
  // cap 256 with the light release (round 5): W = 8 is flat from 128 to 256 workgroups (38.5 / 38.7 us), W = 2 gains 10 us (51.6 -> 41.6);
  // with the system-scope fence every workgroup costs, and 128 is its minimum (50.4 us against 65.6 at 256)
  const int light = chatts_tp_bulk_release(c) == 0;
  // (several ranks on one device - the emulation every test of this box runs: W waiting grids + a persistent prefill GEMM of one 144 KB
  //  workgroup per CU must be resident together, 64 workgroups per rank is what fits)
  const int cap = opt_get(OPT_TP_BULK_BLOCKS, (c->shared_device && !c->p.loopback) ? 64 : (light ? 256 : 128));
  if (blocks > cap) blocks = cap;
  if (blocks > kBulkMaxBlocks) blocks = kBulkMaxBlocks;
  if (blocks < 1) blocks = 1;
  const TpParams tp = tp_issue(c, true);
  switch (opt_get(OPT_TP_BULK_THREADS, 256)) {
    case 1024: hipLaunchKernelGGL(tp_allreduce_bulk_kernel<1024>, dim3((unsigned)blocks), dim3(1024), 0, as_stream(stream), tp, in, x, n, light); break;
    case 512: hipLaunchKernelGGL(tp_allreduce_bulk_kernel<512>, dim3((unsigned)blocks), dim3(512), 0, as_stream(stream), tp, in, x, n, light); break;
    default: hipLaunchKernelGGL(tp_allreduce_bulk_kernel<256>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), tp, in, x, n, light); break;
  }
  CHATTS_CHECK_LAUNCH("tp_allreduce_bulk");
  return CHATTS_OK;
}

extern "C" int chatts_allgather(ChattsTpComm* c, const float* in, float* out, int64_t rows, int64_t row_len, chatts_stream_t stream) {
  CHATTS_REQUIRE(c && in && out, CHATTS_E_BADARG, "allgather: null argument");
  CHATTS_REQUIRE(rows >= 0 && row_len >= 0 && rows * row_len <= c->p.max_elems, CHATTS_E_SHAPE,
                 "allgather: %lld x %lld elements exceed the exchange buffer (%lld)", (long long)rows, (long long)row_len,
                 (long long)c->p.max_elems);
  if (rows * row_len == 0) return CHATTS_OK;
  hipLaunchKernelGGL(tp_allgather_kernel, dim3(tp_blocks(rows * row_len)), dim3(1024), 0, as_stream(stream), tp_issue(c, true), in, out, rows, row_len);
  CHATTS_CHECK_LAUNCH("tp_allgather");
  return CHATTS_OK;
}

extern "C" int chatts_tp_argmax(ChattsTpComm* c, int batch, const float* local_logit, const int64_t* local_token, int64_t* token,
                                float* token_logit, int64_t* out_tokens, int64_t out_stride, int32_t* step_dev, int32_t* pos_dev,
                                int pos_limit, chatts_stream_t stream) {
  CHATTS_REQUIRE(c && local_logit && local_token && token, CHATTS_E_BADARG, "tp_argmax: null argument");
  CHATTS_REQUIRE(batch >= 1 && 2 * (int64_t)batch <= c->p.max_elems, CHATTS_E_SHAPE, "tp_argmax: batch %d", batch);
  hipLaunchKernelGGL(tp_argmax_kernel, dim3(batch), dim3(64), 0, as_stream(stream), tp_issue(c, true), local_logit, local_token, token, token_logit,
                     out_tokens, out_stride, step_dev, pos_dev, pos_limit);
  CHATTS_CHECK_LAUNCH("tp_argmax");
  return CHATTS_OK;
}
