// tp_common.h - the device side of the tensor-parallel exchange protocol (csrc/tp.hip), shared with the kernels that carry an
// exchange in their own epilogue (csrc/gemv.hip: o_proj / down_proj of a decode step push their partial rows to the peers and
// reduce them inside the same launch).  Protocol: see the head of tp.hip.
#pragma once
#include "common.h"

namespace chatts {

constexpr int kMaxWorld = CHATTS_TP_MAX_WORLD;
constexpr uint64_t kSpinTicks = 200000000ull;      // wall_clock64 runs at 100 MHz: 2 s
constexpr uint64_t kSpinTicksBroken = 2000ull;     // 20 us once a timeout has been recorded
// A communicator that has timed out once is broken until chatts_tp_reset: a step holds ~100 collectives, and each of them
// waiting its own 2 s would keep the GPU busy for minutes before the host reads the status at its next sync point.
__device__ __forceinline__ uint64_t spin_limit(const uint32_t* status) { return *status ? kSpinTicksBroken : kSpinTicks; }

struct TpParams {
  uint64_t* peer[kMaxWorld];   // peer[p]: rank p's exchange buffer as mapped in this process (peer[rank] = local)
  uint32_t* ctr;               // device words: [0] completed-call counter (epoch - 1), [1] arrivals, [2] status
  int rank, world;
  int64_t max_elems;           // granules per (slot, source rank)
  // Loop-back (chatts_tp_init_loopback, tools/tp_shard_step.py): ONE rank of a W-rank group alone on a device.  Every push lands in
  // the LOCAL buffer, in the slot of the peer it would have gone to, carrying 0.0 for every peer but the rank itself: the same
  // W stores per element and the same W polls as a real step, zero link latency - what one GPU can measure of a rank's step time.
  int loopback;
  // Epoch of a collective = ctr[0] + 1 + idx.  ctr[0] counts the collectives COMPLETED up to the last counter bump; idx = collectives
  // issued since then whose kernels do not bump it (the exchange-carrying GEMVs of a decode step: 96 per token - their launches would
  // each end with a workgroup barrier and a returning atomic on the arrival counter just to advance a number the host can count
  // too).  The host tracks idx per communicator (tp_issue); a bumping kernel stores its own epoch into ctr[0], which resets idx.  A
  // captured step has its idx values baked in and ends with a bumping collective (the token agreement), so every replay starts from
  // the fresh device counter.
  uint32_t idx;
  // Bulk region (prefill-sized sums, tp_allreduce_bulk_kernel): lies behind the granule slots and the counter words at the same
  // offset in every rank's buffer - flags [2 slots][2 phases][W][kBulkMaxBlocks] uint32, then per slot W "scatter" areas (one per
  // source rank) and one "gather" area of W slices, slice_cap floats each.  bulk_off == 0: the buffer has no bulk region.
  int64_t bulk_off;            // bytes from the buffer base
  int64_t slice_cap;           // floats per slice (multiple of 4)
};
constexpr int kBulkMaxBlocks = 256;
constexpr size_t kBulkFlagBytes = (size_t)2 * 2 * kMaxWorld * kBulkMaxBlocks * sizeof(uint32_t);
__device__ __forceinline__ uint32_t tp_epoch(const TpParams& p) { return p.ctr[0] + 1u + p.idx; }

__device__ __forceinline__ uint64_t* slot_ptr(const TpParams& p, int owner, uint32_t epoch, int src) {
  return p.peer[owner] + ((int64_t)(epoch & 1u) * p.world + src) * p.max_elems;
}
// where this rank's contribution for destination rank q goes (element 0 of the vector)
__device__ __forceinline__ uint64_t* push_ptr(const TpParams& p, int q, uint32_t epoch) {
  return p.loopback ? slot_ptr(p, p.rank, epoch, q) : slot_ptr(p, q, epoch, p.rank);
}
__device__ __forceinline__ uint32_t push_bits(const TpParams& p, int q, uint32_t bits) {
  return (p.loopback && q != p.rank) ? 0u : bits;
}
__device__ __forceinline__ void put(uint64_t* g, uint32_t epoch, uint32_t bits) {
  __hip_atomic_store(g, ((uint64_t)epoch << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// poll one granule of the LOCAL buffer until its tag is `epoch`; false on timeout
__device__ __forceinline__ bool take(uint64_t* g, uint32_t epoch, uint32_t& bits, uint32_t* status) {
  uint64_t v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if ((uint32_t)(v >> 32) != epoch) {
    const uint64_t t0 = wall_clock64();
    do {
      __builtin_amdgcn_s_sleep(1);
      v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((uint32_t)(v >> 32) == epoch) break;
      if (wall_clock64() - t0 > spin_limit(status)) { atomicOr(status, 1u); bits = 0; return false; }
    } while (true);
  }
  bits = (uint32_t)v;
  return true;
}
// the last workgroup to finish publishes the new call count (every workgroup has read ctr[0] before it arrives)
__device__ __forceinline__ void finish_call(const TpParams& p, uint32_t epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t old = atomicAdd(&p.ctr[1], 1u);
    if (old == gridDim.x - 1) { p.ctr[1] = 0; __hip_atomic_store(&p.ctr[0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
}

// host (tp.hip): the kernel parameters of the NEXT collective on this communicator (idx filled in; see TpParams::idx).  bumps =
// the kernel ends with finish_call (stand-alone collectives); false = it leaves the counter to a later one (exchange-carrying GEMVs)
TpParams tp_issue(ChattsTpComm* c, bool bumps);
int64_t tp_capacity(const ChattsTpComm* c);

}  // namespace chatts
