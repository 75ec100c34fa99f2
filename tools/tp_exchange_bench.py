"""GPU micro-benchmark: latency of the one-shot exchange kernels (csrc/tp.hip) with both ranks in THIS process on one device
(two streams of different priority = two hardware queues).  What it measures: kernel launch + push + poll + rank-ordered sum
through uncached device memory - NOT xGMI hops (needs a multi-GPU node); what it bounds: the per-collective software cost.
usage: python tools/tp_exchange_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402
from chatts_amd.tp import P2PExchange  # noqa: E402


def main():
    lib = _lib.load()
    world = 2
    exs = P2PExchange.create_local_group(world, 1 << 20)
    streams = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)]
    for n in (5120, 4096, 16 * 5120, 128, 2):
        x = [torch.randn(n, device="cuda") for _ in range(world)]
        d = [torch.randn(n, device="cuda") for _ in range(world)]
        reps = 300
        for trial in range(2):
            evs = []
            torch.cuda.synchronize()
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        _lib.check(lib.chatts_allreduce(exs[r].handle, d[r].data_ptr(), x[r].data_ptr(), x[r].data_ptr(), n,
                                                        streams[r].cuda_stream))
                    e1.record()
                    evs.append((e0, e1))
            torch.cuda.synchronize()
        us = [e0.elapsed_time(e1) * 1e3 / reps for e0, e1 in evs]
        print(f"allreduce n={n:6d} f32 ({n * 4 / 1024:7.1f} KB): {max(us):6.2f} us per call (rank streams {us[0]:.2f} / {us[1]:.2f}), status {exs[0].status()}")
    for e in exs:
        e.close()


if __name__ == "__main__":
    main()
