"""chatts_allreduce_bulk on ONE rank of TP = W with a loop-back exchange: time per [T, H] sum over (workgroups, threads per workgroup).
    python tools/tp_bulk_sweep.py [W] [T]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402
from chatts_amd.tp import P2PExchange  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 798
H = 5120
n = T * H
ex = P2PExchange.create_loopback(0, W, H * 16, n)
delta = torch.randn(n, device="cuda")
x = torch.zeros(n, device="cuda")
st = torch.cuda.current_stream()
print(f"# one rank of TP = {W}, loop-back, [{T}, {H}] float32 sum; us per chatts_allreduce_bulk (median of 7 x 20 calls)")
FENCE = int(sys.argv[3]) if len(sys.argv) > 3 else 0
_lib.set_option("TP_BULK_FENCE", FENCE)
print(f"# TP_BULK_FENCE={FENCE} ({'__threadfence_system() before the flags (round 4)' if FENCE else 's_waitcnt vmcnt(0) before the flags'})")
print("# blocks \\ threads " + "".join(f"{t:>9d}" for t in (256, 512, 1024)))
for blocks in (32, 64, 96, 128, 192, 256):
    row = []
    for threads in (256, 512, 1024):
        _lib.set_option("TP_BULK_BLOCKS", blocks)
        _lib.set_option("TP_BULK_THREADS", threads)
        for _ in range(5):
            ex.all_reduce_bulk(delta, x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(20):
                ex.all_reduce_bulk(delta, x)
            e1.record(st)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        row.append(float(np.median(ts)))
    print(f"{blocks:>17d} " + "".join(f"{v:9.1f}" for v in row))
assert ex.status() == 0
