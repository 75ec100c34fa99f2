"""GPU probe: what could overlapping consecutive decode GEMVs buy at most?  Two projections of a layer (different weights, no data
dependency here) are run (a) back to back on one stream, (b) concurrently on two streams, each pair over rotating weight copies.
If (b) is not clearly shorter than (a) there is nothing to win by letting kernel N+1 start under kernel N's tail (the idea of a
two-branch graph with completion counters); if it is, the difference bounds the gain before any synchronisation cost.
    python tools/overlap_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
SHAPES = {"qkv": (7168, 5120, _lib.EPI_NONE), "o": (5120, 5120, _lib.EPI_RESID), "gate_up": (27648, 5120, _lib.EPI_SWIGLU),
          "down": (5120, 13824, _lib.EPI_RESID)}
NW = 6


def make(name):
    n, k, epi = SHAPES[name]
    ws = [(torch.randn((n, k), device=DEV) * 0.02).to(torch.bfloat16) for _ in range(NW)]
    x = torch.randn((1, k), device=DEV)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    out = torch.zeros((1, ncols), device=DEV)
    nw = torch.ones(k, device=DEV)

    def launch(i, stream):
        la = _lib.LinearArgs(a=x.data_ptr(), w=ws[i % NW].data_ptr(), bias=None, resid=out.data_ptr() if epi == _lib.EPI_RESID else None,
                             c=out.data_ptr(), norm_w=nw.data_ptr() if name in ("qkv", "gate_up") else None, norm_eps=1e-6, m=1, n=n, k=k,
                             lda=k, ldw=k, ldc=ncols, epilogue=epi, workspace=None, workspace_bytes=0)
        _lib.check(lib.chatts_linear(la, stream.cuda_stream))
    return launch, n * k * 2


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        fn()
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for a, b in (("o", "gate_up"), ("gate_up", "down"), ("down", "qkv"), ("qkv", "o")):
    la, ba = make(a)
    lb, bb = make(b)

    def seq():
        for i in range(NW):
            la(i, s0); lb(i, s0)

    def conc():
        ev = torch.cuda.Event()
        ev.record(s0)
        s1.wait_event(ev); s2.wait_event(ev)
        for i in range(NW):
            la(i, s1); lb(i, s2)
        e1, e2 = torch.cuda.Event(), torch.cuda.Event()
        e1.record(s1); e2.record(s2)
        s0.wait_event(e1); s0.wait_event(e2)

    def only(l):
        def f():
            for i in range(NW):
                l(i, s0)
        return f
    seq(); conc()
    ta, tb, ts, tc = timed(only(la)) / NW, timed(only(lb)) / NW, timed(seq) / NW, timed(conc) / NW
    print(f"{a:8s}+{b:8s}  alone {ta:6.1f} + {tb:6.1f} = {ta + tb:6.1f} us   back to back {ts:6.1f} us   two streams {tc:6.1f} us   "
          f"({(ba + bb) / tc / 1e6:5.2f} TB/s)   gain {ts - tc:5.1f} us per pair")
