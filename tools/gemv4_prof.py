"""GPU: run the 4-bit gate_up GEMV (27648 x 5120, groups of 128) over rotating weight sets - for rocprofv3 counter passes and
quick timing of geometry overrides.  usage: python tools/gemv4_prof.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402
from chatts_amd.modeling import pack_int4, quantize_int4_rows  # noqa: E402

lib = _lib.load()
DEV = "cuda"
st = torch.cuda.current_stream()
n, k, gs = 27648, 5120, 128
sets = []
for i in range(6):
    w = (torch.randn((n, k), device=DEV) * 0.02).to(torch.bfloat16)
    q, sc, z, deq = quantize_int4_rows(w, gs)
    w4, sz = pack_int4(q, sc, z)
    sets.append((deq, w4, sz))
    del w, q
x = torch.randn(k, device=DEV)
nw = torch.ones(k, device=DEV)
out = torch.zeros(n // 2, device=DEV)


def run(use4=True):
    for deq, w4, sz in sets:
        la = _lib.LinearArgs(a=x.data_ptr(), w=deq.data_ptr(), bias=None, resid=None, c=out.data_ptr(), norm_w=nw.data_ptr(), norm_eps=1e-6,
                             m=1, n=n, k=k, lda=k, ldw=k, ldc=n // 2, epilogue=_lib.EPI_SWIGLU, workspace=None, workspace_bytes=0,
                             w4=w4.data_ptr() if use4 else None, w4_sz=sz.data_ptr() if use4 else None, ldw4=k // 2, w4_group=gs)
        _lib.check(lib.chatts_linear(la, st.cuda_stream))


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for use4 in (True, False):
    run(use4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        run(use4)
    e1.record(st)
    torch.cuda.synchronize()
    print(("int4" if use4 else "bf16"), f"{e0.elapsed_time(e1) * 1e3 / (reps * len(sets)):.2f} us per launch")
