"""GPU probe: does the 256 MB Infinity Cache (MALL) serve a GEMV's weights when they were read just before?
Times the o_proj-shaped GEMV (52 MB) (a) over rotating buffers (HBM), (b) over ONE buffer back to back (MALL/L2 warm),
(c) after a plain-load 'prefetch' kernel (torch sum) touched the buffer, with > 1 GB of other traffic before it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
st = torch.cuda.current_stream()


def gemv(w, x, out, resid=True):
    n, k = w.shape
    la = _lib.LinearArgs(a=x.data_ptr(), w=w.data_ptr(), bias=None, resid=out.data_ptr() if resid else None,
                         c=out.data_ptr(), norm_w=None, norm_eps=1e-6, m=1, n=n, k=k, lda=k, ldw=k, ldc=n,
                         epilogue=_lib.EPI_RESID if resid else _lib.EPI_NONE, workspace=None, workspace_bytes=0)
    _lib.check(lib.chatts_linear(la, st.cuda_stream))


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(st)
        n = fn()
        e1.record(st)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for name, (n, k) in {"o_proj 52MB": (5120, 5120), "down 142MB": (5120, 13824), "qkv 73MB": (7168, 5120)}.items():
    ws = [torch.randint(-3000, 3000, (n, k), dtype=torch.int16, device=DEV).view(torch.bfloat16) for _ in range(24)]
    x = torch.randn(k, device=DEV)
    out = torch.zeros(n, device=DEV)
    big = torch.empty(300 * 1024 * 1024, dtype=torch.float32, device=DEV)    # 1.2 GB flusher

    def rotating():
        for w in ws:
            gemv(w, x, out)
        return len(ws)

    def same():
        for _ in range(24):
            gemv(ws[0], x, out)
        return 24

    t_rot, t_same = timed(rotating), timed(same)
    # (c): flush, prefetch by a plain read, then time ONE gemv
    ts = []
    for _ in range(5):
        big.add_(1.0)
        pre = ws[3].view(torch.int16).sum()          # plain loads over the 52 MB
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        gemv(ws[3], x, out)
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    tc = []
    for _ in range(5):
        big.add_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        gemv(ws[5], x, out)
        e1.record(st)
        torch.cuda.synchronize()
        tc.append(e0.elapsed_time(e1) * 1e3)
    print(f"{name:12s} rotating(HBM) {t_rot:6.2f} us | same buffer back-to-back {t_same:6.2f} us | single cold {min(tc):6.2f} us | "
          f"single after plain-read prefetch {min(ts):6.2f} us")
