"""Prefill GEMM: time against K at a fixed tile count -> per-K-step cost s and per-tile fixed cost F of the prefill kernel
(T = rounds * (F + nk * s)).  256 tiles = one full round on 256 CUs; the gate_up shape = 2.95 rounds.
    python tools/gemm_ksweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
st = torch.cuda.current_stream()
os.environ["CHATTS_GEMM_SK"] = "1"
_lib.sync_env()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def sweep(tag, M, n, epi, ks, planes_out):
    pts = []
    for k in ks:
        ws = [(torch.randn((n, k), device=DEV) * 0.02).to(torch.bfloat16) for _ in range(3)]
        a = torch.randn((M, k), device=DEV)
        hi = a.to(torch.bfloat16)
        lo = (a - hi.float()).to(torch.bfloat16)
        ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
        resid = torch.randn((M, ncols), device=DEV)
        out = torch.zeros((M, ncols), device=DEV)
        phi = torch.empty((M, ncols), dtype=torch.bfloat16, device=DEV)
        plo = torch.empty((M, ncols), dtype=torch.bfloat16, device=DEV)

        def run(w):
            la = _lib.LinearArgs(a=None, w=w.data_ptr(), bias=None, resid=resid.data_ptr() if epi == _lib.EPI_RESID else None,
                                 c=out.data_ptr(), norm_w=None, norm_eps=0.0, m=M, n=n, k=k, lda=k, ldw=k, ldc=ncols,
                                 epilogue=epi, workspace=None, workspace_bytes=0)
            la.a_hi, la.a_lo, la.ld_planes = hi.data_ptr(), lo.data_ptr(), k
            if planes_out:
                la.c = None
                la.c_hi, la.c_lo, la.ld_cplanes = phi.data_ptr(), plo.data_ptr(), ncols
            _lib.check(lib.chatts_linear(la, st.cuda_stream))

        t = timed(lambda: [run(w) for w in ws]) / len(ws)
        pts.append((k // 64, t))
        print(f"   {tag} M={M} N={n} K={k:6d} ({k // 64:3d} steps): {t:8.1f} us  {2.0 * M * n * k / t / 1e6:6.0f} TF useful")
    x, y = np.array([p[0] for p in pts], float), np.array([p[1] for p in pts], float)
    s, f = np.polyfit(x, y, 1)
    print(f"   -> fit: T = {f:.1f} us + {s:.3f} us per K-step (128 MFMAs per SIMD: {128 * 16 / s / 1e3:.2f} GHz-equivalent at 16 cycles each)")


KS = [640, 1280, 2560, 5120, 10240]
print("== 256 full tiles, one round (M=1024 N=8192), no epilogue work (EPI_NONE, float32 out)")
sweep("none  ", 1024, 8192, _lib.EPI_NONE, KS, False)
print("== the same with the residual epilogue")
sweep("resid ", 1024, 8192, _lib.EPI_RESID, KS, False)
print("== 140 tiles (o_proj shape M=798 N=5120), residual epilogue, no split")
sweep("o-like", 798, 5120, _lib.EPI_RESID, KS, False)
print("== gate_up shape (756 tiles), SwiGLU -> planes")
sweep("gateup", 798, 27648, _lib.EPI_SWIGLU, [1280, 2560, 5120], True)
print("== 512 full tiles, two rounds (M=1024 N=16384), EPI_NONE")
sweep("2round", 1024, 16384, _lib.EPI_NONE, [1280, 5120], False)
