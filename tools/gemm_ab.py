"""Interleaved A/B of the prefill projections at the ChatTS-14B chunk shapes between option settings of the prefill kernel (round 5
ran it between the round-4 LDS-DMA kernel and gemm_ring_kernel: profiles/r5_gemm_ab_ring_first.txt), R rounds, each round runs every arm once per shape (median and min over the rounds; the epilogue launch of a
split-K projection is inside the timed region, as in the decoder).
    python tools/gemm_ab.py [M] [rounds]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
st = torch.cuda.current_stream()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 798
R = int(sys.argv[2]) if len(sys.argv) > 2 else 7
SHAPES = {"qkv": (7168, 5120, _lib.EPI_NONE), "o": (5120, 5120, _lib.EPI_RESID), "gate_up": (27648, 5120, _lib.EPI_SWIGLU),
          "down": (5120, 13824, _lib.EPI_RESID)}
ARMS = {"auto": {}}
for extra in sys.argv[3:]:          # e.g. ring_T6=T:6  or  ring_o=T:6,SK:2
    name, spec = extra.split("=")
    ARMS[name] = {**{a: int(b) for a, b in (kv.split(":") for kv in spec.split(","))}}


def setenv(env):
    for kk in list(os.environ):
        if kk.startswith("CHATTS_GEMM_"):
            del os.environ[kk]
    os.environ.update({"CHATTS_GEMM_" + a: str(b) for a, b in env.items()})
    _lib.sync_env()


cases = {}
for name, (n, k, epi) in SHAPES.items():
    ws = [(torch.randn((n, k), device=DEV) * 0.02).to(torch.bfloat16) for _ in range(2)]
    a = torch.randn((M, k), device=DEV)
    hi = a.to(torch.bfloat16)
    lo = (a - hi.float()).to(torch.bfloat16)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    cases[name] = dict(n=n, k=k, epi=epi, ws=ws, hi=hi, lo=lo, ncols=ncols, resid=torch.randn((M, ncols), device=DEV),
                       out=torch.zeros((M, ncols), device=DEV), phi=torch.empty((M, ncols), dtype=torch.bfloat16, device=DEV),
                       plo=torch.empty((M, ncols), dtype=torch.bfloat16, device=DEV), nw=torch.rand((n,), device=DEV) + 0.5,
                       nhi=torch.empty((M, n), dtype=torch.bfloat16, device=DEV), nlo=torch.empty((M, n), dtype=torch.bfloat16, device=DEV),
                       bias=torch.randn((n,), device=DEV) if name == "qkv" else None,
                       wsp=torch.empty(16 * M * n * 4, dtype=torch.uint8, device=DEV))


def run(c, w):
    la = _lib.LinearArgs(a=None, w=w.data_ptr(), bias=_lib.ptr(c["bias"]), resid=c["resid"].data_ptr() if c["epi"] == _lib.EPI_RESID else None,
                         c=c["out"].data_ptr(), norm_w=None, norm_eps=0.0, m=M, n=c["n"], k=c["k"], lda=c["k"], ldw=c["k"], ldc=c["ncols"],
                         epilogue=c["epi"], workspace=c["wsp"].data_ptr(), workspace_bytes=c["wsp"].numel())
    la.a_hi, la.a_lo, la.ld_planes = c["hi"].data_ptr(), c["lo"].data_ptr(), c["k"]
    if c["epi"] == _lib.EPI_SWIGLU:
        la.c = None
        la.c_hi, la.c_lo, la.ld_cplanes = c["phi"].data_ptr(), c["plo"].data_ptr(), c["ncols"]
    if c["epi"] == _lib.EPI_RESID:          # as in the decoder: the epilogue also writes the next projection's normed planes
        la.post_norm_w, la.post_norm_eps = c["nw"].data_ptr(), 1e-6
        la.post_hi, la.post_lo, la.ld_post = c["nhi"].data_ptr(), c["nlo"].data_ptr(), c["n"]
    _lib.check(lib.chatts_linear(la, st.cuda_stream))


res = {(s, a): [] for s in SHAPES for a in ARMS}
for rnd in range(R + 1):
    for sname, c in cases.items():
        for aname, env in ARMS.items():
            setenv(env)
            run(c, c["ws"][0])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for w in c["ws"]:
                run(c, w)
            e1.record(st)
            torch.cuda.synchronize()
            if rnd > 0:
                res[(sname, aname)].append(e0.elapsed_time(e1) * 1e3 / len(c["ws"]))
tot = {a: 0.0 for a in ARMS}
for sname in SHAPES:
    line = f"{sname:8s}"
    for aname in ARMS:
        v = res[(sname, aname)]
        tot[aname] += float(np.median(v))
        line += f"  {aname}: median {np.median(v):7.1f} min {min(v):7.1f} us"
    print(line)
print("layer sum (medians): " + "  ".join(f"{a}: {t:7.1f} us" for a, t in tot.items()))
