"""Who waits for whom inside gemm_ring_kernel (diagnostic build: tools/build_variant.py probe -DCHATTS_GEMM_PROBE): per shape of
the ChatTS-14B prefill chunk, the share of a workgroup's cycles its first compute wave spends parked at the half-step barriers and in
the epilogues, and the share its first loader wave spends waiting for its pieces / at barriers / issuing.
    CHATTS_AMD_LIB=chatts_amd/lib/variants/libchatts_amd_probe.so python tools/ring_probe.py [M]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
lib.chatts_debug_ring_probe.restype = C.c_int
lib.chatts_debug_ring_probe.argtypes = [C.c_void_p, C.c_size_t]
lib.chatts_debug_ring_ablate.restype = C.c_int
lib.chatts_debug_ring_ablate.argtypes = [C.c_int]
DEV = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 798
SHAPES = {"qkv": (7168, 5120, _lib.EPI_NONE), "o": (5120, 5120, _lib.EPI_RESID), "gate_up": (27648, 5120, _lib.EPI_SWIGLU),
          "down": (5120, 13824, _lib.EPI_RESID)}
VARIANTS = {"qkv": [{}], "o": [{}], "down": [{}],
            # ablations (timing only, results are garbage): 1 = no LDS-DMA, 2 = no fragment reads in the loop, 4 = no MFMAs
            "gate_up": [{}, {"ABLATE": 1}, {"ABLATE": 2}, {"ABLATE": 3}, {"ABLATE": 4}, {"ABLATE": 5}, {"ABLATE": 6}, {"ABLATE": 7}, {}]}
if len(sys.argv) > 2 and sys.argv[2] == "noablate":
    VARIANTS["gate_up"] = [{}]
st = torch.cuda.current_stream()
NREC = 4096


def read_probe():
    buf = np.zeros(NREC * 16, dtype=np.uint64)
    _lib.check(lib.chatts_debug_ring_probe(buf.ctypes.data, buf.nbytes))
    r = buf.reshape(NREC, 16)
    return r[r[:, 0] != 0]


for name, (n, k, epi) in SHAPES.items():
    for env in VARIANTS[name]:
        for kk in list(os.environ):
            if kk.startswith("CHATTS_GEMM_"):
                del os.environ[kk]
        os.environ.update({"CHATTS_GEMM_" + a: str(b) for a, b in env.items() if a != "ABLATE"})
        _lib.sync_env()
        lib.chatts_debug_ring_ablate(int(env.get("ABLATE", 0)))
        w = (torch.randn((n, k), device=DEV) * 0.02).to(torch.bfloat16)
        a = torch.randn((M, k), device=DEV)
        hi = a.to(torch.bfloat16)
        lo = (a - hi.float()).to(torch.bfloat16)
        ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
        resid = torch.randn((M, ncols), device=DEV)
        out = torch.zeros((M, ncols), device=DEV)
        phi = torch.empty((M, ncols), dtype=torch.bfloat16, device=DEV)
        plo = torch.empty((M, ncols), dtype=torch.bfloat16, device=DEV)
        wsb = max(int(lib.chatts_linear_workspace(M, n, k)), 4 * M * n * 4)
        wsp = torch.empty(wsb, dtype=torch.uint8, device=DEV)

        def run():
            la = _lib.LinearArgs(a=None, w=w.data_ptr(), bias=None, resid=resid.data_ptr() if epi == _lib.EPI_RESID else None,
                                 c=out.data_ptr(), norm_w=None, norm_eps=0.0, m=M, n=n, k=k, lda=k, ldw=k, ldc=ncols,
                                 epilogue=epi, workspace=wsp.data_ptr(), workspace_bytes=wsb)
            la.a_hi, la.a_lo, la.ld_planes = hi.data_ptr(), lo.data_ptr(), k
            if epi == _lib.EPI_SWIGLU:
                la.c = None
                la.c_hi, la.c_lo, la.ld_cplanes = phi.data_ptr(), plo.data_ptr(), ncols
            _lib.check(lib.chatts_linear(la, st.cuda_stream))

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        read_probe()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        run()
        e1.record(st)
        torch.cuda.synchronize()
        t_evt = e0.elapsed_time(e1) * 1e3
        r = read_probe()
        f = r.astype(np.float64)
        life_us = (f[:, 1] - f[:, 0]) * 0.01
        cyc = f[:, 2]
        clk = cyc / np.maximum(life_us * 1e3, 1)
        units = (r[:, 5] >> np.uint64(32)).astype(np.int64)
        halves = (r[:, 5] & np.uint64(0xffffffff)).astype(np.float64)
        span = (f[:, 1].max() - f[:, 0].min()) * 0.01
        tag = " ".join(f"{a}={b}" for a, b in env.items()) or "auto"
        print(f"== {name} [{tag}] M={M} N={n} K={k}: {len(r)} workgroups, units per workgroup {units.min()}..{units.max()}, event {t_evt:.1f} us, "
              f"first entry -> last exit {span:.1f} us, last entry at {(f[:, 0].max() - f[:, 0].min()) * 0.01:.1f} us")
        print(f"   workgroup lifetime median {np.median(life_us):.1f} us (min {life_us.min():.1f}, max {life_us.max():.1f}); clock {np.median(clk):.2f} GHz; "
              f"cycles per half-step (all-in) {np.median(cyc / halves):.0f} = {np.median(life_us / halves * 2):.3f} us per 64-deep step")
        print(f"   compute wave 0: parked at barriers {100 * np.median(f[:, 3] / cyc):.1f} % of its cycles, epilogues {100 * np.median(f[:, 4] / cyc):.1f} % "
              f"({np.median(f[:, 4] / np.maximum(units, 1) / np.maximum(clk, 0.1) / 1e3):.2f} us per unit)")
        print(f"   loader wave 0 : waiting for its pieces {100 * np.median(f[:, 8] / cyc):.1f} %, parked at barriers {100 * np.median(f[:, 9] / cyc):.1f} %, "
              f"issuing {100 * np.median(f[:, 10] / cyc):.1f} %; first half-stage landed {np.median(f[:, 12] - f[:, 11]) * 0.01:.2f} us after entry")
