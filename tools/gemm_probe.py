"""Timeline of the prefill GEMM's workgroups (diagnostic build, tools/build_variant.py probe -DCHATTS_GEMM_PROBE):
where a tile's time goes - dispatch gap, prologue, K loop, epilogue - per shape of the ChatTS-14B prefill chunk.
    CHATTS_AMD_LIB=chatts_amd/lib/variants/libchatts_amd_probe.so python tools/gemm_probe.py [M]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
lib.chatts_debug_gemm_probe.restype = C.c_int
lib.chatts_debug_gemm_probe.argtypes = [C.c_void_p, C.c_size_t]
DEV = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 798
SHAPES = {"qkv": (7168, 5120, _lib.EPI_NONE), "o": (5120, 5120, _lib.EPI_RESID), "gate_up": (27648, 5120, _lib.EPI_SWIGLU),
          "down": (5120, 13824, _lib.EPI_RESID)}
st = torch.cuda.current_stream()
NREC = 1 << 16


def read_probe():
    buf = np.zeros(NREC * 16, dtype=np.uint64)
    _lib.check(lib.chatts_debug_gemm_probe(buf.ctypes.data, buf.nbytes))
    r = buf.reshape(NREC, 16)
    return r[r[:, 0] != 0]


def pct(x, q):
    return float(np.percentile(x, q)) if len(x) else float("nan")


for name, (n, k, epi) in SHAPES.items():
    for force_sk in ([0, 1] if name in ("o", "down") else [0]):
        os.environ.pop("CHATTS_GEMM_SK", None)
        if force_sk:
            os.environ["CHATTS_GEMM_SK"] = str(force_sk)
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
            if epi == _lib.EPI_SWIGLU:      # as in the decoder: the SwiGLU output goes out as planes
                la.c = None
                la.c_hi, la.c_lo, la.ld_cplanes = phi.data_ptr(), plo.data_ptr(), ncols
            _lib.check(lib.chatts_linear(la, st.cuda_stream))

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        read_probe()                       # clear
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        run()
        e1.record(st)
        torch.cuda.synchronize()
        t_evt = e0.elapsed_time(e1) * 1e3
        r = read_probe()
        T = 10.0 / 1e3      # realtime ticks (100 MHz) -> us
        c0, c_clk0, c_pub, c_loop, c_done, c_clk1 = (r[:, i].astype(np.int64) for i in range(6))
        cu = (r[:, 6] >> np.uint64(32)).astype(np.int64)
        nk = (r[:, 7] >> np.uint64(32)).astype(np.int64)
        rows = (r[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
        l0, l_iss, l_land = r[:, 8].astype(np.int64), r[:, 10].astype(np.int64), r[:, 11].astype(np.int64)
        t0 = c0.min()
        full = rows >= 128
        span = (c_done.max() - t0) * T
        print(f"== {name} M={M} N={n} K={k} forced_sk={force_sk}: {len(r)} workgroups on {len(set(cu.tolist()))} CUs, "
              f"event time {t_evt:.1f} us, first entry -> last store {span:.1f} us, last entry at {(c0.max() - t0) * T:.1f} us")
        for tag, sel in (("full tiles", full), ("ragged tiles", ~full)):
            if not sel.any():
                continue
            pro, loop, epi_t = (c_pub - c0)[sel] * T, (c_loop - c_pub)[sel] * T, (c_done - c_loop)[sel] * T
            clk = (c_clk1 - c_clk0)[sel] / np.maximum((c_done - c0)[sel] * 10.0, 1)      # cycles per ns = GHz
            print(f"   {tag:12s} n={int(sel.sum()):4d} nk={int(np.median(nk[sel]))}  prologue {np.median(pro):6.2f} (p90 {pct(pro, 90):6.2f})  "
                  f"K loop {np.median(loop):7.2f} (p10 {pct(loop, 10):7.2f} p90 {pct(loop, 90):7.2f}) = {np.median(loop) / max(np.median(nk[sel]), 1):.3f} us/step  "
                  f"epilogue {np.median(epi_t):6.2f} (p90 {pct(epi_t, 90):6.2f})  clock {np.median(clk):.2f} GHz")
            print(f"   {'':12s} loader: issue of 2 stages {np.median((l_iss - l0)[sel]) * T:5.2f} us, stage 0 landed at +{np.median((l_land - l0)[sel]) * T:5.2f} us")
        # per-CU timelines: gaps between a workgroup's last store and the next workgroup's entry on the same CU, idle time per CU
        gaps, busy = [], []
        for c in set(cu.tolist()):
            s = np.argsort(c0[cu == c])
            a0, a1 = c0[cu == c][s], c_done[cu == c][s]
            gaps += list((a0[1:] - a1[:-1]) * T)
            busy.append(((a1 - a0).sum()) * T)
        if gaps:
            print(f"   same-CU turnover (last store -> next entry): median {np.median(gaps):.2f} us, p90 {pct(gaps, 90):.2f}; "
                  f"busy per CU median {np.median(busy):.1f} us of {span:.1f} (min {min(busy):.1f}, max {max(busy):.1f})")
        else:
            print(f"   one workgroup per CU; busy per CU median {np.median(busy):.1f} us of {span:.1f}")
        print(f"   first entry after launch-side start: event {t_evt:.1f} vs span {span:.1f} -> {t_evt - span:.1f} us outside the workgroups "
              f"(launch, epilogue kernels)")
