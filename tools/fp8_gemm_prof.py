#!/usr/bin/env python
"""The fp8 speed-mode GEMM alone (gemm_fp8.hip) at the two shapes that matter - TS-encoder layer at 8192 patches, gate_up of a 1024-row
prefill chunk - for rocprofv3 passes:  rocprofv3 --kernel-trace [--pmc ...] -- python tools/fp8_gemm_prof.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatts_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    reps = int(os.environ.get("REPS", "10"))
    for m, n, k, epi in [(8192, 5120, 5120, _lib.EPI_GELU), (1024, 27648, 5120, _lib.EPI_SWIGLU), (1024, 5120, 13824, _lib.EPI_RESID)]:
        a8 = torch.randint(0, 120, (m, k), dtype=torch.uint8, device="cuda")
        w8 = torch.randint(0, 120, (n, k), dtype=torch.uint8, device="cuda")
        sa, sw = torch.rand(m, device="cuda") + 0.5, torch.rand(n, device="cuda") * 0.01
        nc = n // 2 if epi == _lib.EPI_SWIGLU else n
        c = torch.zeros((m, nc), device="cuda")
        fa = _lib.LinearFp8Args(a8=a8.data_ptr(), a_scale=sa.data_ptr(), w8=w8.data_ptr(), w_scale=sw.data_ptr(), bias=None,
                                resid=c.data_ptr() if epi == _lib.EPI_RESID else None, c=c.data_ptr(), m=m, n=n, k=k, lda8=k, ldw8=k, ldc=nc, epilogue=epi)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.check(lib.chatts_linear_fp8(C.byref(fa), st))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            _lib.check(lib.chatts_linear_fp8(C.byref(fa), st))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"M={m} N={n} K={k} epi={epi}: {us:.1f} us = {2.0 * m * n * k / us / 1e6:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
