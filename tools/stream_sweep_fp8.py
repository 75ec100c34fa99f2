"""GPU micro-benchmark of the batched-decode projections of BASELINE config 5 (M = 16 sequences, fp8 weights, ChatTS-14B shapes):
chatts_linear on bf16 hi / lo planes (gemm_stream_kernel<., true, 1> + its split-K epilogue launch) against the split count and the
ring depth.  Each line is the time of ONE projection including its epilogue launch, 8 distinct weight copies in rotation (no L2 /
MALL reuse between calls).
    python tools/stream_sweep_fp8.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402
from chatts_amd.modeling import quantize_fp8_rows  # noqa: E402

lib = _lib.load()
DEV = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SHAPES = {"qkv": (7168, 5120, _lib.EPI_NONE), "o": (5120, 5120, _lib.EPI_RESID), "gate_up": (27648, 5120, _lib.EPI_SWIGLU),
          "down": (5120, 13824, _lib.EPI_RESID)}
st = torch.cuda.current_stream()


def setenv(env):
    for kk in list(os.environ):
        if kk.startswith("CHATTS_GEMM_"):
            del os.environ[kk]
    os.environ.update({a: str(b) for a, b in env.items()})
    _lib.sync_env()


for name, (n, k, epi) in SHAPES.items():
    nw = 8
    qs = [quantize_fp8_rows((torch.randn((n, k), device=DEV) * 0.02).to(torch.bfloat16)) for _ in range(nw)]
    wbytes = n * k
    print(f"== {name} N={n} K={k} fp8 weights ({wbytes / 1e6:.0f} MB), M={M}")
    a = torch.randn((M, k), device=DEV)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    out = torch.zeros((M, ncols), device=DEV)
    wsb = max(int(lib.chatts_linear_workspace(M, n, k)), 32 * M * n * 4)
    wsp = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    hi = torch.empty((M, k), dtype=torch.bfloat16, device=DEV)
    lo = torch.empty((M, k), dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.chatts_split_bf16x2(a.data_ptr(), M, k, k, hi.data_ptr(), lo.data_ptr(), k, st.cuda_stream))
    sks = (1,) if name == "gate_up" else (4, 5, 6, 8, 10, 12)
    envs = [{}] + [{"CHATTS_GEMM_STREAM_WAVES": w_, "CHATTS_GEMM_STREAM_STAGES": st_, "CHATTS_GEMM_SK": s}
                   for w_ in (4, 8) for st_ in (3, 4) for s in sks]
    for env in envs:
        setenv(env)

        def run():
            for q, sc, deq in qs:
                la = _lib.LinearArgs(a=None, w=deq.data_ptr(), bias=None, resid=out.data_ptr() if epi == _lib.EPI_RESID else None,
                                     c=out.data_ptr(), norm_w=None, norm_eps=0.0, m=M, n=n, k=k, lda=k, ldw=k, ldc=ncols,
                                     epilogue=epi, workspace=wsp.data_ptr(), workspace_bytes=wsb)
                la.a_hi, la.a_lo, la.ld_planes = hi.data_ptr(), lo.data_ptr(), k
                la.w8, la.w8_scale, la.ldw8 = q.data_ptr(), sc.data_ptr(), k
                _lib.check(lib.chatts_linear(la, st.cuda_stream))
        run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            run()
            e1.record(st)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / nw)
        tag = ",".join(f"{kk.replace('CHATTS_GEMM_', '').replace('STREAM_', '')}={v}" for kk, v in env.items()) or "default"
        print(f"   {tag:22s} {best:6.1f} us  {wbytes / best / 1e6:5.2f} TB/s")
