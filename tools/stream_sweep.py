"""GPU micro-benchmark of chatts_linear for 1 <= M <= 16 on the ChatTS-14B decode shapes: the weight-streaming kernel
(gemm_stream_kernel, bf16 planes in) against the register-staged tiled GEMM (float32 in) and the M = 1 GEMV.
    python tools/stream_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402
from chatts_amd.modeling import quantize_fp8_rows  # noqa: E402

lib = _lib.load()
DEV = "cuda"
FP8 = False
SHAPES = {"qkv": (7168, 5120, _lib.EPI_NONE), "o": (5120, 5120, _lib.EPI_RESID), "gate_up": (27648, 5120, _lib.EPI_SWIGLU),
          "down": (5120, 13824, _lib.EPI_RESID), "lm_head": (152064, 5120, _lib.EPI_NONE)}
st = torch.cuda.current_stream()


def setenv(env):
    for kk in list(os.environ):
        if kk.startswith("CHATTS_GEMM_"):
            del os.environ[kk]
    os.environ.update({a: str(b) for a, b in env.items()})
    _lib.sync_env()


for name, (n, k, epi) in SHAPES.items():
    nw = 2 if name == "lm_head" else 4
    ws = [(torch.randn((n, k), device=DEV) * 0.02).to(torch.bfloat16) for _ in range(nw)]
    w8s = [quantize_fp8_rows(w) for w in ws] if FP8 else None
    wbytes = n * k * (1 if FP8 else 2)
    print(f"== {name} N={n} K={k} {'fp8' if FP8 else 'bf16'} weights ({wbytes / 1e6:.0f} MB)")
    for M in (1, 2, 4, 8, 16):
        a = torch.randn((M, k), device=DEV)
        ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
        out = torch.zeros((M, ncols), device=DEV)
        wsb = max(int(lib.chatts_linear_workspace(M, n, k)), 16 * M * n * 4)
        wsp = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        hi = torch.empty((M, k), dtype=torch.bfloat16, device=DEV)
        lo = torch.empty((M, k), dtype=torch.bfloat16, device=DEV)
        if M > 1:
            _lib.check(lib.chatts_split_bf16x2(a.data_ptr(), M, k, k, hi.data_ptr(), lo.data_ptr(), k, st.cuda_stream))
        envs = [{}] if M == 1 else [{"planes": 0}, {}, {"CHATTS_GEMM_STREAM_STAGES": 3}, {"CHATTS_GEMM_STREAM_STAGES": 5},
                                    {"CHATTS_GEMM_SK": 1}, {"CHATTS_GEMM_SK": 4}, {"CHATTS_GEMM_SK": 8}, {"CHATTS_GEMM_SK": 16}]
        line = []
        for env in envs:
            planes = M > 1 and env.get("planes", 1) != 0
            setenv({kk: vv for kk, vv in env.items() if kk != "planes"})

            def run():
                for i, w in enumerate(ws):
                    la = _lib.LinearArgs(a=a.data_ptr(), w=w.data_ptr(), bias=None, resid=out.data_ptr() if epi == _lib.EPI_RESID else None,
                                         c=out.data_ptr(), norm_w=None, norm_eps=0.0, m=M, n=n, k=k, lda=k, ldw=k, ldc=ncols,
                                         epilogue=epi, workspace=wsp.data_ptr(), workspace_bytes=wsb)
                    if planes:
                        la.a_hi, la.a_lo, la.ld_planes = hi.data_ptr(), lo.data_ptr(), k
                    _lib.check(lib.chatts_linear(la, st.cuda_stream))
            run()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                run()
                e1.record(st)
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / len(ws))
            tag = ",".join(f"{kk.replace('CHATTS_GEMM_', '').replace('STREAM_', '')}={v}" for kk, v in env.items()) or "default"
            line.append(f"{tag} {best:6.1f}us {wbytes / best / 1e6:5.2f}TB/s")
        print(f"   M={M:2d}  " + " | ".join(line))
