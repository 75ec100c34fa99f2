"""GPU micro-benchmark of chatts_linear (M > 1, bf16x2 MFMA GEMM) on the prefill shapes of ChatTS-14B.
Env overrides CHATTS_GEMM_BM / CHATTS_GEMM_SK / CHATTS_GEMM_TARGET select tile height / split-K.
    python tools/gemm_sweep.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 798
SHAPES = {"ts_l0": (5120, 288, _lib.EPI_GELU), "ts_mid": (5120, 5120, _lib.EPI_GELU), "qkv": (7168, 5120, _lib.EPI_NONE), "o": (5120, 5120, _lib.EPI_RESID), "gate_up": (27648, 5120, _lib.EPI_SWIGLU),
          "down": (5120, 13824, _lib.EPI_RESID)}
st = torch.cuda.current_stream()


def bench(name, env):
    n, k, epi = SHAPES[name]
    for kk in list(os.environ):
        if kk.startswith("CHATTS_GEMM_"):
            del os.environ[kk]
    os.environ.update({a: str(b) for a, b in env.items()})
    _lib.sync_env()
    ws = [torch.randint(-3000, 3000, (n, k), dtype=torch.int16, device=DEV).view(torch.bfloat16) for _ in range(4)]
    a = torch.randn((M, k), device=DEV)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    out = torch.zeros((M, ncols), device=DEV)
    wsb = int(lib.chatts_linear_workspace(M, n, k))
    wsp = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)

    def run():
        for w in ws:
            la = _lib.LinearArgs(a=a.data_ptr(), w=w.data_ptr(), bias=None, resid=out.data_ptr() if epi == _lib.EPI_RESID else None,
                                 c=out.data_ptr(), norm_w=None, norm_eps=0.0, m=M, n=n, k=k, lda=k, ldw=k, ldc=ncols,
                                 epilogue=epi, workspace=wsp.data_ptr(), workspace_bytes=wsb)
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
    return best, 2.0 * M * n * k / best / 1e6     # us, TFLOP/s (useful flops, single pass)


for name in SHAPES:
    print(f"== {name} M={M} N={SHAPES[name][0]} K={SHAPES[name][1]}")
    envs = [{}, {"CHATTS_GEMM_BM": 64}, {"CHATTS_GEMM_BM": 128, "CHATTS_GEMM_SK": 1}, {"CHATTS_GEMM_BM": 64, "CHATTS_GEMM_SK": 1},
            {"CHATTS_GEMM_BM": 128, "CHATTS_GEMM_SK": 2}, {"CHATTS_GEMM_BM": 128, "CHATTS_GEMM_SK": 3}]
    if M <= 128:
        envs = [{}] + [{"CHATTS_GEMM_BM": bm, "CHATTS_GEMM_SK": sk} for bm in (32, 64, 128) for sk in (1, 4, 8, 16)]
    for env in envs:
        us, tf = bench(name, env)
        print(f"   {us:8.1f} us  {tf:6.0f} TF useful  " + " ".join(f"{k.replace('CHATTS_GEMM_', '')}={v}" for k, v in env.items()))
