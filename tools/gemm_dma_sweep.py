"""GPU micro-benchmark + bit-equality check of the LDS-DMA GEMM (pre-split bf16 hi/lo planes of A) against the
register-staged bf16x2 GEMM on the ChatTS-14B prefill shapes.
    python tools/gemm_dma_sweep.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 798
SHAPES = {"qkv": (7168, 5120, _lib.EPI_NONE), "o": (5120, 5120, _lib.EPI_RESID), "gate_up": (27648, 5120, _lib.EPI_SWIGLU),
          "down": (5120, 13824, _lib.EPI_RESID)}
st = torch.cuda.current_stream()


def setenv(env):
    for kk in list(os.environ):
        if kk.startswith("CHATTS_GEMM_"):
            del os.environ[kk]
    os.environ.update({a: str(b) for a, b in env.items()})


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


for name, (n, k, epi) in SHAPES.items():
    print(f"== {name} M={M} N={n} K={k}")
    ws = [(torch.randn((n, k), device=DEV) * 0.02).to(torch.bfloat16) for _ in range(4)]
    a = torch.randn((M, k), device=DEV)
    hi = torch.empty((M, k), dtype=torch.bfloat16, device=DEV)
    lo = torch.empty((M, k), dtype=torch.bfloat16, device=DEV)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    resid = torch.randn((M, ncols), device=DEV)
    out = torch.zeros((M, ncols), device=DEV)
    wsb = max(int(lib.chatts_linear_workspace(M, n, k)), 4 * M * n * 4)
    wsp = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)

    def split():
        _lib.check(lib.chatts_split_bf16x2(a.data_ptr(), M, k, k, hi.data_ptr(), lo.data_ptr(), k, st.cuda_stream))

    def run(planes, w):
        la = _lib.LinearArgs(a=a.data_ptr(), w=w.data_ptr(), bias=None, resid=resid.data_ptr() if epi == _lib.EPI_RESID else None,
                             c=out.data_ptr(), norm_w=None, norm_eps=0.0, m=M, n=n, k=k, lda=k, ldw=k, ldc=ncols,
                             epilogue=epi, workspace=wsp.data_ptr(), workspace_bytes=wsb)
        if planes:
            la.a_hi, la.a_lo, la.ld_planes = hi.data_ptr(), lo.data_ptr(), k
        _lib.check(lib.chatts_linear(la, st.cuda_stream))

    t_split = timed(split)
    # the split itself: hi + lo reproduces a to 16 mantissa bits, hi is RNE bf16
    assert torch.equal(hi, a.to(torch.bfloat16)), "hi plane != bf16(a)"
    assert torch.equal(lo, (a - hi.float()).to(torch.bfloat16)), "lo plane != bf16(a - hi)"
    setenv({})
    run(False, ws[0])
    ref = out.clone()
    t_ref = timed(lambda: [run(False, w) for w in ws]) / len(ws)
    print(f"   {t_ref:8.1f} us  {2.0 * M * n * k / t_ref / 1e6:6.0f} TF  register-staged (split {t_split:.1f} us)")
    envs = [{"CHATTS_GEMM_DMA32": 0}, {"CHATTS_GEMM_DMA32": 1}, {"CHATTS_GEMM_DMA32": 1, "CHATTS_GEMM_SK": 1},
            {"CHATTS_GEMM_DMA32": 1, "CHATTS_GEMM_SK": 2}, {"CHATTS_GEMM_DMA32": 1, "CHATTS_GEMM_SK": 3},
            {"CHATTS_GEMM_DMA32": 0, "CHATTS_GEMM_SK": 1}, {"CHATTS_GEMM_DMA32": 0, "CHATTS_GEMM_SK": 2}]
    for env in envs:
        setenv(env)
        out.zero_()
        run(True, ws[0])
        torch.cuda.synchronize()
        sk_forced = "CHATTS_GEMM_SK" in env
        same = torch.equal(out, ref)
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        t = timed(lambda: [run(True, w) for w in ws]) / len(ws)
        tag = " ".join(f"{kk.replace('CHATTS_GEMM_', '').replace('DMA_', '')}={v}" for kk, v in env.items())
        print(f"   {t:8.1f} us  {2.0 * M * n * k / t / 1e6:6.0f} TF  dma {tag}  bit-equal={same} max-rel-diff={err:.2e}")
