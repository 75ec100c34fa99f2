"""GPU micro-benchmark: one TS-MLP layer (P x 5120 x K, bias + GELU, bf16 hi/lo planes in and out) over split-K choices and kernels.
usage: python tools/ts_gemm_sweep.py [P ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
st = torch.cuda.current_stream()


def run(P, K, N, env, planes=True, nbuf=12, reps=5):
    for k in list(os.environ):
        if k.startswith("CHATTS_GEMM_"):
            del os.environ[k]
    os.environ.update({k: str(v) for k, v in env.items()})
    _lib.sync_env()
    ws = [torch.randint(-3000, 3000, (N, K), dtype=torch.int16, device=DEV).view(torch.bfloat16) for _ in range(nbuf)]
    a = torch.randn((P, K), device=DEV)
    hi = a.to(torch.bfloat16)
    lo = (a - hi.float()).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    c = torch.empty((P, N), device=DEV)
    chi = torch.empty((P, N), dtype=torch.bfloat16, device=DEV)
    clo = torch.empty((P, N), dtype=torch.bfloat16, device=DEV)
    wsb = max(int(lib.chatts_linear_workspace(P, N, K)), 16 * P * N * 4)
    wsp = torch.empty(wsb, dtype=torch.uint8, device=DEV)

    def go():
        for w in ws:
            la = _lib.LinearArgs(a=None if planes else a.data_ptr(), w=w.data_ptr(), bias=bias.data_ptr(), resid=None,
                                 c=None if planes else c.data_ptr(), norm_w=None, norm_eps=0.0, m=P, n=N, k=K, lda=K, ldw=K, ldc=N,
                                 epilogue=_lib.EPI_GELU, workspace=wsp.data_ptr(), workspace_bytes=wsb,
                                 a_hi=hi.data_ptr() if planes else None, a_lo=lo.data_ptr() if planes else None, ld_planes=K,
                                 c_hi=chi.data_ptr() if planes else None, c_lo=clo.data_ptr() if planes else None, ld_cplanes=N)
            _lib.check(lib.chatts_linear(la, st.cuda_stream))
    go()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        go()
        e1.record(st)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / len(ws))
    return best


def main():  # noqa
    Ps = [int(v) for v in sys.argv[1:]] or [16, 128, 1100]
    for P in Ps:
        for K in (5120, 320):
            print(f"== P={P} K={K} N=5120 (weights {5120 * K * 2 / 1e6:.1f} MB)")
            res = [("f32 A, register-staged, auto", run(P, K, 5120, {}, planes=False))]
            res.append(("planes, auto", run(P, K, 5120, {})))
            for mbw in (4, 8):
                res.append((f"planes, stream up to 128 rows, {mbw} waves", run(P, K, 5120, {"CHATTS_GEMM_STREAM_MB": 128, "CHATTS_GEMM_STREAM_MB_WAVES": mbw})))
                for sk in (4, 5, 6, 8):
                    if K // sk >= 64:
                        res.append((f"planes, stream up to 128 rows, {mbw} waves, SK={sk}",
                                    run(P, K, 5120, {"CHATTS_GEMM_STREAM_MB": 128, "CHATTS_GEMM_STREAM_MB_WAVES": mbw, "CHATTS_GEMM_SK": sk})))
            for sk in (1, 2, 3, 4, 6, 8, 10, 12, 16, 20):
                if K // sk < 64:
                    continue
                res.append((f"planes, SK={sk}", run(P, K, 5120, {"CHATTS_GEMM_SK": sk})))
            for name, us in res:
                print(f"   {us:8.2f} us  {5120 * K * 2 / us / 1e3:7.0f} GB/s  {name}")
            sys.stdout.flush()


if __name__ == "__main__":
    main()
