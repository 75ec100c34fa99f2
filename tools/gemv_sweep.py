"""GPU micro-benchmark: sweep the GEMV launch geometry (env overrides) over the five decode projection shapes of
ChatTS-14B.  Weights rotate over > 1 GB of distinct buffers per shape so neither L2 nor the 256 MB MALL can
serve them.  Usage (on the GPU box): python tools/gemv_sweep.py [quick]"""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

DEV = "cuda"
lib = _lib.load()
SHAPES = {   # name: (N, K, epilogue, norm, bias)
    "qkv": (7168, 5120, _lib.EPI_NONE, True, True),
    "o": (5120, 5120, _lib.EPI_RESID, False, False),
    "gate_up": (27648, 5120, _lib.EPI_SWIGLU, True, False),
    "down": (5120, 13824, _lib.EPI_RESID, False, False),
    "lm_head": (152064, 5120, _lib.EPI_NONE, True, False),
}


def bench(name, env, reps=3):
    n, k, epi, norm, bias = SHAPES[name]
    nbuf = max(2, int(1.2e9 // (n * k * 2)) + 1)
    ws = [torch.randint(-3000, 3000, (n, k), dtype=torch.int16, device=DEV).view(torch.bfloat16) for _ in range(min(nbuf, 24))]
    x = torch.randn(k, device=DEV)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    out = torch.zeros(ncols, device=DEV)
    nw = torch.ones(k, device=DEV)
    b = torch.zeros(n, device=DEV)
    for kk in list(os.environ):
        if kk.startswith("CHATTS_GEMV_"):
            del os.environ[kk]
    os.environ.update({k2: str(v) for k2, v in env.items()})
    _lib.sync_env()
    st = torch.cuda.current_stream()

    def run():
        for w in ws:
            la = _lib.LinearArgs(a=x.data_ptr(), w=w.data_ptr(), bias=b.data_ptr() if bias else None,
                                 resid=out.data_ptr() if epi == _lib.EPI_RESID else None, c=out.data_ptr(),
                                 norm_w=nw.data_ptr() if norm else None, norm_eps=1e-6, m=1, n=n, k=k, lda=k, ldw=k,
                                 ldc=ncols, epilogue=epi, workspace=None, workspace_bytes=0)
            _lib.check(lib.chatts_linear(la, st.cuda_stream))
    run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        run()
        e1.record(st)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / len(ws))
    return best, n * k * 2 / best / 1e3   # us, GB/s


BALANCED = {   # (ROWS, NW, workgroups per CU): waves per CU divide the tasks per CU exactly -> every wave gets the same work
    "qkv": [(2, 7, 2), (2, 7, 1), (2, 14, 1), (4, 7, 1)],
    "o": [(2, 5, 2), (2, 10, 1), (2, 5, 1), (4, 5, 1)],
    "gate_up": [(2, 9, 3), (2, 9, 2), (2, 9, 1), (2, 6, 3), (4, 9, 3), (4, 9, 1)],
    "down": [(2, 5, 2), (2, 10, 1), (2, 5, 1), (4, 5, 1)],
    "lm_head": [(2, 9, 3), (2, 11, 3), (2, 9, 1), (2, 11, 1), (4, 11, 3)],
}


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    cus = int(lib.chatts_device_cus())
    for name in SHAPES:
        configs = [{}]                                    # the shipped auto geometry
        for rows, nw, per_cu in BALANCED[name]:
            for unr in (2, 4):
                for pad in (0, per_cu):
                    configs.append({"CHATTS_GEMV_ROWS": rows, "CHATTS_GEMV_UNR": unr, "CHATTS_GEMV_NW": nw,
                                    "CHATTS_GEMV_BLOCKS": cus * per_cu, "CHATTS_GEMV_OCC": per_cu, "CHATTS_GEMV_LDSPAD": pad})
        if not quick:
            for rows, unr, nw in itertools.product((2, 4), (2, 4), (4, 8, 16)):
                configs.append({"CHATTS_GEMV_ROWS": rows, "CHATTS_GEMV_UNR": unr, "CHATTS_GEMV_NW": nw})
        res = []
        for env in configs:
            try:
                us, gbs = bench(name, env)
            except Exception as e:
                print(name, env, "ERR", e)
                continue
            res.append((us, gbs, env))
        res.sort(key=lambda r: r[0])
        print(f"== {name} N={SHAPES[name][0]} K={SHAPES[name][1]}")
        for us, gbs, env in res[:10] + res[-2:]:
            print(f"   {us:8.2f} us {gbs:7.0f} GB/s  " + (" ".join(f"{k.replace('CHATTS_GEMV_', '')}={v}" for k, v in env.items()) or "auto"))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
