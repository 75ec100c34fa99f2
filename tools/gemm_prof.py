"""Launch one prefill GEMM shape a few times (for rocprofv3 --pmc passes over a single kernel).
    python tools/gemm_prof.py <shape> [M] [planes]      shape in qkv | o | gate_up | down; env CHATTS_GEMM_* selects geometry"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
SHAPES = {"qkv": (7168, 5120, _lib.EPI_NONE), "o": (5120, 5120, _lib.EPI_RESID), "gate_up": (27648, 5120, _lib.EPI_SWIGLU),
          "down": (5120, 13824, _lib.EPI_RESID)}
name = sys.argv[1]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 798
planes = len(sys.argv) > 3 and sys.argv[3] in ("planes", "tiled")
tiled = len(sys.argv) > 3 and sys.argv[3] == "tiled"      # W in the prefill kernel's tiled layout (chatts_tile_bf16)
n, k, epi = SHAPES[name]
st = torch.cuda.current_stream()
ws = [(torch.randn((n, k), device=DEV) * 0.02).to(torch.bfloat16) for _ in range(4)]
a = torch.randn((M, k), device=DEV)
hi = torch.empty((M, k), dtype=torch.bfloat16, device=DEV)
lo = torch.empty((M, k), dtype=torch.bfloat16, device=DEV)
ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
resid = torch.randn((M, ncols), device=DEV)
out = torch.zeros((M, ncols), device=DEV)
wsb = max(int(lib.chatts_linear_workspace(M, n, k)), 4 * M * n * 4)
wsp = torch.empty(wsb, dtype=torch.uint8, device=DEV)
_lib.check(lib.chatts_split_bf16x2(a.data_ptr(), M, k, k, hi.data_ptr(), lo.data_ptr(), k, st.cuda_stream))
wts = []
if tiled:
    for w in ws:
        t = torch.empty(int(lib.chatts_tile_bf16_elems(n, k)), dtype=torch.bfloat16, device=DEV)
        _lib.check(lib.chatts_tile_bf16(w.data_ptr(), n, k, k, t.data_ptr(), st.cuda_stream))
        wts.append(t)
for it in range(3):
    for wi, w in enumerate(ws):
        la = _lib.LinearArgs(a=a.data_ptr(), w=w.data_ptr(), bias=None, resid=resid.data_ptr() if epi == _lib.EPI_RESID else None,
                             c=out.data_ptr(), norm_w=None, norm_eps=0.0, m=M, n=n, k=k, lda=k, ldw=k, ldc=ncols,
                             epilogue=epi, workspace=wsp.data_ptr(), workspace_bytes=wsb)
        if planes:
            la.a_hi, la.a_lo, la.ld_planes = hi.data_ptr(), lo.data_ptr(), k
        if tiled:
            la.w_tiled = wts[wi].data_ptr()
        _lib.check(lib.chatts_linear(la, st.cuda_stream))
torch.cuda.synchronize()
print("done", name, M, "planes" if planes else "f32")
