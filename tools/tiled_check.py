"""Round 6: the prefill kernel (gemm_ring_kernel through chatts_linear) on TILED operands (chatts_tile_bf16: an LDS-DMA piece is 1 KB of
consecutive memory) against the row-major operands: bit-identity of every output form and an interleaved A/B at the ChatTS-14B chunk shapes.
    python tools/tiled_check.py [M] [rounds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
st = torch.cuda.current_stream()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
M = int(args[0]) if len(args) > 0 else 798
R = int(args[1]) if len(args) > 1 else 7
SHAPES = {"qkv": (7168, 5120, _lib.EPI_NONE), "o": (5120, 5120, _lib.EPI_RESID), "gate_up": (27648, 5120, _lib.EPI_SWIGLU),
          "down": (5120, 13824, _lib.EPI_RESID), "ts": (5120, 5120, _lib.EPI_GELU)}
if not os.environ.get("TILED_SHAPES"):
    SHAPES.pop("ts")
if os.environ.get("TILED_SHAPES"):
    SHAPES = {k: v for k, v in SHAPES.items() if k in os.environ["TILED_SHAPES"].split(",")}
torch.manual_seed(0)


def tile(t):
    rows, k = t.shape
    out = torch.empty(lib.chatts_tile_bf16_elems(rows, k), dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.chatts_tile_bf16(t.data_ptr(), rows, k, k, out.data_ptr(), st.cuda_stream))
    return out


def tile_ref(t):
    """the layout restated with torch indexing (rows padded by repeating the last one)"""
    rows, k = t.shape
    rb = (rows + 15) // 16
    idx = torch.clamp(torch.arange(rb * 16, device=t.device), max=rows - 1)
    x = t[idx].view(rb, 16, k // 32, 4, 8)                       # [b, r, t, c, 8]
    l = torch.arange(64, device=t.device)
    r, c = l >> 2, (l & 3) ^ ((l >> 5) << 1)
    return x.permute(0, 2, 1, 3, 4)[:, :, r, c, :].reshape(-1)   # [b, t, l, 8]


def prep(n, k, epi, name):
    w = (torch.randn((n, k), device=DEV) * 0.02).to(torch.bfloat16)
    a = torch.randn((M, k), device=DEV)
    a[:, ::97] *= 8.0
    c = dict(n=n, k=k, epi=epi, w=w, ncols=n // 2 if epi == _lib.EPI_SWIGLU else n)
    c["hi"] = a.to(torch.bfloat16)
    c["lo"] = (a - c["hi"].float()).to(torch.bfloat16)
    c["wt"], c["hit"], c["lot"] = tile(w), tile(c["hi"]), tile(c["lo"])
    nc = c["ncols"]
    c["resid"] = torch.randn((M, nc), device=DEV)
    c["bias"] = torch.randn((n,), device=DEV) if name == "qkv" else None
    for v in ("rm", "w", "wa"):
        c["out_" + v] = torch.zeros((M, nc), device=DEV)
        c["phi_" + v], c["plo_" + v] = (torch.zeros((M, nc), dtype=torch.bfloat16, device=DEV) for _ in range(2))
        c["nhi_" + v], c["nlo_" + v] = (torch.zeros((M, n), dtype=torch.bfloat16, device=DEV) for _ in range(2))
    c["nw"] = torch.rand((n,), device=DEV) + 0.5
    c["wsp"] = torch.empty(16 * M * n * 4, dtype=torch.uint8, device=DEV)
    return c


def run(c, v):
    la = _lib.LinearArgs(a=None, w=c["w"].data_ptr(), bias=_lib.ptr(c["bias"]), resid=c["resid"].data_ptr() if c["epi"] == _lib.EPI_RESID else None,
                         c=c["out_" + v].data_ptr(), norm_w=None, norm_eps=0.0, m=M, n=c["n"], k=c["k"], lda=c["k"], ldw=c["k"], ldc=c["ncols"],
                         epilogue=c["epi"], workspace=c["wsp"].data_ptr(), workspace_bytes=c["wsp"].numel())
    la.a_hi, la.a_lo, la.ld_planes = c["hi"].data_ptr(), c["lo"].data_ptr(), c["k"]
    if v in ("w", "wa"):
        la.w_tiled = c["wt"].data_ptr()
    if v == "wa":
        la.a_hi, la.a_lo, la.planes_tiled = c["hit"].data_ptr(), c["lot"].data_ptr(), 1
    if c["epi"] in (_lib.EPI_SWIGLU, _lib.EPI_GELU):
        la.c = None
        la.c_hi, la.c_lo, la.ld_cplanes = c["phi_" + v].data_ptr(), c["plo_" + v].data_ptr(), c["ncols"]
    if c["epi"] == _lib.EPI_RESID:
        la.post_norm_w, la.post_norm_eps = c["nw"].data_ptr(), 1e-6
        la.post_hi, la.post_lo, la.ld_post = c["nhi_" + v].data_ptr(), c["nlo_" + v].data_ptr(), c["n"]
    _lib.check(lib.chatts_linear(la, st.cuda_stream))


VARIANTS = (("rm", "row-major"), ("w", "W tiled"), ("wa", "W + planes tiled"))
if os.environ.get("TILED_VARIANTS"):
    VARIANTS = tuple(v for v in VARIANTS if v[0] in os.environ["TILED_VARIANTS"].split(","))
cases = {name: prep(n, k, epi, name) for name, (n, k, epi) in SHAPES.items()}
for name, c in cases.items():
    ok_layout = bool((tile_ref(c["w"]).view(torch.int16) == c["wt"].view(torch.int16)).all()) and \
        bool((tile_ref(c["hi"]).view(torch.int16) == c["hit"].view(torch.int16)).all())
    for v, _ in VARIANTS:
        run(c, v)
    torch.cuda.synchronize()
    same = {}
    for v in [x for x, _ in VARIANTS if x != "rm"]:
        keys = ["out_"] if c["epi"] not in (_lib.EPI_SWIGLU, _lib.EPI_GELU) else ["phi_", "plo_"]
        if c["epi"] == _lib.EPI_RESID:
            keys += ["nhi_", "nlo_"]
        same[v] = all(bool((c[k + v].view(torch.int16 if c[k + v].dtype == torch.bfloat16 else torch.int32) ==
                            c[k + "rm"].view(torch.int16 if c[k + v].dtype == torch.bfloat16 else torch.int32)).all()) for k in keys)
    nz = float(c["out_rm"].abs().sum() + c["phi_rm"].float().abs().sum())
    print(f"{name:8s} tiled layout == restatement: {ok_layout}   outputs bit-identical to row-major: {same}  (|out| {nz:.3e})", flush=True)

res = {(s, v): [] for s in SHAPES for v, _ in VARIANTS}
for rnd in range(R + 1):
    for sname, c in cases.items():
        for v, _ in VARIANTS:
            run(c, v)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(3):
                run(c, v)
            e1.record(st)
            torch.cuda.synchronize()
            if rnd > 0:
                res[(sname, v)].append(e0.elapsed_time(e1) * 1e3 / 3)
tot = {v: 0.0 for v, _ in VARIANTS}
for sname in SHAPES:
    line = f"{sname:8s}"
    for v, label in VARIANTS:
        t = res[(sname, v)]
        tot[v] += float(np.median(t))
        line += f"  {label}: median {np.median(t):7.1f} min {min(t):7.1f} us |"
    print(line)
print("layer sum (medians): " + "  ".join(f"{label}: {tot[v]:7.1f} us" for v, label in VARIANTS))
