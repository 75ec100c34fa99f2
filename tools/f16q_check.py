"""Round 6: the 1.5-pass prefill GEMM (csrc/gemm_f16q.hip) against its float64 reference, and an interleaved A/B against the bf16x2 kernel
(gemm_ring_kernel through chatts_linear) at the ChatTS-14B chunk shapes.
    python tools/f16q_check.py [M] [rounds] [--no-check]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import f16q_ref  # noqa: E402
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
st = torch.cuda.current_stream()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
M = int(args[0]) if len(args) > 0 else 798
R = int(args[1]) if len(args) > 1 else 7
CHECK = "--no-check" not in sys.argv
SHAPES = {"qkv": (7168, 5120, _lib.EPI_NONE), "o": (5120, 5120, _lib.EPI_RESID), "gate_up": (27648, 5120, _lib.EPI_SWIGLU),
          "down": (5120, 13824, _lib.EPI_RESID)}
torch.manual_seed(0)


def prep(n, k, epi, name):
    w = (torch.randn((n, k), device=DEV) * 0.02).to(torch.bfloat16)
    a = torch.randn((M, k), device=DEV)
    a[:, ::97] *= 8.0                                    # outlier columns
    c = dict(n=n, k=k, epi=epi, w=w, a=a, ncols=n // 2 if epi == _lib.EPI_SWIGLU else n)
    # bf16x2 operands
    c["hi"] = a.to(torch.bfloat16)
    c["lo"] = (a - c["hi"].float()).to(torch.bfloat16)
    # f16q operands through the library's own producers
    c["qhi"] = torch.empty((M, k), dtype=torch.float16, device=DEV)
    c["qlo"] = torch.empty((M, k), dtype=torch.uint8, device=DEV)
    c["qsc"] = torch.empty((M, k // 128), dtype=torch.uint8, device=DEV)
    _lib.check(lib.chatts_split_f16q(a.data_ptr(), M, k, k, c["qhi"].data_ptr(), c["qlo"].data_ptr(), c["qsc"].data_ptr(), k, k // 128, 0, st.cuda_stream))
    c["w16"] = torch.empty((n, k), dtype=torch.float16, device=DEV)
    c["w8"] = torch.empty((n, k), dtype=torch.uint8, device=DEV)
    c["w8e"] = torch.empty((n,), dtype=torch.uint8, device=DEV)
    _lib.check(lib.chatts_weights_f16q(w.data_ptr(), n, k, k, c["w16"].data_ptr(), c["w8"].data_ptr(), c["w8e"].data_ptr(), k, st.cuda_stream))
    c["w16t"] = torch.empty(lib.chatts_tile_bf16_elems(n, k), dtype=torch.float16, device=DEV)
    c["w8t"] = torch.empty(lib.chatts_tile_e4m3_bytes(n, k), dtype=torch.uint8, device=DEV)
    _lib.check(lib.chatts_tile_bf16(c["w16"].data_ptr(), n, k, k, c["w16t"].data_ptr(), st.cuda_stream))
    _lib.check(lib.chatts_tile_e4m3(c["w8"].data_ptr(), n, k, k, c["w8t"].data_ptr(), st.cuda_stream))
    c["out_qt"] = torch.zeros((M, c["ncols"]), device=DEV)
    # tiled activation planes (blocks of 16 rows: buffers hold ceil(M / 16) * 16 rows)
    M16 = (M + 15) // 16 * 16
    c["thi"] = torch.zeros((M16, k), dtype=torch.float16, device=DEV)
    c["tlo"] = torch.zeros((M16, k), dtype=torch.uint8, device=DEV)
    _lib.check(lib.chatts_split_f16q(a.data_ptr(), M, k, k, c["thi"].data_ptr(), c["tlo"].data_ptr(), c["qsc"].data_ptr(), k, k // 128, 1, st.cuda_stream))
    c["out_qa"] = torch.zeros((M, c["ncols"]), device=DEV)
    nc16 = c["ncols"]
    c["tchi"] = torch.zeros((M16, nc16), dtype=torch.float16, device=DEV)
    c["tclo"] = torch.zeros((M16, nc16), dtype=torch.uint8, device=DEV)
    c["tnhi"] = torch.zeros((M16, n), dtype=torch.float16, device=DEV)
    c["tnlo"] = torch.zeros((M16, n), dtype=torch.uint8, device=DEV)
    nc = c["ncols"]
    c["resid"] = torch.randn((M, nc), device=DEV)
    c["bias"] = torch.randn((n,), device=DEV) if name == "qkv" else None
    c["out"] = torch.zeros((M, nc), device=DEV)
    c["out_q"] = torch.zeros((M, nc), device=DEV)
    c["phi"], c["plo"] = (torch.empty((M, nc), dtype=torch.bfloat16, device=DEV) for _ in range(2))
    c["chi"] = torch.empty((M, nc), dtype=torch.float16, device=DEV)
    c["clo"] = torch.empty((M, nc), dtype=torch.uint8, device=DEV)
    c["csc"] = torch.empty((M, max(1, nc // 128)), dtype=torch.uint8, device=DEV)
    c["nw"] = torch.rand((n,), device=DEV) + 0.5
    c["nhi"], c["nlo"] = (torch.empty((M, n), dtype=torch.bfloat16, device=DEV) for _ in range(2))
    c["qnhi"] = torch.empty((M, n), dtype=torch.float16, device=DEV)
    c["qnlo"] = torch.empty((M, n), dtype=torch.uint8, device=DEV)
    c["qnsc"] = torch.empty((M, max(1, n // 128)), dtype=torch.uint8, device=DEV)
    c["wsp"] = torch.empty(16 * M * n * 4, dtype=torch.uint8, device=DEV)
    return c


def run_bf16x2(c):
    la = _lib.LinearArgs(a=None, w=c["w"].data_ptr(), bias=_lib.ptr(c["bias"]), resid=c["resid"].data_ptr() if c["epi"] == _lib.EPI_RESID else None,
                         c=c["out"].data_ptr(), norm_w=None, norm_eps=0.0, m=M, n=c["n"], k=c["k"], lda=c["k"], ldw=c["k"], ldc=c["ncols"],
                         epilogue=c["epi"], workspace=c["wsp"].data_ptr(), workspace_bytes=c["wsp"].numel())
    la.a_hi, la.a_lo, la.ld_planes = c["hi"].data_ptr(), c["lo"].data_ptr(), c["k"]
    if c["epi"] == _lib.EPI_SWIGLU and not c.get("plain"):
        la.c = None
        la.c_hi, la.c_lo, la.ld_cplanes = c["phi"].data_ptr(), c["plo"].data_ptr(), c["ncols"]
    if c["epi"] == _lib.EPI_RESID:
        la.post_norm_w, la.post_norm_eps = c["nw"].data_ptr(), 1e-6
        la.post_hi, la.post_lo, la.ld_post = c["nhi"].data_ptr(), c["nlo"].data_ptr(), c["n"]
    _lib.check(lib.chatts_linear(la, st.cuda_stream))


def run_f16q(c, tiled=False, atiled=False):
    qa = _lib.LinearF16qArgs(a_hi=c["thi" if atiled else "qhi"].data_ptr(), a_lo8=c["tlo" if atiled else "qlo"].data_ptr(), a_scale=c["qsc"].data_ptr(), ld_a=c["k"], ld_scale=c["k"] // 128,
                             planes_tiled=1 if atiled else 0,
                             w16=c["w16t" if tiled else "w16"].data_ptr(), w8=c["w8t" if tiled else "w8"].data_ptr(), w8_exp=c["w8e"].data_ptr(), ldw=c["k"],
                             w_tiled=1 if tiled else 0, bias=_lib.ptr(c["bias"]),
                             resid=c["resid"].data_ptr() if c["epi"] == _lib.EPI_RESID else None, c=c["out_qa" if atiled else ("out_qt" if tiled else "out_q")].data_ptr(), m=M, n=c["n"], k=c["k"],
                             ldc=c["ncols"], epilogue=c["epi"], workspace=c["wsp"].data_ptr(), workspace_bytes=c["wsp"].numel())
    if c["epi"] == _lib.EPI_SWIGLU and not c.get("plain"):
        qa.c = None
        qa.c_hi, qa.c_lo8, qa.c_scale, qa.ld_cplanes, qa.ld_cscale = c["tchi" if atiled else "chi"].data_ptr(), c["tclo" if atiled else "clo"].data_ptr(), c["csc"].data_ptr(), c["ncols"], c["ncols"] // 128
    if c["epi"] == _lib.EPI_RESID:
        qa.post_norm_w, qa.post_norm_eps = c["nw"].data_ptr(), 1e-6
        qa.post_hi, qa.post_lo8, qa.post_scale, qa.ld_post, qa.ld_pscale = c["tnhi" if atiled else "qnhi"].data_ptr(), c["tnlo" if atiled else "qnlo"].data_ptr(), c["qnsc"].data_ptr(), c["n"], c["n"] // 128
    _lib.check(lib.chatts_linear_f16q(qa, st.cuda_stream))


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


cases = {name: prep(n, k, epi, name) for name, (n, k, epi) in SHAPES.items()}
if CHECK:
    for name, c in cases.items():
        # producers: bit-exact against the restatement
        rh, rq, rs = f16q_ref.split(c["a"])
        w16, w8, w8e = f16q_ref.weights(c["w"])
        prod = dict(hi=bool((rh.view(torch.int16) == c["qhi"].view(torch.int16)).all()), lo8=bool((rq == c["qlo"]).all()), sc=bool((rs == c["qsc"]).all()),
                    w16=bool((w16.view(torch.int16) == c["w16"].view(torch.int16)).all()), w8=bool((w8 == c["w8"]).all()), w8e=bool((w8e == c["w8e"]).all()))
        # GEMM (float32 output forms) against the float64 product of the dequantised operands, and against the true float32 product
        c["plain"] = True
        run_f16q(c)
        run_f16q(c, tiled=True)
        run_f16q(c, tiled=True, atiled=True)
        run_bf16x2(c)
        torch.cuda.synchronize()
        tiled_same = bool((c["out_q"].view(torch.int32) == c["out_qt"].view(torch.int32)).all()) and bool((c["out_q"].view(torch.int32) == c["out_qa"].view(torch.int32)).all())
        ref = f16q_ref.gemm(c["qhi"], c["qlo"], c["qsc"], c["w16"], c["w8"], c["w8e"])
        true = c["a"].double() @ c["w"].double().t()
        if c["bias"] is not None:
            ref, true = ref + c["bias"].double(), true + c["bias"].double()
        if c["epi"] == _lib.EPI_RESID:
            ref, true = ref + c["resid"].double(), true + c["resid"].double()
        if c["epi"] == _lib.EPI_SWIGLU:
            def sw(t):
                v = t.view(M, c["n"] // 32, 2, 16)
                return (torch.nn.functional.silu(v[:, :, 0]) * v[:, :, 1]).reshape(M, c["n"] // 2)
            ref, true = sw(ref), sw(true)
        line = f"{name:8s} producers {'ok' if all(prod.values()) else prod}  f16q vs its float64 reference {rel(c['out_q'], ref):.2e}  vs the true product {rel(c['out_q'], true):.2e}" \
               f"  (bf16x2 vs true {rel(c['out'], true):.2e})  tiled weights / planes bit-identical: {tiled_same}"
        c["plain"] = False
        if c["epi"] == _lib.EPI_SWIGLU:          # plane output: bit-exact against the split of the float32 output
            run_f16q(c)
            torch.cuda.synchronize()
            eh, eq, es = f16q_ref.split(c["out_q"])
            line += f"  planes: hi {bool((eh.view(torch.int16) == c['chi'].view(torch.int16)).all())} lo8 {bool((eq == c['clo']).all())} scale {bool((es == c['csc']).all())}"
        if c["epi"] == _lib.EPI_RESID:           # post-norm planes against the split of RMSNorm(out)
            x = c["out_q"]
            y = c["nw"] * (x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6))
            eh, eq, es = f16q_ref.split(y)
            dh = (eh.float() - c["qnhi"].float()).abs().max().item()
            line += f"  post-norm planes: scale bytes equal {float((es == c['qnsc']).float().mean()):.4f}  max|hi - ref| {dh:.2e}"
        print(line, flush=True)

ARMS = (("bf16x2", run_bf16x2), ("f16q", run_f16q), ("f16q W tiled", lambda c: run_f16q(c, tiled=True)), ("f16q all tiled", lambda c: run_f16q(c, tiled=True, atiled=True)))
res = {(s, a): [] for s in SHAPES for a, _ in ARMS}
for rnd in range(R + 1):
    for sname, c in cases.items():
        for aname, fn in ARMS:
            fn(c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(3):
                fn(c)
            e1.record(st)
            torch.cuda.synchronize()
            if rnd > 0:
                res[(sname, aname)].append(e0.elapsed_time(e1) * 1e3 / 3)
tot = {a: 0.0 for a, _ in ARMS}
for sname in SHAPES:
    line = f"{sname:8s}"
    for aname in tot:
        v = res[(sname, aname)]
        tot[aname] += float(np.median(v))
        line += f"  {aname}: median {np.median(v):7.1f} min {min(v):7.1f} us"
    print(line)
print("layer sum (medians): " + "  ".join(f"{a}: {t:7.1f} us" for a, t in tot.items()))
