"""The prefill GEMM (gemm_ring_kernel) against float64 and against the register-staged kernel on the float32 operand: values,
bit-equality at equal split-K, and time - over the ChatTS-14B / 8B projection shapes, several M and (T, split-K) choices.
(Round 5 ran it against the round-4 LDS-DMA kernel it replaced: profiles/r5_ring_check_first.txt.)
    python tools/gemm_ring_check.py [quick]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
st = torch.cuda.current_stream()
QUICK = len(sys.argv) > 1 and sys.argv[1] == "quick"
torch.manual_seed(0)


def setenv(**env):
    for kk in list(os.environ):
        if kk.startswith("CHATTS_GEMM_"):
            del os.environ[kk]
    os.environ.update({"CHATTS_GEMM_" + a: str(b) for a, b in env.items()})
    _lib.sync_env()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


class Case:
    def __init__(self, M, n, k, epi, planes_out=False, bias=False, post_norm=False):
        self.M, self.n, self.k, self.epi, self.planes_out, self.post_norm = M, n, k, epi, planes_out, post_norm
        self.ws = [(torch.randn((n, k), device=DEV) * 0.02).to(torch.bfloat16) for _ in range(3)]
        a = torch.randn((M, k), device=DEV)
        self.a = a
        self.hi = a.to(torch.bfloat16)
        self.lo = (a - self.hi.float()).to(torch.bfloat16)
        self.ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
        self.resid = torch.randn((M, self.ncols), device=DEV)
        self.bias = torch.randn((n,), device=DEV) if bias else None
        self.out = torch.zeros((M, self.ncols), device=DEV)
        self.phi = torch.zeros((M, self.ncols), dtype=torch.bfloat16, device=DEV)
        self.plo = torch.zeros((M, self.ncols), dtype=torch.bfloat16, device=DEV)
        self.nw = torch.rand((n,), device=DEV) + 0.5
        self.nhi = torch.zeros((M, n), dtype=torch.bfloat16, device=DEV)
        self.nlo = torch.zeros((M, n), dtype=torch.bfloat16, device=DEV)
        setenv()
        self.wsb = max(int(lib.chatts_linear_workspace(M, n, k)), 16 * M * n * 4)
        self.wsp = torch.empty(self.wsb, dtype=torch.uint8, device=DEV)

    def run(self, w=None, planes=True):
        w = self.ws[0] if w is None else w
        la = _lib.LinearArgs(a=None if planes else self.a.data_ptr(), w=w.data_ptr(), bias=_lib.ptr(self.bias), resid=self.resid.data_ptr() if self.epi == _lib.EPI_RESID else None,
                             c=self.out.data_ptr(), norm_w=None, norm_eps=0.0, m=self.M, n=self.n, k=self.k, lda=self.k, ldw=self.k,
                             ldc=self.ncols, epilogue=self.epi, workspace=self.wsp.data_ptr(), workspace_bytes=self.wsb)
        if planes:
            la.a_hi, la.a_lo, la.ld_planes = self.hi.data_ptr(), self.lo.data_ptr(), self.k
        if self.planes_out:
            la.c = None
            la.c_hi, la.c_lo, la.ld_cplanes = self.phi.data_ptr(), self.plo.data_ptr(), self.ncols
        if self.post_norm:
            la.post_norm_w, la.post_norm_eps = self.nw.data_ptr(), 1e-6
            la.post_hi, la.post_lo, la.ld_post = self.nhi.data_ptr(), self.nlo.data_ptr(), self.n
        _lib.check(lib.chatts_linear(la, st.cuda_stream))

    def result(self):
        torch.cuda.synchronize()
        if self.planes_out:
            return (self.phi.float() + self.plo.float()).clone()
        return self.out.clone()

    def reference(self):
        """float64 on the device: (hi + lo) . W^T, then the epilogue"""
        a = self.hi.double() + self.lo.double()
        acc = a @ self.ws[0].double().t()
        if self.bias is not None:
            acc = acc + self.bias.double()
        if self.epi == _lib.EPI_SWIGLU:
            r = acc.view(self.M, self.n // 32, 2, 16)
            g, u = r[:, :, 0, :], r[:, :, 1, :]
            return (torch.nn.functional.silu(g) * u).reshape(self.M, self.n // 2)
        if self.epi == _lib.EPI_RESID:
            return self.resid.double() + acc
        if self.epi == _lib.EPI_GELU:
            return torch.nn.functional.gelu(acc)
        return acc


def check(tag, c, envs_ring, bits_vs_old=True):
    setenv()
    c.out.zero_(); c.phi.zero_(); c.plo.zero_()
    c.run(planes=False)
    old = c.result()
    t_old = timed(lambda: [c.run(w, planes=False) for w in c.ws]) / len(c.ws)
    ref = c.reference()
    tol = 3e-5 if not c.planes_out else 2e-4      # planes carry 16 mantissa bits
    e_old = ((old.double() - ref).norm() / ref.norm()).item()
    print(f"== {tag}: M={c.M} N={c.n} K={c.k}  register-staged kernel {t_old:8.1f} us  rel err {e_old:.2e}")
    ok = True
    for env in envs_ring:
        setenv(**env)
        c.out.fill_(float("nan")); c.phi.zero_(); c.plo.zero_()
        c.run()
        new = c.result()
        e_new = ((new.double() - ref).norm() / ref.norm()).item()
        same = torch.equal(new, old)
        finite = bool(torch.isfinite(new).all())
        t_new = timed(lambda: [c.run(w) for w in c.ws]) / len(c.ws)
        flag = "" if (finite and e_new < tol) else "   <-- WRONG"
        ok = ok and finite and e_new < tol
        et = " ".join(f"{a}={b}" for a, b in env.items()) or "auto"
        print(f"   ring {et:12s} {t_new:8.1f} us  ({t_old / t_new:4.2f}x)  {2.0 * c.M * c.n * c.k / t_new / 1e6:6.0f} TF useful  rel err {e_new:.2e}  "
              f"bit-equal to the register-staged kernel: {same}{flag}")
    return ok


ok = True
E = _lib
if QUICK:
    ok &= check("qkv", Case(798, 7168, 5120, E.EPI_NONE, bias=True), [{}])
    ok &= check("gate_up", Case(798, 27648, 5120, E.EPI_SWIGLU, planes_out=True), [{}])
    ok &= check("down", Case(798, 5120, 13824, E.EPI_RESID, post_norm=True), [{}])
else:
    ok &= check("qkv", Case(798, 7168, 5120, E.EPI_NONE, bias=True), [{}, {"T": 7}, {"T": 9}, {"T": 6}, {"T": 12}, {"T": 7, "SK": 1}])
    ok &= check("o", Case(798, 5120, 5120, E.EPI_RESID, post_norm=True), [{}, {"T": 6, "SK": 2}, {"T": 7, "SK": 3}, {"T": 12, "SK": 1}, {"T": 9, "SK": 1}, {"T": 5, "SK": 2}])
    ok &= check("gate_up", Case(798, 27648, 5120, E.EPI_SWIGLU, planes_out=True), [{}, {"T": 7}, {"T": 6}, {"T": 5}, {"T": 9}])
    ok &= check("down", Case(798, 5120, 13824, E.EPI_RESID, post_norm=True), [{}, {"T": 6, "SK": 2}, {"T": 7, "SK": 3}, {"T": 12, "SK": 1}, {"T": 6, "SK": 4}])
    # other M: one fragment short of / beyond tile multiples, short chunks, long chunks
    for M in (96, 130, 257, 512, 1024, 1185):
        ok &= check(f"qkv M={M}", Case(M, 7168, 5120, E.EPI_NONE, bias=True), [{}])
        ok &= check(f"down M={M}", Case(M, 5120, 13824, E.EPI_RESID), [{}])
    ok &= check("gate_up f32 out M=300", Case(300, 27648, 5120, E.EPI_SWIGLU, bias=True), [{}])
    ok &= check("gelu planes M=200", Case(200, 5120, 5120, E.EPI_GELU, planes_out=True, bias=True), [{}])
    ok &= check("8B qkv", Case(144, 6144, 4096, E.EPI_NONE), [{}])
    ok &= check("8B gate_up", Case(144, 24576, 4096, E.EPI_SWIGLU, planes_out=True), [{}])
    ok &= check("8B down", Case(144, 4096, 12288, E.EPI_RESID, post_norm=True), [{}])
    ok &= check("N not a panel multiple", Case(798, 5120 + 48, 5120, E.EPI_NONE, bias=True), [{}])
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
