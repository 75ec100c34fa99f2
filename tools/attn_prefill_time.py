"""GPU: time the prefill attention of the benchmark prompt (T = 798, 40 / 8 heads) - float32-MFMA kernel vs the bf16x3 kernel,
with and without the 2-way key split.  usage: python tools/attn_prefill_time.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
T, nq, nkv, d, ctx = 798, 40, 8, 128, 1024
qkv = torch.randn((T, (nq + 2 * nkv) * d), device=DEV)
kc = torch.randn((nkv, ctx, d), device=DEV)
vc = torch.randn((nkv, ctx, d), device=DEV)
out = torch.empty((T, nq * d), device=DEV)
cache = _lib.KvCache(k=kc.data_ptr(), v=vc.data_ptr(), max_ctx=ctx)
st = torch.cuda.current_stream()
ref = None
for mode in ("0", "1"):
    os.environ["CHATTS_ATTN_BF16X3"] = mode
    _lib.sync_env()
    for NS in (1, 2):
        wsb = int(lib.chatts_attn_workspace(T, nq, NS))
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)

        def run():
            _lib.check(lib.chatts_attention(qkv.data_ptr(), T, nq, nkv, 0, None, C.byref(cache), out.data_ptr(), NS, ws.data_ptr(), wsb, st.cuda_stream))
        run()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        err = float((out - ref).norm() / ref.norm())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            run()
        e1.record(st)
        torch.cuda.synchronize()
        print("bf16x3=%s key-splits %d: %.1f us  (rel. diff to the float32-MFMA kernel %.1e)" % (mode, NS, e0.elapsed_time(e1) * 1e3 / 20, err))
# the same with the K / V rows split into bf16 planes once (kv_planes_kernel + the copy-staging kernel: workspace large enough)
split_out = out.clone()
wsb = ((T + 31) // 32) * nkv * 4 * 32 * d * 2
ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
NS = 1
run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(20):
    run()
e1.record(st)
torch.cuda.synchronize()
os.environ["CHATTS_ATTN_BF16X3"] = "1"
_lib.sync_env()
ws16 = torch.empty(16, dtype=torch.uint8, device=DEV)
_lib.check(lib.chatts_attention(qkv.data_ptr(), T, nq, nkv, 0, None, C.byref(cache), split_out.data_ptr(), 1, ws16.data_ptr(), 16, st.cuda_stream))
torch.cuda.synchronize()
print("K / V planes + transposed tiles (kv_planes_kernel + attn_prefill_planes_kernel): %.1f us, rel. diff to split-while-staging %.1e"
      % (e0.elapsed_time(e1) * 1e3 / 20, float((out - split_out).norm() / split_out.norm())))
