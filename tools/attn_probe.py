"""GPU probe: time chatts_attention_decode_fused (both kernels) at several context lengths, rotating over many
KV buffers so the cache rows come from HBM like in a real decode step.   python tools/attn_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatts_amd import _lib  # noqa: E402

lib = _lib.load()
DEV = "cuda"
nq, nkv, max_ctx, NB = 40, 8, 2048, 48
raw = torch.randn((1, (nq + 2 * nkv) * 128), device=DEV)
caches = [(torch.randn((nkv, max_ctx, 128), device=DEV), torch.randn((nkv, max_ctx, 128), device=DEV)) for _ in range(NB)]
cos = torch.rand((max_ctx, 64), device=DEV)
sin = torch.rand((max_ctx, 64), device=DEV)
out = torch.empty((1, nq * 128), device=DEV)
st = torch.cuda.current_stream()
flush = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=DEV)     # 1 GB: evicts L2 + MALL between reps


def run(pos, splits, use_dev_pos):
    wsb = int(lib.chatts_attn_workspace(1, nq, splits))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    pos_dev = torch.tensor([pos], dtype=torch.int32, device=DEV)
    best = 1e9
    for rep in range(3):
        flush.add_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for k, v in caches:
            kc = _lib.KvCache(k=k.data_ptr(), v=v.data_ptr(), max_ctx=max_ctx)
            _lib.check(lib.chatts_attention_decode_fused(raw.data_ptr(), nq, nkv, None, None, 1e-6, cos.data_ptr(),
                                                         sin.data_ptr(), pos, pos_dev.data_ptr() if use_dev_pos else None,
                                                         C.byref(kc), out.data_ptr(), splits, ws.data_ptr(), wsb,
                                                         st.cuda_stream))
        e1.record(st)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / NB)
    return best


for pos in (15, 255, 830, 2000):
    for splits in (16, 64):
        for dev_pos in (False, True):
            print(f"pos={pos:5d} slots={splits:3d} pos_on_device={int(dev_pos)}  {run(pos, splits, dev_pos):7.2f} us per call (decode + combine, back to back)")
