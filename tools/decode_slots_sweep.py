"""GPU: batch-1 decode step time vs the number of 16-key slots of the decode attention (ChatTS-14B, 8x256 prompt).
    python tools/decode_slots_sweep.py [n_splits ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_inputs  # noqa: E402
from chatts_amd import config as cfgmod  # noqa: E402
from chatts_amd.modeling import ChatTSForCausalLM  # noqa: E402

cfg = cfgmod.preset("chatts-14b")
proc, prompt, series, lengths = build_inputs(cfg)
inp = proc(text=[prompt], timeseries=series, return_tensors="pt")
ids, ser = inp["input_ids"][0].tolist(), inp["timeseries"]
model = ChatTSForCausalLM.from_synthetic(cfg, seed=0, max_ctx=1024, max_prefill_tokens=1024)
for ns in [int(a) for a in sys.argv[1:]] or [64, 32, 16, 8]:
    model.n_splits = ns
    model._graph = None
    toks = model.generate_one(ids, ser.cuda(), lengths, 8, eos_token_id=None)
    for _ in range(8):
        model.decode_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(64):
        model.decode_step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 64 * 1e3
    print(f"n_splits={ns:3d}: {ms:.3f} ms/step = {1e3 / ms:.1f} tok/s   first tokens {toks[:4]}")
