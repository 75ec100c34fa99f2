"""GPU: aggregate decode throughput of continuous batching vs batch size (ChatTS-14B, 8x256 prompt per request).
    python tools/batch_bench.py [B ...]"""
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
FMT = "fp8" if "fp8" in sys.argv[1:] else "bf16"
print(f"# weight_format = {FMT}")
for B in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4, 8, 16]:
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=0, max_ctx=1024, max_prefill_tokens=1024, max_batch=B,
                                             weight_format=FMT)
    reqs = [(ids, ser, lengths)] * B
    new = 48
    model.generate_batch(reqs, max_new_tokens=4)          # warm-up + graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = model.generate_batch(reqs, max_new_tokens=new, eos_token_id=None, sync_every=64)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # decode-only rate: time B-wide steps directly
    for _ in range(4):
        model.batched_step() if B > 1 else model.decode_step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(32):
        model.batched_step() if B > 1 else model.decode_step()
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t1) / 32 * 1e3
    print(f"B={B:3d}  end-to-end {B * new / dt:8.1f} tok/s (prefill of {B} prompts included)   decode step {step_ms:6.2f} ms "
          f"= {B / step_ms * 1e3:8.1f} tok/s aggregate   same tokens across slots: {all(o == outs[0] for o in outs)}")
    del model
    torch.cuda.empty_cache()
