#!/usr/bin/env python
"""GPU-box check of the persistent decode step: correctness against the multi-kernel schedule (bitwise) and step time of both,
at ChatTS-14B widths.  python tools/mega_check.py [--layers N] [--steps K]  -> one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="chatts-14b")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--new", type=int, default=6)
    args = ap.parse_args()
    import torch
    import bench
    from chatts_amd import config as cfgmod
    from chatts_amd.modeling import ChatTSForCausalLM
    over = {} if args.layers is None else {"num_hidden_layers": args.layers}
    cfg = cfgmod.preset(args.model, **over)
    proc, prompt, series, lengths = bench.build_inputs(cfg, 8, 256)
    inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
    ids = inputs["input_ids"][0].tolist()
    ser = inputs["timeseries"].cuda()
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=0, max_ctx=2048, max_prefill_tokens=1024, enable_prefix_caching=False)
    res = {"model": args.model, "layers": cfg.num_hidden_layers, "available": bool(model.enable_persistent_decode(True))}
    out = {}
    for mega in (True, False):
        model.enable_persistent_decode(mega)
        model.use_graph = True
        toks, l0 = model.generate_one(ids, ser, proc.last_lengths, args.new, eos_token_id=None, return_logits=True)
        last = model.buf["logits"].clone()
        for _ in range(4):
            model.decode_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model.decode_step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        out[mega] = (toks, l0, last)
        res["ms_per_step_" + ("persistent" if mega else "multi_kernel")] = ms
        res["status_" + ("persistent" if mega else "multi_kernel")] = model.persistent_decode_status() if mega else 0
    res["tokens_equal"] = out[True][0] == out[False][0]
    res["first_logits_bitwise"] = bool(torch.equal(out[True][1], out[False][1]))
    res["last_logits_bitwise"] = bool(torch.equal(out[True][2], out[False][2]))
    res["last_logits_max_abs_diff"] = float((out[True][2] - out[False][2]).abs().max())
    res["tokens_persistent"] = out[True][0]
    res["tokens_multi_kernel"] = out[False][0]
    print(json.dumps(res), flush=True)
    return 0 if res["tokens_equal"] and res["last_logits_bitwise"] else 1


if __name__ == "__main__":
    sys.exit(main())
