#!/usr/bin/env python
"""Where a persistent decode step spends its time: s_memtime stamps of two workgroups at every phase edge
(chatts_decoder_mega_profile), averaged per phase kind over the layers.  python tools/mega_profile.py [--layers N]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="chatts-14b")
    ap.add_argument("--layers", type=int, default=None)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from chatts_amd import _lib, config as cfgmod
    from chatts_amd.modeling import ChatTSForCausalLM
    over = {} if args.layers is None else {"num_hidden_layers": args.layers}
    cfg = cfgmod.preset(args.model, **over)
    proc, prompt, series, lengths = bench.build_inputs(cfg, 8, 256)
    inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
    ids = inputs["input_ids"][0].tolist()
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=0, max_ctx=2048, max_prefill_tokens=1024, enable_prefix_caching=False)
    model.use_graph = False
    assert model.enable_persistent_decode(True), "persistent decode step not available for this model"
    model.generate_one(ids, inputs["timeseries"].cuda(), proc.last_lengths, 4, eos_token_id=None)
    L = cfg.num_hidden_layers
    n_ph = 6 * L + 1
    buf = torch.zeros(2 * n_ph * 16 + 12 * 1024, dtype=torch.int64, device="cuda")
    _lib.check(model.lib.chatts_decoder_mega_profile(model._decoder, buf.data_ptr(), buf.numel() * 8))
    for _ in range(3):
        model.decode_step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    model.decode_step()
    e1.record()
    torch.cuda.synchronize()
    step_us = e0.elapsed_time(e1) * 1e3
    allw = buf.cpu().numpy().astype(np.float64)
    st = allw[:2 * n_ph * 16].reshape(2, n_ph, 16)
    nwg = int(model.lib.chatts_device_cus())
    fin = allw[2 * n_ph * 16:2 * n_ph * 16 + 6 * nwg].reshape(6, nwg)
    beg = allw[2 * n_ph * 16 + 6 * nwg:2 * n_ph * 16 + 12 * nwg].reshape(6, nwg)
    _lib.check(model.lib.chatts_decoder_mega_profile(model._decoder, None, 0))
    span = st[0, n_ph - 1, 5] - st[0, 0, 0]
    tick_us = step_us / span if span > 0 else 0.0          # (kernel time ~ step time: launch overhead is a few us of ~5 ms)
    names = ["qkv", "attn", "combine", "o", "gate_up", "down"]
    out = {"model": args.model, "layers": L, "step_us": step_us, "ticks_per_us": 1.0 / tick_us if tick_us else None, "phases": {}}
    for wg in (0, 1):
        for k, name in enumerate(names + ["lm_head"]):
            rows = [st[wg, l * 6 + k] for l in range(1, L)] if k < 6 else [st[wg, n_ph - 1]]
            if not rows:
                continue
            r = np.mean(np.stack(rows), axis=0)
            d = lambda a, b: round(float((r[b] - r[a]) * tick_us), 2)
            out["phases"][f"wg{wg}.{name}"] = {"work": d(0, 1), "publish": d(1, 2), "grid_barrier": d(2, 3), "to_B": d(3, 4),
                                              "stage": d(4, 5), "total": d(0, 5), "wave0_loop": d(6, 7),
                                              "wave0_start_after_phase_start": d(0, 6),
                                              "wave0_block_starts_after_loop_start": [round(float((r[8 + i] - r[6]) * tick_us), 2) for i in range(8) if r[8 + i] > 0]}
    # layer 1: when each workgroup's own part of a phase was done, relative to the earliest workgroup (us): spread = the imbalance
    out["finish_spread_layer1"] = {}
    for k, name in enumerate(names):
        # (s_memtime is per XCD: only differences taken on ONE workgroup's clock mean anything) own work time of every workgroup
        f = (fin[k] - beg[k]) * tick_us
        by_x = [round(float(f[x::8].mean()), 2) for x in range(8)]
        out["finish_spread_layer1"][name] = {"min": round(float(f.min()), 2), "p50": round(float(np.percentile(f, 50)), 2),
                                             "p90": round(float(np.percentile(f, 90)), 2), "max": round(float(f.max()), 2),
                                             "mean_by_b_mod_8": by_x, "slowest_wgs": [int(i) for i in np.argsort(f)[-6:]]}
    per_layer = sum(out["phases"][f"wg0.{n}"]["total"] for n in names)
    out["per_layer_us_wg0"] = per_layer
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
