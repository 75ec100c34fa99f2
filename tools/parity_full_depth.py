#!/usr/bin/env python
"""One-off GPU-box job: FULL-DEPTH parity of the headline configuration against the CPU float32 oracle.

    python tools/parity_full_depth.py --model chatts-14b --out gpurun_out/r2_parity_14b_full.json

Runs the exact bench.py workload (bench.build_inputs, seed 0, all 48 layers) on the HIP path and through the oracle
(oracle/ - TEST INFRASTRUCTURE; layer-streamed: weights are copied back from the device, un-packed into HF names and
widened one layer at a time), and writes first-token logits error, per-step logits errors and both token lists.
bench.py compares the tokens it generates with `tokens_oracle` of the committed file and reports `parity_checked`.
"""
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
    ap.add_argument("--series", type=int, default=8)
    ap.add_argument("--length", type=int, default=256)
    ap.add_argument("--new", type=int, default=9, help="first token + decode steps compared")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/r2_parity_full.json")
    args = ap.parse_args()

    import numpy as np
    import psutil
    import torch
    import bench
    from chatts_amd import config as cfgmod
    from chatts_amd.modeling import ChatTSForCausalLM
    from oracle import from_device, protocol, ts_embedding
    from oracle.qwen_decoder import QwenOracle

    def rel_err(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

    over = {} if args.layers is None else {"num_hidden_layers": args.layers}
    cfg = cfgmod.preset(args.model, **over)
    proc, prompt, series, lengths = bench.build_inputs(cfg, args.series, args.length)
    inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
    ids = inputs["input_ids"][0].tolist()
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=0, max_ctx=2048, max_prefill_tokens=1024)
    ser = inputs["timeseries"].cuda()

    # ---- HIP path, driven like bench.py ---------------------------------------------------------------------
    model.use_graph = True
    toks_gpu, logits0 = model.generate_one(ids, ser, proc.last_lengths, args.new, eos_token_id=None, return_logits=True)
    model.generate_one(ids, ser, proc.last_lengths, 1, eos_token_id=None)
    step_logits = [logits0.cpu().numpy()]
    for _ in range(1, args.new):
        model.decode_step()
        step_logits.append(model.buf["logits"].cpu().numpy())
    torch.cuda.synchronize()

    # ---- oracle, layer-streamed ---------------------------------------------------------------------------
    ncores = os.cpu_count() or 1
    nthreads = args.threads or min(ncores, 64)
    torch.set_num_threads(nthreads)
    avail = psutil.virtual_memory().available
    need_f32 = cfg.param_counts()["decoder"] * 4
    keep = "f32" if avail > 2.2 * need_f32 else ("bf16" if avail > 1.5 * need_f32 / 2 else "none")
    t0 = time.time()
    w = from_device.LayerStreamedWeights(model, keep=keep)
    t_copy = time.time() - t0
    tsw = {k[len("ts_encoder."):]: v for k, v in from_device.ts_encoder_state_dict(model).items()}
    feats, pc = ts_embedding.ts_embedding_forward(inputs["timeseries"].numpy().astype(np.float32), cfg.ts, tsw)
    full = protocol.expand_placeholders(ids, pc, cfg.ts_token_start_index)
    emb = protocol.merge_embeddings(full, w["model.embed_tokens.weight"].numpy(), feats, cfg.ts_token_start_index)
    o = QwenOracle(cfg.oracle_dict(), w)
    t0 = time.time()
    toks_ref, logits_ref = o.greedy(torch.from_numpy(emb), args.new)
    t_oracle = time.time() - t0

    errs = [rel_err(step_logits[i], logits_ref[i].numpy()) for i in range(min(len(step_logits), len(logits_ref)))]
    # the GPU continuation equals the oracle's only while the tokens agree; errors after a divergence are meaningless
    agree = next((i for i, (a, b) in enumerate(zip(toks_gpu, toks_ref)) if a != b), len(toks_ref))
    res = {
        "what": "full-depth parity of the bench.py workload: HIP path vs CPU float32 oracle (layer-streamed)",
        "model": args.model, "layers": cfg.num_hidden_layers, "series": args.series, "length": args.length,
        "prompt_tokens": len(full), "seed": 0, "tolerance": 1e-3,
        "first_token_logits_rel_err": errs[0], "step_logits_rel_err": errs, "max_logits_rel_err_while_tokens_agree": max(errs[:max(agree, 1)]),
        "tokens_gpu": toks_gpu, "tokens_oracle": toks_ref, "identical_tokens": toks_gpu == toks_ref,
        "top2_margin_first_token": float(torch.topk(logits_ref[0], 2).values.diff().abs()),
        "oracle": {"threads": nthreads, "host_cores": ncores, "weights_kept_as": keep, "copy_back_s": t_copy,
                   "greedy_wall_s": t_oracle},
        "passed": bool(toks_gpu == toks_ref and max(errs) < 1e-3),
    }
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))
    return 0 if res["passed"] else 1


if __name__ == "__main__":
    sys.exit(main())
