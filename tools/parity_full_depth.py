#!/usr/bin/env python
"""One-off GPU-box job: FULL-DEPTH parity of a bench.py workload against the CPU float32 oracle.

    python tools/parity_full_depth.py --model chatts-14b --out gpurun_out/r3_parity_14b_8x256_bf16_b1_full.json            # headline (config 3)
    python tools/parity_full_depth.py --series 30 --lengths mixed --out gpurun_out/r3_parity_14b_30xmixed_bf16_b1_full.json   # config 4
    python tools/parity_full_depth.py --series 8 --length 1024 --batch 16 --weights fp8 --oracle-slots 0,7,15 \
        --out gpurun_out/r3_parity_14b_8x1024_fp8_b16_full.json                                                            # config 5

Runs the exact bench.py workload (bench.build_inputs / bench.build_batched_requests, seed 0, all layers) on the HIP path, driven
the way bench.py drives it (chunked prefill + graph-replayed decode steps; for --batch: packed admission + the batched decode
graph), and through the oracle (oracle/ - TEST INFRASTRUCTURE; layer-streamed: weights are copied back from the device - for
fp8 these are the dequantised values the fp8 copy encodes losslessly - un-packed into HF names and widened one layer at a time).
Writes per-step logits errors (norm-wise AND max|delta| / max|logit|) and both token lists; bench.py compares the tokens it
generates with `tokens_oracle` of the committed file (profiles/r3_parity_<workload key>_full.json) and prints `parity_checked`.
For --batch the oracle covers --oracle-slots (each slot costs a full-depth CPU prefill); the GPU runs all slots together.
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
    ap.add_argument("--lengths", default="uniform", choices=["uniform", "mixed"])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8", "int8", "int4"])
    ap.add_argument("--oracle-slots", default="0", help="--batch > 1: cache slots the oracle recomputes (comma separated)")
    ap.add_argument("--new", type=int, default=9, help="first token + decode steps compared")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--precision", default=None, choices=[None, "bf16x2", "f16q"], help="f16q: the opt-in parity-grade prefill on the f16 + fp8 matrix pipes")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/r3_parity_full.json")
    args = ap.parse_args()

    import numpy as np
    import psutil
    import torch
    import bench
    from chatts_amd import config as cfgmod
    from chatts_amd.modeling import ChatTSForCausalLM
    from oracle import from_device, protocol, ts_embedding
    from oracle.qwen_decoder import QwenOracle

    def errs_of(a, b):
        """(norm-wise relative error, max|a - b| / max|b|) of two logits vectors"""
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return (float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)), float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)))

    over = {} if args.layers is None else {"num_hidden_layers": args.layers}
    cfg = cfgmod.preset(args.model, **over)
    B = max(1, args.batch)
    if B == 1:
        proc, prompt, series, lengths = bench.build_inputs(cfg, args.series, args.length, args.lengths)
        reqs = [series]
    else:
        proc, prompt, reqs, lengths = bench.build_batched_requests(cfg, B, args.series, args.length)
    enc = [proc(text=[prompt], timeseries=s, padding=True, return_tensors="pt") for s in reqs]
    ids = enc[0]["input_ids"][0].tolist()
    T = len(ids) - 2 * len(lengths) + sum((L + 15) // 16 for L in lengths)
    max_ctx = max(2048, -(-(T + args.new + 16) // 256) * 256)
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=0, max_ctx=max_ctx, max_prefill_tokens=1024, weight_format=args.weights,
                                             max_batch=B, precision=args.precision)
    model.use_graph = True

    # ---- HIP path, driven like bench.py ---------------------------------------------------------------------
    if B == 1:
        ser = enc[0]["timeseries"].cuda()
        toks_gpu, logits0 = model.generate_one(ids, ser, proc.last_lengths, args.new, eos_token_id=None, return_logits=True)
        model.generate_one(ids, ser, proc.last_lengths, 1, eos_token_id=None)
        step_logits = {0: [logits0.cpu().numpy()]}
        for _ in range(1, args.new):
            model.decode_step()
            step_logits[0].append(model.buf["logits"].cpu().numpy())
        toks_gpu = {0: toks_gpu}
        slots = [0]
    else:
        slots = sorted({int(s) for s in args.oracle_slots.split(",") if s != ""})
        Bf = model.buf
        Bf["pos_all"].zero_(); Bf["step_all"].zero_(); Bf["token_all"].zero_()
        bench.admit_batched(model, proc, prompt, reqs, args.new)
        step_logits = {s: [None] for s in slots}           # first-token logits of a packed admission are not kept per slot
        for _ in range(1, args.new):
            model.batched_step()
            la = Bf["logits_all"]
            for s in slots:
                step_logits[s].append(la[s].cpu().numpy())
        all_toks = Bf["out_tokens_all"][:, :args.new].tolist()
        toks_gpu = {s: all_toks[s] for s in range(B)}
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
    per_slot, toks_ref, t_oracle = {}, {}, 0.0
    worst_rel, worst_abs, ok = 0.0, 0.0, True
    for s in slots:
        feats, pc = ts_embedding.ts_embedding_forward(enc[s]["timeseries"].numpy().astype(np.float32), cfg.ts, tsw)
        full = protocol.expand_placeholders(enc[s]["input_ids"][0].tolist(), pc, cfg.ts_token_start_index)
        emb = protocol.merge_embeddings(full, w["model.embed_tokens.weight"].numpy(), feats, cfg.ts_token_start_index)
        o = QwenOracle(cfg.oracle_dict(), w)
        t0 = time.time()
        tr, logits_ref = o.greedy(torch.from_numpy(emb), args.new)
        t_oracle += time.time() - t0
        toks_ref[s] = tr
        # the GPU continuation equals the oracle's only while the tokens agree; errors after a divergence are meaningless
        agree = next((i for i, (a, b) in enumerate(zip(toks_gpu[s], tr)) if a != b), len(tr))
        pairs = [errs_of(g, logits_ref[i].numpy()) if g is not None else None for i, g in enumerate(step_logits[s][:len(logits_ref)])]
        rel = [p[0] if p else None for p in pairs]
        mab = [p[1] if p else None for p in pairs]
        live = [p for p in pairs[:max(agree, 1)] if p]
        worst_rel = max([worst_rel] + [p[0] for p in live])
        worst_abs = max([worst_abs] + [p[1] for p in live])
        ok = ok and toks_gpu[s][:len(tr)] == tr and all(p[0] < 1e-3 for p in pairs if p)
        per_slot[str(s)] = {"step_logits_rel_err": rel, "step_max_abs_err_over_max_logit": mab, "prompt_tokens": len(full),
                            "top2_margin_first_token": float(torch.topk(logits_ref[0], 2).values.diff().abs())}

    key = bench.workload_key(argparse.Namespace(model=args.model, series=args.series, length=args.length, lengths=args.lengths,
                                                weights=args.weights, batch=B))
    res = {
        "what": "full-depth parity of a bench.py workload: HIP path vs CPU float32 oracle (layer-streamed)",
        "workload_key": key, "precision": args.precision or "bf16x2", "model": args.model, "layers": cfg.num_hidden_layers, "series": args.series,
        "length": None if args.lengths == "mixed" else args.length, "lengths": lengths, "batch": B, "weights": args.weights,
        "prompt_tokens": per_slot[str(slots[0])]["prompt_tokens"], "seed": 0, "tolerance": 1e-3,
        "max_step_logits_rel_err": worst_rel, "max_abs_err_over_max_logit": worst_abs,
        "rel_err_definition": "norm-wise ||gpu - oracle||_2 / ||oracle||_2 per step; max_abs_err_over_max_logit = max|gpu - oracle| / max|oracle|",
        "oracle": {"threads": nthreads, "host_cores": ncores, "weights_kept_as": keep, "copy_back_s": t_copy, "greedy_wall_s": t_oracle,
                   "weights": "the device's bf16 tensors (for fp8: the dequantised values the fp8 copy encodes exactly)"},
        "passed": bool(ok and worst_rel < 1e-3),
    }
    if B == 1:
        p0 = per_slot["0"]
        res.update(first_token_logits_rel_err=p0["step_logits_rel_err"][0], step_logits_rel_err=p0["step_logits_rel_err"],
                   step_max_abs_err_over_max_logit=p0["step_max_abs_err_over_max_logit"],
                   top2_margin_first_token=p0["top2_margin_first_token"],
                   tokens_gpu=toks_gpu[0], tokens_oracle=toks_ref[0], identical_tokens=toks_gpu[0] == toks_ref[0])
    else:
        res.update(oracle_slots=slots, per_slot=per_slot, tokens_gpu={str(s): toks_gpu[s] for s in range(B)},
                   tokens_oracle={str(s): toks_ref[s] for s in slots},
                   identical_tokens=all(toks_gpu[s][:len(toks_ref[s])] == toks_ref[s] for s in slots),
                   note="first-token logits of the packed admission are not kept per slot: the first token is compared as a token, "
                        "steps 1.. as logits (logits_all of the batched decode graph)")
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("tokens_gpu", "lengths")}))
    return 0 if res["passed"] else 1


if __name__ == "__main__":
    sys.exit(main())
