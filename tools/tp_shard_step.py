#!/usr/bin/env python
"""What ONE GPU can measure of tensor parallelism: the step time of ONE RANK of TP = W at the real ChatTS-14B shard shapes.

    python tools/tp_shard_step.py --worlds 1,2,4,8 --out gpurun_out/r4_tp_shard_step.json
    CHATTS_TP_FUSE=0 python tools/tp_shard_step.py --worlds 8 ...        # the stand-alone exchange kernels (A/B)
    rocprofv3 --kernel-trace --stats ... -- python tools/tp_shard_step.py --worlds 8 --steps 8

Rank 0 of a W-rank group is built at FULL depth (48 layers; 28 GB / W of decoder weights) and given a LOOP-BACK exchange
(chatts_tp_init_loopback): every push of a collective lands in the rank's own buffer, in the slot of the peer it would have gone to,
so the captured decode step issues exactly the stores, polls and launches of a real TP step - only the xGMI hop is missing (and the
"sums" are this rank's partials alone: TIMING ONLY, the tokens mean nothing).  Measured per W:
  * decode: ms per graph-replayed step (batch 1, ctx = the 798-token bench prompt), the DESIGN.md section 6 model's compute term;
  * prefill: ms for this rank's share of the 798-token prompt as ONE chatts_decoder_prefill_last call: layer halves + the two-shot
    [T, H] sums between them (chatts_allreduce_bulk, looped back: the kernel's stores, flags and local passes are real, the 2 x 14 MB
    that would cross the links are local writes - the link time is modelled separately);
  * optional --batch B --weights fp8: the B-wide decode step of config 5 at ctx --ctx.
The link term (96 exchanges x one xGMI hop, prefill all-reduce bandwidth) stays the only modelled part of a TP estimate."""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--model", default="chatts-14b")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"])
    ap.add_argument("--ctx", type=int, default=1207, help="--batch > 1: context length the batched step is timed at")
    ap.add_argument("--prefill-runs", type=int, default=3)
    ap.add_argument("--out", default="gpurun_out/r4_tp_shard_step.json")
    args = ap.parse_args()

    import torch
    import bench
    from chatts_amd import config as cfgmod
    from chatts_amd.modeling import ChatTSForCausalLM
    from chatts_amd.tp import LocalComm, P2PExchange

    class LoneRank(LocalComm):
        """rank 0 of a `world`-rank group with nobody to talk to: host-driven collectives are no-ops"""

        def __init__(self, world):
            self.rank, self.world, self.group, self.dist = 0, world, None, None

        def all_reduce(self, t):
            return t

        def barrier(self):
            pass

    over = {} if args.layers is None else {"num_hidden_layers": args.layers}
    cfg = cfgmod.preset(args.model, **over)
    proc, prompt, series, lengths = bench.build_inputs(cfg, 8, 256)
    inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
    ids = inputs["input_ids"][0].tolist()
    fuse = os.environ.get("CHATTS_TP_FUSE", "1") != "0"
    res = {"model": args.model, "layers": cfg.num_hidden_layers, "batch": args.batch, "weights": args.weights,
           "exchange": "in the o_proj / down_proj GEMV launches (ChattsLinearArgs.tp_reduce)" if fuse else "stand-alone chatts_allreduce kernels",
           "note": "ONE rank on ONE GPU, loop-back exchange: real launches / stores / polls of a TP step (decode) and of the two-shot "
                   "[T, H] sums (prefill), zero link latency / local instead of remote writes", "worlds": {}}
    B = max(1, args.batch)
    for W in [int(w) for w in args.worlds.split(",")]:
        t0 = time.time()
        model = ChatTSForCausalLM.from_synthetic(cfg, seed=0, comm=LoneRank(W) if W > 1 else None, max_ctx=2048, max_prefill_tokens=1024,
                                                 weight_format=args.weights, max_batch=B)
        if W > 1:
            model.attach_exchange(P2PExchange.create_loopback(0, W, model.exchange_elems(), model.exchange_bulk_elems()))
        torch.cuda.synchronize()
        row = {"weight_gb_this_rank": model.weight_bytes_local() / 1e9, "build_s": time.time() - t0}
        ser = inputs["timeseries"].cuda()
        mm = model.get_multimodal_embeddings(timeseries=ser, valid_lengths=proc.last_lengths)
        full = model.expand_input_ids(ids, [(L + 15) // 16 for L in lengths])
        T = len(full)
        emb = model.get_input_embeddings(torch.tensor(full, dtype=torch.int64), mm)
        # ---- prefill of this rank's share ---------------------------------------------------------------------------
        pre = []
        for i in range(args.prefill_runs + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.reset()
            last = model.prefill(emb, 0, for_next_token=True)
            model.buf["pos"].fill_(T)
            model._first_token(last)
            torch.cuda.synchronize()
            if i:
                pre.append((time.perf_counter() - t0) * 1e3)
        row["prompt_tokens"] = T
        row["prefill_ms"] = sorted(pre)[len(pre) // 2]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if B == 1:
            for _ in range(args.warmup):
                model.decode_step()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.steps):
                model.decode_step()
            e1.record()
            torch.cuda.synchronize()
            row["decode_ms_per_step"] = e0.elapsed_time(e1) / args.steps
            row["decode_graph"] = model.graph_capturable()
        else:
            Bf = model.buf
            Bf["pos_all"].fill_(args.ctx); Bf["step_all"].zero_(); Bf["token_all"].fill_(11)
            for _ in range(args.warmup):
                model.batched_step()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.steps):
                model.batched_step()
            e1.record()
            torch.cuda.synchronize()
            row["batched_decode_ms_per_step"] = e0.elapsed_time(e1) / args.steps
            row["ctx"] = args.ctx
        row["exchange_status"] = model._tp.status() if model._tp is not None else 0
        res["worlds"][str(W)] = row
        print(f"[tp_shard_step] W={W}: {json.dumps(row)}", file=sys.stderr, flush=True)
        if model._tp is not None:
            ex = model._tp
            model.attach_exchange(None)
            ex.close()
        del model
        gc.collect()
        torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
