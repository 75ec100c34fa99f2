#!/usr/bin/env python
"""The precision="fp8" SPEED mode (gemm_fp8.hip: fp8 x fp8 on v_mfma_scale_f32_16x16x128_f8f6f4) next to the parity-grade default, on
BASELINE.json config 5's shapes: ChatTS-14B with fp8 weights, an 8 x 1024 prompt (1.2k tokens), the TS encoder at the patch count of the
whole 16-prompt batch (8192).  Same process, same synthetic weights, one model per mode.  Reports TTFT-relevant times, how far the
speed mode's first-token logits are from the default's (which is within 5e-5 of the float32 oracle: profiles/r3_parity_14b_8x1024_fp8_b16_full.json)
and how many greedy tokens agree.  -> gpurun_out/r4_fp8_speed_mode.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chatts_amd import config as cfgmod  # noqa: E402
from chatts_amd.modeling import ChatTSForCausalLM  # noqa: E402


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else None
    n_new, runs, B = 16, 4, 16
    cfg = cfgmod.preset("chatts-14b", **({} if layers is None else {"num_hidden_layers": layers}))
    proc, prompt, reqs, lengths = bench.build_batched_requests(cfg, B, 8, 1024)
    enc = [proc(text=[prompt], timeseries=r, padding=True, return_tensors="pt") for r in reqs]
    ids, ser = enc[0]["input_ids"][0].tolist(), enc[0]["timeseries"].cuda()
    ser_all = torch.cat([e["timeseries"] for e in enc], dim=0).cuda()
    out = {}
    for mode in ("bf16x2", "fp8"):
        m = ChatTSForCausalLM.from_synthetic(cfg, seed=0, max_ctx=2048, max_prefill_tokens=1024, weight_format="fp8", enable_prefix_caching=False,
                                             precision=None if mode == "bf16x2" else "fp8")
        ttft = []
        for _ in range(runs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m._prefill_request(ids, ser, list(proc.last_lengths), n_new)
            first = m.buf["out_tokens"][:1].tolist()     # noqa: F841  (the sync)
            ttft.append((time.perf_counter() - t0) * 1e3)
        toks, lg = m.generate_one(ids, ser, list(proc.last_lengths), n_new, return_logits=True)
        feats = torch.cat(m.get_multimodal_embeddings(timeseries=ser_all, valid_lengths=list(lengths) * B)).double().cpu()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        m.ts_encoder.replay_last()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            m.ts_encoder.replay_last()
        e1.record()
        torch.cuda.synchronize()
        out[mode] = {"ttft_ms_p50": sorted(ttft[1:])[len(ttft[1:]) // 2], "tokens": toks, "logits": lg.double().cpu(), "ts_feats": feats,
                     "ts_encode_ms_8192_patches": e0.elapsed_time(e1) / 5}
        del m
        torch.cuda.empty_cache()
    a, b = out["bf16x2"], out["fp8"]
    rel = float((b["logits"] - a["logits"]).norm() / a["logits"].norm())
    mab = float((b["logits"] - a["logits"]).abs().max() / a["logits"].abs().max())
    ts_rel = float((b["ts_feats"] - a["ts_feats"]).norm() / a["ts_feats"].norm())
    agree = next((i for i, (x, y) in enumerate(zip(a["tokens"], b["tokens"])) if x != y), len(a["tokens"]))
    res = {"model": "chatts-14b", "layers": cfg.num_hidden_layers, "weights": "fp8 (e4m3, pow2 row scales)", "prompt": "8 series x 1024 steps",
           "prompt_tokens": len(ids) - 2 * len(lengths) + sum((L + 15) // 16 for L in lengths),
           "prefill_plus_first_token_ms_p50": {"bf16x2 (default, parity grade)": a["ttft_ms_p50"], "fp8 (speed mode)": b["ttft_ms_p50"]},
           "ts_encode_ms_8192_patches": {"bf16x2": a["ts_encode_ms_8192_patches"], "fp8": b["ts_encode_ms_8192_patches"]},
           "first_token_logits_rel_diff_speed_vs_default": rel, "first_token_logits_max_abs_diff_over_max_logit": mab,
           "ts_features_rel_diff_speed_vs_default": ts_rel,
           "greedy_tokens_agreeing_before_first_difference": agree, "of": len(a["tokens"]),
           "note": "SPEED MODE, NOT the parity-grade line: activations are quantised per row to e4m3 (3 mantissa bits) for the prefill GEMMs and the "
                   "TS-encoder MLP; decode steps, attention, norms, KV cache identical in both modes; processor time not included"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r4_fp8_speed_mode.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
