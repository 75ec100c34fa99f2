"""The optional precision="bf16" speed mode next to the parity-grade default, same process, same 14B synthetic model and bench prompt:
TTFT (p50 of a few runs each), first-token logits of one mode against the other (the default is within 5e-5 of the float32 oracle:
profiles/r2_parity_14b_full.json), and how many of the first greedy tokens agree.  One JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from chatts_amd import _lib, config as cfgmod  # noqa: E402
from chatts_amd.modeling import ChatTSForCausalLM  # noqa: E402


def main():
    model_name = sys.argv[1] if len(sys.argv) > 1 else "chatts-14b"
    n_new, runs = 16, 4
    cfg = cfgmod.preset(model_name)
    proc, prompt, series, lengths = bench.build_inputs(cfg, 8, 256)
    m = ChatTSForCausalLM.from_synthetic(cfg, seed=0, max_ctx=2048, max_prefill_tokens=1024, enable_prefix_caching=False)
    inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
    ids, ser = inputs["input_ids"][0].tolist(), inputs["timeseries"]
    out = {}
    for mode in ("bf16x2", "bf16", "bf16x2"):
        if mode == "bf16":
            os.environ["CHATTS_GEMM_PRECISION"] = "bf16"
            _lib.sync_env()
        else:
            os.environ.pop("CHATTS_GEMM_PRECISION", None)
            _lib.sync_env()
        ttft = []
        for _ in range(runs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m._prefill_request(ids, ser.to("cuda"), list(proc.last_lengths), n_new)
            first = m.buf["out_tokens"][:1].tolist()
            ttft.append((time.perf_counter() - t0) * 1e3)
        toks, lg = m.generate_one(ids, ser, list(proc.last_lengths), n_new, return_logits=True)
        out.setdefault(mode, []).append({"ttft_ms_p50": sorted(ttft[1:])[len(ttft[1:]) // 2], "tokens": toks, "logits": lg.double().cpu()})
    a, b, a2 = out["bf16x2"][0], out["bf16"][0], out["bf16x2"][1]
    rel = float((b["logits"] - a["logits"]).norm() / a["logits"].norm())
    agree = next((i for i, (x, y) in enumerate(zip(a["tokens"], b["tokens"])) if x != y), len(a["tokens"]))
    print(json.dumps({"model": model_name, "prompt_tokens": m.request_tokens(ids, ser, list(proc.last_lengths)),
                      "ttft_ms_p50": {"bf16x2 (default, parity grade)": a["ttft_ms_p50"], "bf16 (speed mode)": b["ttft_ms_p50"],
                                      "bf16x2 again": a2["ttft_ms_p50"]},
                      "first_token_logits_rel_diff_speed_vs_default": rel,
                      "default_mode_is_reproducible": a["tokens"] == a2["tokens"] and bool(torch.equal(a["logits"], a2["logits"])),
                      "greedy_tokens_agreeing_before_first_difference": agree, "of": len(a["tokens"]),
                      "note": "processor time is not in these TTFTs (inputs prepared once); decode kernels are the same in both modes"}))


if __name__ == "__main__":
    main()
