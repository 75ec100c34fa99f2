#!/usr/bin/env python
"""LLM(model, tensor_parallel_size=2) called from ONE plain process - the reference's call shape (NetManAIOps/ChatTS
demo/demo_vllm.py:30) - spawns its follower rank itself; tokens must equal the TP=1 engine's and the oracle's.
On a single-GPU box: CHATTS_FORCE_DEVICE=0 CHATTS_DIST_BACKEND=gloo python tools/llm_tp_spawn_check.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from chatts_amd import LLM, SamplingParams
    rng = np.random.default_rng(1234)
    lengths = [64, 30]
    series = [(50 + 2 * np.cumsum(rng.standard_normal(L))).tolist() for L in lengths]
    prompt = ("<|im_start|>system\nYou are a helpful assistant.<|im_end|><|im_start|>user\nI have 2 time series. "
              "TS0 is of length 64: <ts><ts/>; TS1 is of length 30: <ts><ts/>; Please analyze.<|im_end|><|im_start|>assistant\n")
    reqs = [{"prompt": prompt, "multi_modal_data": {"timeseries": series}}] * 2
    sp = SamplingParams(max_tokens=8, ignore_eos=True)
    llm2 = LLM("tiny-qwen2", tensor_parallel_size=2, max_model_len=512, seed=3)
    spawned = llm2._tp_group is not None
    out2 = [o.outputs[0].token_ids for o in llm2.generate(reqs, sampling_params=sp)]
    out2b = [o.outputs[0].token_ids for o in llm2.generate(reqs[:1], sampling_params=SamplingParams(max_tokens=5, ignore_eos=True))]
    exch = llm2.model._tp is not None
    llm2.shutdown()
    llm1 = LLM("tiny-qwen2", tensor_parallel_size=1, max_model_len=512, seed=3)
    out1 = [o.outputs[0].token_ids for o in llm1.generate(reqs, sampling_params=sp)]
    from chatts_amd import config as cfgmod, synth
    from oracle import pipeline, synth as osynth
    cfg = cfgmod.preset("tiny-qwen2")
    sd = osynth.state_dict(synth.all_specs(cfg), 3)
    inputs = llm1.processor(text=[prompt], timeseries=[np.asarray(s) for s in series], return_tensors="pt")
    want = pipeline.generate(cfg, sd, inputs["input_ids"][0].tolist(), inputs["timeseries"].numpy(), 8)["tokens"]
    res = {"what": "LLM(tensor_parallel_size=2) from one plain process (spawned follower), two processes on one device",
           "spawned_followers": spawned, "p2p_exchange_attached": exch, "tokens_tp2": out2[0], "tokens_tp1": out1[0], "tokens_oracle": want,
           "second_call_prefix_ok": out2b[0] == want[:5],
           "passed": bool(spawned and out2[0] == out1[0] == want and out2[1] == want and out2b[0] == want[:5])}
    print(json.dumps(res))
    return 0 if res["passed"] else 1


if __name__ == "__main__":
    sys.exit(main())
