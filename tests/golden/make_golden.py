"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and transformers); the GPU box never executes it.
    python tests/golden/make_golden.py

What is imported / executed from the reference (nothing is copied into the repo):
  * chatts.utils.encoding_utils (sp_encoding, eval_prompt_to_encoding)      - imported as a module
  * chatts/vllm/chatts_vllm.py class TimeSeriesEmbedding (:61-193)          - AST-sliced and exec'd
    (the module itself cannot be imported: it needs vLLM)
  * stock transformers Qwen2ForCausalLM / Qwen3ForCausalLM, CPU float32     - the decoder the reference
    delegates to (requirements.txt:7 pins 4.52.4; this container has 5.x, same math)
Outputs (small .npz files, committed):
  sp_encoding.npz, ts_embedding_{posemb,posidx,raw,single}.npz, qwen2_tiny.npz, qwen3_tiny.npz
"""
import ast
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_ts_embedding():
    src = open(os.path.join(REF, "chatts/vllm/chatts_vllm.py")).read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TimeSeriesEmbedding")
    code = ast.get_source_segment(src, node)
    ns = {"torch": torch, "nn": torch.nn}
    exec(compile(code, "chatts_vllm.py:TimeSeriesEmbedding", "exec"), ns)
    return ns["TimeSeriesEmbedding"]


def demo_series():
    x = np.arange(256)
    ts1 = np.sin(x / 10) * 5.0
    ts1[100:] -= 10.0
    ts2 = x * 0.05
    ts2[103] += 10.0
    return ts1, ts2


def gen_sp():
    sys.path.insert(0, REF)
    from chatts.utils import encoding_utils as eu
    rng = np.random.default_rng(1234)
    series = list(demo_series())
    for L in (1, 15, 16, 17, 64, 100, 1000):
        series.append(50 + 2 * np.cumsum(rng.standard_normal(L)))
    series.append(np.linspace(-1.0, 1.0, 33))          # no scaling branch (|x-mean| < 3)
    series.append(np.array([0.0, 3.0, -3.0, 0.0]))     # boundary: |dev| == 3 exactly
    out = {"n": len(series)}
    for i, s in enumerate(series):
        enc, prompt, meta = eu.sp_encoding(np.array(s))
        out[f"in_{i}"] = np.asarray(s, dtype=np.float64)
        out[f"enc_{i}"] = enc
        out[f"prompt_{i}"] = np.array(prompt)
        out[f"offset_{i}"] = meta["offset"]
        out[f"scale_{i}"] = meta["scale_factor"]
    prompt = "I have 3 time series. A: <ts><ts/>; B: <ts><ts/>; C: <ts><ts/>. Describe."
    tri = [series[0], series[6], series[3]]
    rp, arr = eu.eval_prompt_to_encoding(prompt, [t.tolist() for t in tri], "sp")
    out["batch_prompt_in"] = np.array(prompt)
    out["batch_prompt_out"] = np.array(rp)
    out["batch_arr"] = arr
    out["batch_idx"] = np.array([0, 6, 3])
    np.savez_compressed(os.path.join(OUT, "sp_encoding.npz"), **out)
    print("sp_encoding.npz", len(series), "series")


def gen_ts(name, cfg, lengths, seed):
    sys.path.insert(0, REF)
    from chatts.utils import encoding_utils as eu
    TSE = load_reference_ts_embedding()
    torch.manual_seed(seed)
    m = TSE(cfg).float().eval()
    rng = np.random.default_rng(seed)
    encs = []
    for L in lengths:
        if L == 0:
            encs.append(np.zeros((1, 0, 1)))
            continue
        enc, _, _ = eu.sp_encoding(50 + 2 * np.cumsum(rng.standard_normal(L)))
        encs.append(enc[None])
    lmax = max(e.shape[1] for e in encs)
    x = np.concatenate([np.pad(e, ((0, 0), (0, lmax - e.shape[1]), (0, 0))) for e in encs], axis=0)
    with torch.no_grad():
        feats, pc = m(torch.from_numpy(x).float())
    out = {"x": x.astype(np.float32), "features": feats.numpy(), "patch_cnt": pc.numpy(),
           "lengths": np.array(lengths), "config": np.array(json.dumps(cfg))}
    for k, v in m.state_dict().items():
        out["w:" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, f"ts_embedding_{name}.npz"), **out)
    print(f"ts_embedding_{name}.npz", x.shape, "->", tuple(feats.shape), pc.tolist())


def gen_qwen(kind, seed):
    from transformers import Qwen2Config, Qwen2ForCausalLM, Qwen3Config, Qwen3ForCausalLM
    common = dict(vocab_size=320, hidden_size=64, intermediate_size=160, num_hidden_layers=2, rope_theta=1e6,
                  rms_norm_eps=1e-6, tie_word_embeddings=False, max_position_embeddings=512)
    torch.manual_seed(seed)
    if kind == "qwen2":
        c = Qwen2Config(num_attention_heads=4, num_key_value_heads=2, **common)
        m = Qwen2ForCausalLM(c)
        head_dim = 16
    else:
        c = Qwen3Config(num_attention_heads=4, num_key_value_heads=2, head_dim=32, **common)
        m = Qwen3ForCausalLM(c)
        head_dim = 32
    m = m.float().eval()
    with torch.no_grad():      # make biases / norms non-trivial so every term is exercised
        for n, p in m.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.05)
            if "norm" in n:
                p.add_(torch.randn_like(p) * 0.1)
    T, new = 11, 6
    emb = torch.randn(1, T, 64) * 0.5
    with torch.no_grad():
        o = m(inputs_embeds=emb, use_cache=True)
        logits = [o.logits[0].numpy()]
        past = o.past_key_values
        toks = []
        cur = o.logits[0, -1]
        for _ in range(new):
            t = int(torch.argmax(cur))
            toks.append(t)
            o = m(input_ids=torch.tensor([[t]]), past_key_values=past, use_cache=True)
            past = o.past_key_values
            cur = o.logits[0, -1]
            logits.append(o.logits[0].numpy())
    cfg = dict(hidden_size=64, num_attention_heads=4, num_key_value_heads=2, head_dim=head_dim, rms_norm_eps=1e-6,
               rope_theta=1e6, num_hidden_layers=2, intermediate_size=160, vocab_size=320)
    out = {"config": np.array(json.dumps(cfg)), "embeds": emb[0].numpy(), "prefill_logits": logits[0],
           "decode_logits": np.concatenate(logits[1:], axis=0), "tokens": np.array(toks)}
    for k, v in m.state_dict().items():
        out["w:" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, f"{kind}_tiny.npz"), **out)
    print(f"{kind}_tiny.npz tokens", toks)


if __name__ == "__main__":
    torch.set_num_threads(4)
    gen_sp()
    base = dict(patch_size=16, num_layers=5, hidden_size=64, num_features=2, max_sequence_length=1024,
                use_position_embedding=True, embedding_dim=16)
    gen_ts("posemb", base, [256, 17, 0, 64, 1000, 1, 15, 16, 1024], 1)
    gen_ts("single", dict(base, num_layers=1, hidden_size=32), [40, 33], 2)
    gen_ts("posidx", dict(patch_size=16, num_layers=3, hidden_size=64, num_features=2, max_sequence_length=1024,
                          use_position_idx=True), [96, 64, 0, 16], 3)   # ragged tails crash the reference here too
    # raw mode: the reference only survives lengths that are multiples of 16 here (AttributeError otherwise)
    gen_ts("raw", dict(patch_size=16, num_layers=2, hidden_size=32, num_features=2, max_sequence_length=1024),
           [64, 32, 0, 16], 4)
    gen_qwen("qwen2", 5)
    gen_qwen("qwen3", 6)
