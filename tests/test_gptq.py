"""GPTQ-Int4 checkpoint handling (chatts_amd/gptq.py): the AutoGPTQ tensor layout of the published ChatTS-14B-GPTQ-Int4
(NetManAIOps/ChatTS README.md:52,262-263) is unpacked to the nn.Linear weight the engine runs on.  CPU part: the vectorised
dequantiser against a scalar restatement of the published formula, both checkpoint formats, act-order g_idx, the
streaming (name, tensor) transformation of load_weights.  GPU part: a quantised checkpoint on disk -> from_pretrained ->
greedy tokens equal to the oracle's on the dequantised weights."""
import numpy as np
import pytest
import torch

from chatts_amd import gptq


def _scalar_dequant(qweight, qzeros, scales, g_idx, v2):
    qw, qz, sc = qweight.numpy().view(np.uint32), qzeros.numpy().view(np.uint32), scales.float().numpy()
    K, N = qw.shape[0] * 8, qw.shape[1]
    w = np.zeros((N, K), dtype=np.float32)
    for k in range(K):
        g = int(g_idx[k])
        for n in range(N):
            q = (int(qw[k // 8, n]) >> (4 * (k % 8))) & 15
            z = ((int(qz[g, n // 8]) >> (4 * (n % 8))) & 15) + (0 if v2 else 1)
            w[n, k] = sc[g, n] * np.float32(q - z)
    return w


@pytest.mark.parametrize("v2", [False, True])
def test_dequantize_matches_the_published_formula(v2):
    g = torch.Generator().manual_seed(1)
    K, N, gs = 64, 16, 32
    qweight = torch.randint(-2**31, 2**31 - 1, (K // 8, N), generator=g, dtype=torch.int64).to(torch.int32)
    qzeros = torch.randint(-2**31, 2**31 - 1, (K // gs, N // 8), generator=g, dtype=torch.int64).to(torch.int32)
    scales = (torch.rand((K // gs, N), generator=g) * 0.02 + 0.001).to(torch.float16)
    g_idx = torch.arange(K) // gs
    got = gptq.dequantize(qweight, qzeros, scales, g_idx, gs, v2=v2)
    assert got.shape == (N, K) and np.array_equal(got.numpy(), _scalar_dequant(qweight, qzeros, scales, g_idx, v2))
    # act-order: an arbitrary channel -> group map
    perm = torch.randperm(K, generator=g) % (K // gs)
    got = gptq.dequantize(qweight, qzeros, scales, perm, gs, v2=v2)
    assert np.array_equal(got.numpy(), _scalar_dequant(qweight, qzeros, scales, perm, v2))


def test_quantize_then_dequantize_and_pair_stream():
    g = torch.Generator().manual_seed(2)
    w = torch.randn((32, 256), generator=g) * 0.02
    t = gptq.quantize_rows(w, 128)
    assert t["qweight"].shape == (32, 32) and t["qzeros"].shape == (2, 4) and t["scales"].shape == (2, 32)
    deq = gptq.dequantize(t["qweight"], t["qzeros"], t["scales"], t["g_idx"], 128)
    step = t["scales"].float().t().repeat_interleave(128, dim=1)
    assert (deq - w).abs().max() <= 0.5 * step.max() * 1.01 + 1e-6            # round-to-nearest: within half a step
    pairs = [("model.norm.weight", torch.ones(4))] + [(f"model.layers.0.mlp.up_proj.{k}", v) for k, v in t.items()] + \
            [("model.layers.0.mlp.up_proj.bias", torch.zeros(32))]
    out = dict(gptq.dequantized_pairs(iter(pairs), {"bits": 4, "group_size": 128, "quant_method": "gptq"}))
    assert set(out) == {"model.norm.weight", "model.layers.0.mlp.up_proj.weight", "model.layers.0.mlp.up_proj.bias",
                        "model.layers.0.mlp.up_proj.gptq_codes"}
    assert out["model.layers.0.mlp.up_proj.weight"].dtype == torch.bfloat16
    assert torch.equal(out["model.layers.0.mlp.up_proj.weight"], deq.to(torch.bfloat16))
    # the codes that travel to the int4 decode GEMV describe the same matrix: scale * (code - zero), [N, K] orientation
    q, sc, z = out["model.layers.0.mlp.up_proj.gptq_codes"]
    assert q.shape == (32, 256) and q.dtype == torch.uint8 and sc.shape == z.shape == (32, 2)
    rebuilt = sc.repeat_interleave(128, 1) * (q.float() - z.repeat_interleave(128, 1))
    assert torch.equal(rebuilt, deq)
    # act-order checkpoints (g_idx not sequential) only yield the dequantised matrix
    shuffled = dict(t)
    shuffled["g_idx"] = t["g_idx"].flip(0)
    pairs2 = [(f"m.{k}", v) for k, v in shuffled.items()]
    assert set(dict(gptq.dequantized_pairs(iter(pairs2), {"bits": 4, "group_size": 128, "desc_act": True}))) == {"m.weight"}
    # a checkpoint WITHOUT g_idx tensors (desc_act = false) is emitted by the final flush - with its codes, like the others
    no_gidx = [(f"m.{k}", v) for k, v in t.items() if k != "g_idx"]
    out3 = dict(gptq.dequantized_pairs(iter(no_gidx), {"bits": 4, "group_size": 128, "desc_act": False}))
    assert set(out3) == {"m.weight", "m.gptq_codes"} and torch.equal(out3["m.weight"], deq.to(torch.bfloat16))
    assert all(torch.equal(a, b) for a, b in zip(out3["m.gptq_codes"], out["model.layers.0.mlp.up_proj.gptq_codes"]))
    # ... while an act-order checkpoint that misses a g_idx is an error, not a guess
    with pytest.raises(ValueError, match="incomplete"):
        list(gptq.dequantized_pairs(iter(no_gidx), {"bits": 4, "group_size": 128, "desc_act": True}))
    # per-column groups (group_size = -1: one group of K per row, K differs per module): no codes, the bf16 matrix is streamed
    K = 256
    tc = gptq.quantize_rows(w, K)
    outc = dict(gptq.dequantized_pairs(iter([(f"m.{k}", v) for k, v in tc.items()]), {"bits": 4, "group_size": -1}))
    assert set(outc) == {"m.weight"}
    assert torch.equal(outc["m.weight"], gptq.dequantize(tc["qweight"], tc["qzeros"], tc["scales"], tc["g_idx"], K).to(torch.bfloat16))
    with pytest.raises(ValueError, match="bits"):
        list(gptq.dequantized_pairs(iter(pairs), {"bits": 8}))


@pytest.mark.gpu
def test_gptq_checkpoint_generates_like_the_oracle(tmp_path):
    from safetensors.torch import save_file
    from chatts_amd import config as cfgmod, synth
    from chatts_amd.modeling import ChatTSForCausalLM
    from chatts_amd.processing import ChatTSProcessor
    from oracle import pipeline, synth as osynth
    from tests.util import chat_prompt, random_walk_series
    cfg = cfgmod.preset("tiny-qwen2")
    sd = osynth.state_dict(synth.all_specs(cfg), 13)
    ckpt = tmp_path / "gptq"
    cfg.extra["quantization_config"] = {"quant_method": "gptq", "bits": 4, "group_size": 128, "desc_act": False, "sym": False}
    cfg.save_pretrained(str(ckpt))
    tensors, deq_sd = {}, dict(sd)
    for name, t in sd.items():
        if name.endswith("_proj.weight"):                      # every decoder projection, like the published checkpoint
            q = gptq.quantize_rows(t, 128)
            base = name[:-len(".weight")]
            for k, v in q.items():
                tensors[f"{base}.{k}"] = v.contiguous()
            deq_sd[name] = gptq.dequantize(q["qweight"], q["qzeros"], q["scales"], q["g_idx"], 128).to(torch.bfloat16).float()
        else:
            tensors[name] = t.to(torch.bfloat16).contiguous()
    save_file(tensors, str(ckpt / "model.safetensors"))
    model = ChatTSForCausalLM.from_pretrained(str(ckpt), device_map="cuda:0", max_ctx=256, max_prefill_tokens=256)
    assert torch.equal(model.layers[1]["down"].float().cpu(), deq_sd["model.layers.1.mlp.down_proj.weight"])
    # the checkpoint's own 4-bit codes were packed for the decode GEMV (q|k|v fused, gate/up interleaved): decode streams them
    for name in ("qkv4", "o4", "gate_up4", "down4"):
        assert name in model.layers[0] and model.layers[0][name].dtype == torch.uint8
    assert model.layers[0]["down4"].shape == (cfg.hidden_size, cfg.intermediate_size // 2)
    bf16_bytes = sum(model.layers[0][k].numel() * 2 for k in ("qkv", "o", "gate_up", "down"))
    assert model.weight_bytes_local() < 0.45 * (bf16_bytes * cfg.num_hidden_layers) + model._tensors["lm_head"].numel() * 2 + 1e5
    proc = ChatTSProcessor.from_pretrained(str(ckpt))
    rng = np.random.default_rng(4)
    series = [random_walk_series(rng, 48)]
    inputs = proc(text=[chat_prompt([48])], timeseries=series, return_tensors="pt")
    ids = inputs["input_ids"][0].tolist()
    want = pipeline.generate(cfg, deq_sd, ids, inputs["timeseries"].numpy(), 8)
    out = model.generate(**inputs.to("cuda"), max_new_tokens=8, eos_token_id=[])
    assert out[0, len(ids):].tolist() == want["tokens"]
