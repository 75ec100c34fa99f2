"""LoRA adapters merged at load (chatts_amd/lora.py): peft directory format, merge arithmetic, error cases (CPU);
end-to-end generation with a merged adapter against the oracle on the merged weights (GPU)."""
import json

import numpy as np
import pytest
import torch

from chatts_amd import config as cfgmod, lora, synth
from oracle import synth as osynth


def _write_adapter(path, shapes, r=4, alpha=8.0, seed=0, rslora=False, prefix="base_model.model.", default_tag=False):
    from safetensors.torch import save_file
    path.mkdir(parents=True, exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    tensors, mats = {}, {}
    for name, (out_f, in_f) in shapes.items():
        a = torch.randn((r, in_f), generator=g) * 0.05
        b = torch.randn((out_f, r), generator=g) * 0.05
        mod = name[:-len(".weight")]
        tag = ".default" if default_tag else ""
        tensors[f"{prefix}{mod}.lora_A{tag}.weight"] = a
        tensors[f"{prefix}{mod}.lora_B{tag}.weight"] = b
        mats[name] = (a, b)
    save_file(tensors, str(path / "adapter_model.safetensors"))
    (path / "adapter_config.json").write_text(json.dumps({"peft_type": "LORA", "r": r, "lora_alpha": alpha, "use_rslora": rslora,
                                                          "target_modules": sorted({n.split(".")[-2] for n in shapes})}))
    return mats


def test_merge_arithmetic_and_format(tmp_path):
    base = {"model.layers.0.self_attn.q_proj.weight": torch.randn(32, 16), "model.layers.0.mlp.down_proj.weight": torch.randn(16, 48),
            "model.norm.weight": torch.ones(16)}
    shapes = {k: tuple(v.shape) for k, v in base.items() if v.dim() == 2}
    mats = _write_adapter(tmp_path / "ad", shapes, r=4, alpha=8.0)
    out = dict(lora.merged(base.items(), str(tmp_path / "ad")))
    for k in shapes:
        a, b = mats[k]
        assert torch.allclose(out[k], base[k] + 2.0 * (b @ a), atol=1e-6)          # scale = alpha / r
    assert torch.equal(out["model.norm.weight"], base["model.norm.weight"])      # untouched tensors pass through
    mats = _write_adapter(tmp_path / "rs", shapes, r=4, alpha=8.0, rslora=True, default_tag=True)
    out = dict(lora.merged(base.items(), str(tmp_path / "rs")))
    k = "model.layers.0.self_attn.q_proj.weight"
    assert torch.allclose(out[k], base[k] + 4.0 * (mats[k][1] @ mats[k][0]), atol=1e-6)   # rslora: alpha / sqrt(r)


def test_merge_errors(tmp_path):
    base = {"model.layers.0.self_attn.q_proj.weight": torch.randn(32, 16)}
    _write_adapter(tmp_path / "missing", {"model.layers.9.self_attn.q_proj.weight": (32, 16)})
    with pytest.raises(ValueError, match="not in the checkpoint"):
        dict(lora.merged(base.items(), str(tmp_path / "missing")))
    _write_adapter(tmp_path / "shape", {"model.layers.0.self_attn.q_proj.weight": (32, 24)})
    with pytest.raises(ValueError, match="checkpoint shape"):
        dict(lora.merged(base.items(), str(tmp_path / "shape")))
    (tmp_path / "shape" / "adapter_config.json").write_text(json.dumps({"peft_type": "IA3", "r": 4}))
    with pytest.raises(ValueError, match="unsupported adapter type"):
        dict(lora.merged(base.items(), str(tmp_path / "shape")))


@pytest.mark.gpu
def test_generate_with_merged_lora_adapter_matches_oracle(tmp_path):
    """demo/demo_lora.ipynb flow: base checkpoint + adapter directory -> PeftModel.from_pretrained -> generate; the oracle
    runs on W + scale * B A rounded to bf16 (the engine's weight format)."""
    from safetensors.torch import save_file
    from chatts_amd import PeftModel
    from chatts_amd.modeling import ChatTSForCausalLM
    from chatts_amd.processing import ChatTSProcessor
    from oracle import pipeline
    from tests.util import chat_prompt, random_walk_series
    cfg = cfgmod.preset("tiny-qwen2")
    sd = osynth.state_dict(synth.all_specs(cfg), 11)
    ckpt = tmp_path / "ckpt"
    cfg.save_pretrained(str(ckpt))
    save_file({k: v.to(torch.bfloat16).contiguous() for k, v in sd.items()}, str(ckpt / "model.safetensors"))
    targets = {}
    for l in range(cfg.num_hidden_layers):
        for mod in ("self_attn.q_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.down_proj"):
            name = f"model.layers.{l}.{mod}.weight"
            targets[name] = tuple(sd[name].shape)
    targets["ts_encoder.mlp.2.weight"] = tuple(sd["ts_encoder.mlp.2.weight"].shape)       # the TS encoder can be adapted too
    mats = _write_adapter(tmp_path / "adapter", targets, r=8, alpha=64.0, seed=3)
    base = ChatTSForCausalLM.from_pretrained(str(ckpt), device_map="cuda:0", max_ctx=256, max_prefill_tokens=256)
    model = PeftModel.from_pretrained(base, str(tmp_path / "adapter"))
    merged_sd = dict(sd)
    for name, (a, b) in mats.items():
        merged_sd[name] = (sd[name].float() + 8.0 * (b @ a)).to(torch.bfloat16).float()
    proc = ChatTSProcessor.from_pretrained(str(ckpt))
    rng = np.random.default_rng(5)
    inputs = proc(text=[chat_prompt([40, 64])], timeseries=[random_walk_series(rng, 40), random_walk_series(rng, 64)], return_tensors="pt")
    ids = inputs["input_ids"][0].tolist()
    want = pipeline.generate(cfg, merged_sd, ids, inputs["timeseries"].numpy(), 8)
    plain = pipeline.generate(cfg, sd, ids, inputs["timeseries"].numpy(), 8)
    out = model.generate(**inputs.to("cuda"), max_new_tokens=8, eos_token_id=[])
    assert out[0, len(ids):].tolist() == want["tokens"]
    assert want["tokens"] != plain["tokens"]                                     # the adapter is strong enough to matter
    assert base.generate(**inputs.to("cuda"), max_new_tokens=8, eos_token_id=[])[0, len(ids):].tolist() == plain["tokens"]
