"""LoRA adapters, merged at load time.

The reference's LoRA demo wraps the HF model with peft (`PeftModel.from_pretrained(base_model, LORA_ADAPTER_PATH)`,
demo/demo_lora.ipynb cells 2-3; peft itself is a third-party dependency, NOT IN REFERENCE).  A peft LoRA adapter directory
holds `adapter_config.json` (r, lora_alpha, use_rslora, target_modules) and `adapter_model.safetensors` with, per target
module, `base_model.model.<module>.lora_A.weight` [r, in] and `...lora_B.weight` [out, r]; the adapted layer computes
W x + (lora_alpha / r) B A x.  This engine runs merged weights only: W' = W + scale * B @ A (float32), rounded to the
engine's bf16 weight format like any other checkpoint tensor.  Merged inference is what `merge_and_unload()` gives in peft.
"""
import json
import math
import os

import torch


def read_adapter(adapter_dir):
    """-> (dict base-weight-name -> (A [r,in] f32, B [out,r] f32), scale)"""
    from safetensors import safe_open
    with open(os.path.join(adapter_dir, "adapter_config.json")) as f:
        cfg = json.load(f)
    if cfg.get("peft_type", "LORA").upper() != "LORA":
        raise ValueError(f"unsupported adapter type {cfg.get('peft_type')!r} (only LoRA can be merged)")
    r, alpha = int(cfg["r"]), float(cfg.get("lora_alpha", cfg["r"]))
    scale = alpha / math.sqrt(r) if cfg.get("use_rslora") else alpha / r
    path = os.path.join(adapter_dir, "adapter_model.safetensors")
    parts = {}
    with safe_open(path, framework="pt", device="cpu") as h:
        for key in h.keys():
            name = key
            for prefix in ("base_model.model.", "base_model."):
                if name.startswith(prefix):
                    name = name[len(prefix):]
                    break
            for tag, slot in ((".lora_A.weight", 0), (".lora_B.weight", 1), (".lora_A.default.weight", 0), (".lora_B.default.weight", 1)):
                if name.endswith(tag):
                    parts.setdefault(name[:-len(tag)] + ".weight", [None, None])[slot] = h.get_tensor(key).float()
                    break
            else:
                raise ValueError(f"adapter tensor {key!r} is not a LoRA A/B matrix (modules_to_save etc. are not supported)")
    for name, (a, b) in parts.items():
        if a is None or b is None:
            raise ValueError(f"adapter is missing the {'A' if a is None else 'B'} matrix of {name}")
        if a.shape[0] != r or b.shape[1] != r:
            raise ValueError(f"{name}: LoRA shapes {tuple(a.shape)} / {tuple(b.shape)} do not match r={r}")
    return {k: (a, b) for k, (a, b) in parts.items()}, scale


def merged(weights, adapter_dir):
    """Wrap an iterable of (name, tensor) checkpoint pairs: tensors the adapter targets come out as W + scale * B @ A
    (float32).  Raises if the adapter targets a module the checkpoint does not contain."""
    deltas, scale = read_adapter(adapter_dir)
    seen = set()
    for name, t in weights:
        if name in deltas:
            a, b = deltas[name]
            if t.shape != (b.shape[0], a.shape[1]):
                raise ValueError(f"{name}: checkpoint shape {tuple(t.shape)} vs adapter {b.shape[0]}x{a.shape[1]}")
            t = t.float() + scale * (b @ a)
            seen.add(name)
        yield name, t
    missing = sorted(set(deltas) - seen)
    if missing:
        raise ValueError(f"the adapter targets modules that are not in the checkpoint: {missing[:4]}{' ...' if len(missing) > 4 else ''}")


class PeftModel:
    """`PeftModel.from_pretrained(base_model, adapter_dir)` of the reference's notebook: returns a ChatTSForCausalLM whose
    weights are the base checkpoint with the adapter merged (the base model must have come from `from_pretrained`)."""

    @staticmethod
    def from_pretrained(base_model, adapter_dir, **kw):
        src = getattr(base_model, "_checkpoint_path", None)
        if src is None:
            raise ValueError("PeftModel.from_pretrained needs a base model loaded with ChatTSForCausalLM.from_pretrained")
        return type(base_model).from_pretrained(src, lora_adapter=adapter_dir, **base_model._checkpoint_kw)
