"""Model configuration: the HF ``config.json`` keys the reference reads.

Keys and their reference call sites (NetManAIOps/ChatTS):
  ts{patch_size,num_layers,hidden_size,num_features,max_sequence_length,use_position_embedding,
     use_position_idx,embedding_dim}            chatts/vllm/chatts_vllm.py:64-71
  ts_token_start_index / ts_token_end_index    chatts_vllm.py:376,441,573
  decoder keys                                  Qwen2Config / Qwen3Config (transformers, NOT IN REFERENCE)
``model.config.ts['patch_size']`` must stay readable (chatts/utils/inference_tsmllm_deepspeed.py:86).
"""
import copy
import json
import os

# Qwen2.5 tokenizer ids; ChatTS appends <ts>, <ts/> to the vocabulary as adjacent ids (chatts_vllm.py:441)
IM_START_ID, IM_END_ID, EOS_ID = 151644, 151645, 151643
TS_START_ID, TS_END_ID = 151665, 151666

_DEFAULT_TS = dict(patch_size=16, num_layers=5, hidden_size=None, num_features=2, max_sequence_length=8192,
                   use_position_embedding=True, use_position_idx=False, embedding_dim=16)


# The trust_remote_code entry points save_pretrained() drops next to config.json: transformers copies them into its module
# cache and imports them, so they only re-export the installed package (which must be importable: PYTHONPATH / pip -e).
REMOTE_CODE = {
    "configuration_chatts_amd.py": (
        '"""AutoConfig entry point of the MI355X-native ChatTS engine (chatts_amd)."""\n'
        "from transformers import PretrainedConfig\n\n\n"
        "class ChatTSAmdConfig(PretrainedConfig):\n"
        '    model_type = "chatts_amd"\n\n'
        "    def __init__(self, ts=None, **kw):\n"
        "        self.ts = ts or {}            # model.config.ts['patch_size'] stays readable (inference_tsmllm_deepspeed.py:86)\n"
        "        super().__init__(**kw)\n"),
    "modeling_chatts_amd.py": (
        '"""AutoModelForCausalLM entry point: the hand-written HIP engine behind the reference\'s HF surface."""\n'
        "from chatts_amd.modeling import ChatTSForCausalLM  # noqa: F401\n"),
    "processing_chatts_amd.py": (
        '"""AutoProcessor entry point: sp encoding + <ts><ts/> splicing on the host (chatts_amd.processing)."""\n'
        "from chatts_amd.processing import ChatTSProcessor  # noqa: F401\n"),
}


class ChatTSConfig:
    model_type_default = "qwen2"

    def __init__(self, **kw):
        kw.pop("auto_map", None); kw.pop("architectures", None)
        decoder_type = kw.pop("decoder_type", None)               # written by save_pretrained next to model_type "chatts_amd"
        self.model_type = kw.pop("model_type", "qwen2")           # "qwen2" (bias, no qk-norm) | "qwen3"
        if self.model_type == "chatts_amd":
            self.model_type = decoder_type or "qwen2"
        if self.model_type in ("chatts", "qwen2_ts"):
            self.model_type = "qwen2"
        if self.model_type == "qwen3_ts":
            self.model_type = "qwen3"
        self.vocab_size = kw.pop("vocab_size", 152064)
        self.hidden_size = kw.pop("hidden_size", 5120)
        self.intermediate_size = kw.pop("intermediate_size", 13824)
        self.num_hidden_layers = kw.pop("num_hidden_layers", 48)
        self.num_attention_heads = kw.pop("num_attention_heads", 40)
        self.num_key_value_heads = kw.pop("num_key_value_heads", 8)
        self.head_dim = kw.pop("head_dim", None) or self.hidden_size // self.num_attention_heads
        self.rope_theta = float(kw.pop("rope_theta", 1e6))
        self.rms_norm_eps = float(kw.pop("rms_norm_eps", 1e-6))
        self.max_position_embeddings = kw.pop("max_position_embeddings", 32768)
        self.tie_word_embeddings = bool(kw.pop("tie_word_embeddings", False))
        self.attention_bias = bool(kw.pop("attention_bias", self.model_type == "qwen2"))
        self.qk_norm = bool(kw.pop("qk_norm", self.model_type == "qwen3"))
        raw_ts = kw.pop("ts", {}) or {}
        ts = dict(_DEFAULT_TS)
        ts.update(raw_ts)
        if "max_length" in raw_ts and "max_sequence_length" not in raw_ts:
            ts["max_sequence_length"] = raw_ts["max_length"]       # chatts_vllm.py:245 accepts either key, prefers this one
        if ts.get("hidden_size") is None:
            ts["hidden_size"] = self.hidden_size
        self.ts = ts
        self.ts_token_start_index = kw.pop("ts_token_start_index", TS_START_ID)
        self.ts_token_end_index = kw.pop("ts_token_end_index", self.ts_token_start_index + 1)
        eos = kw.pop("eos_token_id", [IM_END_ID, EOS_ID])           # Qwen configs store a scalar or a list
        self.eos_token_id = [] if eos is None else ([int(e) for e in eos] if isinstance(eos, (list, tuple)) else [int(eos)])
        self.pad_token_id = kw.pop("pad_token_id", EOS_ID)
        self.torch_dtype = kw.pop("torch_dtype", "bfloat16")
        self.name = kw.pop("name", "custom")
        self.extra = kw

    # --- HF-style helpers -------------------------------------------------------------------------
    def to_dict(self):
        d = {k: copy.deepcopy(v) for k, v in self.__dict__.items() if k != "extra"}
        d.update(copy.deepcopy(self.extra))       # unknown keys round-trip (e.g. im_start_token_id of the tiny presets)
        return d

    @classmethod
    def from_dict(cls, d):
        return cls(**copy.deepcopy(d))

    @classmethod
    def from_pretrained(cls, path):
        with open(os.path.join(path, "config.json")) as f:
            return cls.from_dict(json.load(f))

    def save_pretrained(self, path, remote_code=True):
        """config.json (+ the trust_remote_code entry points).  With remote_code the directory also gets three one-line modules
        and an `auto_map`, so that the reference's UNMODIFIED loading cell (README.md:88-90)
            AutoModelForCausalLM.from_pretrained(path, trust_remote_code=True, device_map=0, torch_dtype='float16')
            AutoProcessor.from_pretrained(path, trust_remote_code=True, tokenizer=tokenizer)
        resolves to this engine (the published checkpoints point their auto_map at the HF-hub PyTorch implementation)."""
        os.makedirs(path, exist_ok=True)
        d = self.to_dict()
        if remote_code:
            d["model_type"] = "chatts_amd"
            d["decoder_type"] = self.model_type
            d["architectures"] = ["ChatTSForCausalLM"]
            d["auto_map"] = {"AutoConfig": "configuration_chatts_amd.ChatTSAmdConfig",
                             "AutoModelForCausalLM": "modeling_chatts_amd.ChatTSForCausalLM",
                             "AutoProcessor": "processing_chatts_amd.ChatTSProcessor"}
            for name, body in REMOTE_CODE.items():
                with open(os.path.join(path, name), "w") as f:
                    f.write(body)
            with open(os.path.join(path, "processor_config.json"), "w") as f:
                json.dump({"processor_class": "ChatTSProcessor",
                           "auto_map": {"AutoProcessor": "processing_chatts_amd.ChatTSProcessor"}}, f, indent=1)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(d, f, indent=1)

    def oracle_dict(self):
        """The keys oracle/qwen_decoder.py reads."""
        return dict(hidden_size=self.hidden_size, num_attention_heads=self.num_attention_heads,
                    num_key_value_heads=self.num_key_value_heads, head_dim=self.head_dim,
                    rms_norm_eps=self.rms_norm_eps, rope_theta=self.rope_theta,
                    num_hidden_layers=self.num_hidden_layers, intermediate_size=self.intermediate_size,
                    vocab_size=self.vocab_size)

    def param_counts(self):
        H, I, d = self.hidden_size, self.intermediate_size, self.head_dim
        nq, nkv = self.num_attention_heads, self.num_key_value_heads
        attn = H * nq * d + 2 * H * nkv * d + nq * d * H
        if self.attention_bias:
            attn += (nq + 2 * nkv) * d
        if self.qk_norm:
            attn += 2 * d
        layer = attn + 3 * H * I + 2 * H
        return dict(layer=layer, decoder=layer * self.num_hidden_layers + H, lm_head=self.vocab_size * H,
                    embed=self.vocab_size * H)

    def decode_weight_bytes(self):
        """bf16 bytes streamed per generated token at batch 1."""
        p = self.param_counts()
        return 2 * (p["decoder"] + p["lm_head"])


PRESETS = {
    # Qwen2.5-14B dims (public config; NOT IN REFERENCE) = ChatTS-14B decoder
    "chatts-14b": dict(model_type="qwen2", vocab_size=152064, hidden_size=5120, intermediate_size=13824,
                       num_hidden_layers=48, num_attention_heads=40, num_key_value_heads=8, head_dim=128),
    # Qwen3-8B dims = ChatTS-8B decoder (Qwen3TSForCausalLM, chatts_vllm.py:633-668)
    "chatts-8b": dict(model_type="qwen3", vocab_size=151936, hidden_size=4096, intermediate_size=12288,
                      num_hidden_layers=36, num_attention_heads=32, num_key_value_heads=8, head_dim=128),
    # small shapes for parity tests (head_dim stays 128: the attention kernels are specialised for it)
    # small vocabulary: specials are remapped to the top of it (4000.. ; <ts/> stays <ts>+1)
    "tiny-qwen2": dict(model_type="qwen2", vocab_size=4096, hidden_size=512, intermediate_size=1024,
                       num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                       ts_token_start_index=4000, eos_token_id=[4003, 4002], pad_token_id=4002,
                       im_start_token_id=4004, ts=dict(max_sequence_length=2048)),
    "tiny-qwen3": dict(model_type="qwen3", vocab_size=4096, hidden_size=384, intermediate_size=768,
                       num_hidden_layers=3, num_attention_heads=8, num_key_value_heads=2, head_dim=128,
                       ts_token_start_index=4000, eos_token_id=[4003, 4002], pad_token_id=4002,
                       im_start_token_id=4004, ts=dict(max_sequence_length=2048)),
}


def preset(name, **overrides):
    d = copy.deepcopy(PRESETS[name])
    ts = d.pop("ts", {})
    ts.update(overrides.pop("ts", {}) or {})
    d.update(overrides)
    return ChatTSConfig(name=name, ts=ts, **d)
