"""Synthetic ("random-init") checkpoints defined by a counter-based hash.

No checkpoint or tokenizer can reach the GPU box (no network, snapshot size limits), so weights are
DEFINED as a pure function of (seed, tensor name, element index):

    key   = mix32(seed * 0x9E3779B9 + fnv1a32(name))
    h(i)  = mix32(lo32(i) ^ mix32(key + hi32(i) * 0x85ebca6b))
    n(i)  = byte0 + byte1 + byte2 + byte3 - 510                     (Irwin-Hall, ~normal, |n| <= 510)
    w(i)  = bf16_rne(base + n(i) * 2^-shift)

The HIP kernel ``chatts_fill_hash`` evaluates it on the device, shard by shard; ``oracle/synth.py``
evaluates the same definition with numpy for the CPU oracle, so both arms hold bit-identical bf16
values without any file I/O.  This module only holds the *specification* (names, shapes, base, shift)
and the device-side materialisation.
"""
import numpy as np

M32 = 0xFFFFFFFF


def mix32(x):
    x &= M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


def fnv1a32(name):
    h = 0x811C9DC5
    for b in name.encode("utf-8"):
        h = ((h ^ b) * 0x01000193) & M32
    return h


def tensor_key(seed, name):
    return mix32((seed * 0x9E3779B9 + fnv1a32(name)) & M32)


class TensorSpec:
    """One HF-named checkpoint tensor: shape [rows, cols] (1-D tensors have rows = 1)."""
    __slots__ = ("name", "rows", "cols", "base", "shift", "is_1d")

    def __init__(self, name, rows, cols, base, shift, is_1d=False):
        self.name, self.rows, self.cols, self.base, self.shift, self.is_1d = name, rows, cols, base, shift, is_1d

    @property
    def shape(self):
        return (self.cols,) if self.is_1d else (self.rows, self.cols)


# std of n(i) is 147.8; shift 13 -> std 0.018 (HF initializer_range 0.02)
W_SHIFT, B_SHIFT, NORM_SHIFT, TS_SHIFT = 13, 15, 12, 13


def decoder_specs(cfg):
    """Every decoder tensor with its HF checkpoint name (checkpoint row)."""
    H, I, d = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    nq, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    s = [TensorSpec("model.embed_tokens.weight", cfg.vocab_size, H, 0.0, W_SHIFT)]
    for l in range(cfg.num_hidden_layers):
        p = f"model.layers.{l}."
        s += [TensorSpec(p + "self_attn.q_proj.weight", nq * d, H, 0.0, W_SHIFT),
              TensorSpec(p + "self_attn.k_proj.weight", nkv * d, H, 0.0, W_SHIFT),
              TensorSpec(p + "self_attn.v_proj.weight", nkv * d, H, 0.0, W_SHIFT),
              TensorSpec(p + "self_attn.o_proj.weight", H, nq * d, 0.0, W_SHIFT),
              TensorSpec(p + "mlp.gate_proj.weight", I, H, 0.0, W_SHIFT),
              TensorSpec(p + "mlp.up_proj.weight", I, H, 0.0, W_SHIFT),
              TensorSpec(p + "mlp.down_proj.weight", H, I, 0.0, W_SHIFT),
              TensorSpec(p + "input_layernorm.weight", 1, H, 1.0, NORM_SHIFT, True),
              TensorSpec(p + "post_attention_layernorm.weight", 1, H, 1.0, NORM_SHIFT, True)]
        if cfg.attention_bias:
            s += [TensorSpec(p + "self_attn.q_proj.bias", 1, nq * d, 0.0, B_SHIFT, True),
                  TensorSpec(p + "self_attn.k_proj.bias", 1, nkv * d, 0.0, B_SHIFT, True),
                  TensorSpec(p + "self_attn.v_proj.bias", 1, nkv * d, 0.0, B_SHIFT, True)]
        if cfg.qk_norm:
            s += [TensorSpec(p + "self_attn.q_norm.weight", 1, d, 1.0, NORM_SHIFT, True),
                  TensorSpec(p + "self_attn.k_norm.weight", 1, d, 1.0, NORM_SHIFT, True)]
    s.append(TensorSpec("model.norm.weight", 1, H, 1.0, NORM_SHIFT, True))
    if not cfg.tie_word_embeddings:
        s.append(TensorSpec("lm_head.weight", cfg.vocab_size, H, 0.0, W_SHIFT))
    return s


def ts_encoder_specs(cfg):
    """ts_encoder.* tensors: nn.Sequential indexing gives mlp.{0,2,4,...} (chatts_vllm.py:83-91)."""
    ts = cfg.ts
    ps, H, n = ts["patch_size"], ts["hidden_size"], ts["num_layers"]
    s = []
    if ts.get("use_position_embedding"):
        s.append(TensorSpec("ts_encoder.position_embedding.weight", ts["max_sequence_length"] + 1,
                            ts.get("embedding_dim", 16), 0.0, 8))       # nn.Embedding init ~ N(0,1): std 0.58
        k = ps + ps * ts.get("embedding_dim", 16)
    elif ts.get("use_position_idx"):
        k = 2 * ps
    else:
        k = ps
    for l in range(n):
        # first layer sees O(1)-magnitude inputs: larger weights keep activations O(1) through the GELUs
        shift = 11 if l == 0 else TS_SHIFT
        s.append(TensorSpec(f"ts_encoder.mlp.{2 * l}.weight", H, k, 0.0, shift))
        s.append(TensorSpec(f"ts_encoder.mlp.{2 * l}.bias", 1, H, 0.0, B_SHIFT, True))
        k = H
    return s


def all_specs(cfg):
    return ts_encoder_specs(cfg) + decoder_specs(cfg)


def fill_device(dst, spec, seed, row0=0, col0=0, rows=None, cols=None, ld=None):
    """Materialise a [rows, cols] block of ``spec`` into the torch tensor ``dst`` (bf16 or f32, on the GPU)."""
    import torch
    from . import _lib
    lib = _lib.load()
    rows = spec.rows - row0 if rows is None else rows
    cols = spec.cols - col0 if cols is None else cols
    ld = cols if ld is None else ld
    assert dst.is_cuda and dst.is_contiguous() and dst.numel() >= (rows - 1) * ld + cols
    assert dst.dtype in (torch.bfloat16, torch.float32)
    _lib.check(lib.chatts_fill_hash(dst.data_ptr(), int(dst.dtype == torch.float32), tensor_key(seed, spec.name),
                                    float(spec.base), int(spec.shift), rows, cols, ld, row0, col0, spec.cols,
                                    _lib.stream_ptr()))
    return dst
