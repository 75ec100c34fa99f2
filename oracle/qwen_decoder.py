"""Oracle: Qwen2 / Qwen3 decoder, CPU float32.  TEST INFRASTRUCTURE ONLY.

The reference delegates the decoder to a third-party dependency that is NOT under /root/reference
(vLLM 0.8.5 Qwen2/Qwen3, selected at chatts_vllm.py:483-488 / :664-669; HF path: transformers==4.52.4,
requirements.txt:7).  This restates the published transformers algorithm:
  models/qwen2/modeling_qwen2.py  Qwen2RMSNorm.forward (fp32 variance, weight * x),
      Qwen2RotaryEmbedding (inv_freq = theta^(-2i/d), cat(freqs, freqs)), rotate_half,
      apply_rotary_pos_emb, repeat_kv, eager_attention_forward (scale d^-1/2, fp32 softmax),
      Qwen2Attention (q/k/v bias=True, o bias=False), Qwen2MLP (down(silu(gate(x)) * up(x))),
      Qwen2DecoderLayer (pre-norm, two residuals), Qwen2Model (final norm), lm_head.
  models/qwen3/modeling_qwen3.py  delta: q_norm / k_norm RMSNorm over head_dim before RoPE,
      attention_bias=False, explicit head_dim.
Pinned against stock transformers (5.15.0 in the build container) by tests/golden/make_golden.py.

Weights: dict name -> torch.float32 tensor with the HF checkpoint names
  model.embed_tokens.weight, model.layers.N.self_attn.{q,k,v,o}_proj.{weight,bias},
  model.layers.N.self_attn.{q,k}_norm.weight (qwen3), model.layers.N.mlp.{gate,up,down}_proj.weight,
  model.layers.N.{input,post_attention}_layernorm.weight, model.norm.weight, lm_head.weight
"""
import math

import torch


def rms_norm(x, w, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps))


def rope_cos_sin(positions, head_dim, theta):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = positions.float()[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


class QwenOracle:
    """Single-sequence decoder with a growing KV cache (lists of [n_kv, ctx, d] tensors)."""

    def __init__(self, cfg, weights, num_layers=None):
        self.c = cfg
        self.w = weights
        self.L = cfg["num_hidden_layers"] if num_layers is None else num_layers
        self.k = [None] * self.L
        self.v = [None] * self.L
        self.ctx = 0

    def reset(self):
        self.k = [None] * self.L
        self.v = [None] * self.L
        self.ctx = 0

    def embed(self, ids):
        return self.w["model.embed_tokens.weight"][torch.as_tensor(ids, dtype=torch.long)]

    def layer(self, l, x, positions):
        c, w = self.c, self.w
        p = f"model.layers.{l}."
        H, nq, nkv, d = c["hidden_size"], c["num_attention_heads"], c["num_key_value_heads"], c["head_dim"]
        eps = c["rms_norm_eps"]
        T = x.shape[0]
        h = rms_norm(x, w[p + "input_layernorm.weight"], eps)
        q = h @ w[p + "self_attn.q_proj.weight"].T
        k = h @ w[p + "self_attn.k_proj.weight"].T
        v = h @ w[p + "self_attn.v_proj.weight"].T
        if p + "self_attn.q_proj.bias" in w:
            q = q + w[p + "self_attn.q_proj.bias"]
            k = k + w[p + "self_attn.k_proj.bias"]
            v = v + w[p + "self_attn.v_proj.bias"]
        q = q.view(T, nq, d)
        k = k.view(T, nkv, d)
        v = v.view(T, nkv, d)
        if p + "self_attn.q_norm.weight" in w:
            q = rms_norm(q, w[p + "self_attn.q_norm.weight"], eps)
            k = rms_norm(k, w[p + "self_attn.k_norm.weight"], eps)
        cos, sin = rope_cos_sin(positions, d, c["rope_theta"])
        q = q * cos[:, None, :] + rotate_half(q) * sin[:, None, :]
        k = k * cos[:, None, :] + rotate_half(k) * sin[:, None, :]
        k = k.transpose(0, 1)
        v = v.transpose(0, 1)
        if self.k[l] is not None:
            k = torch.cat([self.k[l], k], dim=1)
            v = torch.cat([self.v[l], v], dim=1)
        self.k[l], self.v[l] = k, v
        S = k.shape[1]
        g = nq // nkv
        kk = k.repeat_interleave(g, dim=0)
        vv = v.repeat_interleave(g, dim=0)
        att = (q.transpose(0, 1) @ kk.transpose(1, 2)) * (1.0 / math.sqrt(d))     # [nq, T, S]
        kpos = torch.arange(S)[None, :]
        causal = kpos > positions[:, None]
        att = att.masked_fill(causal[None], float("-inf"))
        att = torch.softmax(att, dim=-1, dtype=torch.float32)
        o = (att @ vv).transpose(0, 1).reshape(T, nq * d)
        x = x + o @ w[p + "self_attn.o_proj.weight"].T
        h = rms_norm(x, w[p + "post_attention_layernorm.weight"], eps)
        gate = h @ w[p + "mlp.gate_proj.weight"].T
        up = h @ w[p + "mlp.up_proj.weight"].T
        x = x + (torch.nn.functional.silu(gate) * up) @ w[p + "mlp.down_proj.weight"].T
        return x

    @torch.no_grad()
    def forward_embeds(self, x, return_hidden=False, last_only=False):
        """x [T,H] appended at positions ctx..ctx+T-1 -> logits [T,V] (or final-norm hidden).
        last_only: final norm + lm_head on the last row only -> logits [1,V] (what compute_logits needs after a prefill)."""
        T = x.shape[0]
        positions = torch.arange(self.ctx, self.ctx + T)
        x = x.float()
        for l in range(self.L):
            x = self.layer(l, x, positions)
        self.ctx += T
        if last_only:
            x = x[-1:]
        h = rms_norm(x, self.w["model.norm.weight"], self.c["rms_norm_eps"])
        if return_hidden:
            return h
        return h @ self.w["lm_head.weight"].T

    @torch.no_grad()
    def greedy(self, prefill_embeds, max_new_tokens, eos_ids=(), forced_tokens=None):
        """Greedy decode after a prefill with embeddings -> (token ids, per-step last logits).
        forced_tokens (teacher forcing): follow this continuation instead of the argmax - the per-step logits are then
        the ones a sampler saw along that continuation."""
        logits = self.forward_embeds(prefill_embeds, last_only=True)[-1]
        toks, all_logits = [], [logits]
        for i in range(max_new_tokens):
            t = int(torch.argmax(logits)) if forced_tokens is None else int(forced_tokens[i])
            toks.append(t)
            if t in eos_ids or len(toks) == max_new_tokens:
                break
            logits = self.forward_embeds(self.embed([t]))[-1]
            all_logits.append(logits)
        return toks, all_logits
