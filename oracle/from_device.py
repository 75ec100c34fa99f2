"""Oracle weights read back from a materialised model.  TEST INFRASTRUCTURE ONLY.

The synthetic checkpoints are DEFINED by a hash (oracle/synth.py evaluates it on the host, bit-identically to the HIP
fill kernel; tests spot-check blocks of the device tensors against it).  Evaluating 14 G elements with numpy takes
minutes, so the large-config parity runs copy the device tensors back instead and UN-PACK them into the HF checkpoint
names oracle/qwen_decoder.py reads - which also exercises the packing conventions of load_weights
(NetManAIOps/ChatTS chatts/vllm/chatts_vllm.py:454-470,612-625: q|k|v fused, gate/up fused).
layer_tensors / decoder_state_dict work on a tensor_parallel_size=1 model (a rank-local shard is not a checkpoint);
sharded_state_dict re-assembles the checkpoint from ALL W rank-local shards of a tensor-parallel group (ShardPlan's contiguous head /
intermediate / vocabulary ranges: chatts_amd/tp.py) - for 8-bit weight formats that is the only way to know the weights the shards
hold, since every shard picks its row scales over its own K-slice.
"""
import torch


def head_tensors(model, embed_rows=None):
    T = model._tensors
    emb = T["embed"] if embed_rows is None else T["embed"][:embed_rows]
    return {"model.embed_tokens.weight": emb.float().cpu(), "lm_head.weight": T["lm_head"].float().cpu(),
            "model.norm.weight": T["final_norm"].float().cpu()}


def layer_tensors(model, l, dtype=torch.float32):
    """HF-named tensors of decoder layer l (dtype float32, or bfloat16 to halve host memory: upcast before use)."""
    cfg, lw = model.config, model.layers[l]
    assert model.plan.world == 1, "un-packing needs the full (TP=1) weights"
    d, nq, nkv, I = cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size
    p = f"model.layers.{l}."
    out = {}
    qkv = lw["qkv"].cpu().to(dtype)
    out[p + "self_attn.q_proj.weight"] = qkv[:nq * d]
    out[p + "self_attn.k_proj.weight"] = qkv[nq * d:(nq + nkv) * d]
    out[p + "self_attn.v_proj.weight"] = qkv[(nq + nkv) * d:]
    if "qkv_bias" in lw:
        b = lw["qkv_bias"].float().cpu()
        out[p + "self_attn.q_proj.bias"], out[p + "self_attn.k_proj.bias"], out[p + "self_attn.v_proj.bias"] = \
            b[:nq * d], b[nq * d:(nq + nkv) * d], b[(nq + nkv) * d:]
    if "q_norm" in lw:
        out[p + "self_attn.q_norm.weight"], out[p + "self_attn.k_norm.weight"] = lw["q_norm"].float().cpu(), lw["k_norm"].float().cpu()
    out[p + "self_attn.o_proj.weight"] = lw["o"].cpu().to(dtype)
    gu = lw["gate_up"].cpu().to(dtype).view(I // 16, 2, 16, -1)          # gate/up interleaved in blocks of 16 rows
    out[p + "mlp.gate_proj.weight"] = gu[:, 0].reshape(I, -1).contiguous()
    out[p + "mlp.up_proj.weight"] = gu[:, 1].reshape(I, -1).contiguous()
    out[p + "mlp.down_proj.weight"] = lw["down"].cpu().to(dtype)
    out[p + "input_layernorm.weight"] = lw["input_norm"].float().cpu()
    out[p + "post_attention_layernorm.weight"] = lw["post_norm"].float().cpu()
    return out


def decoder_state_dict(model, num_layers=None, embed_rows=None):
    sd = head_tensors(model, embed_rows)
    for l in range(model.config.num_hidden_layers if num_layers is None else num_layers):
        sd.update(layer_tensors(model, l))
    return sd


def sharded_state_dict(models, num_layers=None):
    """HF-named float32 tensors of the WHOLE decoder from the W rank-local shard models of one tensor-parallel group (rank order)."""
    m0 = models[0]
    cfg = m0.config
    assert [m.plan.rank for m in models] == list(range(m0.plan.world)), "pass every rank's model, in rank order"
    d = cfg.head_dim
    sd = {"model.embed_tokens.weight": m0._tensors["embed"].float().cpu(),             # replicated
          "lm_head.weight": torch.cat([m._tensors["lm_head"].float().cpu() for m in models], dim=0),      # vocab-parallel
          "model.norm.weight": m0._tensors["final_norm"].float().cpu()}
    for l in range(cfg.num_hidden_layers if num_layers is None else num_layers):
        p = f"model.layers.{l}."
        q, k, v, o, g, u, dn, bq, bk, bv = [], [], [], [], [], [], [], [], [], []
        for m in models:
            lw, nq, nkv, I = m.layers[l], m.plan.nq, m.plan.nkv, m.plan.inter
            qkv = lw["qkv"].float().cpu()
            q.append(qkv[:nq * d]); k.append(qkv[nq * d:(nq + nkv) * d]); v.append(qkv[(nq + nkv) * d:])
            if "qkv_bias" in lw:
                b = lw["qkv_bias"].float().cpu()
                bq.append(b[:nq * d]); bk.append(b[nq * d:(nq + nkv) * d]); bv.append(b[(nq + nkv) * d:])
            o.append(lw["o"].float().cpu())                                  # column (K) slice
            gu = lw["gate_up"].float().cpu().view(I // 16, 2, 16, -1)
            g.append(gu[:, 0].reshape(I, -1)); u.append(gu[:, 1].reshape(I, -1))
            dn.append(lw["down"].float().cpu())                              # column (K) slice
        sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"] = \
            torch.cat(q), torch.cat(k), torch.cat(v)
        if bq:
            sd[p + "self_attn.q_proj.bias"], sd[p + "self_attn.k_proj.bias"], sd[p + "self_attn.v_proj.bias"] = \
                torch.cat(bq), torch.cat(bk), torch.cat(bv)
        lw0 = m0.layers[l]
        if "q_norm" in lw0:
            sd[p + "self_attn.q_norm.weight"], sd[p + "self_attn.k_norm.weight"] = lw0["q_norm"].float().cpu(), lw0["k_norm"].float().cpu()
        sd[p + "self_attn.o_proj.weight"] = torch.cat(o, dim=1).contiguous()
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = torch.cat(g).contiguous(), torch.cat(u).contiguous()
        sd[p + "mlp.down_proj.weight"] = torch.cat(dn, dim=1).contiguous()
        sd[p + "input_layernorm.weight"] = lw0["input_norm"].float().cpu()
        sd[p + "post_attention_layernorm.weight"] = lw0["post_norm"].float().cpu()
    return sd


def ts_encoder_state_dict(model):
    """'ts_encoder.*' tensors (numpy float32) with the K padding of layer 0 removed."""
    enc = model.ts_encoder
    sd = {}
    if enc.position_embedding is not None:
        sd["ts_encoder.position_embedding.weight"] = enc.position_embedding.float().cpu().numpy()
    for l in range(enc.num_layers):
        sd[f"ts_encoder.mlp.{2 * l}.weight"] = enc.weights[l][:, :enc.layer_in_features(l)].float().cpu().numpy()
        sd[f"ts_encoder.mlp.{2 * l}.bias"] = enc.biases[l].float().cpu().numpy()
    return sd


class LayerStreamedWeights:
    """dict-like weight store for QwenOracle that holds ONE layer in float32 at a time (full-depth 14B = 53 GB in f32).

    keep="bf16": all layers are copied to the host once as bf16 (26 GB for 14B) and widened layer by layer;
    keep="f32" : everything is kept widened (fastest, needs ~2x that);
    keep="none": every access of a new layer copies it back from the device again."""

    def __init__(self, model, keep="bf16", embed_rows=None):
        self.model, self.keep = model, keep
        self.head = head_tensors(model, embed_rows)
        self.L = model.config.num_hidden_layers
        self.store = {}
        if keep in ("bf16", "f32"):
            for l in range(self.L):
                self.store[l] = layer_tensors(model, l, torch.float32 if keep == "f32" else torch.bfloat16)
        self.cur, self.cur_l = None, -1

    def _layer(self, l):
        if l != self.cur_l:
            if self.keep == "f32":
                self.cur = self.store[l]
            elif self.keep == "bf16":
                self.cur = {k: v.float() for k, v in self.store[l].items()}
            else:
                self.cur = layer_tensors(self.model, l)
            self.cur_l = l
        return self.cur

    @staticmethod
    def _layer_of(name):
        return int(name.split(".")[2]) if name.startswith("model.layers.") else None

    def __getitem__(self, name):
        l = self._layer_of(name)
        return self.head[name] if l is None else self._layer(l)[name]

    def __contains__(self, name):
        l = self._layer_of(name)
        return name in self.head if l is None else name in self._layer(l)
