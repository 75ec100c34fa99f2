"""The reference's own decoder dependency as a CPU baseline / cross-check.  TEST INFRASTRUCTURE ONLY.

The ChatTS decoder is stock `transformers` Qwen2ForCausalLM / Qwen3ForCausalLM (NetManAIOps/ChatTS requirements.txt:7
transformers==4.52.4; the HF-hub remote code subclasses it, README.md:88; 5.x is installed here and on the GPU box - the
Qwen2/Qwen3 math is unchanged).  This module instantiates that class at arbitrary widths WITHOUT random initialisation
(meta device) and binds an HF-named float32 state dict to it, so that bench.py can time the real third-party code on the
host cores (`cpu_baseline.kind = "reference"`) and tests can cross-check oracle/qwen_decoder.py against it at any size.
"""
import torch


def build(cfg, sd, num_layers=None, threads=None):
    """cfg: chatts_amd ChatTSConfig-like; sd: HF-named tensors (float32; tensors are bound, not copied)."""
    import transformers
    L = cfg.num_hidden_layers if num_layers is None else num_layers
    common = dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                  num_hidden_layers=L, num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads,
                  rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, max_position_embeddings=cfg.max_position_embeddings,
                  tie_word_embeddings=False, attn_implementation="eager")
    if cfg.model_type == "qwen3":
        hc = transformers.Qwen3Config(head_dim=cfg.head_dim, attention_bias=False, **common)
        cls = transformers.Qwen3ForCausalLM
    else:
        hc = transformers.Qwen2Config(**common)
        cls = transformers.Qwen2ForCausalLM
    with torch.device("meta"):
        m = cls(hc)
    m.to_empty(device="cpu")
    own = dict(m.named_parameters())
    missing = []
    for name, p in own.items():
        if name not in sd:
            missing.append(name)
            continue
        t = sd[name]
        t = t if isinstance(t, torch.Tensor) else torch.from_numpy(t)
        assert tuple(t.shape) == tuple(p.shape), (name, t.shape, p.shape)
        p.data = t.float()
    assert not missing, missing
    # non-persistent buffers (rotary inv_freq) were created on the meta device: rebuild them for real
    rot = type(m.model.rotary_emb)(config=hc)
    m.model.rotary_emb = rot
    m.eval()
    return m


@torch.no_grad()
def prefill_and_decode(m, embeds, n_decode, embed_table_rows=None):
    """-> (logits of the last prompt row, [decode logits...], seconds prefill, [seconds per decode step])."""
    import time
    t0 = time.time()
    out = m(inputs_embeds=embeds[None], use_cache=True, logits_to_keep=1)
    t_pre = time.time() - t0
    logits = out.logits[0, -1]
    past = out.past_key_values
    steps, times = [], []
    rows = m.get_input_embeddings().weight.shape[0] if embed_table_rows is None else embed_table_rows
    for _ in range(n_decode):
        tok = int(torch.argmax(logits)) % rows
        t0 = time.time()
        out = m(input_ids=torch.tensor([[tok]]), past_key_values=past, use_cache=True)
        times.append(time.time() - t0)
        logits = out.logits[0, -1]
        past = out.past_key_values
        steps.append(logits)
    return out, steps, t_pre, times
