"""Oracle: the Value-Preserved Time Series Encoder.  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/chatts/vllm/chatts_vllm.py:
  TimeSeriesEmbedding.__init__  :61-91   (MLP = Linear+GELU x (n-1) + Linear; exact-erf GELU)
  TimeSeriesEmbedding.forward   :93-193
  get_patch_cnt                 :198-207
in numpy float32 (float64 optional), patch-major instead of the reference's per-series loop.

Weights are passed as a dict with the reference's state-dict names:
  position_embedding.weight [max_sequence_length+1, embedding_dim]
  mlp.{0,2,4,...}.weight [out,in], mlp.{0,2,4,...}.bias [out]
"""
import math

import numpy as np

try:                                    # scipy is in the image; keep a pure-python fallback
    from scipy.special import erf as _erf
except Exception:                       # pragma: no cover
    _erf = np.vectorize(math.erf, otypes=[np.float64])


def gelu_erf(x):
    """nn.GELU() default = exact erf form (chatts_vllm.py:87)."""
    x64 = x.astype(np.float64)
    return (0.5 * x64 * (1.0 + _erf(x64 / math.sqrt(2.0)))).astype(x.dtype)


def get_patch_cnt(x, ts_config):
    """chatts_vllm.py:198-207: ceil(sum(mask)/patch_size) per series; x [N, 2*Lmax, 1]."""
    n = x.shape[0]
    x = np.asarray(x).reshape(n, -1, ts_config["num_features"])
    vl = x[:, :, -1].sum(axis=1).astype(np.int64)
    p = int(ts_config["patch_size"])
    return (vl + p - 1) // p


def patch_features(x, ts_config, pos_table=None):
    """chatts_vllm.py:94-183: build the MLP input rows.

    x [N, 2*Lmax, 1] interleaved (value, mask).  Returns (feat [P, F] float, patch_cnt [N] int64)
    with F = 16+16*emb (use_position_embedding), 32 (use_position_idx) or 16 (neither).
    """
    cfg = ts_config
    ps = int(cfg["patch_size"])
    n = x.shape[0]
    x = np.asarray(x)
    dt = x.dtype if x.dtype in (np.float32, np.float64) else np.float32
    x = x.reshape(n, -1, cfg["num_features"]).astype(dt)
    mask = x[:, :, -1].astype(np.int64)           # :98  .long() truncates toward zero
    vl = mask.sum(axis=1)                          # :99
    pc = (vl + ps - 1) // ps                       # :100
    use_pe = bool(cfg.get("use_position_embedding", False))
    use_pi = bool(cfg.get("use_position_idx", False))
    max_vl = int(vl.max()) if n else 0
    rows = []
    for i in range(n):
        v, c = int(vl[i]), int(pc[i])
        if c == 0:                                 # :110-111
            continue
        vals = x[i, :v, 0]                         # :114 prefix assumption
        pad = c * ps - v
        pos = np.arange(v, dtype=np.int64)
        if pad > 0:                                # :121-129 last-value pad, padding_idx position
            vals = np.concatenate([vals, np.full(pad, vals[-1], dtype=dt)])
        vals = vals.reshape(c, ps)
        if use_pe:
            pad_idx = int(cfg["max_sequence_length"])
            posp = np.concatenate([pos, np.full(pad, pad_idx, dtype=np.int64)]).reshape(c, ps)
            if posp.max() > pad_idx:
                raise IndexError("index out of range in self")      # nn.Embedding, :165
            emb = pos_table[posp].astype(dt)       # [c, ps, emb]
            rows.append(np.concatenate([vals, emb.reshape(c, -1)], axis=1))   # :176-181
        elif use_pi:
            pi = (pos / max(1, max_vl - 1)).astype(dt)                        # :146-147
            pi = np.concatenate([pi, np.full(pad, -1, dtype=dt)])             # :148-151
            comb = np.stack([vals.reshape(-1), pi], axis=1)                   # :153
            rows.append(comb.reshape(c, ps * 2))                              # :154
        else:
            rows.append(vals)                                                 # :157
    f = (ps + ps * int(cfg.get("embedding_dim", 16))) if use_pe else (2 * ps if use_pi else ps)
    feat = np.concatenate(rows, axis=0) if rows else np.zeros((0, f), dtype=dt)
    return feat, pc


def mlp(feat, weights, num_layers):
    """chatts_vllm.py:83-91,188: y = L_{n-1}(gelu(...gelu(L_0(x))))."""
    h = feat
    for l in range(num_layers):
        w = weights[f"mlp.{2 * l}.weight"].astype(h.dtype)
        b = weights[f"mlp.{2 * l}.bias"].astype(h.dtype)
        h = h @ w.T + b
        if l < num_layers - 1:
            h = gelu_erf(h)
    return h


def ts_embedding_forward(x, ts_config, weights):
    """TimeSeriesEmbedding.forward (chatts_vllm.py:93-193) -> (features [P,H], patch_cnt [N])."""
    feat, pc = patch_features(x, ts_config, weights.get("position_embedding.weight"))
    if feat.shape[0] == 0:
        return np.zeros((0, int(ts_config["hidden_size"])), dtype=feat.dtype), pc
    return mlp(feat, weights, int(ts_config["num_layers"])), pc
