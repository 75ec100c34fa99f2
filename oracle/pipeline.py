"""Oracle: the assembled greedy-generate path.  TEST INFRASTRUCTURE ONLY.

   sp-encoded series [N,2Lmax,1] --ts_embedding--> [P,H] --+
   un-expanded prompt ids --expand_placeholders--> ids ----+--merge_embeddings--> [T,H] --QwenOracle--> greedy
Follows chatts/vllm/chatts_vllm.py:538-610 for the order of the stages.
"""
import numpy as np
import torch

from . import protocol, ts_embedding
from .qwen_decoder import QwenOracle


def split_state_dict(sd):
    """HF-named state dict -> (ts_encoder weights as numpy without prefix, decoder weights as torch)."""
    ts = {k[len("ts_encoder."):]: (v.numpy() if hasattr(v, "numpy") else v) for k, v in sd.items()
          if k.startswith("ts_encoder.")}
    dec = {k: (v if hasattr(v, "numpy") else torch.from_numpy(v)) for k, v in sd.items()
           if not k.startswith("ts_encoder.")}
    return ts, dec


def generate(cfg, sd, ids, series, max_new_tokens, num_layers=None, forced_tokens=None):
    """cfg: chatts_amd ChatTSConfig-like (uses .ts, .ts_token_start_index, .oracle_dict()).
    ids: un-expanded prompt ids; series: [N, 2*Lmax, 1] float array or None.
    Returns dict(tokens, logits (list of [V] per step), embeds [T,H], ts_features [P,H], expanded_ids)."""
    tsw, dec = split_state_dict(sd)
    ts0 = cfg.ts_token_start_index
    if series is not None and len(series):
        feats, pc = ts_embedding.ts_embedding_forward(np.asarray(series, dtype=np.float32), cfg.ts, tsw)
    else:
        feats, pc = np.zeros((0, cfg.hidden_size), dtype=np.float32), np.zeros(0, dtype=np.int64)
    full = protocol.expand_placeholders(ids, pc, ts0)
    table = dec["model.embed_tokens.weight"].numpy()
    emb = protocol.merge_embeddings(full, table, feats, ts0)
    m = QwenOracle(cfg.oracle_dict(), dec, num_layers=num_layers)
    toks, logits = m.greedy(torch.from_numpy(emb), max_new_tokens, forced_tokens=forced_tokens)
    return dict(tokens=toks, logits=logits, embeds=emb, ts_features=feats, expanded_ids=full, patch_cnt=pc)
