"""Oracle: the <ts><ts/> token protocol and the embedding merge.  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/chatts/vllm/chatts_vllm.py:
  _get_prompt_updates       :369-444  every [<ts>, <ts/>] token pair is replaced by the series'
                                      prefix tokens + <ts> repeated until #<ts> == patch_cnt
                                      (:402-410); embeddings go only on <ts> ids (:412-415)
  get_input_embeddings      :564-574  E = embed(ids); E[ids == ts_token_start_index] = TS rows, in
                                      order (vLLM merge_multimodal_embeddings raises ValueError on
                                      a count mismatch)
and the HF flavour described by its call sites: callers see the un-expanded
ids; the model expands each pair into patch_cnt rows internally.
"""
import numpy as np


def expand_placeholders(ids, patch_cnt, ts_start, ts_tokens=None):
    """ids: 1-D int sequence containing one [ts_start, ts_start+1] pair per series."""
    ids = [int(t) for t in ids]
    out, k, i = [], 0, 0
    while i < len(ids):
        if ids[i] == ts_start and i + 1 < len(ids) and ids[i + 1] == ts_start + 1:
            if k >= len(patch_cnt):
                raise ValueError("more <ts><ts/> pairs than time series")
            toks = list(ts_tokens[k]) if ts_tokens is not None else []
            have = sum(1 for t in toks if t == ts_start)
            if have < int(patch_cnt[k]):
                toks.extend([ts_start] * (int(patch_cnt[k]) - have))
            out.extend(toks)
            k += 1
            i += 2
        else:
            out.append(ids[i])
            i += 1
    if k != len(patch_cnt):
        raise ValueError(f"{len(patch_cnt)} time series but {k} <ts><ts/> pairs")
    return np.asarray(out, dtype=np.int64)


def merge_embeddings(ids, embed_table, ts_rows, ts_start):
    """chatts_vllm.py:564-574 -> [T, H]."""
    ids = np.asarray(ids, dtype=np.int64)
    out = np.array(embed_table[ids], copy=True)
    sel = ids == ts_start
    if int(sel.sum()) != ts_rows.shape[0]:
        raise ValueError(
            f"Attempted to assign {ts_rows.shape[0]} multimodal tokens to {int(sel.sum())} placeholders")
    out[sel] = ts_rows.astype(out.dtype)
    return out
