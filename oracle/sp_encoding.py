"""Oracle: value-preserved ("sp") normalisation and batch padding.  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/chatts/utils/encoding_utils.py:
  sp_encoding              :23-37
  eval_prompt_to_encoding  :65-86  (method == 'sp')
and the richer prefix the HF-hub remote processor prints, whose format is only
visible in a stored notebook output (/root/reference/demo/demo_lora.ipynb:147):
  [offset=6.0772|scaling=3.6917|length=256|max=4.9979|min=-15.0000|left=0.0000|right=-8.2047]<ts><ts/>
"""
import numpy as np


def sp_stats(series):
    """(mean, scale_factor) exactly as encoding_utils.py:25-31 computes them (float64)."""
    x = np.asarray(series, dtype=np.float64)
    mean = np.mean(x)
    dev = x - mean
    factor = 1.0
    if np.any(np.abs(dev) >= 3.0):
        factor = np.max(np.abs(dev)) / 3.0
    return mean, factor


def sp_encoding(series):
    """encoding_utils.py:23-37 -> (array [2L,1] interleaved (value, 1.0), prompt, meta)."""
    x = np.asarray(series, dtype=np.float64)
    mean, factor = sp_stats(x)
    scaled = x - mean
    if np.any(np.abs(scaled) >= 3.0):
        scaled = scaled / factor
    prompt = f"[Value Offset: {-mean:.4f}|Value Scaling: {factor:.4f}]<ts><ts/>"
    out = np.stack([scaled, np.ones_like(scaled)], axis=-1).reshape(-1, 1)
    return out, prompt, {"offset": float(-mean), "scale_factor": float(factor)}


def hf_prefix(series):
    """Prefix text of the HF-hub processor (demo_lora.ipynb:147); raw-series statistics."""
    x = np.asarray(series, dtype=np.float64)
    mean, factor = sp_stats(x)
    return (f"[offset={-mean:.4f}|scaling={factor:.4f}|length={len(x)}|max={np.max(x):.4f}|"
            f"min={np.min(x):.4f}|left={x[0]:.4f}|right={x[-1]:.4f}]<ts><ts/>")


def eval_prompt_to_encoding(prompt, timeseries, prefix="sp"):
    """encoding_utils.py:65-86: splice one prefix per '<ts><ts/>' and zero-pad/stack the series.

    Returns (prompt_with_prefixes, array [N, 2*Lmax, 1] float64).
    """
    parts = prompt.split("<ts><ts/>")
    assert len(timeseries) == len(parts) - 1
    result = parts[0]
    encoded = []
    for i, ts in enumerate(timeseries):
        arr, pfx, _ = sp_encoding(np.array(ts))
        if prefix == "hf":
            pfx = hf_prefix(np.array(ts))
        result += pfx + parts[i + 1]
        encoded.append(arr[None])
    lmax = max(a.shape[1] for a in encoded)
    padded = [np.pad(a, ((0, 0), (0, lmax - a.shape[1]), (0, 0))) for a in encoded]
    return result, np.concatenate(padded, axis=0)
