import numpy as np
import torch


def rel_err(a, b):
    """norm-wise relative error ||a-b|| / ||b|| (the 'relative' of the 1e-3 logit tolerance)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def bf16_round(a):
    """float32 numpy -> float32 numpy holding bf16-representable values (RNE)."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def random_walk_series(rng, n):
    return 50 + 2 * np.cumsum(rng.standard_normal(n))


def chat_prompt(lengths):
    body = f"I have {len(lengths)} time series. " + " ".join(
        f"TS{i} is of length {L}: <ts><ts/>;" for i, L in enumerate(lengths)) + \
        " Please analyze the local changes in these time series."
    return ("<|im_start|>system\nYou are a helpful assistant.<|im_end|><|im_start|>user\n" + body +
            "<|im_end|><|im_start|>assistant\n")
