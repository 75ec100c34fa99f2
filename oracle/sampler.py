"""CPU restatement of next-token sampling (temperature / top-k / top-p).  TEST INFRASTRUCTURE ONLY.

The reference does not contain a sampler: its drivers hand SamplingParams(temperature=0.2)
(chatts/utils/inference_tsmllm_vllm.py:43-46), temperature=0.5 + top_p=0.95 (chatts/utils/llm_utils.py:94,153) to
vLLM (pinned vllm==0.8.5, requirements.txt:30; NOT IN REFERENCE).  vLLM's published rule
(vllm/v1/sample/ops/topk_topp_sampler.py, apply_top_k_top_p): logits /= temperature; mask everything below the k-th
largest logit; sort ascending, softmax, cumulative sum, mask tokens whose cumulative mass <= 1 - top_p (never the
last one) - i.e. keep the smallest set of most probable tokens whose mass reaches top_p; softmax of what is left; draw.
Parity is unpinned against vLLM itself (absent here, and its random stream is torch's Philox): what the tests pin is
this rule - the kept set, and that the drawn token is the one the uniform variate selects in token-id order.

Differences stated in include/chatts_amd.h: ties at either cut are all kept; a token whose probability is below 2^-40
of the most probable one carries no mass (the HIP kernel accumulates masses as 2^-40 fixed point).
"""
import numpy as np

MASS_FLOOR = 2.0 ** -40


def mix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def uniform24(seed, seq, step):
    """The 24-bit variate the HIP sampler derives from (seed, sequence, draw counter)."""
    inner = mix32((seq * 0x9E3779B9 + step * 0x85EBCA6B + 0x68BC21EB) & 0xFFFFFFFF)
    return mix32((seed ^ inner) & 0xFFFFFFFF) >> 8


def kept_set(logits, temperature, top_k=0, top_p=1.0):
    """-> (probabilities over the full vocabulary in float64 with zeros outside the kept set, NOT renormalised;
           mass of the kept set relative to the top-k set)."""
    l = np.asarray(logits, dtype=np.float64)
    keep = np.ones(l.shape[0], dtype=bool)
    if 0 < top_k < l.shape[0]:
        kth = np.sort(l)[-top_k]
        keep &= l >= kth
    e = np.where(keep, np.exp((l - l.max()) / temperature), 0.0)
    e = np.where(e >= MASS_FLOOR, e, 0.0)
    z = e.sum()
    if top_p < 1.0:
        order = np.argsort(-e, kind="stable")
        cum = np.cumsum(e[order])
        n = int(np.searchsorted(cum, top_p * z, side="left")) + 1        # smallest prefix with mass >= top_p * z
        cut = e[order[min(n, len(order)) - 1]]
        keep &= e >= cut                                                   # ties at the cut are kept
        e = np.where(keep, e, 0.0)
    return e / z, e.sum() / z


def draw(probs_unnormalised, u):
    """First token id (ascending) whose running share of the kept mass exceeds u in [0, 1)."""
    cum = np.cumsum(probs_unnormalised)
    return int(np.searchsorted(cum, u * cum[-1], side="right"))


def sample(logits, temperature, top_k, top_p, seed, seq, step):
    p, _ = kept_set(logits, temperature, top_k, top_p)
    return draw(p, uniform24(seed, seq, step) / float(1 << 24))
