"""CPU checks of the sampling oracle (oracle/sampler.py) - the rule the HIP sampler is tested against."""
import numpy as np

from oracle import sampler


def test_kept_set_rules():
    logits = np.log(np.array([0.5, 0.2, 0.15, 0.1, 0.05]))
    p, mass = sampler.kept_set(logits, 1.0, 0, 1.0)
    assert np.allclose(p, [0.5, 0.2, 0.15, 0.1, 0.05]) and np.isclose(mass, 1.0)
    p, mass = sampler.kept_set(logits, 1.0, 2, 1.0)                 # top-k 2
    assert (p > 0).tolist() == [True, True, False, False, False]
    p, mass = sampler.kept_set(logits, 1.0, 0, 0.8)                 # 0.5 + 0.2 = 0.7 < 0.8 <= 0.85
    assert (p > 0).tolist() == [True, True, True, False, False] and np.isclose(mass, 0.85)
    p, mass = sampler.kept_set(logits, 1.0, 0, 0.5)                 # the boundary token itself reaches top_p
    assert (p > 0).tolist() == [True, False, False, False, False]
    p, mass = sampler.kept_set(logits, 1.0, 4, 0.8)                 # top-k first, top-p on the renormalised rest
    assert (p > 0).tolist() == [True, True, True, False, False]     # 0.85 / 0.95 >= 0.8 > 0.7 / 0.95
    # temperature sharpens: at T = 0.2 the first token alone holds > 0.95
    p, mass = sampler.kept_set(logits, 0.2, 0, 0.95)
    assert (p > 0).tolist() == [True, False, False, False, False]
    # ties at the cut are all kept
    p, _ = sampler.kept_set(np.log(np.array([0.4, 0.2, 0.2, 0.2])), 1.0, 2, 1.0)
    assert (p > 0).all()


def test_draw_is_inverse_cdf_in_token_order():
    p = np.array([0.0, 0.25, 0.0, 0.5, 0.25])
    assert [sampler.draw(p, u) for u in (0.0, 0.2499, 0.25, 0.74, 0.75, 0.999)] == [1, 1, 3, 3, 4, 4]


def test_uniform24_stream():
    us = np.array([sampler.uniform24(7, 0, s) for s in range(4096)], dtype=np.float64) / (1 << 24)
    assert 0 <= us.min() and us.max() < 1 and abs(us.mean() - 0.5) < 0.02 and len(set(us.tolist())) > 4000
    assert sampler.uniform24(7, 0, 5) != sampler.uniform24(7, 1, 5) != sampler.uniform24(8, 1, 5)
    # pinned values of the stream (the HIP kernel is checked against the oracle, the oracle against these constants)
    assert [sampler.uniform24(1234, 0, s) for s in range(3)] == [16190760, 11032913, 8685192]
    assert sampler.uniform24(1234, 3, 17) == 12540707


def test_kept_set_is_pinned_by_the_transformers_warpers():
    """Third-party pin: the reference's HF drivers sample through transformers' logits warpers
    (model.generate(..., temperature=0.2), chatts/utils/inference_tsmllm_deepspeed.py:95-100; transformers is the
    requirements.txt:7 dependency and IS installed here).  TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper
    (generate()'s order) must keep exactly the token set oracle/sampler.kept_set keeps, with the same renormalised
    probabilities, on real decoder logits (the committed reference-generated fixture) and on random ones."""
    import os

    import torch
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "qwen2_tiny.npz"), allow_pickle=False)
    rows = [np.asarray(g[k], dtype=np.float64).reshape(-1, np.asarray(g[k]).shape[-1])[-1] for k in g.files if "logits" in k]
    assert rows, g.files
    rng = np.random.default_rng(0)
    rows += [rng.standard_normal(4096) * s for s in (1.0, 3.0, 6.0)]
    checked = 0
    for logits in rows:
        for T, K, P in [(1.0, 50, 1.0), (0.2, 0, 0.95), (0.5, 0, 0.95), (0.8, 50, 0.95), (1.5, 20, 0.5), (1.0, 0, 0.3), (0.7, 1, 1.0),
                        (1.0, 5, 0.999)]:
            s = torch.from_numpy(logits)[None].double()
            ids = torch.zeros((1, 1), dtype=torch.long)
            s = TemperatureLogitsWarper(T)(ids, s)
            if K:
                s = TopKLogitsWarper(K)(ids, s)
            if P < 1.0:
                s = TopPLogitsWarper(P)(ids, s)
            hf_keep = torch.isfinite(s[0]).numpy()
            hf_p = torch.softmax(s[0], -1).numpy()
            p, _ = sampler.kept_set(logits, T, K, P)
            ours = p > 0
            if not np.array_equal(ours, hf_keep):
                # the only admissible difference: a token sitting on the top-p boundary to within rounding of the cumulative sum
                diff = np.flatnonzero(ours != hf_keep)
                e = np.exp((logits - logits.max()) / T)
                srt = np.sort(e[ours | hf_keep])[::-1]
                cum = np.cumsum(srt) / e[(np.sort(logits)[-K] <= logits) if K else np.ones_like(ours)].sum()
                assert len(diff) == 1 and np.min(np.abs(cum - P)) < 1e-9, (T, K, P, diff)
                continue
            assert np.allclose(p / p.sum(), hf_p, rtol=1e-9, atol=1e-15)
            checked += 1
    assert checked >= 8 * (len(rows) - 1)
