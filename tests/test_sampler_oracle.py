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
