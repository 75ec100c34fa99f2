"""GPU parity of chatts_sample_batched (through the C-ABI) against oracle/sampler.py."""
import ctypes as C

import numpy as np
import pytest
import torch

from chatts_amd import _lib
from oracle import sampler as osamp

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def st():
    return torch.cuda.current_stream().cuda_stream


def _sample(lib, logits, temperature, top_k, top_p, seed, steps=None, vocab_offset=0):
    """logits [B, V] -> (tokens [B], token_logits [B], n_kept [B], kept_mass [B]); steps = draw counters [B]."""
    b, v = logits.shape
    tok = torch.full((b,), -1, dtype=torch.int64, device=DEV)
    tl = torch.zeros(b, dtype=torch.float32, device=DEV)
    nk = torch.zeros(b, dtype=torch.int32, device=DEV)
    km = torch.zeros(b, dtype=torch.float32, device=DEV)
    step = torch.zeros(b, dtype=torch.int32, device=DEV) if steps is None else torch.tensor(steps, dtype=torch.int32, device=DEV)
    step0 = step.clone()
    out = torch.full((b, 8), -1, dtype=torch.int64, device=DEV)
    sa = _lib.SamplingArgs(temperature=temperature, top_k=top_k, top_p=top_p, seed=seed, n_kept=nk.data_ptr(),
                           kept_mass=km.data_ptr())
    _lib.check(lib.chatts_sample_batched(logits.data_ptr(), b, v, v, vocab_offset, C.byref(sa), tok.data_ptr(), tl.data_ptr(),
                                         out.data_ptr(), 8, step.data_ptr(), None, 0, st()))
    torch.cuda.synchronize()
    assert torch.equal(step, step0 + 1)
    for i in range(b):
        if step0[i] < 8:
            assert out[i, step0[i]] == tok[i]
    return tok.cpu().numpy(), tl.cpu().numpy(), nk.cpu().numpy(), km.cpu().numpy()


def _check_draw(logits_row, tok, temperature, top_k, top_p, seed, seq, step, n_kept, kept_mass, vocab_offset=0):
    """The device's choice is valid under the oracle's float64 rule (cuts and CDF boundaries get a float32 margin)."""
    p, mass = osamp.kept_set(logits_row, temperature, top_k, top_p)
    kept = p > 0
    t = int(tok) - vocab_offset
    assert 0 <= t < len(p)
    # kept set: identical unless some probability sits within rounding distance of a cut
    pf, _ = osamp.kept_set(logits_row, temperature, 0, 1.0)
    cut = pf[kept].min()
    ambiguous = int((np.abs(pf - cut) <= 2e-5 * cut).sum())
    e_rel = np.exp((logits_row.astype(np.float64) - logits_row.max()) / temperature)
    ambiguous += int((np.abs(e_rel - osamp.MASS_FLOOR) <= 1e-4 * osamp.MASS_FLOOR).sum())     # float32 exp around the mass floor
    assert abs(int(n_kept) - int(kept.sum())) <= ambiguous, (n_kept, kept.sum(), ambiguous)
    assert abs(float(kept_mass) - mass) <= 1e-4 + ambiguous * cut
    assert pf[t] >= cut * (1 - 2e-5), "a token outside the kept set was drawn"
    u = osamp.uniform24(seed, seq, step) / float(1 << 24)
    cum = np.cumsum(p) / p.sum()
    lo = cum[t] - p[t] / p.sum()
    assert lo - 1e-4 <= u <= cum[t] + 1e-4, (u, lo, cum[t])


@pytest.mark.parametrize("temperature,top_k,top_p", [(1.0, 0, 1.0), (0.2, 0, 1.0), (0.5, 0, 0.95), (1.0, 50, 1.0), (0.7, 20, 0.9),
                                                     (1.5, 1000, 0.99), (1.0, 1, 1.0), (1.0, 0, 1e-6)])
def test_sampler_parity_full_vocab(lib, temperature, top_k, top_p):
    g = torch.Generator().manual_seed(int(temperature * 10) + top_k)
    v = 152064
    logits = (torch.randn((4, v), generator=g) * 2.5).to(DEV)
    logits[1, 777] = 30.0                                     # a dominant token
    logits[2] = torch.round(logits[2] * 2) / 2                # heavy ties
    steps = [0, 5, 123, 4000]
    tok, tl, nk, km = _sample(lib, logits, temperature, top_k, top_p, seed=99, steps=steps)
    ln = logits.cpu().numpy()
    for b in range(4):
        assert tl[b] == ln[b, tok[b]]
        if b == 2:
            continue                                          # ties: checked by the dedicated test below
        _check_draw(ln[b], tok[b], temperature, top_k, top_p, 99, b, steps[b], nk[b], km[b])
    if top_k == 1 or top_p < 1e-3:
        assert all(tok[b] == ln[b].argmax() for b in (0, 1, 3))


def test_sampler_ties_are_kept(lib):
    v = 4096
    logits = torch.full((1, v), -5.0, device=DEV)
    logits[0, [10, 200, 3000, 4095]] = 2.0                    # four tied leaders
    logits[0, 7] = 1.0
    tok, _, nk, km = _sample(lib, logits, 1.0, 2, 1.0, seed=1)    # top-k 2 cuts through the tie: all four stay
    assert nk[0] == 4 and tok[0] in (10, 200, 3000, 4095)
    tok, _, nk, km = _sample(lib, logits, 1.0, 0, 0.3, seed=1)    # top-p inside the tie: all four stay
    assert nk[0] == 4
    seen = set()
    for s in range(64):
        seen.add(int(_sample(lib, logits, 1.0, 4, 1.0, seed=3, steps=[s])[0][0]))
    assert seen == {10, 200, 3000, 4095}


def test_sampler_matches_distribution(lib):
    """20 000 draws (draw counter 0..) from a 7-token distribution: chi-square against the renormalised top-p set."""
    probs = np.array([0.35, 0.25, 0.15, 0.1, 0.08, 0.05, 0.02])
    v = 1024
    row = np.full(v, -40.0, dtype=np.float32)
    ids = [3, 100, 101, 500, 777, 1000, 1023]
    row[ids] = np.log(probs).astype(np.float32)
    n = 20000
    logits = torch.from_numpy(row).to(DEV)[None].repeat(n, 1)
    # one launch, n sequences: sequence index b is part of the random stream
    tok, _, nk, _ = _sample(lib, logits[:, :], 1.0, 0, 0.9, seed=2024)
    want = probs[:5] / probs[:5].sum()                       # 0.35+0.25+0.15+0.1 = 0.85 < 0.9 <= 0.93
    assert (nk == 5).all()
    counts = np.array([(tok == i).sum() for i in ids[:5]])
    assert counts.sum() == n
    chi2 = (((counts - n * want) ** 2) / (n * want)).sum()
    assert chi2 < 25.0, (counts, n * want)                    # 4 dof: P(chi2 > 25) ~ 5e-5


def test_sampler_is_deterministic_and_counter_driven(lib):
    g = torch.Generator().manual_seed(5)
    logits = torch.randn((2, 32000), generator=g).to(DEV) * 3
    a = _sample(lib, logits, 0.8, 40, 0.95, seed=11, steps=[7, 7])[0]
    b = _sample(lib, logits, 0.8, 40, 0.95, seed=11, steps=[7, 7])[0]
    assert (a == b).all()
    draws = {tuple(_sample(lib, logits, 0.8, 40, 0.95, seed=11, steps=[s, s])[0]) for s in range(16)}
    assert len(draws) > 8                                     # the draw counter changes the variate
    off = _sample(lib, logits, 0.8, 40, 0.95, seed=11, steps=[7, 7], vocab_offset=1000)[0]
    assert (off == a + 1000).all()


def test_sampler_argument_errors(lib):
    logits = torch.zeros((1, 64), device=DEV)
    for bad in (dict(temperature=0.0, top_p=1.0), dict(temperature=1.0, top_p=0.0), dict(temperature=-1.0, top_p=0.5)):
        sa = _lib.SamplingArgs(top_k=0, seed=0, n_kept=None, kept_mass=None, **bad)
        tok = torch.zeros(1, dtype=torch.int64, device=DEV)
        rc = lib.chatts_sample_batched(logits.data_ptr(), 1, 64, 64, 0, C.byref(sa), tok.data_ptr(), None, None, 0, None, None, 0, st())
        assert rc == _lib.E_BADARG


def test_per_row_sampling_parameters(lib):
    """ChattsSamplingArgs.*_rows: every row draws with its own (temperature, top_k, top_p, seed) read from device arrays; a row with
    temperature 0 decodes greedily (torch.argmax's token, ties -> first index); the variate of a row depends on (seed, step) only, so
    the same settings give the same token in ANY row - and equal the scalar call's row 0."""
    v = 152064
    g = torch.Generator().manual_seed(11)
    base = (torch.randn((1, v), generator=g) * 2.5)
    logits = base.repeat(5, 1).to(DEV)
    logits[3, 500] = 40.0; logits[3, 499] = 40.0                # greedy row with a tie
    rows = [(0.7, 20, 0.9, 1234), (1.0, 0, 1.0, 77), (0.7, 20, 0.9, 1234), (0.0, 0, 1.0, 5), (0.2, 0, 0.5, 9)]
    temp = torch.tensor([r[0] for r in rows], dtype=torch.float32, device=DEV)
    topk = torch.tensor([r[1] for r in rows], dtype=torch.int32, device=DEV)
    topp = torch.tensor([r[2] for r in rows], dtype=torch.float32, device=DEV)
    seed = torch.tensor([r[3] for r in rows], dtype=torch.int32, device=DEV)
    steps = [3, 3, 3, 0, 7]
    b = len(rows)
    tok = torch.full((b,), -1, dtype=torch.int64, device=DEV)
    tl = torch.zeros(b, dtype=torch.float32, device=DEV)
    step = torch.tensor(steps, dtype=torch.int32, device=DEV)
    pos = torch.tensor([4, 4, -1, 4, 4], dtype=torch.int32, device=DEV)
    out = torch.full((b, 8), -1, dtype=torch.int64, device=DEV)
    sa = _lib.SamplingArgs(temperature=0.0, top_k=0, top_p=0.0, seed=0, n_kept=None, kept_mass=None, temperature_rows=temp.data_ptr(),
                           top_k_rows=topk.data_ptr(), top_p_rows=topp.data_ptr(), seed_rows=seed.data_ptr())
    _lib.check(lib.chatts_sample_batched(logits.data_ptr(), b, v, v, 0, C.byref(sa), tok.data_ptr(), tl.data_ptr(), out.data_ptr(), 8,
                                         step.data_ptr(), pos.data_ptr(), 0, st()))
    torch.cuda.synchronize()
    tok = tok.cpu().numpy()
    assert tok[0] == tok[2]                                      # same settings, same step, another row: the same token
    assert tok[3] == 499 and float(tl[3]) == 40.0                # greedy row: first index of the maximum
    assert step.tolist() == [4, 4, 4, 1, 8] and pos.tolist() == [5, 5, -1, 5, 5]
    assert all(int(out[i, steps[i]]) == int(tok[i]) for i in range(b))
    ln = logits.cpu().numpy()
    for i in (0, 1, 4):                                          # each row valid under ITS settings (row salt 0 in per-row mode)
        nk, km = _sample_rows_diag(lib, logits[i:i + 1], rows[i], steps[i])
        _check_draw(ln[i], tok[i], rows[i][0], rows[i][1], rows[i][2], rows[i][3], 0, steps[i], nk, km)
    # the scalar call draws row 0 with salt 0 too: identical token
    t0 = _sample(lib, logits[:1], rows[0][0], rows[0][1], rows[0][2], rows[0][3], steps=[3])[0][0]
    assert t0 == tok[0]


def _sample_rows_diag(lib, logits1, row, step):
    """kept-set diagnostics of one row through the scalar interface (same cuts as the per-row call)."""
    _, _, nk, km = _sample(lib, logits1, row[0], row[1], row[2], row[3], steps=[step])
    return nk[0], km[0]
