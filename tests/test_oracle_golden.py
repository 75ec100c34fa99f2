"""CPU: pin the oracle against vectors produced by the reference itself (tests/golden/make_golden.py)
and against the one known-answer the reference stores (demo/demo_lora.ipynb:147)."""
import json

import numpy as np
import pytest
import torch

from oracle import protocol, qwen_decoder, sp_encoding as osp, ts_embedding as ots


def _demo_series():
    x = np.arange(256)
    ts1 = np.sin(x / 10) * 5.0
    ts1[100:] -= 10.0
    ts2 = x * 0.05
    ts2[103] += 10.0
    return ts1, ts2


def test_sp_known_answer_from_notebook():
    # /root/reference/demo/demo_lora.ipynb:147 (stored cell output)
    ts1, ts2 = _demo_series()
    assert osp.hf_prefix(ts1) == ("[offset=6.0772|scaling=3.6917|length=256|max=4.9979|min=-15.0000|"
                                  "left=0.0000|right=-8.2047]<ts><ts/>")
    assert osp.hf_prefix(ts2) == ("[offset=-6.4141|scaling=2.9120|length=256|max=15.1500|min=0.0000|"
                                  "left=0.0000|right=12.7500]<ts><ts/>")


def test_sp_encoding_matches_reference(golden):
    g = golden("sp_encoding")
    for i in range(int(g["n"])):
        enc, prompt, meta = osp.sp_encoding(g[f"in_{i}"])
        assert np.array_equal(enc, g[f"enc_{i}"]), i          # float64, bit exact
        assert prompt == str(g[f"prompt_{i}"])
        assert meta["offset"] == float(g[f"offset_{i}"]) and meta["scale_factor"] == float(g[f"scale_{i}"])
    idx = g["batch_idx"]
    rp, arr = osp.eval_prompt_to_encoding(str(g["batch_prompt_in"]), [g[f"in_{i}"] for i in idx])
    assert rp == str(g["batch_prompt_out"])
    assert np.array_equal(arr, g["batch_arr"])


@pytest.mark.parametrize("name", ["posemb", "single", "posidx", "raw"])
def test_ts_embedding_matches_reference(golden, name):
    g = golden("ts_embedding_" + name)
    cfg = json.loads(str(g["config"]))
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    feats, pc = ots.ts_embedding_forward(g["x"], cfg, w)
    assert np.array_equal(pc, g["patch_cnt"])
    assert np.array_equal(ots.get_patch_cnt(g["x"], cfg), g["patch_cnt"])
    assert feats.shape == g["features"].shape
    np.testing.assert_allclose(feats, g["features"], rtol=2e-5, atol=2e-5)
    # structural invariants
    L = g["lengths"]
    assert np.array_equal(pc, (L + 15) // 16) and feats.shape[0] == pc.sum()


def test_ts_embedding_tail_pad_is_last_value(golden):
    g = golden("ts_embedding_posemb")
    cfg = json.loads(str(g["config"]))
    feat, pc = ots.patch_features(g["x"], cfg, g["w:position_embedding.weight"])
    # series 1 has length 17 -> 2 patches; the second holds value[16] followed by 15 copies of it
    row = int(pc[0]) + 1
    v = g["x"][1, 32, 0]
    assert np.all(feat[row, :16] == v)
    pad_emb = g["w:position_embedding.weight"][cfg["max_sequence_length"]]
    assert np.array_equal(feat[row, 16 + 16:16 + 32], pad_emb)        # slot 1 is padding -> padding_idx row


@pytest.mark.parametrize("kind", ["qwen2", "qwen3"])
def test_decoder_matches_transformers(golden, kind):
    g = golden(kind + "_tiny")
    cfg = json.loads(str(g["config"]))
    w = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")}
    m = qwen_decoder.QwenOracle(cfg, w)
    logits = m.forward_embeds(torch.from_numpy(g["embeds"]))
    np.testing.assert_allclose(logits.numpy(), g["prefill_logits"], rtol=1e-4, atol=2e-5)
    m.reset()
    toks, step_logits = m.greedy(torch.from_numpy(g["embeds"]), len(g["tokens"]) + 1)
    assert toks[:len(g["tokens"])] == g["tokens"].tolist()
    for i in range(len(g["tokens"])):
        np.testing.assert_allclose(step_logits[i + 1].numpy(), g["decode_logits"][i], rtol=1e-4, atol=2e-5)


def test_protocol_expand_and_merge():
    ts = 100
    ids = [1, 2, ts, ts + 1, 3, ts, ts + 1, 4]
    out = protocol.expand_placeholders(ids, [3, 0], ts)
    assert out.tolist() == [1, 2, ts, ts, ts, 3, 4]
    out = protocol.expand_placeholders(ids, [2, 1], ts, ts_tokens=[[7, 8], [9]])
    assert out.tolist() == [1, 2, 7, 8, ts, ts, 3, 9, ts, 4]
    table = np.arange(200 * 4, dtype=np.float32).reshape(200, 4)
    rows = -np.ones((3, 4), dtype=np.float32) * np.arange(1, 4)[:, None]
    m = protocol.merge_embeddings([1, ts, ts, 5, ts], table, rows, ts)
    assert np.array_equal(m[1], rows[0]) and np.array_equal(m[2], rows[1]) and np.array_equal(m[4], rows[2])
    assert np.array_equal(m[0], table[1]) and np.array_equal(m[3], table[5])
    with pytest.raises(ValueError):
        protocol.merge_embeddings([1, ts, 5], table, rows, ts)
    with pytest.raises(ValueError):
        protocol.expand_placeholders(ids, [1], ts)
