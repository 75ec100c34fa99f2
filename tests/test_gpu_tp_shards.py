"""Parity of the tensor-parallel SHARD SHAPES at ChatTS-14B widths: BASELINE.json's configs 4 and 5 are TP=8
configurations, and a rank of TP = 8 / 4 runs kernels whose geometry has nothing in common with TP = 1 - qkv N = 896 / 1792,
gate_up N = 3456 / 6912, o_proj K = 640 / 1280, down_proj K = 1728 / 3456, ONE kv head (group of 5 query heads) per rank,
vocabulary 19008 / 38016.  Every launch-geometry decision (launch_gemv's balanced layouts, the LDS-DMA GEMM's tile ranges and
split-K counts, the weight-streaming kernel's splits, attn_decode_kernel<5> with one kv head) is taken at those sizes here, on
the bench inputs, and checked against the UNSHARDED float32 oracle: logits within 1e-3 (norm-wise and max-abs over max logit),
identical greedy tokens, bit-identical residual streams on all ranks.  The W shards run one after the other on this one GPU
(tests/tp_emulation.py: the collectives are played between the kernels in the exchange kernels' rank order); the exchange
TRANSPORT is the subject of tests/test_gpu_tp_p2p.py and tools/tp_parity_worker.py.  Depth 4 keeps the CPU oracle to seconds.
Reference: vLLM tensor_parallel_size=k, NetManAIOps/ChatTS demo/demo_vllm.py:30, chatts/utils/llm_utils.py:154."""
import numpy as np
import pytest
import torch

import bench
from chatts_amd import config as cfgmod
from oracle import from_device, pipeline
from tests.tp_emulation import EmulatedTP
from tests.util import rel_err

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3
DEPTH = 4


def _max_abs_over_max(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _check_logits(got, want, what, tight=2e-4):
    g, w = got.float().cpu().numpy(), want.numpy()
    e, ea = rel_err(g, w), _max_abs_over_max(g, w)
    assert e < LOGIT_TOL and ea < LOGIT_TOL, (what, e, ea)
    assert e < tight, (what, e)            # what the f32 / bf16x2 design delivers at this depth
    return e


def _oracle_and_embeddings(tp, cfg, proc, prompt, series, new):
    """oracle run on the checkpoint re-assembled from the shards + the device-side embeddings of the same request"""
    m0 = tp.ms[0]
    inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
    ids = inputs["input_ids"][0].tolist()
    if getattr(tp, "_oracle_sd", None) is None:         # the checkpoint the shards hold, re-assembled once per group
        tp._oracle_sd = {**from_device.ts_encoder_state_dict(m0), **from_device.sharded_state_dict(tp.ms)}
    sd = tp._oracle_sd
    want = pipeline.generate(cfg, sd, ids, inputs["timeseries"].numpy(), new)
    ser = inputs["timeseries"].cuda()
    lengths = list(proc.last_lengths)
    mm = m0.get_multimodal_embeddings(timeseries=ser, valid_lengths=lengths)        # the TS encoder is replicated
    full = m0.expand_input_ids(ids, [(L + 15) // 16 for L in lengths])
    assert full == list(want["expanded_ids"])
    emb = m0.get_input_embeddings(torch.tensor(full, dtype=torch.int64), mm)
    return want, emb, len(full)


def _shard_shapes(tp, world):
    m = tp.ms[-1]
    assert (m.plan.nq, m.plan.nkv, m.plan.inter, m.plan.vocab) == (40 // world, 8 // world, 13824 // world, 152064 // world)
    assert m.layers[0]["qkv"].shape == ((40 + 16) // world * 128, 5120) and m.layers[0]["down"].shape == (5120, 13824 // world)
    assert m.plan.v0 == (world - 1) * m.plan.vocab


@pytest.mark.parametrize("world", [8, 4])
def test_headline_prompt_on_tp_shards_vs_oracle(world):
    """The bench prompt (8 x 256 -> 798 tokens: one prefill chunk through the LDS-DMA GEMMs at the shard widths, MFMA attention with
    40 / W query heads), first token, then 8 decode steps (batch-1 GEMVs + one-kv-head decode attention + vocab slices)."""
    new = 9
    cfg = cfgmod.preset("chatts-14b", num_hidden_layers=DEPTH)
    proc, prompt, series, lengths = bench.build_inputs(cfg, 8, 256)
    tp = EmulatedTP(cfg, world, seed=0, max_ctx=1024, max_prefill_tokens=1024)
    _shard_shapes(tp, world)
    want, emb, T = _oracle_and_embeddings(tp, cfg, proc, prompt, series, new)
    assert T >= 700
    last = tp.prefill(emb)
    assert tp.residual_streams_identical(last)
    tok, lg = tp.first_token(last, T)
    worst = _check_logits(lg, want["logits"][0], "first token")
    toks = [tok]
    for i in range(1, new):
        tok, lg = tp.decode_step()
        toks.append(tok)
        worst = max(worst, _check_logits(lg, want["logits"][i], f"step {i}"))
        assert tp.residual_streams_identical(1)
    assert toks == want["tokens"]
    assert all(m.buf["out_tokens"][:new].tolist() == want["tokens"] for m in tp.ms)
    print(f"[tp shards] W={world} headline prompt: worst logits error {worst:.2e}")


def test_config4_mixed_lengths_on_tp8_shards_vs_oracle():
    """BASELINE.json config 4 in the form it is stated (TP=8): 30 series of mixed lengths 64..1024 -> a 3.5k-token prompt that goes
    in as FOUR chunks (1024 rows each at the shard widths, attention against a growing cache), then decode steps at ctx 3.5k."""
    world, new = 8, 6
    cfg = cfgmod.preset("chatts-14b", num_hidden_layers=DEPTH)
    proc, prompt, series, lengths = bench.build_inputs(cfg, 30, 256, "mixed")
    assert any(L % 16 for L in lengths)
    tp = EmulatedTP(cfg, world, seed=0, max_ctx=4096, max_prefill_tokens=1024)
    want, emb, T = _oracle_and_embeddings(tp, cfg, proc, prompt, series, new)
    assert T > 3 * 1024
    last = tp.prefill(emb)
    assert tp.residual_streams_identical(last)
    tok, lg = tp.first_token(last, T)
    _check_logits(lg, want["logits"][0], "first token")
    toks = [tok]
    for i in range(1, new):
        tok, lg = tp.decode_step()
        toks.append(tok)
        _check_logits(lg, want["logits"][i], f"step {i}")
    assert toks == want["tokens"]
    assert tp.residual_streams_identical(1)


def test_config5_fp8_batch16_on_tp8_shards_vs_oracle():
    """BASELINE.json config 5 in the form it is stated (TP=8): fp8 weights (every shard quantises its own rows / K-slices: the oracle
    runs on the values the shards actually hold), 16 DIFFERENT prompts of 8 x 1024 steps (~1.2k tokens: two chunks each), then the
    16-wide decode step at the shard widths (M = 16 fp8 weight-streaming GEMMs with N = 896 / 3456, K = 640 / 1728; per-sequence
    attention on one kv head).  The oracle recomputes slots 0, 7 and 15."""
    world, B, new = 8, 16, 5
    cfg = cfgmod.preset("chatts-14b", num_hidden_layers=DEPTH)
    proc, prompt, reqs, lengths = bench.build_batched_requests(cfg, B, 8, 1024)
    tp = EmulatedTP(cfg, world, seed=0, max_ctx=2048, max_prefill_tokens=1024, weight_format="fp8", max_batch=B)
    assert "gate_up8" in tp.ms[0].layers[0] and tp.ms[0].layers[0]["gate_up8"].shape == (2 * 13824 // world, 5120)
    wants, first = {}, {}
    for s in range(B):
        if s in (0, 7, 15):
            wants[s], emb, T = _oracle_and_embeddings(tp, cfg, proc, prompt, reqs[s], new)
        else:
            m0 = tp.ms[0]
            inputs = proc(text=[prompt], timeseries=reqs[s], padding=True, return_tensors="pt")
            ln = list(proc.last_lengths)
            mm = m0.get_multimodal_embeddings(timeseries=inputs["timeseries"].cuda(), valid_lengths=ln)
            full = m0.expand_input_ids(inputs["input_ids"][0].tolist(), [(L + 15) // 16 for L in ln])
            emb, T = m0.get_input_embeddings(torch.tensor(full, dtype=torch.int64), mm), len(full)
        assert T >= 1190
        first[s] = tp.admit(s, emb, T)
    for s, w in wants.items():
        _check_logits(first[s][1], w["logits"][0], f"slot {s} first token")
    worst = 0.0
    for i in range(1, new):
        toks, lg = tp.batched_step()
        for s, w in wants.items():
            worst = max(worst, _check_logits(lg[s], w["logits"][i], f"slot {s} step {i}"))
        assert tp.residual_streams_identical(B)
    out = tp.ms[0].buf["out_tokens_all"][:, :new].tolist()
    for s, w in wants.items():
        assert out[s] == w["tokens"], (s, out[s], w["tokens"])
    assert all(m.buf["out_tokens_all"][:, :new].tolist() == out for m in tp.ms)
    assert len({tuple(t) for t in out}) > 1          # the prompts really differ
    print(f"[tp shards] W={world} config 5: worst decode logits error {worst:.2e}")
