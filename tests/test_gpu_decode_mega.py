"""The persistent decode step (csrc/decode_mega.hip: one launch per token, weight stream running across projection boundaries)
against the multi-kernel schedule it replaces: the two must agree BIT FOR BIT - same row accumulation order, same RMSNorm
partial-sum geometry, same attention / combine functions - on logits and tokens, eagerly and as a replayed hipGraph, with the
contiguous and the block-paged cache, greedy and sampled; and against the CPU oracle like every other path."""
import numpy as np
import pytest
import torch

import bench
from chatts_amd import config as cfgmod, synth
from chatts_amd.modeling import ChatTSForCausalLM
from chatts_amd.processing import ChatTSProcessor
from oracle import pipeline, synth as osynth
from tests.util import chat_prompt, random_walk_series, rel_err

pytestmark = pytest.mark.gpu


def _inputs(cfg, lengths, seed=11):
    proc = ChatTSProcessor.from_pretrained(cfg)
    rng = np.random.default_rng(seed)
    series = [random_walk_series(rng, L) for L in lengths]
    inputs = proc(text=[chat_prompt(lengths)], timeseries=series, padding=True, return_tensors="pt")
    return proc, inputs


def _run(model, ids, ser, lens, new, graph):
    """tokens + the logits of every step (first token's included)"""
    model.use_graph = graph
    model._graph = None
    toks, l0 = model.generate_one(ids, ser, lens, new, eos_token_id=None, return_logits=True)
    model.generate_one(ids, ser, lens, 1, eos_token_id=None)
    steps = [l0.clone()]
    for _ in range(1, new):
        model.decode_step()
        steps.append(model.buf["logits"].clone())
    return toks, steps


@pytest.mark.parametrize("preset,kv_block", [("tiny-qwen2", None), ("tiny-qwen3", None), ("tiny-qwen2", 64)])
def test_persistent_step_is_bit_identical_to_the_multi_kernel_schedule(preset, kv_block):
    cfg = cfgmod.preset(preset)
    proc, inputs = _inputs(cfg, [100, 37, 64])
    ids = inputs["input_ids"][0].tolist()
    ser = inputs["timeseries"].cuda()
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=9, max_ctx=512, max_prefill_tokens=512, kv_block_size=kv_block,
                                             enable_prefix_caching=False)
    assert model.enable_persistent_decode(True), "the persistent step should be available for a TP = 1 bf16 model"
    new = 12
    runs = {}
    for mega in (True, False):
        assert model.enable_persistent_decode(mega) == mega
        for graph in (False, True):
            runs[(mega, graph)] = _run(model, ids, ser, proc.last_lengths, new, graph)
    assert model.enable_persistent_decode(True)
    assert model.persistent_decode_status() == 0
    ref_t, ref_l = runs[(False, False)]
    for key, (t, l) in runs.items():
        assert t == ref_t, key
        for i in range(new):
            assert torch.equal(l[i], ref_l[i]), (key, i, rel_err(l[i].cpu().numpy(), ref_l[i].cpu().numpy()))
    sd = osynth.state_dict(synth.all_specs(cfg), 9)
    want = pipeline.generate(cfg, sd, ids, inputs["timeseries"].numpy(), new)
    assert ref_t == want["tokens"]
    assert rel_err(ref_l[-1].cpu().numpy(), want["logits"][-1].numpy()) < 1e-3


def test_persistent_step_with_sampling_and_across_requests():
    """sampled decoding ends the launch at the logits (selection + embedding are the ordinary kernels): same draws as the
    multi-kernel schedule; and a second, different request on the same model reuses the attached state."""
    cfg = cfgmod.preset("tiny-qwen2")
    proc, inputs = _inputs(cfg, [80, 20])
    ids = inputs["input_ids"][0].tolist()
    ser = inputs["timeseries"].cuda()
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=4, max_ctx=512, max_prefill_tokens=512, enable_prefix_caching=False)
    out = {}
    for mega in (True, False):
        model.enable_persistent_decode(mega)
        model.set_sampling(0.8, 40, 0.95, seed=5)
        out[("s", mega)] = model.generate_one(ids, ser, proc.last_lengths, 10, eos_token_id=None)
        model.set_sampling(0.0)
        out[("g", mega)] = model.generate_one(ids, ser, proc.last_lengths, 10, eos_token_id=None)
        proc2, inputs2 = _inputs(cfg, [33], seed=5)
        out[("g2", mega)] = model.generate_one(inputs2["input_ids"][0].tolist(), inputs2["timeseries"].cuda(), proc2.last_lengths, 7,
                                               eos_token_id=None)
    for k in ("s", "g", "g2"):
        assert out[(k, True)] == out[(k, False)], k
    assert out[("s", True)] != out[("g", True)]
    assert model.persistent_decode_status() == 0


def test_persistent_step_full_width_14b_4_layers():
    """ChatTS-14B widths (the shapes the plan is tuned for: 14 / 10 / 54 / 10 / 297 row-pair tasks per workgroup), 4 layers, the
    bench prompt: bit-identical to the multi-kernel schedule, graph replay included; the multi-kernel schedule's oracle parity at this size is
    tests/test_gpu_parity_real_size.py."""
    cfg = cfgmod.preset("chatts-14b", num_hidden_layers=4)
    proc, prompt, series, lengths = bench.build_inputs(cfg, 8, 256)
    inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
    ids = inputs["input_ids"][0].tolist()
    ser = inputs["timeseries"].cuda()
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=0, max_ctx=2048, max_prefill_tokens=1024, enable_prefix_caching=False)
    assert model.enable_persistent_decode(True)
    runs = {}
    for mega in (True, False):
        model.enable_persistent_decode(mega)
        runs[mega] = _run(model, ids, ser, proc.last_lengths, 9, True)
    assert model.enable_persistent_decode(True) and model.persistent_decode_status() == 0
    assert runs[True][0] == runs[False][0]
    for i in range(9):
        assert torch.equal(runs[True][1][i], runs[False][1][i]), (i, rel_err(runs[True][1][i].cpu().numpy(), runs[False][1][i].cpu().numpy()))
