"""Parity at the BASELINE.json configurations' REAL widths and prompt.

The exact bench prompt (bench.build_inputs: 8 series x 256 steps -> 798 tokens for ChatTS-14B; 1 x 256 for ChatTS-8B,
config 2) runs processor -> chatts_ts_encode -> merge -> chatts_decoder_prefill -> graph-replayed decode steps at full
width (H = 5120 / 4096, I = 13824 / 12288, V = 152k), so the kernels the benchmark times are the ones checked:
gemm_dma_kernel on bf16 planes, the key-split MFMA prefill attention, the fused split-K + post-norm epilogue, the
full-width GEMVs.  Depth is truncated to 4 layers so that the CPU float32 oracle finishes in seconds; the full-depth
(48-layer) comparison is the one-off job tools/parity_full_depth.py -> profiles/r2_parity_14b_full.json, which bench.py
checks its first tokens against.  Bar (BASELINE.json north_star): logits within 1e-3 relative, identical greedy tokens.
"""
import numpy as np
import pytest
import torch

import bench
from chatts_amd import config as cfgmod, synth
from chatts_amd.modeling import ChatTSForCausalLM
from oracle import from_device, pipeline, synth as osynth
from tests.util import rel_err

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3
NEW = 9           # first token + 8 graph-replayed decode steps


def _spot_check_hash(model, cfg, seed):
    """The device tensors ARE the hash-defined checkpoint: compare blocks of them with the host evaluation."""
    specs = {s.name: s for s in synth.all_specs(cfg)}
    s = specs["model.layers.1.mlp.up_proj.weight"]
    bits = osynth.bf16_bits(osynth.tensor_key(seed, s.name), 0.0, s.shift, 16, s.cols, row0=4096, full_cols=s.cols)
    want = (bits.astype(np.uint32) << 16).view(np.float32)
    got = from_device.layer_tensors(model, 1)["model.layers.1.mlp.up_proj.weight"][4096:4112].numpy()
    assert np.array_equal(want, got)
    s = specs["ts_encoder.mlp.4.weight"]
    bits = osynth.bf16_bits(osynth.tensor_key(seed, s.name), 0.0, s.shift, 8, s.cols, row0=1000, full_cols=s.cols)
    want = (bits.astype(np.uint32) << 16).view(np.float32)
    assert np.array_equal(want, from_device.ts_encoder_state_dict(model)["ts_encoder.mlp.4.weight"][1000:1008])


@pytest.mark.parametrize("preset,n_series,length,min_tokens", [("chatts-14b", 8, 256, 700), ("chatts-8b", 1, 256, 96),
                                                               ("chatts-8b", 8, 256, 700)])
def test_bench_prompt_full_width_4_layers_vs_oracle(preset, n_series, length, min_tokens):
    seed, depth = 0, 4
    cfg = cfgmod.preset(preset, num_hidden_layers=depth)
    proc, prompt, series, lengths = bench.build_inputs(cfg, n_series, length)
    inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
    ids = inputs["input_ids"][0].tolist()
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=seed, max_ctx=1024, max_prefill_tokens=1024)
    _spot_check_hash(model, cfg, seed)
    sd = {**from_device.ts_encoder_state_dict(model), **from_device.decoder_state_dict(model)}
    want = pipeline.generate(cfg, sd, ids, inputs["timeseries"].numpy(), NEW)
    T = len(want["expanded_ids"])
    assert T >= min_tokens                     # M >= 96: the LDS-DMA GEMM / MFMA attention chain, not the short-chunk kernels
    # stage boundaries at full width
    ser = inputs["timeseries"].cuda()
    mm = model.get_multimodal_embeddings(timeseries=ser, valid_lengths=proc.last_lengths)
    assert rel_err(torch.cat(mm).cpu().numpy(), want["ts_features"]) < 5e-5
    # generation exactly as bench.py drives it: prefill at once, first token, then hipGraph replays
    model.use_graph = True
    toks, logits0 = model.generate_one(ids, ser, proc.last_lengths, NEW, eos_token_id=None, return_logits=True)
    e0 = rel_err(logits0.cpu().numpy(), want["logits"][0].numpy())
    assert e0 < LOGIT_TOL, e0
    assert toks == want["tokens"]
    # per-step logits of the graph-replayed decode steps (tokens are identical, so the continuation is the oracle's)
    model.generate_one(ids, ser, proc.last_lengths, 1, eos_token_id=None)
    worst = e0
    for i in range(1, NEW):
        model.decode_step()
        e = rel_err(model.buf["logits"].cpu().numpy(), want["logits"][i].numpy())
        worst = max(worst, e)
        assert e < LOGIT_TOL, (i, e)
    assert worst < 2e-4, worst                 # what the f32 / bf16x2 design delivers at this depth


def _max_abs_over_max(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_config4_mixed_lengths_full_width_4_layers_vs_oracle():
    """BASELINE.json config 4 at ChatTS-14B widths: 30 series of mixed lengths 64..1024 (rng 1234, ragged tails forced; the
    envelope of NetManAIOps/ChatTS README.md:108) -> a > 2048-token prompt, i.e. THREE prefill chunks through the LDS-DMA GEMMs
    and the MFMA attention against a growing cache, then graph-replayed decode steps."""
    seed, depth = 0, 4
    cfg = cfgmod.preset("chatts-14b", num_hidden_layers=depth)
    proc, prompt, series, lengths = bench.build_inputs(cfg, 30, 256, "mixed")
    assert any(L % 16 for L in lengths) and min(lengths) >= 64 and max(lengths) <= 1024
    inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
    ids = inputs["input_ids"][0].tolist()
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=seed, max_ctx=4096, max_prefill_tokens=1024)
    sd = {**from_device.ts_encoder_state_dict(model), **from_device.decoder_state_dict(model)}
    want = pipeline.generate(cfg, sd, ids, inputs["timeseries"].numpy(), NEW)
    assert len(want["expanded_ids"]) > 2048
    ser = inputs["timeseries"].cuda()
    mm = model.get_multimodal_embeddings(timeseries=ser, valid_lengths=proc.last_lengths)
    assert rel_err(torch.cat(mm).cpu().numpy(), want["ts_features"]) < 5e-5          # ~1.1k patches: the M > 16 encoder GEMMs
    model.use_graph = True
    toks, logits0 = model.generate_one(ids, ser, proc.last_lengths, NEW, eos_token_id=None, return_logits=True)
    e0 = rel_err(logits0.cpu().numpy(), want["logits"][0].numpy())
    assert e0 < LOGIT_TOL and _max_abs_over_max(logits0.cpu().numpy(), want["logits"][0].numpy()) < LOGIT_TOL, e0
    assert toks == want["tokens"]
    model.generate_one(ids, ser, proc.last_lengths, 1, eos_token_id=None)
    for i in range(1, NEW):
        model.decode_step()
        e = rel_err(model.buf["logits"].cpu().numpy(), want["logits"][i].numpy())
        assert e < 2e-4, (i, e)


def test_config5_fp8_batch16_full_width_4_layers_vs_oracle():
    """BASELINE.json config 5 at ChatTS-14B widths: fp8 weights, 16 DIFFERENT prompts of 8 x 1024 steps (1207 tokens each)
    admitted the way bench.py --batch 16 admits them (plan_pack / _admit_packed where they fit, single admissions otherwise),
    then the batched decode graph (M = 16 fp8 weight-streaming GEMMs, per-sequence attention).  Slots 0, 7 and 15 (first / middle / last
    of the packed admissions; each oracle run of a 1207-token prompt costs ~9 s of CPU): identical tokens, every decode step's logits
    within 1e-3 (norm-wise and max-abs over max logit) of the oracle run on the dequantised weights; all 16 token rows differ pairwise
    where the prompts do (the full-depth, all-slot comparison is the committed run profiles/r3_parity_14b_8x1024_fp8_b16_full.json)."""
    seed, depth, B, new = 0, 4, 16, 5
    cfg = cfgmod.preset("chatts-14b", num_hidden_layers=depth)
    proc, prompt, reqs, lengths = bench.build_batched_requests(cfg, B, 8, 1024)
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=seed, max_ctx=2048, max_prefill_tokens=1024, weight_format="fp8", max_batch=B)
    assert "gate_up8" in model.layers[0]
    model.use_graph = True
    Bf = model.buf
    Bf["pos_all"].zero_(); Bf["step_all"].zero_(); Bf["token_all"].zero_()
    T, _, _ = bench.admit_batched(model, proc, prompt, reqs, new)
    assert T >= 1200
    step_logits = []
    for _ in range(1, new):
        model.batched_step()
        step_logits.append(Bf["logits_all"].cpu().numpy().copy())
    toks = Bf["out_tokens_all"][:, :new].tolist()
    # oracle on what the device holds: the bf16 tensors ARE the dequantised fp8 values
    sd = {**from_device.ts_encoder_state_dict(model), **from_device.decoder_state_dict(model)}
    worst = 0.0
    for s in (0, 7, 15):
        inp = proc(text=[prompt], timeseries=reqs[s], padding=True, return_tensors="pt")
        want = pipeline.generate(cfg, sd, inp["input_ids"][0].tolist(), inp["timeseries"].numpy(), new)
        assert toks[s] == want["tokens"], (s, toks[s], want["tokens"])
        for i in range(1, new):
            g, w = step_logits[i - 1][s], want["logits"][i].numpy()
            e = rel_err(g, w)
            worst = max(worst, e)
            assert e < LOGIT_TOL and _max_abs_over_max(g, w) < LOGIT_TOL, (s, i, e)
    assert len({tuple(t) for t in toks}) > 1          # the prompts really differ
    assert worst < 2e-4, worst


@pytest.mark.parametrize("preset", ["chatts-14b", "chatts-8b"])
def test_rope_in_the_qkv_epilogue_writes_the_same_bits(preset, monkeypatch):
    """The qkv projection's split-K epilogue that also rotates q / k and fills the cache (splitk_epilogue_rope_kernel, prefill) ==
    the separate epilogue + rope_kv_kernel launches: same K / V cache rows, same first-token logits, bit for bit (Qwen2 with
    bias, Qwen3 with q_norm / k_norm; 14B / 8B widths so that the projection really splits K)."""
    cfg = cfgmod.preset(preset, num_hidden_layers=2)
    proc, prompt, series, lengths = bench.build_inputs(cfg, 8, 256)
    inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
    ids = inputs["input_ids"][0].tolist()
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=3, max_ctx=1024, max_prefill_tokens=1024, enable_prefix_caching=False)
    ser = inputs["timeseries"].cuda()
    runs = []
    monkeypatch.setenv("CHATTS_GEMM_SK", "2")      # the 798-row chunk alone would run qkv unsplit (196 tiles): force the split both times
    for fuse in ("0", "1"):
        monkeypatch.setenv("CHATTS_ROPE_FUSE", fuse)
        model.buf["kv_k"].fill_(float("nan")); model.buf["kv_v"].fill_(float("nan"))
        toks, logits0 = model.generate_one(ids, ser, proc.last_lengths, 3, eos_token_id=None, return_logits=True)
        torch.cuda.synchronize()
        runs.append((toks, logits0.clone(), model.buf["kv_k"].clone(), model.buf["kv_v"].clone()))
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1])
    for a, b in ((runs[0][2], runs[1][2]), (runs[0][3], runs[1][3])):
        assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
    assert not torch.isnan(runs[1][2][..., :700, :]).all()
