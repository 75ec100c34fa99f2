"""CPU: host-side product logic (processor, tokenizer, protocol expansion, config, shard plan) and the
C-ABI library: it loads and exports every symbol include/chatts_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from chatts_amd import _lib, config as cfgmod, synth
from chatts_amd.processing import ChatTSProcessor, sp_normalise
from chatts_amd.tokenizer import SyntheticTokenizer
from chatts_amd.tp import ShardPlan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "chatts_amd.h")).read()
    declared = set(re.findall(r"\b(chatts_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"chatts_stream_t", "chatts_bf16"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.chatts_abi_version() == 9
    # host-only validation paths: negative codes + message, no device touched
    assert lib.chatts_linear(None, None) == _lib.E_BADARG
    assert b"null args" in lib.chatts_last_error()
    assert lib.chatts_ts_patchify(None, None) == _lib.E_BADARG


def test_embed_merge_count_mismatch_is_value_error():
    lib = _lib.load()
    ids = torch.tensor([1, 7, 7, 2], dtype=torch.int64)
    rc = lib.chatts_embed_merge(None, ids.data_ptr(), 4, None, 100, 64, ctypes.c_void_p(8), 3, 7, None, None, None, None)
    assert rc == _lib.E_COUNT_MISMATCH
    with pytest.raises(ValueError, match="Attempted to assign 3 multimodal tokens to 2 placeholders"):
        _lib.check(rc)


def test_processor_matches_reference_vectors(golden):
    g = golden("sp_encoding")
    proc = ChatTSProcessor(SyntheticTokenizer(), cfgmod.preset("chatts-14b"), prefix_format="sp")
    for i in range(int(g["n"])):
        enc, pfx, meta = proc.encode_series(g[f"in_{i}"])
        assert pfx == str(g[f"prompt_{i}"])
        assert np.array_equal(enc[0], g[f"enc_{i}"].astype(np.float32))
        assert meta["offset"] == float(g[f"offset_{i}"]) and meta["scale_factor"] == float(g[f"scale_{i}"])
        scaled, _, _ = sp_normalise(g[f"in_{i}"])
        assert np.array_equal(scaled, g[f"enc_{i}"][0::2, 0])           # float64 bit exact
    idx = g["batch_idx"]
    text, encs, lens = proc.splice(str(g["batch_prompt_in"]), [g[f"in_{i}"] for i in idx])
    assert text == str(g["batch_prompt_out"])
    assert np.array_equal(proc.pad_stack(encs), g["batch_arr"].astype(np.float32))
    assert lens == [len(g[f"in_{i}"]) for i in idx]


def test_processor_hf_prefix_known_answer():
    x = np.arange(256)
    ts1 = np.sin(x / 10) * 5.0
    ts1[100:] -= 10.0
    proc = ChatTSProcessor(SyntheticTokenizer(), cfgmod.preset("chatts-14b"))
    _, pfx, _ = proc.encode_series(ts1)     # /root/reference/demo/demo_lora.ipynb:147
    assert pfx == "[offset=6.0772|scaling=3.6917|length=256|max=4.9979|min=-15.0000|left=0.0000|right=-8.2047]<ts><ts/>"


def test_processor_call_surface():
    cfg = cfgmod.preset("chatts-14b")
    proc = ChatTSProcessor.from_pretrained(cfg)
    p1 = "<|im_start|>user\nA: <ts><ts/> B: <ts><ts/><|im_end|><|im_start|>assistant\n"
    p2 = "<|im_start|>user\nOnly text<|im_end|>"
    a, b = np.arange(40.0), np.ones(17)
    out = proc(text=[p1, p2], timeseries=[a, b], padding=True, return_tensors="pt")
    assert set(out) == {"input_ids", "attention_mask", "timeseries"}
    assert out["timeseries"].shape == (2, 80, 1) and out["timeseries"].dtype == torch.float32
    assert out["input_ids"].shape == out["attention_mask"].shape and out["input_ids"].shape[0] == 2
    assert out["attention_mask"][1, 0] == 0            # left padded
    assert proc.last_lengths == [40, 17] and proc.last_series_per_prompt == [2, 0]
    ts0 = cfg.ts_token_start_index
    row = out["input_ids"][0][out["attention_mask"][0].bool()].tolist()
    assert sum(1 for i in range(len(row) - 1) if row[i] == ts0 and row[i + 1] == ts0 + 1) == 2
    with pytest.raises(ValueError):
        proc(text=[p1], timeseries=[a])
    with pytest.raises(TypeError):
        proc(text=["<ts><ts/>"], timeseries=["not a series"])
    moved = out.to("cpu")
    assert moved["timeseries"].shape == (2, 80, 1)
    v = proc(text=[p1], timeseries=[a, b], vllm_flag=True)
    assert len(v["timeseries"]) == 2 and v["timeseries"][0][1].shape == (1, 80, 1)


def test_tokenizer_roundtrip_and_specials():
    t = SyntheticTokenizer()
    s = "<|im_start|>system\nYou are a helpful assistant.<|im_end|> TS1 is of length 256: <ts><ts/>; x=-3.25"
    ids = t.encode(s)
    assert t.decode(ids) == s
    assert t.special["<ts/>"] == t.special["<ts>"] + 1 == cfgmod.TS_END_ID
    assert all(0 <= i < t.vocab_size for i in ids)
    assert t.decode(ids, skip_special_tokens=True).count("<") == 0
    small = SyntheticTokenizer.for_config(cfgmod.preset("tiny-qwen2"))
    assert max(small.encode(s)) < cfgmod.preset("tiny-qwen2").vocab_size


def test_config_presets_and_param_counts():
    c14, c8 = cfgmod.preset("chatts-14b"), cfgmod.preset("chatts-8b")
    assert c14.param_counts()["layer"] == 275_268_608 - 0
    assert c14.decode_weight_bytes() == 2 * (48 * 275_268_608 + 778_567_680 + 5120)
    assert c8.qk_norm and not c8.attention_bias and c14.attention_bias and not c14.qk_norm
    assert c14.ts["patch_size"] == 16 and c14.ts["hidden_size"] == 5120
    assert c14.ts_token_end_index == c14.ts_token_start_index + 1
    rt = cfgmod.ChatTSConfig.from_dict(c8.to_dict())
    assert rt.to_dict() == c8.to_dict()
    n_ts = sum(s.rows * s.cols for s in synth.ts_encoder_specs(c14))
    assert n_ts == 106_406_928                                   # SURVEY.md section 2.1 K3 [PROBED]


def test_shard_plan_covers_everything_once():
    cfg = cfgmod.preset("chatts-14b")
    for world in (1, 2, 4, 8):
        plans = [ShardPlan(cfg, r, world) for r in range(world)]
        assert sum(p.nq for p in plans) == 40 and sum(p.nkv for p in plans) == 8
        assert [p.q0 for p in plans] == [r * 40 // world for r in range(world)]
        assert sum(p.inter for p in plans) == 13824 and all(p.inter % 16 == 0 for p in plans)
        assert sum(p.vocab for p in plans) == 152064 and plans[-1].v0 + plans[-1].vocab == 152064
    with pytest.raises(ValueError):
        ShardPlan(cfg, 0, 3)


def test_synth_key_and_spec_names():
    from oracle import synth as osynth
    for name in ("model.embed_tokens.weight", "ts_encoder.mlp.0.weight", "model.layers.47.mlp.down_proj.weight"):
        assert synth.tensor_key(3, name) == osynth.tensor_key(3, name)
    names = [s.name for s in synth.all_specs(cfgmod.preset("tiny-qwen3"))]
    assert "ts_encoder.position_embedding.weight" in names and "ts_encoder.mlp.8.bias" in names
    assert "model.layers.2.self_attn.q_norm.weight" in names and "lm_head.weight" in names
    assert not any("bias" in n for n in names if "self_attn" in n)
    a = osynth.materialize(synth.TensorSpec("t", 64, 512, 0.0, 13), 1)
    assert 0.015 < a.std() < 0.021 and abs(a.mean()) < 1e-3
    assert np.array_equal(a, a.astype(np.float32)) and np.all((a.view(np.uint32) & 0xFFFF) == 0)   # bf16 exact
    blk = osynth.bf16_bits(osynth.tensor_key(1, "t"), 0.0, 13, 8, 100, row0=5, col0=17, full_cols=512)
    full = osynth.bf16_bits(osynth.tensor_key(1, "t"), 0.0, 13, 64, 512)
    assert np.array_equal(blk, full[5:13, 17:117])


def test_auto_map_drop_in(tmp_path, monkeypatch):
    """The reference's UNMODIFIED loading cell (README.md:88-90) - AutoModelForCausalLM / AutoProcessor with trust_remote_code -
    resolves to this engine through the auto_map ChatTSConfig.save_pretrained writes (stock transformers does the resolving).
    No GPU here, so the model constructor is expected to refuse with chatts_amd's own 'no CPU fallback' error."""
    from transformers import AutoConfig, AutoModelForCausalLM, AutoProcessor
    monkeypatch.setenv("PYTHONPATH", ROOT)
    cfg = cfgmod.preset("tiny-qwen3")
    d = str(tmp_path / "ckpt")
    cfg.save_pretrained(d)
    back = cfgmod.ChatTSConfig.from_pretrained(d)
    assert back.model_type == "qwen3" and back.qk_norm and back.ts["patch_size"] == 16 and back.eos_token_id == cfg.eos_token_id
    ac = AutoConfig.from_pretrained(d, trust_remote_code=True)
    assert type(ac).__name__ == "ChatTSAmdConfig" and ac.ts["patch_size"] == 16          # model.config.ts['patch_size'] stays readable
    proc = AutoProcessor.from_pretrained(d, trust_remote_code=True, tokenizer=SyntheticTokenizer.for_config(cfg))
    assert isinstance(proc, ChatTSProcessor)
    out = proc(text=["a <ts><ts/> b"], timeseries=[np.arange(20.0)], return_tensors="pt")
    assert out["timeseries"].shape == (1, 40, 1)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="chatts_amd needs a ROCm GPU"):
            AutoModelForCausalLM.from_pretrained(d, trust_remote_code=True, device_map=0, dtype="float16")


def test_config_ts_key_aliases_and_scalar_eos():
    """ts.max_sequence_length wins over ts.max_length when both are present (chatts_vllm.py:68,76,245 reads both);
    a scalar eos_token_id (common in Qwen configs) is normalised to a list."""
    c = cfgmod.ChatTSConfig(ts={"max_length": 4096, "max_sequence_length": 8192}, eos_token_id=151645)
    assert c.ts["max_sequence_length"] == 8192 and c.eos_token_id == [151645]
    c = cfgmod.ChatTSConfig(ts={"max_length": 4096})
    assert c.ts["max_sequence_length"] == 4096


def test_decode_gemv_kernel_choice_is_pinned_by_shape():
    """Every TP = 1 projection of ChatTS-14B and ChatTS-8B runs the whole-K decode GEMV (the kernel the full-depth parity runs were
    recorded with; the K-split form sums a row in another order), the column-parallel projections of a TP = 8 rank take the K-split
    form (chatts_gemv_ksplit: the launcher's own decision function, evaluated on the host for 256 CUs)."""
    lib = _lib.load()
    NONE, RESID, SWIGLU = _lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU
    tp1 = {"14b": [(7168, 5120, NONE, 1), (5120, 5120, RESID, 0), (27648, 5120, SWIGLU, 1), (5120, 13824, RESID, 0), (152064, 5120, NONE, 1)],
           "8b": [(6144, 4096, NONE, 1), (4096, 4096, RESID, 0), (24576, 4096, SWIGLU, 1), (4096, 12288, RESID, 0), (151936, 4096, NONE, 1)]}
    for model, shapes in tp1.items():
        for n, k, epi, norm in shapes:
            assert lib.chatts_gemv_ksplit(n, k, epi, norm) == 1, (model, n, k)
    # one rank of TP = 8 (14B): qkv 896 x 5120 and gate_up 3456 x 5120 split K; the row-parallel o / down run the whole-K kernel
    # only when they carry the exchange (tp_reduce) - their shape alone would split, which the decoder never asks for under TP
    assert lib.chatts_gemv_ksplit(896, 5120, NONE, 1) > 1
    assert lib.chatts_gemv_ksplit(3456, 5120, SWIGLU, 1) > 1



def test_tuning_options_table_set_get_unset_and_the_environment_mirror(monkeypatch):
    """chatts_set_option / _unset_option / _get_option / _option_name (include/chatts_amd.h): the table is host code - every name the
    header documents exists, unknown names are refused with a message, a leading CHATTS_ is ignored, unset(NULL) clears everything, and
    the Python binding mirrors CHATTS_<NAME> variables into it (the library itself never reads the environment)."""
    lib = _lib.load()
    names = _lib.option_names()
    assert len(names) == len(set(names)) and "GEMM_SK" in names and "ATTN_EXACT" in names and "TP_BULK_FENCE" in names
    hdr = open(os.path.join(ROOT, "include", "chatts_amd.h")).read()
    doc = hdr[hdr.index("Names (a leading"):hdr.index("int chatts_set_option")]
    documented = set(re.findall(r"\b([A-Z][A-Z0-9]*(?:_[A-Z0-9]+)+)\b", doc)) - {"CHATTS_", "DESIGN"}
    probes = set()                                             # (the prefill kernel's ablation switch left the table with its probe: tools/probes/gemm_ring_probe.hip)
    assert documented - {"CHATTS_"} <= set(names), documented - set(names)
    assert set(names) - documented <= probes, set(names) - documented
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    sec = design[design.index("## 11. Tuning options"):design.index("## 12. ")]
    for n in set(names) - probes:                               # DESIGN.md section 11 says what each one selects
        assert n in sec or n.rsplit("_", 1)[0] + " / _" in sec or ("GEMV_" in n and "GEMV_ROWS / _UNR" in sec), n
    try:
        _lib.set_option("GEMM_SK", 3)
        assert _lib.get_option("GEMM_SK") == 3 and _lib.get_option("CHATTS_GEMM_SK") == 3
        _lib.set_option("CHATTS_GEMM_T", 7)
        assert _lib.get_option("GEMM_T") == 7
        _lib.set_option("GEMM_SK", None)
        assert _lib.get_option("GEMM_SK") is None and _lib.get_option("GEMM_T") == 7
        assert lib.chatts_set_option(b"NO_SUCH_KNOB", 1) == _lib.E_BADARG and b"NO_SUCH_KNOB" in lib.chatts_last_error()
        assert lib.chatts_set_option(None, 1) == _lib.E_BADARG
        v, isset = ctypes.c_int(5), ctypes.c_int(5)
        assert lib.chatts_get_option(b"NO_SUCH_KNOB", ctypes.byref(v), ctypes.byref(isset)) == _lib.E_BADARG
        assert lib.chatts_unset_option(None) == 0 and _lib.get_option("GEMM_T") is None
        with _lib.options(EPI_V4=0, ATTN_EXACT=0):
            assert _lib.get_option("EPI_V4") == 0 and _lib.get_option("ATTN_EXACT") == 0
        assert _lib.get_option("EPI_V4") is None and _lib.get_option("ATTN_EXACT") is None
        # the environment reaches the table only through sync_env (tests/conftest.py calls it on every monkeypatch.setenv)
        monkeypatch.setenv("CHATTS_TP_BULK_BLOCKS", "64")
        monkeypatch.setenv("CHATTS_GEMM_PRECISION", "bf16")   # alias of 1
        assert _lib.get_option("TP_BULK_BLOCKS") == 64 and _lib.get_option("GEMM_PRECISION") == 1
        monkeypatch.delenv("CHATTS_TP_BULK_BLOCKS")
        assert _lib.get_option("TP_BULK_BLOCKS") is None
        # ... and leaves options set programmatically alone (a model built with precision='bf16' keeps GEMM_PRECISION across a later
        # sync_env), while an option it mirrored itself follows its variable back to unset
        monkeypatch.delenv("CHATTS_GEMM_PRECISION")
        assert _lib.get_option("GEMM_PRECISION") is None
        _lib.set_option("ATTN_ROWS", 1)
        monkeypatch.setenv("CHATTS_GEMM_T", "5")
        assert _lib.get_option("ATTN_ROWS") == 1 and _lib.get_option("GEMM_T") == 5
        monkeypatch.delenv("CHATTS_GEMM_T")
        assert _lib.get_option("ATTN_ROWS") == 1 and _lib.get_option("GEMM_T") is None
        _lib.set_option("ATTN_ROWS", None)
        monkeypatch.setenv("CHATTS_GEMM_SK", "three")
    except RuntimeError as e:
        assert "integers" in str(e)
        monkeypatch.delenv("CHATTS_GEMM_SK")
    else:
        raise AssertionError("a non-integer option value must be refused")
    finally:
        lib.chatts_unset_option(None)


def test_c_abi_argument_validation_needs_no_gpu():
    """Everything the C-ABI checks BEFORE it enqueues a kernel is host code: null / mis-shaped arguments come back as error codes with a
    message, size queries are pure functions.  (Also what tests/test_asan_host.py drives under the host-AddressSanitizer build.)"""
    lib = _lib.load()
    la = _lib.LinearArgs()
    assert lib.chatts_linear(None, None) == _lib.E_BADARG and b"linear" in lib.chatts_last_error()
    la.m, la.n, la.k = 4, 0, 64
    assert lib.chatts_linear(ctypes.byref(la), None) < 0
    assert lib.chatts_tile_bf16_elems(798, 5120) == 800 * 5120 and lib.chatts_tile_bf16_elems(0, 64) == 0
    assert lib.chatts_tile_bf16(None, 16, 60, 64, None, None) == _lib.E_SHAPE
    assert lib.chatts_tile_bf16(None, 16, 64, 64, None, None) == _lib.E_BADARG
    assert lib.chatts_tile_bf16(None, 0, 64, 64, None, None) == 0                        # empty input: nothing to do
    assert lib.chatts_linear_f16q(None, None) == _lib.E_BADARG
    assert lib.chatts_split_f16q(None, 4, 100, 100, None, None, None, 100, 1, 0, None) == _lib.E_SHAPE
    assert lib.chatts_tp_buffer_bytes(9, 1024) == 0 and lib.chatts_tp_buffer_bytes(2, 1024) == 2 * 2 * 1024 * 8 + 256
    assert lib.chatts_tp_buffer_bytes_bulk(2, 1024, 4096) > lib.chatts_tp_buffer_bytes(2, 1024)
    assert lib.chatts_tp_bulk_release(None) == -1 and lib.chatts_tp_cross_device(None) == -1
    assert lib.chatts_tp_set_bulk_release(None, 0) == _lib.E_BADARG and b"tp_set_bulk_release" in lib.chatts_last_error()
    assert lib.chatts_tp_init(0, 2, None, None, 0, 64) is None and b"tp_init" in lib.chatts_last_error()
    assert lib.chatts_tp_rank(None) == -1 and lib.chatts_tp_pending(None) == 0
    assert lib.chatts_option_name(10 ** 6) is None and lib.chatts_option_name(-1) is None
