"""The host side with a REAL Hugging Face fast tokenizer (not the synthetic stand-in): a byte-level BPE tokenizer with the
ChatML + <ts> / <ts/> special tokens is trained in memory, saved like a checkpoint's tokenizer files and loaded back the way
LLM(model=dir) does it.  Checks the processor's call surface (README.md:98-100), the placeholder protocol on real ids
(chatts_vllm.py:402-415) and the server's incremental detokeniser on multi-byte text.  No GPU."""
import json
import os

import numpy as np
import pytest

from chatts_amd import config as cfgmod
from chatts_amd.llm import _load_checkpoint_tokenizer
from chatts_amd.processing import ChatTSProcessor
from chatts_amd.server import IncrementalDecoder
from oracle import protocol
from tests.util import chat_prompt, random_walk_series

SPECIALS = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<ts>", "<ts/>"]


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    corpus = [chat_prompt([256, 64]), "I have 3 time series. Please analyze the local changes in these time series.",
              "[offset=-6.0772|scaling=3.6917|length=256|max=4.9979|min=-15.0000|left=0.0000|right=-8.2047]",
              "ünïcode text, 时间序列 异常 检测, numbers 0123456789 and punctuation ;:,.!?()[]|=-"] * 4
    trainer = trainers.BpeTrainer(vocab_size=700, special_tokens=SPECIALS, initial_alphabet=pre_tokenizers.ByteLevel.alphabet())
    tok.train_from_iterator(corpus, trainer)
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|im_end|>", pad_token="<|endoftext|>",
                                   additional_special_tokens=["<|im_start|>", "<ts>", "<ts/>"], padding_side="left")
    d = str(tmp_path_factory.mktemp("ckpt"))
    fast.save_pretrained(d)
    ts0 = fast.convert_tokens_to_ids("<ts>")
    assert fast.convert_tokens_to_ids("<ts/>") == ts0 + 1               # the protocol's pair of adjacent ids
    cfg = cfgmod.preset("tiny-qwen2", vocab_size=len(fast), ts_token_start_index=ts0,
                        eos_token_id=[fast.convert_tokens_to_ids("<|im_end|>"), fast.convert_tokens_to_ids("<|endoftext|>")])
    cfg.save_pretrained(d)
    with open(os.path.join(d, "config.json")) as f:
        c = json.load(f)
    c.pop("synthetic_tokenizer", None)                                  # this directory carries real tokenizer files
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(c, f)
    return d, cfg


def test_checkpoint_tokenizer_is_loaded_not_the_stand_in(ckpt, tmp_path):
    d, cfg = ckpt
    tok = _load_checkpoint_tokenizer(d)
    from chatts_amd.tokenizer import SyntheticTokenizer
    assert not isinstance(tok, SyntheticTokenizer) and hasattr(tok, "backend_tokenizer")      # a transformers fast tokenizer
    assert tok.convert_tokens_to_ids("<ts>") == cfg.ts_token_start_index
    # a checkpoint directory WITHOUT tokenizer files must not silently fall back to the synthetic tokenizer
    empty = tmp_path / "no_tok"
    empty.mkdir()
    (empty / "config.json").write_text(json.dumps({"model_type": "chatts"}))
    with pytest.raises(ValueError, match="tokenizer"):
        _load_checkpoint_tokenizer(str(empty))


def test_processor_surface_and_placeholder_protocol_with_hf_tokenizer(ckpt):
    d, cfg = ckpt
    tok = _load_checkpoint_tokenizer(d)
    proc = ChatTSProcessor.from_pretrained(cfg, tokenizer=tok)
    rng = np.random.default_rng(1)
    lengths = [[256, 33], [64]]
    series = [[random_walk_series(rng, L) for L in ls] for ls in lengths]
    flat = [s for ss in series for s in ss]
    out = proc(text=[chat_prompt(ls) for ls in lengths], timeseries=flat, padding=True, return_tensors="pt")
    ids, mask, ts = out["input_ids"], out["attention_mask"], out["timeseries"]
    assert ids.shape == mask.shape and ids.shape[0] == 2 and ts.shape == (3, 512, 1)
    assert mask[1, 0] == 0 and mask[1, -1] == 1 and mask[0].all()        # the shorter prompt is LEFT padded (generation)
    assert ids[1, 0] == tok.pad_token_id
    ts0 = cfg.ts_token_start_index
    for b, ls in enumerate(lengths):
        row = ids[b][mask[b].bool()].tolist()
        pairs = [i for i in range(len(row) - 1) if row[i] == ts0 and row[i + 1] == ts0 + 1]
        assert len(pairs) == len(ls)                                      # one <ts><ts/> pair per series, as single special ids
        full = protocol.expand_placeholders(row, [(L + 15) // 16 for L in ls], ts0).tolist()
        assert full.count(ts0) == sum((L + 15) // 16 for L in ls) and ts0 + 1 not in full
        text = tok.decode(row, skip_special_tokens=False)
        assert text.count("<ts><ts/>") == len(ls) and "[offset=" in text and f"length={ls[0]}" in text
    # the value-preserving prefix survives the round trip through a real BPE vocabulary, digit for digit
    one = proc(text=["x <ts><ts/> y"], timeseries=[np.array([1.0, 2.0, 30.0])], return_tensors="pt")
    assert "[offset=-11.0000|scaling=6.3333|length=3|max=30.0000|min=1.0000|left=1.0000|right=30.0000]" in tok.decode(one["input_ids"][0])
    assert proc.batch_decode(one["input_ids"], skip_special_tokens=True)[0].startswith("x [offset=")


def test_incremental_decoder_with_byte_level_bpe(ckpt):
    d, _ = ckpt
    tok = _load_checkpoint_tokenizer(d)
    text = "Spike at t=17: ünïcode, 时间序列 异常 检测 ✓ done."
    ids = tok.encode(text)
    dec = IncrementalDecoder(tok)
    pieces = [dec.push([t]) for t in ids[:-1]] + [dec.push([ids[-1]], final=True)]
    assert "".join(pieces) == tok.decode(ids, skip_special_tokens=True) == text
    assert all("�" not in p for p in pieces)                        # no half-decoded UTF-8 sequence was ever emitted
    assert any(p == "" for p in pieces)                                   # ... because incomplete pieces were held back
