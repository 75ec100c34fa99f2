"""Deterministic stand-in tokenizer for the synthetic-weight setting.

No ChatTS/Qwen tokenizer files exist offline, so prompts are tokenised by a self-contained scheme that
reproduces what matters for the hot path: realistic token COUNTS (Qwen2-style pre-tokenisation: words
with their leading space, single digits, punctuation runs) and the special tokens of the protocol as
single ids with ``id(<ts/>) == id(<ts>) + 1`` (NetManAIOps/ChatTS chatts/vllm/chatts_vllm.py:441).
Word pieces are mapped to ids by an FNV-1a hash into the non-special id range; single bytes map to
0..255.  ``decode`` inverts every piece this instance has seen and prints ``<|id|>`` otherwise.
Any HF tokenizer can be used instead: ChatTSProcessor only needs ``encode``/``decode``/``__call__``.
"""
import re

from . import config as _cfg

_SPECIAL = re.compile(r"(<\|im_start\|>|<\|im_end\|>|<\|endoftext\|>|<ts/>|<ts>)")
_PRE = re.compile(r"'s|'t|'re|'ve|'m|'ll|'d| ?[^\W\d_]+|\d| ?[^\s\w]+|_|\s+(?!\S)|\s+", re.UNICODE)


class SyntheticTokenizer:
    padding_side = "left"

    def __init__(self, vocab_size=152064, ts_start=_cfg.TS_START_ID, im_start=_cfg.IM_START_ID,
                 im_end=_cfg.IM_END_ID, eos=_cfg.EOS_ID):
        self.vocab_size = vocab_size
        self.special = {"<|endoftext|>": eos, "<|im_start|>": im_start, "<|im_end|>": im_end,
                        "<ts>": ts_start, "<ts/>": ts_start + 1}
        self.special_inv = {v: k for k, v in self.special.items()}
        self.word_lo, self.word_hi = 256, min(self.special.values())
        assert self.word_hi > self.word_lo and max(self.special.values()) < vocab_size
        self.pad_token_id = eos
        self.eos_token_id = im_end
        self._inv = {}
        self._ids = {}                             # piece -> id memo (a serving loop re-tokenises the same words)

    @classmethod
    def for_config(cls, cfg):
        e = cfg.eos_token_id if isinstance(cfg.eos_token_id, (list, tuple)) else [cfg.eos_token_id]
        im_end = e[0]
        eos = e[1] if len(e) > 1 else e[0]
        im_start = cfg.extra.get("im_start_token_id", _cfg.IM_START_ID) if hasattr(cfg, "extra") else _cfg.IM_START_ID
        return cls(cfg.vocab_size, cfg.ts_token_start_index, im_start, im_end, eos)

    def _piece_id(self, piece):
        tid = self._ids.get(piece)
        if tid is not None:
            return tid
        b = piece.encode("utf-8")
        if len(b) == 1:
            self._ids[piece] = b[0]
            return b[0]
        h = 0x811C9DC5
        for x in b:
            h = ((h ^ x) * 0x01000193) & 0xFFFFFFFF
        tid = self.word_lo + h % (self.word_hi - self.word_lo)
        self._inv.setdefault(tid, piece)
        if len(self._ids) < (1 << 20):
            self._ids[piece] = tid
        return tid

    def encode(self, text, add_special_tokens=False):
        ids = []
        for seg in _SPECIAL.split(text):          # specials first, so punctuation runs cannot swallow their '<'
            if seg in self.special:
                ids.append(self.special[seg])
            else:
                ids.extend(map(self._piece_id, _PRE.findall(seg)))   # _PRE has no groups: findall yields the whole matches
        return ids

    def convert_tokens_to_ids(self, tok):
        return self.special.get(tok, None) if isinstance(tok, str) else [self.convert_tokens_to_ids(t) for t in tok]

    def decode(self, ids, skip_special_tokens=False):
        out = []
        for t in ids:
            t = int(t)
            if t in self.special_inv:
                if not skip_special_tokens:
                    out.append(self.special_inv[t])
            elif t < 256:
                out.append(bytes([t]).decode("latin-1"))
            else:
                out.append(self._inv.get(t, f"<|{t}|>"))
        return "".join(out)

    def batch_decode(self, seqs, skip_special_tokens=False):
        return [self.decode(s, skip_special_tokens) for s in seqs]

    def __call__(self, text, padding=False, return_tensors=None, padding_side=None, **_):
        import torch
        single = isinstance(text, str)
        seqs = [self.encode(t) for t in ([text] if single else text)]
        side = padding_side or self.padding_side
        n = max(len(s) for s in seqs) if seqs else 0
        if padding or return_tensors:
            ids, mask = [], []
            for s in seqs:
                pad = n - len(s)
                if side == "left":
                    ids.append([self.pad_token_id] * pad + s); mask.append([0] * pad + [1] * len(s))
                else:
                    ids.append(s + [self.pad_token_id] * pad); mask.append([1] * len(s) + [0] * pad)
        else:
            ids, mask = seqs, [[1] * len(s) for s in seqs]
        if return_tensors == "pt":
            return {"input_ids": torch.tensor(ids, dtype=torch.long), "attention_mask": torch.tensor(mask, dtype=torch.long)}
        return {"input_ids": ids, "attention_mask": mask}
