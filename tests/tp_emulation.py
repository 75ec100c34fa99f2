"""Tensor parallelism of ANY degree on ONE GPU, for the parity tests: the W rank-local shard models of one synthetic checkpoint
live in this process and run one after the other on one stream; the sum all-reduce after o_proj / down_proj is formed here, on
the device, in RANK ORDER (delta_0 + delta_1 + ...: the order of csrc/tp.hip's kernels), and the vocab-parallel token agreement is
an argmax over the concatenated logits slices (lowest id on ties = torch.argmax = chatts_tp_argmax's rule).

What this exercises is every kernel of a rank at the SHARD SHAPES of TP = W (qkv N = (n_q + 2 n_kv) 128 / W, gate_up N = 2 I / W,
down K = I / W, one kv head per rank at W = 8, vocab V / W, ...) against the unsharded float32 oracle.  What it does NOT exercise
is the exchange transport itself: that is tests/test_gpu_tp_p2p.py (two concurrent ranks in one process, real kernels) and
tools/tp_parity_worker.py (one PROCESS per rank, IPC-mapped buffers).  Reference: vLLM tensor_parallel_size=k
(NetManAIOps/ChatTS demo/demo_vllm.py:30, chatts/utils/llm_utils.py:154)."""
import torch

from chatts_amd import _lib
from chatts_amd.modeling import ChatTSForCausalLM
from chatts_amd.tp import LocalComm


class FakeComm(LocalComm):
    """rank / world of a shard without a process group (the emulation below plays the collectives)"""

    def __init__(self, rank, world):
        self.rank, self.world, self.group, self.dist = rank, world, None, None


class EmulatedTP:
    def __init__(self, cfg, world, seed, **kw):
        self.cfg, self.world = cfg, world
        self.ms = [ChatTSForCausalLM.from_synthetic(cfg, seed=seed, comm=FakeComm(r, world), **kw) for r in range(world)]
        self.lib = self.ms[0].lib

    # ---- the two collectives ------------------------------------------------------------------------------------------
    def _all_reduce_into_x(self, rows):
        total = self.ms[0].buf["delta"][:rows].clone()
        for m in self.ms[1:]:
            total += m.buf["delta"][:rows]               # rank order
        for m in self.ms:
            m.buf["x"][:rows] += total

    def residual_streams_identical(self, rows):
        x0 = self.ms[0].buf["x"][:rows]
        return all(torch.equal(x0, m.buf["x"][:rows]) for m in self.ms[1:])

    def full_logits(self, key="logits"):
        """vocab-parallel slices -> the full vocabulary (dim -1)"""
        return torch.cat([m.buf[key] for m in self.ms], dim=-1)

    # ---- single sequence (cache slot `slot`) ----------------------------------------------------------------------------
    def prefill(self, emb, pos0=0, slot=0):
        """chunked prefill of [T, H] embeddings through chatts_decoder_layer_part on every shard"""
        st = _lib.stream_ptr()
        T, t_max = emb.shape[0], self.ms[0].t_max
        for m in self.ms:
            m.select_sequence(slot)
        done = 0
        while done < T:
            n = min(t_max, T - done)
            for m in self.ms:
                m.buf["x"][:n].copy_(emb[done:done + n])
            for l in range(self.cfg.num_hidden_layers):
                for part in (0, 1):
                    for m in self.ms:
                        _lib.check(self.lib.chatts_decoder_layer_part(m._decoder, l, part, n, pos0 + done, None, 1, st))
                    self._all_reduce_into_x(n)
            done += n
        return n          # rows of the last chunk: its last row yields the next token

    def logits_of_row(self, row):
        st = _lib.stream_ptr()
        for m in self.ms:
            _lib.check(self.lib.chatts_decoder_logits(m._decoder, int(row), st))
        return self.full_logits()

    def first_token(self, last_rows, T):
        """logits of the last prompt row -> first token on every shard, decode-loop state set (single-sequence buffers)"""
        lg = self.logits_of_row(last_rows - 1)
        tok = int(torch.argmax(lg))
        for m in self.ms:
            B = m.buf
            B["pos"].fill_(T)
            B["step"].fill_(1)
            B["token"].fill_(tok)
            B["out_tokens"][0] = tok
            m._load_token_embedding()
        return tok, lg

    def decode_step(self):
        """one token: the batch-1 GEMVs / decode attention of every shard, exchanges and token agreement played here"""
        st = _lib.stream_ptr()
        for l in range(self.cfg.num_hidden_layers):
            for part in (0, 1):
                for m in self.ms:
                    _lib.check(self.lib.chatts_decoder_layer_part(m._decoder, l, part, 1, 0, _lib.ptr(m.buf["pos"]), m.n_splits, st))
                self._all_reduce_into_x(1)
        lg = self.logits_of_row(0)
        tok = int(torch.argmax(lg))
        for m in self.ms:
            B = m.buf
            step = int(B["step"].item())
            B["out_tokens"][step] = tok
            B["step"] += 1
            B["pos"] += 1
            B["token"].fill_(tok)
            m._load_token_embedding()
        return tok, lg

    # ---- batched decode (continuous batching under TP) ------------------------------------------------------------------
    def admit(self, slot, emb, T):
        """prefill a request into cache slot `slot` on every shard and produce its first token there"""
        last = self.prefill(emb, 0, slot)
        lg = self.logits_of_row(last - 1)
        tok = int(torch.argmax(lg))
        for m in self.ms:
            B = m.buf
            B["pos_all"][slot] = T
            B["step_all"][slot] = 1
            B["token_all"][slot] = tok
            B["out_tokens_all"][slot, 0] = tok
            m.select_sequence(0)
        return tok, lg

    def batched_step(self):
        """one token for every cache slot: M = max_batch weight-streaming GEMMs + per-sequence attention on every shard"""
        st = _lib.stream_ptr()
        m0 = self.ms[0]
        Bn, H = m0.max_batch, self.cfg.hidden_size
        for m in self.ms:
            B = m.buf
            _lib.check(self.lib.chatts_embed_token_batched(_lib.ptr(B["token_all"]), Bn, _lib.ptr(m._tensors["embed"]), 0,
                                                           self.cfg.vocab_size, H, _lib.ptr(B["x"]), st))
        for l in range(self.cfg.num_hidden_layers):
            for part in (0, 1):
                for m in self.ms:
                    _lib.check(self.lib.chatts_decoder_layer_part_batched(m._decoder, l, part, Bn, _lib.ptr(m.buf["pos_all"]),
                                                                          m._n_splits_batched(), st))
                self._all_reduce_into_x(Bn)
        for m in self.ms:
            _lib.check(self.lib.chatts_decoder_logits_batched(m._decoder, Bn, _lib.ptr(m.buf["logits_all"]), st))
        lg = self.full_logits("logits_all")                 # [B, V]
        toks = torch.argmax(lg, dim=1)
        for m in self.ms:
            B = m.buf
            steps = B["step_all"].to(torch.int64)
            B["out_tokens_all"].scatter_(1, steps[:, None], toks[:, None])
            B["step_all"] += 1
            B["pos_all"] += 1
            B["token_all"].copy_(toks)
        return toks.tolist(), lg
