"""The request scheduler (chatts_amd/engine.py) on a stub model: admission order, packed admission, sampling groups, the
block-pool gate of an oversubscribed paged KV cache, stop / length retirement, failed requests.  No GPU: the stub produces
token t = 1000 * request + step for the sequence in a slot, and keeps the same slot / pool bookkeeping the real model keeps."""
import pytest
import torch

from chatts_amd.engine import Engine
from chatts_amd.kv_blocks import BlockPool, KvPoolExhausted


class _Cfg:
    eos_token_id = [7]
    ts = {"patch_size": 16}


class _Tok:
    def encode(self, text):
        return [ord(c) for c in text]


class _Proc:
    tokenizer = _Tok()

    def splice(self, prompt, series):
        if "BAD" in prompt:
            raise ValueError("malformed request")
        return prompt, [], []

    def pad_stack(self, encs):
        return None


class StubModel:
    def __init__(self, max_batch, pool=None, pack=True, t_max=64):
        self.max_batch, self.config, self.t_max, self.pack = max_batch, _Cfg(), t_max, pack
        self.buf = {"pos_all": torch.zeros(max_batch, dtype=torch.int32), "out_tokens_all": torch.zeros((max_batch, 64), dtype=torch.int64),
                    "out_tokens": torch.zeros(64, dtype=torch.int64)}
        self.seq = [None] * max_batch                 # (request tag, steps produced) per slot
        self._kv = pool
        self._kv_dynamic = pool is not None
        self.sampling, self.log, self.steps = [], [], 0
        self.stop_at = {}                             # request tag -> step at which it emits the eos token

    # ---- what Engine calls ---------------------------------------------------------------------------------
    def set_sampling(self, *a):
        self.sampling.append(a)

    def _request_idents(self, ids, series, lens):
        return None

    def pick_slot(self, free, idents=None):
        return free[0]

    def request_tokens(self, ids, series=None, lens=None):
        return len(ids)

    def kv_fits(self, counts):
        return self._kv is None or self._kv.fits(counts)

    def kv_stats(self):
        return None if self._kv is None else {"free": len(self._kv.free)}

    def plan_pack(self, cands, free):
        if not self.pack or len(cands) < 2 or len(free) < 2:
            return []
        take, rows = [], 0
        for i, (ids, _, _, _) in enumerate(cands):
            if len(take) < len(free) and rows + len(ids) <= self.t_max:
                take.append(i); rows += len(ids)
        return take if len(take) >= 2 else []

    def _start(self, slot, ids, max_new):
        if self._kv is not None:
            self._kv.reserve(slot, len(ids) + max_new)
        tag = ids[0]
        self.seq[slot] = [tag, 1]
        self.buf["out_tokens_all"][slot, 0] = self._token(tag, 0)
        self.buf["pos_all"][slot] = len(ids)

    def _token(self, tag, step):
        return 7 if self.stop_at.get(tag) == step else 1000 * tag + step

    def _admit(self, slot, ids, series, lens, max_new):
        if len(ids) + max_new > 100:
            raise ValueError("exceeds max_ctx")
        self.log.append(("admit", slot, ids[0]))
        self._start(slot, ids, max_new)

    def admit_begin(self, slot, ids, series, lens, max_new):
        if len(ids) + max_new > 100:
            raise ValueError("exceeds max_ctx")
        if self._kv is not None:
            self._kv.reserve(slot, len(ids) + max_new)
        return {"slot": slot, "T": len(ids), "done": 0, "ids": ids, "max_new": max_new}

    def admit_step(self, st, max_rows=None):
        a = st["done"]
        b = st["T"] if max_rows is None else min(st["T"], a + max_rows)
        self.log.append(("chunk", st["slot"], a, b))
        st["done"] = b
        if b >= st["T"]:
            self.log.append(("admit", st["slot"], st["ids"][0]))
            tag = st["ids"][0]
            self.seq[st["slot"]] = [tag, 1]
            self.buf["out_tokens_all"][st["slot"], 0] = self._token(tag, 0)
            self.buf["pos_all"][st["slot"]] = st["T"]
            return True
        return False

    def _admit_packed(self, items):
        if self._kv is not None and not self._kv.fits([len(i[1]) + i[4] for i in items]):
            raise KvPoolExhausted("pack")
        self.log.append(("pack", [i[0] for i in items], [i[1][0] for i in items]))
        for slot, ids, _, _, max_new in items:
            self._start(slot, ids, max_new)

    def _prefill_request(self, ids, series, lens, max_new):
        self._admit(0, ids, series, lens, max_new)
        self.buf["out_tokens"][0] = self.buf["out_tokens_all"][0, 0]

    def batched_step(self):
        self.steps += 1
        for s, st in enumerate(self.seq):
            if st is not None and int(self.buf["pos_all"][s]) >= 0:
                self.buf["out_tokens_all"][s, st[1]] = self._token(st[0], st[1])
                st[1] += 1

    def decode_step(self):
        self.batched_step()
        self.buf["out_tokens"][:] = self.buf["out_tokens_all"][0]

    def note_generated(self, slot, tokens):
        self.log.append(("retire", slot, len(tokens)))
        self.seq[slot] = None
        if self._kv is not None:
            self._kv.retire(slot)


def _run(eng, reqs):
    out = {}
    for tag, n, kw in reqs:
        eng.add_request(chr(tag) * 10, max_tokens=n, on_tokens=lambda r, new, fin, tag=tag: out.setdefault(tag, []).extend(new), **kw)
    done = eng.run_until_done()
    return out, done


def test_continuous_batching_order_length_and_stop():
    m = StubModel(max_batch=2, pack=False)
    m.stop_at[66] = 2                                  # request 'B' stops at its third token
    eng = Engine(m, _Proc(), sync_every=2)
    out, done = _run(eng, [(65, 5, {}), (66, 9, {}), (67, 3, {})])
    assert out[65] == [65000 + i for i in range(5)]
    assert out[66] == [66000, 66001, 7]                # the eos token is delivered, nothing after it
    assert out[67] == [67000, 67001, 67002]
    assert [r.finish_reason for r in sorted(done, key=lambda r: r.rid)] == ["length", "stop", "length"]
    admits = [e for e in m.log if e[0] == "admit"]
    assert [a[2] for a in admits] == [65, 66, 67] and admits[2][1] in (0, 1)      # the third request took a freed slot
    assert all(s is None for s in eng.slots) and int(m.buf["pos_all"].max()) == -1   # every slot parked again


def test_packed_admission_and_single_slot_path():
    m = StubModel(max_batch=3, pack=True, t_max=25)    # two 10-token prompts fit one packed pass, the third waits its turn
    eng = Engine(m, _Proc())
    out, _ = _run(eng, [(65, 2, {}), (66, 2, {}), (67, 2, {})])
    assert m.log[0] == ("pack", [0, 1], [65, 66]) and ("admit", 2, 67) in m.log
    assert all(out[t] == [1000 * t, 1000 * t + 1] for t in (65, 66, 67))
    m1 = StubModel(max_batch=1)
    eng1 = Engine(m1, _Proc())
    out1, _ = _run(eng1, [(65, 3, {}), (66, 2, {})])
    assert out1[65] == [65000, 65001, 65002] and out1[66] == [66000, 66001]


def test_sampling_groups_do_not_mix():
    m = StubModel(max_batch=2, pack=False)
    eng = Engine(m, _Proc())
    out, _ = _run(eng, [(65, 3, {}), (66, 3, {"temperature": 0.5, "seed": 4}), (67, 3, {})])
    # greedy A and C share a batch; the sampled B waits until they drained, then the step is re-configured once
    order = [e[2] for e in m.log if e[0] == "admit"]
    assert order == [65, 67, 66]
    assert m.sampling[-1][0] == 0.5 and len(out[66]) == 3


def test_per_slot_sampling_lets_different_settings_decode_together():
    """A model with per-slot sampling arrays (set_slot_sampling / set_sampling_rows): no sampling groups - requests with different
    temperatures are admitted into one batch, every slot gets its request's settings before its first token, the step runs in per-row
    mode only while somebody samples, and seed=None gives every request its own stream."""
    class RowsModel(StubModel):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.slot_sampling, self.modes = {}, []

        def set_slot_sampling(self, slot, *a):
            self.slot_sampling[slot] = a

        def set_sampling_rows(self):
            self.modes.append("rows")

        def set_sampling(self, *a):
            self.modes.append("greedy" if not a[0] else a)

    m = RowsModel(3, pack=False)
    eng = Engine(m, _Proc(), sync_every=1)
    assert eng.mixed
    got = {}
    cb = lambda r, new, fin: got.setdefault(r.rid, []).extend(new)
    a = eng.add_request("A" * 10, max_tokens=4, on_tokens=cb)                                           # greedy
    b = eng.add_request("B" * 10, max_tokens=4, temperature=0.7, top_p=0.9, seed=5, on_tokens=cb)
    c = eng.add_request("C" * 10, max_tokens=4, temperature=0.2, top_k=40, on_tokens=cb)               # seed None
    eng.step()
    assert [r.rid for r in eng.slots] == [a.rid, b.rid, c.rid]                  # all three admitted together
    assert m.slot_sampling[0] == (0.0, 0, 1.0, 0) and m.slot_sampling[1] == (0.7, 0, 0.9, 5)
    assert m.slot_sampling[2][:3] == (0.2, 40, 1.0) and m.slot_sampling[2][3] not in (0, 5)
    assert m.modes == ["rows"]
    eng.run_until_done()
    assert all(len(got[r.rid]) == 4 for r in (a, b, c))
    d = eng.add_request("D" * 10, max_tokens=2, on_tokens=cb)      # nobody samples any more: the step STAYS in per-row mode (greedy rows take
    eng.run_until_done()                                            # the argmax token there; every switch would re-capture both graphs)
    assert m.modes == ["rows"] and len(got[d.rid]) == 2


def test_block_pool_gate_defers_then_admits():
    pool = BlockPool(n_blocks=3, block_size=64, n_slots=3, blocks_per_slot=4)
    m = StubModel(max_batch=3, pool=pool, pack=True, t_max=64)
    eng = Engine(m, _Proc(), sync_every=1)
    out, fin = {}, {}

    def add(tag, prompt_len, n):
        def cb(r, new, finished, tag=tag):
            out.setdefault(tag, []).extend(new)
            if finished:
                fin[tag] = r
        eng.add_request(chr(tag) * prompt_len, max_tokens=n, ignore_eos=True, on_tokens=cb)
    add(65, 10, 6)          # A: 1 block
    add(66, 10, 3)          # B: 1 block
    add(68, 30, 40)         # D: 70 positions = 2 blocks: must wait for B's block to become evictable
    add(67, 10, 2)          # C: 1 block, queued behind D
    add(69, 10, 95)         # E: 105 positions fit two blocks, but prompt + max_tokens > the stub's max_ctx: refused by _admit
    while eng.has_work():
        eng.step()
        assert pool.check()
    assert out[65] == [65000 + i for i in range(6)] and out[66] == [66000 + i for i in range(3)]
    assert out[68] == [68000 + i for i in range(40)] and out[67] == [67000, 67001]
    tags = []
    for e in m.log:                                                      # admission order, whatever mix of packed / single passes
        if e[0] == "admit":
            tags.append(e[2])
        elif e[0] == "pack":
            tags += e[2]
    assert tags == [65, 66, 68, 67]                                      # D before C: nobody overtakes the head of the queue
    d_admit = next(i for i, e in enumerate(m.log) if e[0] == "admit" and e[2] == 68)
    assert d_admit > m.log.index(("retire", 1, 3))                       # ... and D only after B had finished and freed a block
    assert fin[69].error is not None and fin[69].finish_reason == "error" and 69 not in tags
    assert not pool.active and pool.check()

    # a request larger than the whole pool is refused once nothing is left to wait for
    eng2 = Engine(StubModel(max_batch=2, pool=BlockPool(2, 64, 2, 4), pack=False), _Proc())
    got = {}
    eng2.add_request("Z" * 20, max_tokens=15, ignore_eos=True, on_tokens=lambda r, new, f: got.setdefault("ok", r))
    eng2.add_request("Y" * 130, max_tokens=10, ignore_eos=True, on_tokens=lambda r, new, f: got.setdefault("big", r))
    eng2.run_until_done()
    assert got["ok"].error is None and got["big"].error is not None and "block pool" in str(got["big"].error)


def test_chunked_prefill_interleaves_with_the_running_batch():
    m = StubModel(max_batch=2, pack=False)
    eng = Engine(m, _Proc(), sync_every=1, prefill_chunk_tokens=16)
    out = {}
    eng.add_request("A" * 10, max_tokens=12, ignore_eos=True, on_tokens=lambda r, new, f: out.setdefault("A", []).extend(new))
    eng.step()                                             # A is running
    eng.add_request("B" * 50, max_tokens=3, ignore_eos=True, on_tokens=lambda r, new, f: out.setdefault("B", []).extend(new))
    eng.add_request("C" * 10, max_tokens=2, ignore_eos=True, on_tokens=lambda r, new, f: out.setdefault("C", []).extend(new))
    steps_before = m.steps
    while eng.has_work():
        eng.step()
    chunks = [e for e in m.log if e[0] == "chunk" and e[1] == 1]
    assert [(c[2], c[3]) for c in chunks[:4]] == [(0, 16), (16, 32), (32, 48), (48, 50)]       # B came in 16 rows at a time
    # ... and A kept decoding in between: one batched step per scheduler iteration while B was still being prefilled
    first, last = m.log.index(chunks[0]), m.log.index(chunks[3])
    assert out["A"] == [65000 + i for i in range(12)] and out["B"] == [66000, 66001, 66002] and out["C"] == [67000, 67001]
    assert m.steps - steps_before >= 3 and last > first
    # C (short, arrived behind B) was not admitted while B's prefill was in flight
    assert m.log.index(("admit", 1, 66)) < next(i for i, e in enumerate(m.log) if e[0] == "admit" and e[2] == 67)
    # nothing else is running: a long prompt goes in with one call (no reason to chunk)
    m2 = StubModel(max_batch=2, pack=False)
    eng2 = Engine(m2, _Proc(), prefill_chunk_tokens=16)
    eng2.add_request("D" * 50, max_tokens=2, ignore_eos=True)
    eng2.run_until_done()
    assert not [e for e in m2.log if e[0] == "chunk"] or [e for e in m2.log if e[0] == "chunk"] == [("chunk", 0, 0, 50)]


def test_failed_chunk_and_pool_requeue():
    m = StubModel(max_batch=2, pack=False)
    eng = Engine(m, _Proc(), sync_every=1, prefill_chunk_tokens=16)
    got = {}
    eng.add_request("A" * 10, max_tokens=6, ignore_eos=True)
    eng.step()
    eng.add_request("B" * 50, max_tokens=3, ignore_eos=True, on_tokens=lambda r, new, f: got.setdefault("B", r))
    eng.step()
    assert eng.prefilling is not None
    def boom(st, max_rows=None):
        raise RuntimeError("device lost")
    m.admit_step = boom
    eng.step()
    assert eng.prefilling is None and got["B"].error is not None and got["B"].finish_reason == "error"
    eng.run_until_done()                                    # A still finishes
    # an admission that finds the pool exhausted after all goes back to the HEAD of the queue (deque), not to an error
    pool = BlockPool(n_blocks=2, block_size=64, n_slots=2, blocks_per_slot=2)
    m3 = StubModel(max_batch=2, pool=pool, pack=False)
    m3.kv_fits = lambda counts: True                        # the optimistic pre-check lets it through ...
    eng3 = Engine(m3, _Proc(), sync_every=1)
    order = []
    eng3.add_request("A" * 70, max_tokens=10, ignore_eos=True, on_tokens=lambda r, new, f: f and order.append("A"))
    eng3.add_request("B" * 70, max_tokens=10, ignore_eos=True, on_tokens=lambda r, new, f: f and order.append("B"))
    eng3.add_request("C" * 10, max_tokens=2, ignore_eos=True, on_tokens=lambda r, new, f: f and order.append("C"))
    eng3.run_until_done()                                   # ... B's reservation raises KvPoolExhausted while A runs: requeued
    assert order == ["A", "B", "C"] and pool.check()


def test_out_of_range_sampling_parameters_are_refused_before_the_engine():
    """{"temperature": -1} used to reach set_sampling inside step() and kill the engine thread for everybody."""
    from chatts_amd.engine import validate_sampling
    from chatts_amd.server import sampling_from_body
    for bad in ({"temperature": -1}, {"temperature": float("nan")}, {"top_p": 1.5}, {"top_p": 0}, {"top_p": float("nan")},
                {"top_k": -5}, {"max_tokens": -3}, {"temperature": "hot"}):
        with pytest.raises(ValueError):
            sampling_from_body(bad)
    ok = sampling_from_body({"temperature": 0.2, "top_p": 0.95, "top_k": -1, "max_tokens": 7})
    assert ok["temperature"] == 0.2 and ok["top_p"] == 0.95 and ok["max_tokens"] == 7
    validate_sampling(max_tokens=1, temperature=0.0, top_p=1.0, top_k=0)


def test_a_sampling_configuration_the_library_refuses_fails_only_that_request():
    model = StubModel(max_batch=2)
    eng = Engine(model, _Proc())

    def refuse(*a, **k):
        if a and a[0] == 0.123:
            raise ValueError("refused by the library")
    model.set_sampling = refuse
    bad = eng.add_request("p1", max_tokens=3, temperature=0.123)
    good = eng.add_request("p2", max_tokens=3)
    done = eng.run_until_done()
    assert bad in done and bad.error is not None and bad.finish_reason == "error"
    assert good in done and good.error is None and len(good.tokens) == 3
    with pytest.raises(ValueError):
        eng.add_request("p3", temperature=-1.0)


def test_head_of_the_queue_with_another_sampling_key_is_overtaken_a_bounded_number_of_times():
    """a request waiting for another sampling group could be starved by a steady stream of same-key arrivals"""
    model = StubModel(max_batch=2, pack=False)
    eng = Engine(model, _Proc(), sync_every=1)
    eng.max_bypass = 3
    eng.add_request("A", max_tokens=2)
    eng.step()
    b = eng.add_request("B", max_tokens=2, temperature=0.5)      # waits for the greedy batch to drain
    finished, fed = [], 0
    for _ in range(200):
        if fed < 40:                                             # a steady stream of greedy requests arriving after B
            eng.add_request(chr(ord("C") + fed % 20), max_tokens=2)
            fed += 1
        finished += [r.rid for r in eng.step()]
        if b.finished:
            break
    assert b.finished and b.error is None and 1 <= b.bypassed <= 3 + 1
    later = [r for r in finished if r > b.rid]
    assert len(later) <= 3 + 2                                   # only the bounded number of latecomers finished before B
