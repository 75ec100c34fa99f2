"""Request-level engine: incremental continuous batching over the model's KV-cache slots.

The role vLLM's LLMEngine / AsyncLLMEngine plays for the reference (NetManAIOps/ChatTS demo/demo_vllm.py:49-63 offline,
scripts/start_vllm_server.sh + chatts/utils/vllm_stream_qa.py:41-52 streaming): requests arrive at any time, are admitted
into free cache slots between decode steps, advance together in one batched step, and hand their new tokens back as
they are produced.  One engine owns one model (= one GPU, or one TP rank group); it is driven by a single thread
(`EngineThread`) because the HIP stream, the captured graphs and the slot state are not re-entrant.

Sampling: with a model that keeps per-slot sampling settings on the device (`set_slot_sampling`, ChattsSamplingArgs.*_rows) requests
with DIFFERENT temperature / top-k / top-p / seed decode together in one captured step - each slot's settings are written when the
request is admitted; the step selects greedily as long as nobody samples.  With a model that only has one baked-in configuration
(`set_sampling`) the requests decoded together share it: a request with a different one waits until the running batch has drained.
"""
import queue
import threading
import time
from collections import deque

import numpy as np


def validate_sampling(max_tokens=64, temperature=0.0, top_p=1.0, top_k=0, **_):
    """The ranges vLLM's SamplingParams accepts (what the reference's drivers pass: inference_tsmllm_vllm.py:43-46,
    llm_utils.py:153).  Raises ValueError - the server turns it into a 400 - BEFORE the request can reach the engine thread, where
    an out-of-range value would otherwise surface inside set_sampling / chatts_decoder_set_sampling."""
    import math
    t = 0.0 if temperature is None else float(temperature)
    if not math.isfinite(t) or t < 0.0:
        raise ValueError(f"temperature must be a finite number >= 0, got {temperature}")
    p = 1.0 if top_p is None else float(top_p)
    if not math.isfinite(p) or not (0.0 < p <= 1.0):
        raise ValueError(f"top_p must be in (0, 1], got {top_p}")
    if int(top_k or 0) < -1:
        raise ValueError(f"top_k must be -1 / 0 (off) or positive, got {top_k}")
    if int(max_tokens) < 1:
        raise ValueError(f"max_tokens must be at least 1, got {max_tokens}")


class ExchangeBroken(RuntimeError):
    """The tensor-parallel exchange timed out on THIS rank.  Fatal for the whole rank group: the ranks can no longer be assumed to hold
    the same engine state (a peer that arrived late saw no timeout and kept decoding), so the engine stops instead of serving
    on - EngineThread fails every request, releases the followers, and a supervisor restarts all ranks."""


class Request:
    def __init__(self, rid, prompt, timeseries, max_tokens, sampling_key, eos, on_tokens):
        self.rid, self.prompt, self.timeseries, self.max_tokens = rid, prompt, timeseries, int(max_tokens)
        self.sampling_key, self.eos, self.on_tokens = sampling_key, set(eos or ()), on_tokens
        self.tokens, self.sent, self.finished, self.finish_reason = [], 0, False, None
        self.prompt_tokens = 0
        self.t_arrival, self.t_first = time.perf_counter(), None
        self.error = None
        self.bypassed = 0                    # later arrivals admitted past this request while it waited for its sampling group


class Engine:
    """add_request() / step() like vLLM's LLMEngine.  `on_tokens(request, new_token_ids, finished)` is called from step()."""

    def __init__(self, model, processor, sync_every=4, prefill_chunk_tokens=None, greedy_return_steps=256):
        self.model, self.processor = model, processor
        # per-slot sampling: scheduler iterations without any sampling request (running or waiting) after which the step goes back from the
        # per-row sampler to the plain argmax tail (a graph re-capture); 0 = stay in per-row mode once entered.  Every rank of a
        # tensor-parallel group counts the same iterations (followers replay the leader's), so all of them switch together.
        self.greedy_return_steps = max(0, int(greedy_return_steps))
        self._greedy_iters = 0
        self.sync_every = max(1, int(sync_every))       # decode steps between device->host token reads
        # chunked-prefill scheduling (vLLM: enable_chunked_prefill / max_num_batched_tokens): while other sequences are decoding, a
        # long prompt is prefilled this many rows per scheduler iteration, each followed by one decode step of the running batch,
        # instead of stalling them for the whole prefill.  None: a prompt is always prefilled in one go.
        self.prefill_chunk_tokens = None if not prefill_chunk_tokens else max(16, int(prefill_chunk_tokens))
        self.prefilling = None                          # (request, slot, admission state) of the prompt being prefilled in chunks
        self.max_bypass = 8                             # admissions that may overtake a request waiting for another sampling group
        self.waiting = deque()
        self.nslots = max(1, model.max_batch)
        self.slots = [None] * self.nslots
        self.produced = [0] * self.nslots
        self.active_key = None
        self.mixed = hasattr(model, "set_slot_sampling")   # per-slot sampling settings on the device: no sampling groups
        self._since_sync = 0
        self._rid = 0
        if self.nslots > 1:
            model.buf["pos_all"].fill_(-1)            # every slot starts parked

    # ---- requests ---------------------------------------------------------------------------------
    def add_request(self, prompt, timeseries=None, max_tokens=64, temperature=0.0, top_p=1.0, top_k=0, seed=None,
                    stop_token_ids=None, ignore_eos=False, on_tokens=None):
        """seed None: per-slot mode gives every request its own stream (a hash of its arrival number - vLLM's seed=None means "not
        reproducible across requests"; here two identical prompts still differ, and a rerun of the same arrival order repeats);
        grouped mode uses 0 (requests that decode together must share ONE configuration there)."""
        validate_sampling(max_tokens, temperature, top_p, top_k)
        if seed is None:
            seed = ((self._rid + 1) * 2654435761) & 0xFFFFFFFF if self.mixed else 0
        e = self.model.config.eos_token_id
        eos = [] if ignore_eos else (list(e) if isinstance(e, (list, tuple)) else [e])
        eos += list(stop_token_ids or [])
        key = None if not temperature else (float(temperature), max(int(top_k or 0), 0), float(top_p), int(seed))
        self._rid += 1
        r = Request(self._rid, prompt, list(timeseries or []), max_tokens, key, eos, on_tokens)
        r.sampling = (0.0, 0, 1.0, 0) if key is None else key
        if self.mixed:
            r.sampling_key = "rows" if key is not None else None     # one group: who samples is a per-slot matter
        self.waiting.append(r)
        return r

    def has_work(self):
        return bool(self.waiting) or self.prefilling is not None or any(s is not None for s in self.slots)

    def _encode(self, r):
        import torch
        text, encs, lens = self.processor.splice(r.prompt, [np.asarray(s, dtype=np.float64) for s in r.timeseries])
        ids = self.processor.tokenizer.encode(text)
        ser = torch.from_numpy(self.processor.pad_stack(encs)) if encs else None
        return ids, ser, lens

    def _apply_sampling(self, key):
        if key is None:
            self.model.set_sampling(0.0)
        elif key == "rows":
            self.model.set_sampling_rows()
        else:
            self.model.set_sampling(key[0], key[1], key[2], key[3])
        self.active_key = key

    def _mixed_mode(self):
        """Per-slot mode: the step draws per row (greedy rows included) while any running or waiting request samples, and goes back
        to the plain argmax tail when nobody does.  Every slot's settings are current at all times, so the switch is safe whenever."""
        want = "rows" if (any(r is not None and r.sampling_key for r in self.slots) or
                          (self.prefilling is not None and self.prefilling[0].sampling_key) or
                          any(q.sampling_key for q in self.waiting)) else None
        if want is None and self.active_key == "rows":
            # Hysteresis instead of either extreme.  Every switch drops both captured hipGraphs (re-capture + a warm eager step on all
            # ranks), so alternating traffic must not flip the mode per request; but staying in per-row mode for good would leave later
            # greedy traffic on the sampler path (under TP: a full-vocabulary all-gather + the sampler kernel per step instead of the
            # 2-word tp_argmax).  So: back to the argmax tail after `greedy_return_steps` consecutive iterations without any sampling
            # request running or waiting (0 = never switch back).
            self._greedy_iters = getattr(self, "_greedy_iters", 0) + 1
            if not self.greedy_return_steps or self._greedy_iters < self.greedy_return_steps:
                return
        if want == "rows":
            self._greedy_iters = 0
        if want != self.active_key:
            self._apply_sampling(want)

    def _emit(self, r, finished, reason=None):
        new = r.tokens[r.sent:]
        r.sent = len(r.tokens)
        if finished:
            r.finished, r.finish_reason = True, reason
        if r.on_tokens is not None and (new or finished):
            r.on_tokens(r, new, finished)

    # ---- one scheduler iteration ---------------------------------------------------------------------
    def step(self):
        """Admit what fits, advance every running sequence by one token, deliver finished / newly read tokens.
        Returns the requests that finished in this call."""
        m = self.model
        done = []
        # (a prompt that is being prefilled in chunks draws its first token with the CURRENT configuration: it counts as running)
        running = any(s is not None for s in self.slots) or self.prefilling is not None
        if self.mixed:
            self._mixed_mode()
        while not self.mixed and not running and self.waiting and self.waiting[0].sampling_key != self.active_key:
            try:
                self._apply_sampling(self.waiting[0].sampling_key)
            except Exception as e:                       # a configuration the library refuses fails THAT request, not the engine
                r = self.waiting.popleft()
                r.error = e
                self._emit(r, True, "error")
                done.append(r)
        if self.prefilling is not None:                  # a long prompt is on its way in: its next chunk, no other admission meanwhile
            r, s, st = self.prefilling
            try:
                if m.admit_step(st, self.prefill_chunk_tokens):
                    self.prefilling = None
                    self.slots[s], self.produced[s] = r, 1
                    self._since_sync = self.sync_every   # read the first token right away
            except Exception as e:
                self.prefilling = None
                r.error = e
                self._emit(r, True, "error")
                done.append(r)
        # admission: free slots take waiting requests that share the running batch's sampling configuration
        while self.prefilling is None and self.waiting and any(v is None for v in self.slots):
            free = [i for i, v in enumerate(self.slots) if v is None]
            # requests that share the running batch's sampling configuration may overtake a head of the queue that waits for ANOTHER
            # one (it joins when the batch has drained) - but only `max_bypass` times, or a steady stream of same-key arrivals
            # would starve it
            head = self.waiting[0]
            if self.mixed:
                cands = list(self.waiting)[:len(free)]
            else:
                if head.sampling_key != self.active_key and head.bypassed >= self.max_bypass:
                    break
                cands = [q for q in self.waiting if q.sampling_key == self.active_key][:len(free)]
                if not cands:
                    break
                if head.sampling_key != self.active_key:
                    head.bypassed += len(cands)
            ready = []
            for r in cands:                              # tokenise / sp-encode once per request
                if getattr(r, "_enc", None) is None:
                    try:
                        r._enc = self._encode(r)
                        r.prompt_tokens = len(r._enc[0])
                    except Exception as e:               # a bad request must not take the engine down
                        self.waiting.remove(r)
                        r.error = e
                        self._emit(r, True, "error")
                        done.append(r)
                        continue
                ready.append(r)
            if not ready:
                continue
            pack = m.plan_pack([r._enc + (r.max_tokens,) for r in ready], free) if self.nslots > 1 else []
            group = [ready[j] for j in pack] if pack else ready[:1]
            if getattr(m, "_kv_dynamic", False):         # block-paged cache with an oversubscribed pool: admit only what it can hold
                def need(q):
                    return m.request_tokens(*q._enc) + q.max_tokens
                if len(group) > 1 and not m.kv_fits([need(q) for q in group]):
                    group = group[:1]
                if not m.kv_fits([need(group[0])]):
                    if any(v is not None for v in self.slots):
                        break                            # wait until a running sequence finishes: its blocks become evictable
                    r = group[0]                         # nothing is running: this request can never fit
                    self.waiting.remove(r)
                    r.error = RuntimeError(f"request needs {need(r)} KV positions, more than the block pool can hold ({m.kv_stats()})")
                    self._emit(r, True, "error")
                    done.append(r)
                    continue
            items = []
            for r in group:
                ids, ser, lens = r._enc
                s = m.pick_slot(free, m._request_idents(ids, ser, lens))      # the slot whose resident prefix matches best
                free.remove(s)
                items.append((s, ids, ser, lens, r.max_tokens))
                self.waiting.remove(r)
                if self.mixed:                           # the slot's settings, before its first token is drawn
                    m.set_slot_sampling(s, *r.sampling)
            try:
                if len(items) > 1:
                    m._admit_packed(items)               # several short prompts: one packed prefill pass
                elif self.nslots == 1:
                    m._prefill_request(*items[0][1:])
                elif self.prefill_chunk_tokens and any(v is not None for v in self.slots):
                    st = m.admit_begin(*items[0])         # others are decoding: feed a long prompt in chunks between their steps
                    if st["T"] - st["done"] > self.prefill_chunk_tokens:
                        self.prefilling = (group[0], items[0][0], st)
                        m.admit_step(st, self.prefill_chunk_tokens)
                        break
                    m.admit_step(st)
                else:
                    m._admit(*items[0])
            except Exception as e:
                self.prefilling = None
                if type(e).__name__ == "KvPoolExhausted" and any(v is not None for v in self.slots):
                    self.waiting.extendleft(reversed(group))      # the block pool is full (resident-prefix claims of the touched slots were dropped): back to the head
                    break                                         # of the queue until a running sequence finishes
                for r in group:
                    r.error = e
                    self._emit(r, True, "error")
                    done.append(r)
                continue
            for r, it in zip(group, items):
                self.slots[it[0]], self.produced[it[0]] = r, 1
            self._since_sync = self.sync_every          # read the first token right away (time to first token)
        live = [s for s in range(self.nslots) if self.slots[s] is not None]
        if not live:
            return done
        if self._since_sync >= self.sync_every or any(self.produced[s] >= self.slots[s].max_tokens for s in live):
            done += self._harvest()
            live = [s for s in range(self.nslots) if self.slots[s] is not None]
            if not live:
                return done
        if self.nslots == 1:
            m.decode_step()
        else:
            m.batched_step()
        for s in live:
            if self.produced[s] < self.slots[s].max_tokens:
                self.produced[s] += 1
        self._since_sync += 1
        return done

    def _harvest(self):
        """One device->host read of every slot's tokens; deliver the new ones; retire finished sequences."""
        m, B = self.model, self.model.buf
        done = []
        self._since_sync = 0
        toks_all = B["out_tokens_all"].cpu() if self.nslots > 1 else B["out_tokens"][None].cpu()
        tp = getattr(m, "_tp", None)
        if tp is not None and tp.status():
            # a peer's contribution did not arrive (csrc/tp.hip sets a sticky bit and every later collective sums garbage): the
            # tokens read above are worthless - fail what is running instead of serving them, and reset the exchange
            err = RuntimeError("tensor-parallel exchange timed out (a peer rank did not reach a collective): tokens discarded")
            for s in range(self.nslots):
                r = self.slots[s]
                if r is None:
                    continue
                self.slots[s] = None
                if self.nslots > 1:
                    B["pos_all"][s] = -1
                m._slot_idents[s] = []
                r.error = err
                self._emit(r, True, "error")
                done.append(r)
            # Each rank reads only ITS OWN sticky status: a peer that arrived late saw no timeout and would keep decoding these
            # slots while this rank hands them to other requests - slot choice and packing would differ between the ranks and every
            # later collective would be mismatched.  So this is fatal for the rank group, not for the running requests only: raise
            # out of step() (EngineThread._run fails everything, publishes stop to the followers; a follower's loop ends with the
            # exception) and let the supervisor restart all ranks.
            raise ExchangeBroken(str(err))
        for s in range(self.nslots):
            r = self.slots[s]
            if r is None:
                continue
            toks = toks_all[s, :self.produced[s]].tolist()
            cut = next((i + 1 for i, t in enumerate(toks) if t in r.eos), None) if r.eos else None
            if r.t_first is None:
                r.t_first = time.perf_counter()
            if cut is not None:
                r.tokens = toks[:cut]
                self._retire(s, "stop", done)
            elif self.produced[s] >= r.max_tokens:
                r.tokens = toks[:r.max_tokens]
                self._retire(s, "length", done)
            else:
                r.tokens = toks
                self._emit(r, False)
        return done

    def _retire(self, s, reason, done):
        r = self.slots[s]
        self.slots[s] = None
        if self.nslots > 1:
            self.model.buf["pos_all"][s] = -1         # parked: the batched step neither attends nor writes this slot's cache
        self.model.note_generated(s, r.tokens)
        self._emit(r, True, reason)
        done.append(r)

    def run_until_done(self):
        out = []
        while self.has_work():
            out += self.step()
        return out


class ControlPlane:
    """Tensor-parallel serving: rank 0 owns the HTTP front end, the other ranks follow.  Every rank's Engine must see the SAME
    sequence of add_request() / step() calls (slot choice, packing and harvest points are deterministic functions of it, and the
    decode steps rendezvous inside the exchange kernels), so before every engine iteration the leader publishes what it is about
    to do - the requests it admits now (usually none) or a shutdown - over a CPU (gloo) group: followers block there without
    spinning a GPU.  One small broadcast per iteration (a count), a pickled list only when there is something to say."""

    def __init__(self, rank, world, group=None, dist=None):
        self.rank, self.world, self.group = rank, world, group
        if dist is None:
            import torch.distributed as dist
        self.dist = dist

    @classmethod
    def create(cls):
        """collective: a gloo group over all ranks of the default process group"""
        import torch.distributed as dist
        group = dist.new_group(backend="gloo")
        return cls(dist.get_rank(), dist.get_world_size(), group, dist)

    def publish(self, new_requests, stop=False):
        """leader: new_requests = list of add_request kwargs (picklable: no callbacks)"""
        import torch
        head = torch.tensor([-1 if stop else len(new_requests)], dtype=torch.int64)
        self.dist.broadcast(head, src=0, group=self.group)
        if new_requests and not stop:
            self.dist.broadcast_object_list([new_requests], src=0, group=self.group)

    def receive(self):
        """follower: -> (list of add_request kwargs, stop)"""
        import torch
        head = torch.zeros(1, dtype=torch.int64)
        self.dist.broadcast(head, src=0, group=self.group)
        n = int(head.item())
        if n < 0:
            return [], True
        if n == 0:
            return [], False
        box = [None]
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0], False


def follow(engine, control):
    """The loop of a non-zero TP rank: replay the leader's admissions and step whenever it steps."""
    while True:
        new, stop = control.receive()
        if stop:
            return
        for kw in new:
            engine.add_request(**kw)
        if engine.has_work():
            engine.step()


class EngineThread:
    """Owns an Engine on one thread; other threads (the HTTP server's event loop) submit requests through a queue and get
    their tokens through the per-request callback.  With a ControlPlane (tensor parallel) this is the LEADER: every iteration is
    announced to the follower ranks first."""

    def __init__(self, engine, device=None, control=None):
        self.engine, self.device, self.control = engine, device, control
        self.inbox = queue.Queue()
        self.error = None
        self._stop = threading.Event()
        self.thread = threading.Thread(target=self._run, name="chatts-engine", daemon=True)
        self.thread.start()

    def submit(self, **kw):
        """Queue a request for the engine thread (kw: Engine.add_request arguments + `holder`, a list that receives the Request)."""
        if self.error is not None or not self.thread.is_alive():
            raise RuntimeError(f"engine thread is not running: {self.error}")
        self.inbox.put(kw)

    def _run(self):
        try:
            self._loop()
        except BaseException as e:           # the engine must never die silently: every waiting / later request is failed
            self.error = e
            import traceback
            traceback.print_exc()
            self._fail_all(e)
            if self.control is not None:     # the follower ranks block in control.receive(): release them
                try:
                    self.control.publish([], stop=True)
                except Exception:
                    pass

    def _fail_all(self, e):
        eng = self.engine
        pending = list(eng.waiting) + [r for r in eng.slots if r is not None]
        if getattr(eng, "prefilling", None) is not None:
            pending.append(eng.prefilling[0])
            eng.prefilling = None
        eng.waiting.clear()
        eng.slots = [None] * eng.nslots
        while True:
            try:
                kw = self.inbox.get_nowait()
            except queue.Empty:
                break
            holder = kw.pop("holder", None)
            r = eng.add_request(**kw)
            eng.waiting.clear()
            pending.append(r)
            if holder is not None:
                holder.append(r)
        for r in pending:
            r.error = RuntimeError(f"engine thread died: {type(e).__name__}: {e}")
            if r.on_tokens is not None:
                r.finished = True
                r.on_tokens(r, [], True)

    def _reject_invalid(self, new):
        """validate_sampling on every queued request; the offending ones are answered with an error here and never reach the engine
        (nor, under tensor parallelism, the followers - add_request raising after control.publish would take every rank down)."""
        ok = []
        for kw in new:
            try:
                validate_sampling(**{k: kw[k] for k in ("max_tokens", "temperature", "top_p", "top_k") if k in kw})
                ok.append(kw)
            except (ValueError, TypeError) as e:
                r = Request(-1, kw.get("prompt"), list(kw.get("timeseries") or []), 1, None, [], kw.get("on_tokens"))
                r.error, r.finished, r.finish_reason = e, True, "error"
                holder = kw.get("holder")
                if holder is not None:
                    holder.append(r)
                if r.on_tokens is not None:
                    r.on_tokens(r, [], True)
        return ok

    def _loop(self):
        import torch
        idx = getattr(self.device, "index", self.device) if self.device is not None else None
        if isinstance(idx, int):
            torch.cuda.set_device(idx)
        while not self._stop.is_set():
            new = []
            try:
                block = not self.engine.has_work()
                while True:
                    kw = self.inbox.get(timeout=0.05) if block else self.inbox.get_nowait()
                    new.append(kw)
                    block = False
            except queue.Empty:
                pass
            new = self._reject_invalid(new)        # BEFORE anything is announced: a bad value fails that request only, on no rank
            if self.control is not None:
                if not new and not self.engine.has_work():
                    continue                       # idle: nothing to announce, the followers keep waiting
                self.control.publish([{k: v for k, v in kw.items() if k not in ("holder", "on_tokens")} for kw in new])
            for kw in new:
                holder = kw.pop("holder", None)
                r = self.engine.add_request(**kw)
                if holder is not None:
                    holder.append(r)
            if self.engine.has_work():
                self.engine.step()
        if self.control is not None:
            self.control.publish([], stop=True)

    def close(self):
        self._stop.set()
        self.thread.join(timeout=5)
