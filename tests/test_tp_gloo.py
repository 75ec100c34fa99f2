"""CPU, world_size 2, gloo: the tensor-parallel PLAN and COMM of chatts_amd/tp.py.

Each rank evaluates its shard of the decoder with the oracle's float32 math (heads / MLP columns / vocabulary
sliced exactly as ShardPlan says), exchanges partial sums through tp.Comm.all_reduce and picks the greedy token
with tp.Comm.argmax_pair; rank 0 compares with the unsharded oracle.  (The HIP kernels themselves cannot run on
CPU; on the GPU the same plan is checked by tests/test_gpu_e2e.py::test_emulated_tensor_parallel_matches_tp1.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chatts_amd import config as cfgmod, synth
from chatts_amd.tp import Comm, ShardPlan
from oracle import qwen_decoder as qd, sampler as osamp, synth as osynth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sharded_forward(cfg, sd, plan, comm, x):
    """One pass of all layers on this rank's shard; returns the (replicated) final hidden states."""
    c = cfg.oracle_dict()
    d, eps, H = c["head_dim"], c["rms_norm_eps"], c["hidden_size"]
    T = x.shape[0]
    pos = torch.arange(T)
    cos, sin = qd.rope_cos_sin(pos, d, c["rope_theta"])
    for l in range(cfg.num_hidden_layers):
        p = f"model.layers.{l}."
        h = qd.rms_norm(x, sd[p + "input_layernorm.weight"], eps)
        qs, ks = slice(plan.q0 * d, (plan.q0 + plan.nq) * d), slice(plan.kv0 * d, (plan.kv0 + plan.nkv) * d)
        q = (h @ sd[p + "self_attn.q_proj.weight"][qs].T).view(T, plan.nq, d)
        k = (h @ sd[p + "self_attn.k_proj.weight"][ks].T).view(T, plan.nkv, d)
        v = (h @ sd[p + "self_attn.v_proj.weight"][ks].T).view(T, plan.nkv, d)
        if p + "self_attn.q_proj.bias" in sd:
            q = q + sd[p + "self_attn.q_proj.bias"][qs].view(plan.nq, d)
            k = k + sd[p + "self_attn.k_proj.bias"][ks].view(plan.nkv, d)
            v = v + sd[p + "self_attn.v_proj.bias"][ks].view(plan.nkv, d)
        if p + "self_attn.q_norm.weight" in sd:
            q = qd.rms_norm(q, sd[p + "self_attn.q_norm.weight"], eps)
            k = qd.rms_norm(k, sd[p + "self_attn.k_norm.weight"], eps)
        q = q * cos[:, None] + qd.rotate_half(q) * sin[:, None]
        k = k * cos[:, None] + qd.rotate_half(k) * sin[:, None]
        g = plan.nq // plan.nkv
        kk, vv = k.transpose(0, 1).repeat_interleave(g, 0), v.transpose(0, 1).repeat_interleave(g, 0)
        att = (q.transpose(0, 1) @ kk.transpose(1, 2)) / np.sqrt(d)
        att = att.masked_fill(torch.arange(T)[None, :] > pos[:, None], float("-inf"))
        o = (torch.softmax(att, -1) @ vv).transpose(0, 1).reshape(T, plan.nq * d)
        delta = o @ sd[p + "self_attn.o_proj.weight"][:, qs].T              # row-parallel partial sum
        x = x + comm.all_reduce(delta.contiguous())
        h = qd.rms_norm(x, sd[p + "post_attention_layernorm.weight"], eps)
        isl = slice(plan.i0, plan.i0 + plan.inter)
        act = torch.nn.functional.silu(h @ sd[p + "mlp.gate_proj.weight"][isl].T) * (h @ sd[p + "mlp.up_proj.weight"][isl].T)
        delta = act @ sd[p + "mlp.down_proj.weight"][:, isl].T
        x = x + comm.all_reduce(delta.contiguous())
    return qd.rms_norm(x, sd["model.norm.weight"], eps)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        cfg = cfgmod.preset("tiny-qwen3", num_hidden_layers=2)
        sd = osynth.state_dict(synth.decoder_specs(cfg), 4)
        comm = Comm()
        assert (comm.rank, comm.world) == (rank, world)
        plan = ShardPlan(cfg, comm.rank, comm.world)
        g = torch.Generator().manual_seed(0)
        x = torch.randn((9, cfg.hidden_size), generator=g) * 0.5
        h = _sharded_forward(cfg, sd, plan, comm, x)
        logits = h[-1] @ sd["lm_head.weight"][plan.v0:plan.v0 + plan.vocab].T        # vocab-parallel
        val, idx = logits.max(0)
        tok = comm.argmax_pair(val.reshape(1), (idx + plan.v0).reshape(1))
        # tie rule: equal maxima on both ranks -> the lowest token id wins
        tie = comm.argmax_pair(torch.tensor([1.0]), torch.tensor([100 + 7 * (world - rank)]))
        # sampling under TP: every rank gathers the vocab-parallel logits in rank order and draws from the full vocabulary
        # with the same (seed, sequence, step) -> the same token everywhere (here: the oracle's draw on the gathered row)
        full = comm.all_gather_cat(logits.contiguous())
        drawn = osamp.sample(full.numpy(), 0.5, 50, 0.95, seed=9, seq=0, step=3)
        comm.barrier()
        if rank == 0:
            ref = qd.QwenOracle(cfg.oracle_dict(), sd).forward_embeds(x)
            q.put((int(tok), int(torch.argmax(ref[-1])), float((h - _ref_hidden(cfg, sd, x)).abs().max()), int(tie)))
            q.put((float((full - ref[-1]).abs().max()), drawn, osamp.sample(ref[-1].numpy(), 0.5, 50, 0.95, seed=9, seq=0, step=3)))
            q.put(float(np.linalg.norm((logits - ref[-1][plan.v0:plan.v0 + plan.vocab]).numpy()) /
                        np.linalg.norm(ref[-1][plan.v0:plan.v0 + plan.vocab].numpy())))
    finally:
        dist.destroy_process_group()


def _ref_hidden(cfg, sd, x):
    return qd.QwenOracle(cfg.oracle_dict(), sd).forward_embeds(x, return_hidden=True)


def test_tensor_parallel_plan_and_comm_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    tok, ref_tok, hid_err, tie = q.get(timeout=10)
    gather_err, drawn, ref_drawn = q.get(timeout=10)
    rel = q.get(timeout=10)
    assert gather_err < 1e-4 and drawn == ref_drawn
    assert tok == ref_tok
    assert hid_err < 1e-4
    assert rel < 1e-5
    assert tie == 107            # min(100 + 7*2, 100 + 7*1): lowest id among equal maxima


# ---- tensor-parallel SERVING: the leader's engine iterations are replayed by the follower ranks -------------------------
class _ScriptEngine:
    """stands in for chatts_amd.engine.Engine: records the call sequence; a request lives for max_tokens steps"""

    def __init__(self):
        self.log, self.live = [], []

    def add_request(self, prompt, max_tokens=2, **kw):
        self.log.append(("add", prompt, max_tokens, tuple(sorted(kw))))
        self.live.append(max_tokens)

        class R:
            pass
        return R()

    def has_work(self):
        return bool(self.live)

    def step(self):
        self.log.append(("step", len(self.live)))
        self.live = [n - 1 for n in self.live if n > 1]


def _serve_worker(rank, world, port, q):
    import time
    from chatts_amd.engine import ControlPlane, EngineThread, follow
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        control = ControlPlane.create()
        eng = _ScriptEngine()
        if rank == 0:
            et = EngineThread(eng, control=control)
            et.submit(prompt="a <ts><ts/>", timeseries=[[1.0, 2.0]], max_tokens=3, temperature=0.5, on_tokens=lambda *a: None, holder=[])
            time.sleep(0.3)                       # an idle gap: nothing is announced, followers wait on the CPU
            et.submit(prompt="b", max_tokens=2, holder=[])
            et.submit(prompt="c", max_tokens=4, holder=[])
            deadline = time.time() + 20
            while (eng.has_work() or not et.inbox.empty() or len([e for e in eng.log if e[0] == "add"]) < 3) and time.time() < deadline:
                time.sleep(0.02)
            et.close()                            # publishes the shutdown
        else:
            follow(eng, control)
        q.put((rank, eng.log))
    finally:
        dist.destroy_process_group()


def test_tensor_parallel_serving_followers_replay_the_leader():
    """chatts_amd.engine.ControlPlane: rank 0 (HTTP front end + EngineThread) announces every engine iteration over a gloo
    group; the follower's Engine sees exactly the same add_request / step sequence - callbacks and holders stay on rank 0."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_serve_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    logs = dict(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    strip = lambda log: [e[:3] + (tuple(k for k in e[3] if k != "on_tokens"),) if e[0] == "add" else e for e in log]
    assert strip(logs[0]) == logs[1]              # same sequence; only the leader holds callbacks
    adds = [e for e in logs[1] if e[0] == "add"]
    assert [a[1] for a in adds] == ["a <ts><ts/>", "b", "c"]
    assert adds[0][3] == ("temperature", "timeseries")                                        # no callback crossed the wire
    assert sum(1 for e in logs[0] if e[0] == "step") >= 4


# ---- the release-form decision of the bulk sums (P2PExchange.first_contact), driven on CPU -----------------------------------------------
class _FakeExchange:
    """P2PExchange with its device-facing hooks replaced: the bulk sum is a gloo all-reduce, optionally corrupted under the light form on
    one rank, the status word and the library's knobs are Python attributes.  first_contact itself is the shipped method."""

    def __init__(self, comm, ident, corrupt_light_on=None, stall_on=None, forced=None, lib_cross=False):
        self.comm, self.ident, self.corrupt, self.stall, self.forced, self.lib_cross = comm, ident, corrupt_light_on, stall_on, forced, lib_cross
        self.bulk_elems, self.release_note, self.mode, self.cross_set, self.resets = 1 << 12, None, None, False, 0

    _tensor_device = "cpu"

    def device_identity(self):
        return self.ident

    def _set_cross_device(self):
        self.cross_set = True

    def _library_saw_cross_device(self):
        return self.lib_cross

    def _forced_release_option(self):
        return self.forced

    def _reset_after_stall(self):
        self.resets += 1

    def set_bulk_release(self, mode):
        self.mode = mode

    def bulk_release(self):
        return "fence" if (self.forced == 1 or (self.forced is None and self.mode == "fence")) else "light"

    def status(self):
        return 1 if self.stall == self.comm.rank else 0

    def all_reduce_bulk(self, inp, x):
        s = inp.clone()
        dist.all_reduce(s)
        if self.mode == "light" and self.corrupt == self.comm.rank:
            s[7] += 1.0                           # one element of one sum on one rank arrives wrong
        x += s
        return x


def _first_contact_worker(rank, world, port, q):
    from chatts_amd.tp import P2PExchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("CHATTS_TP_ASSUME_CROSS_DEVICE", "CHATTS_TP_INJECT_RELEASE_MISMATCH"):
        os.environ.pop(k, None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = Comm()
        out = {}
        cases = {"same_device": dict(ident=("host", "gpu0")),
                 "cross_ok": dict(ident=("host", f"gpu{rank}")),
                 "cross_light_wrong_on_rank1": dict(ident=("host", f"gpu{rank}"), corrupt_light_on=1),
                 "cross_peer_stalled_on_rank0": dict(ident=("host", f"gpu{rank}"), stall_on=0),
                 "forced_fence": dict(ident=("host", f"gpu{rank}"), forced=1),
                 "library_unsure": dict(ident=("host", "gpu0"), lib_cross=True)}
        for name, kw in cases.items():
            ex = _FakeExchange(comm, **kw)
            P2PExchange.first_contact(ex, comm, rounds=4)
            out[name] = (ex.release_note, ex.mode, ex.cross_set, ex.resets)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_first_contact_decides_the_release_form_identically_on_every_rank():
    """chatts_amd.tp.P2PExchange.first_contact (the guard every serving path runs when it builds its exchange): ranks that share a device
    keep the light release untested; ranks on different devices run the test sums and get the light form only if EVERY rank saw every
    sum exact under both forms - one wrong element on one rank, or one timed-out peer (then every rank also resets its buffers), leaves
    the fence on ALL ranks; TP_BULK_FENCE set wins; a library that could not place a peer buffer counts as cross-device."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_first_contact_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for name in res[0]:
        assert res[0][name][:2] == res[1][name][:2], (name, res[0][name], res[1][name])      # same verdict, same note, on both ranks
    r = res[0]
    assert r["same_device"][0] == "light (all ranks on one device)" and r["same_device"][1] is None and not r["same_device"][2]
    assert r["cross_ok"][0].startswith("light (validated at first contact: 4 sums") and r["cross_ok"][1] == "light" and r["cross_ok"][2]
    assert r["cross_light_wrong_on_rank1"][0].startswith("fence (first contact: 1 of 2 ranks") and r["cross_light_wrong_on_rank1"][1] == "fence"
    assert r["cross_peer_stalled_on_rank0"][1] == "fence" and res[0]["cross_peer_stalled_on_rank0"][3] == 1 and res[1]["cross_peer_stalled_on_rank0"][3] == 1
    assert r["forced_fence"][0] == "fence (TP_BULK_FENCE=1 set)" and r["forced_fence"][1] is None
    assert r["library_unsure"][0].startswith("light (validated") and not r["library_unsure"][2]
