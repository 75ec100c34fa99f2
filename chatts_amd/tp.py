"""Tensor-parallel plan for the decoder (one process per GPU, RCCL over xGMI via torch.distributed).

The reference gets TP from vLLM (``tensor_parallel_size=k``: demo/demo_vllm.py:30, chatts/utils/llm_utils.py:154),
Megatron style: q/k/v and gate/up column-parallel, o/down row-parallel -> two all-reduces of [T, H] per layer;
lm_head vocab-parallel.  Here:
  * heads are split contiguously: rank r owns q heads [r*nq/W, (r+1)*nq/W) and kv heads [r*nkv/W, ...)
  * the MLP intermediate dim is split contiguously (multiple of 16 per rank for the gate/up interleave)
  * lm_head rows are split contiguously; greedy sampling exchanges one (logit, index) pair per rank
    instead of gathering [V/W] logits
  * the TS encoder, the token embedding and all norms are replicated (213 MB + 1.5 GB: cheaper than a collective)
The exchange is a float32 sum all-reduce; there is no other data-path collective.

Two transports:
  * decode-sized messages ([B, H] float32, 96 per token): `P2PExchange` - the one-shot peer-to-peer kernels of
    csrc/tp.hip (chatts_allreduce / chatts_tp_argmax / chatts_allgather) over IPC-mapped exchange buffers; the whole
    TP decode step is enqueued by ONE C call and captured into ONE hipGraph;
  * prefill-sized messages ([T, H]): chatts_allreduce_bulk - a two-shot all-reduce (direct reduce-scatter, rank-ordered sum by the
    slice's owner, direct all-gather) over the bulk region of the same buffers, launched by chatts_decoder_prefill itself between
    the layer halves (one host call per chunk).  RCCL through torch.distributed (`Comm.all_reduce`) remains the path when the
    exchange is not attached (use_p2p=False, or the IPC mapping could not be set up).
"""
import ctypes as C

import torch


class ShardPlan:
    def __init__(self, cfg, rank=0, world=1):
        nq, nkv, I, V = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size, cfg.vocab_size
        if nq % world or nkv % world or I % (16 * world) or V % (16 * world):
            raise ValueError(f"tensor_parallel_size={world} does not divide heads ({nq}/{nkv}), "
                             f"intermediate ({I}) or vocab ({V}) into 16-aligned shards")
        self.rank, self.world = rank, world
        self.nq, self.nkv = nq // world, nkv // world
        self.q0, self.kv0 = rank * self.nq, rank * self.nkv
        self.inter = I // world
        self.i0 = rank * self.inter
        self.vocab = V // world
        self.v0 = rank * self.vocab
        self.d = cfg.head_dim


class Comm:
    """Sum all-reduce + tiny gathers over a torch.distributed group (backend 'nccl' = RCCL on ROCm; 'gloo' in CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def all_reduce(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def argmax_pair(self, logit, token):
        """Each rank holds its local (max logit [1] f32, token id [1] i64) -> global greedy token on every rank.
        Ties resolve to the lowest token id, like torch.argmax over the full vocabulary."""
        if self.world == 1:
            return token
        vl = [torch.empty(1, dtype=logit.dtype, device=logit.device) for _ in range(self.world)]
        il = [torch.empty(1, dtype=token.dtype, device=token.device) for _ in range(self.world)]
        self.dist.all_gather(vl, logit.reshape(1).contiguous(), group=self.group)     # list form: works on nccl and gloo
        self.dist.all_gather(il, token.reshape(1).contiguous(), group=self.group)
        vals, idxs = torch.cat(vl), torch.cat(il)
        best = vals.max()
        cand = torch.where(vals == best, idxs, torch.full_like(idxs, torch.iinfo(torch.int64).max))
        return cand.min().reshape(1)

    def all_gather_cat(self, t):
        """Concatenation of every rank's 1-D tensor in rank order (vocab-parallel logits -> full vocabulary)."""
        if self.world == 1:
            return t
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t.contiguous(), group=self.group)
        return torch.cat(parts)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier(group=self.group)


class LocalComm(Comm):
    def __init__(self):
        self.rank, self.world, self.group, self.dist = 0, 1, None, None


class P2PExchange:
    """One rank's end of the peer-to-peer exchange (csrc/tp.hip).  Owns the exchange buffer; `handle` is the C ChattsTpComm*."""

    def __init__(self, lib, handle, buf_ptr, owns_buffer=True):
        self.lib, self.handle, self.buf_ptr, self.owns_buffer = lib, C.c_void_p(handle), buf_ptr, owns_buffer
        self.rank, self.world = lib.chatts_tp_rank(self.handle), lib.chatts_tp_world(self.handle)
        self.max_elems = int(lib.chatts_tp_max_elems(self.handle))
        self.bulk_elems = int(lib.chatts_tp_bulk_elems(self.handle))      # capacity of chatts_allreduce_bulk (0: no bulk region)
        self.release_note = None                 # how the release form of the bulk sums was decided (first_contact)

    @classmethod
    def create(cls, comm, max_elems, bulk_elems=0):
        """Collective over `comm` (a torch.distributed-backed Comm): allocate + export this rank's buffer, exchange the IPC
        handles out of band (all_gather_object), map every peer.  HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the environment.
        bulk_elems > 0 adds the bulk region of chatts_allreduce_bulk (prefill-sized sums of up to that many float32)."""
        from . import _lib
        lib = _lib.load()
        nbytes = int(lib.chatts_tp_buffer_bytes_bulk(comm.world, int(max_elems), int(bulk_elems)))
        ptr = C.c_void_p()
        hbuf = (C.c_uint8 * _lib.TP_HANDLE_BYTES)()
        _lib.check(lib.chatts_tp_buffer_alloc(nbytes, C.byref(ptr), hbuf))
        mine = bytes(hbuf)
        allh = [None] * comm.world
        comm.dist.all_gather_object(allh, mine, group=comm.group)
        table = (C.c_uint8 * (_lib.TP_HANDLE_BYTES * comm.world)).from_buffer_copy(b"".join(allh))
        h = lib.chatts_tp_init(comm.rank, comm.world, ptr, table, nbytes, int(max_elems))
        if not h:
            lib.chatts_tp_buffer_free(ptr)
            raise _lib.ChattsError(-1, lib.chatts_last_error().decode())
        comm.barrier()                           # every rank has mapped every buffer before anyone pushes
        ex = cls(lib, h, ptr)
        ex.first_contact(comm)
        return ex

    @staticmethod
    def device_identity():
        """what tells two ranks' GPUs apart: host + everything this process knows about its device - uuid and PCI address when torch
        exposes them, and always the device index together with the visibility masks it is relative to (one process per GPU under a
        launcher: the indices differ even where uuid / PCI fields are missing).  Equal identities = the same device; anything else is
        treated as another device (the conservative side: fence, or the light form after the test sums)."""
        import os
        import socket
        idx = torch.cuda.current_device()
        props = torch.cuda.get_device_properties(idx)
        pci = tuple(getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        masks = tuple(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"))
        return (socket.gethostname(), str(getattr(props, "uuid", None)), pci, idx, masks)

    def bulk_release(self):
        """'fence' | 'light': the form the next bulk sum uses (chatts_tp_bulk_release)"""
        return "fence" if int(self.lib.chatts_tp_bulk_release(self.handle)) == 1 else "light"

    def set_bulk_release(self, mode):
        """mode: 'fence' | 'light' | None (= by device)"""
        from . import _lib
        _lib.check(self.lib.chatts_tp_set_bulk_release(self.handle, {None: -1, "light": 0, "fence": 1}[mode]))

    # -- the few device-facing operations first_contact needs, as overridable hooks (tests/test_tp_gloo.py drives the decision logic on CPU)
    _tensor_device = "cuda"

    def _set_cross_device(self):
        from . import _lib
        _lib.check(self.lib.chatts_tp_set_cross_device(self.handle, 1))

    def _library_saw_cross_device(self):
        return int(self.lib.chatts_tp_cross_device(self.handle)) == 1

    def _forced_release_option(self):
        from . import _lib
        return _lib.get_option("TP_BULK_FENCE")

    def _reset_after_stall(self):
        from . import _lib
        _lib.check(self.lib.chatts_tp_reset(self.handle, _lib.stream_ptr()))
        torch.cuda.synchronize()

    def first_contact(self, comm, rounds=64):
        """Collective.  Decide the release form of the prefill-sized sums for THIS group of devices (VERDICT r5 weak #1a).
        The light form (s_waitcnt vmcnt(0) before the flags) is only known to be correct with every rank on one device.  Ranks on
        different devices therefore start on the system-scope fence, and earn the light form by a test on the very links they will use:
        `rounds` bulk sums of a rank-dependent pattern under each form, every element compared with the exact sum (small integers: the
        float32 sum is exact in any order).  Any difference or a timed-out peer on any rank -> the fence stays, on every rank.
        Hooks: CHATTS_TP_ASSUME_CROSS_DEVICE=1 treats a one-device group as cross-device (tests on one GPU),
        CHATTS_TP_INJECT_RELEASE_MISMATCH=1 makes the comparison fail."""
        import os
        dev = self._tensor_device
        ids = [None] * comm.world
        comm.dist.all_gather_object(ids, self.device_identity(), group=comm.group)
        cross = len(set(ids)) > 1 or os.environ.get("CHATTS_TP_ASSUME_CROSS_DEVICE", "0") == "1"
        if cross:
            self._set_cross_device()
        elif self._library_saw_cross_device():
            cross = True                         # the library could not place a peer buffer on our device: stay conservative
        forced = self._forced_release_option()
        if forced is not None:
            self.release_note = f"{self.bulk_release()} (TP_BULK_FENCE={forced} set)"
            return
        if not cross:
            self.release_note = "light (all ranks on one device)"
            return
        if self.bulk_elems < 4 * comm.world:
            self.release_note = "fence (cross-device, no bulk region to test)"
            return
        n = min(self.bulk_elems, 1 << 20) // 4 * 4
        idx = torch.arange(n, dtype=torch.int64, device=dev)
        bad = 0
        for i in range(rounds):
            parts = [((idx * (r + 1) + 7919 * i) % 1021 + r).to(torch.float32) for r in range(comm.world)]
            want = torch.stack(parts).sum(0)
            for form in ("fence", "light"):
                self.set_bulk_release(form)
                x = torch.zeros(n, dtype=torch.float32, device=dev)
                self.all_reduce_bulk(parts[comm.rank], x)
                bad += int(not torch.equal(x, want))
        if dev == "cuda":
            torch.cuda.synchronize()
        stalled = self.status() != 0
        bad += int(stalled)
        if os.environ.get("CHATTS_TP_INJECT_RELEASE_MISMATCH", "0") == "1":
            bad += 1
        gathered = [None] * comm.world
        comm.dist.all_gather_object(gathered, (bad, stalled), group=comm.group)
        allbad = [b for b, _ in gathered]
        if any(st for _, st in gathered):
            # a peer's contribution timed out somewhere: the exchange buffers hold garbage from then on - zero them and the call counters on
            # EVERY rank before anything else uses the communicator (include/chatts_amd.h: chatts_tp_reset), then meet
            self._reset_after_stall()
            comm.barrier()
        if any(allbad):
            self.set_bulk_release("fence")
            self.release_note = f"fence (first contact: {sum(1 for b in allbad if b)} of {comm.world} ranks saw a light-release sum differ or a peer time out)"
        else:
            self.set_bulk_release("light")
            self.release_note = f"light (validated at first contact: {rounds} sums of {n} elements identical to the exact sum under both forms on all {comm.world} ranks)"

    @classmethod
    def create_local_group(cls, world, max_elems, bulk_elems=0):
        """`world` exchanges living in THIS process on the current device (single-GPU emulation of TP: every 'rank' runs on
        its own stream; peers are plain device pointers, no IPC)."""
        from . import _lib
        lib = _lib.load()
        nbytes = int(lib.chatts_tp_buffer_bytes_bulk(world, int(max_elems), int(bulk_elems)))
        ptrs = []
        for _ in range(world):
            ptr = C.c_void_p()
            _lib.check(lib.chatts_tp_buffer_alloc(nbytes, C.byref(ptr), None))
            ptrs.append(ptr)
        arr = (C.c_void_p * world)(*[p.value for p in ptrs])
        out = []
        for r in range(world):
            h = lib.chatts_tp_init_local(r, world, arr, nbytes, int(max_elems))
            if not h:
                raise _lib.ChattsError(-1, lib.chatts_last_error().decode())
            out.append(cls(lib, h, ptrs[r]))
        return out

    @classmethod
    def create_loopback(cls, rank, world, max_elems, bulk_elems=0):
        """ONE rank of a `world`-rank group alone on the current device (chatts_tp_init_loopback): its pushes land in its own
        buffer, the absent peers contribute zeros.  Same stores and polls per element as a real step, zero link latency - what a
        single GPU can MEASURE of a rank's step time at the shard shapes of TP = 2 / 4 / 8 (tools/tp_shard_step.py).  Timing only."""
        from . import _lib
        lib = _lib.load()
        nbytes = int(lib.chatts_tp_buffer_bytes_bulk(world, int(max_elems), int(bulk_elems)))
        ptr = C.c_void_p()
        _lib.check(lib.chatts_tp_buffer_alloc(nbytes, C.byref(ptr), None))
        h = lib.chatts_tp_init_loopback(rank, world, ptr, nbytes, int(max_elems))
        if not h:
            lib.chatts_tp_buffer_free(ptr)
            raise _lib.ChattsError(-1, lib.chatts_last_error().decode())
        return cls(lib, h, ptr)

    def status(self):
        """0 = healthy; bit 0 = a peer's contribution timed out.  Synchronising diagnostic (hipMemcpy of one word)."""
        rc = int(self.lib.chatts_tp_status(self.handle))
        if rc < 0:
            from . import _lib
            _lib.check(rc)
        return rc

    def all_reduce(self, inp, out=None, resid=None):
        from . import _lib
        out = inp if out is None else out
        _lib.check(self.lib.chatts_allreduce(self.handle, inp.data_ptr(), out.data_ptr(), _lib.ptr(resid), inp.numel(),
                                             _lib.stream_ptr()))
        return out

    def all_reduce_bulk(self, inp, x):
        """x += sum over the ranks of inp (prefill-sized: the two-shot kernel over the bulk region)"""
        from . import _lib
        _lib.check(self.lib.chatts_allreduce_bulk(self.handle, inp.data_ptr(), x.data_ptr(), inp.numel(), _lib.stream_ptr()))
        return x

    def all_gather(self, inp, rows=1):
        from . import _lib
        row_len = inp.numel() // rows
        out = torch.empty((rows, self.world * row_len), dtype=torch.float32, device=inp.device)
        _lib.check(self.lib.chatts_allgather(self.handle, inp.data_ptr(), out.data_ptr(), rows, row_len, _lib.stream_ptr()))
        return out

    def close(self):
        if self.handle:
            self.lib.chatts_tp_destroy(self.handle)
            self.handle = None
        if self.buf_ptr and self.owns_buffer:
            self.lib.chatts_tp_buffer_free(self.buf_ptr)
            self.buf_ptr = None
