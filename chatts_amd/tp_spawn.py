"""Tensor-parallel workers spawned from ONE plain Python process.

The reference never launches its TP ranks itself: ``LLM(model, tensor_parallel_size=k)`` (NetManAIOps/ChatTS demo/demo_vllm.py:30,
chatts/utils/llm_utils.py:154) is called from an ordinary script and vLLM spawns k - 1 worker processes
(README.md:141: VLLM_WORKER_MULTIPROC_METHOD=spawn).  This module is that piece for this engine: the calling process becomes
rank 0 of a fresh torch.distributed group, k - 1 followers are started with the `spawn` method (one per GPU), every rank builds
the same object with the same arguments, and from then on the leader announces each call over a CPU (gloo) group before it makes
it itself - the followers replay it, so all ranks issue the same kernels and meet inside the exchange collectives.

    group = TpGroup.launch(world, factory, factory_args)    # leader side; returns after every rank has built its object
    group.call("generate", prompts, params)                 # followers run obj.generate(prompts, params) concurrently
    group.shutdown()

`bench.py --gpus N` uses the other standard route (re-exec under torch.distributed.run); both end in the same Comm().
Test hooks (single-GPU boxes, CPU tests): CHATTS_FORCE_DEVICE pins every rank to one device, CHATTS_DIST_BACKEND=gloo replaces RCCL.
"""
import os
import socket


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_RENDEZVOUS_VARS = ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK")


def _save_env():
    return {k: os.environ.get(k) for k in _RENDEZVOUS_VARS}


def _restore_env(saved):
    """put the rendezvous variables back the way the caller's process had them (a stale WORLD_SIZE would make the next
    LLM(tensor_parallel_size=k) of this process skip its spawn branch, and every child process would inherit it)"""
    for k, v in (saved or {}).items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def init_rank(rank, world, port, use_cuda=True):
    """Join the process group as `rank` (env-style rendezvous on 127.0.0.1:port) -> the CPU control group (gloo)."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("CHATTS_DIST_BACKEND", "nccl" if use_cuda else "gloo")
    if use_cuda:
        dev = int(os.environ.get("CHATTS_FORCE_DEVICE", rank))
        if dev >= torch.cuda.device_count():
            raise RuntimeError(f"tensor-parallel rank {rank} needs GPU {dev}, but only {torch.cuda.device_count()} are visible")
        torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{torch.cuda.current_device()}"))
    else:
        dist.init_process_group(backend=backend)
    return dist.new_group(backend="gloo")


def _follower(rank, world, port, use_cuda, factory, factory_args, factory_kwargs):
    import torch.distributed as dist
    control = init_rank(rank, world, port, use_cuda)
    obj, err = None, None
    try:
        obj = factory(*factory_args, **factory_kwargs)
    except BaseException as e:               # report instead of leaving the leader blocked in a collective
        err = f"{type(e).__name__}: {e}"
    oks = [None] * world
    dist.all_gather_object(oks, err, group=control)
    if any(o is not None for o in oks):
        dist.destroy_process_group()
        return
    while True:
        box = [None]
        dist.broadcast_object_list(box, src=0, group=control)
        name, args, kwargs = box[0]
        if name == "__stop__":
            break
        try:
            getattr(obj, name)(*args, **kwargs)
        except Exception as e:               # argument errors are raised identically (and before any collective) on the leader,
            import sys                       # which reports them to the caller: stay alive for the next call
            print(f"[chatts_amd tp rank {rank}] {name} raised {type(e).__name__}: {e}", file=sys.stderr, flush=True)
    dist.destroy_process_group()


class TpGroup:
    """Leader-side handle of the spawned ranks."""

    def __init__(self, world, procs, control, saved_env=None):
        self.world, self.procs, self.control = world, procs, control
        self.closed = False
        self.saved_env = saved_env          # the leader process's rendezvous variables before launch(): restored by shutdown()

    @classmethod
    def launch(cls, world, factory, factory_args=(), factory_kwargs=None, use_cuda=True):
        """-> (group, obj): spawns world - 1 followers, joins as rank 0, builds the leader's object with the same factory.
        `factory` must be a picklable top-level callable; its arguments must be picklable."""
        import torch.distributed as dist
        import torch.multiprocessing as mp
        if dist.is_initialized():
            raise RuntimeError("a torch.distributed group already exists in this process: launch the ranks with torchrun instead")
        port = free_port()
        ctx = mp.get_context("spawn")
        kw = dict(factory_kwargs or {})
        procs = [ctx.Process(target=_follower, args=(r, world, port, use_cuda, factory, tuple(factory_args), kw), daemon=True)
                 for r in range(1, world)]
        saved = _save_env()
        for p in procs:
            p.start()
        try:
            control = init_rank(0, world, port, use_cuda)
        except BaseException:
            _restore_env(saved)
            raise
        obj, err = None, None
        try:
            obj = factory(*factory_args, **kw)
        except BaseException as e:
            err = f"{type(e).__name__}: {e}"
        oks = [None] * world
        dist.all_gather_object(oks, err, group=control)
        bad = {r: o for r, o in enumerate(oks) if o is not None}
        if bad:
            for p in procs:
                p.join(timeout=30)
            dist.destroy_process_group()
            _restore_env(saved)
            raise RuntimeError(f"tensor-parallel start-up failed on ranks {bad}")
        return cls(world, procs, control, saved), obj

    def call(self, name, *args, **kwargs):
        """announce a method call to the followers (they start executing it now); the leader then makes the same call itself"""
        import torch.distributed as dist
        if self.closed:
            raise RuntimeError("the tensor-parallel group has been shut down")
        dist.broadcast_object_list([(name, args, kwargs)], src=0, group=self.control)

    def shutdown(self):
        import torch.distributed as dist
        if self.closed:
            return
        self.closed = True
        try:
            dist.broadcast_object_list([("__stop__", (), {})], src=0, group=self.control)
        except Exception:
            pass
        for p in self.procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
        if dist.is_initialized():
            dist.destroy_process_group()
        _restore_env(self.saved_env)
