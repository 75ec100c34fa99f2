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
"""
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
