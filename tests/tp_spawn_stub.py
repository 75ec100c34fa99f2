"""Picklable stand-in for an engine object, used by tests/test_launchers.py through chatts_amd.tp_spawn (spawned followers import it)."""
import torch
import torch.distributed as dist


class Stub:
    def __init__(self, out_dir, scale):
        self.out_dir, self.scale = out_dir, scale
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def generate(self, values, tag="x"):
        """every rank contributes (rank + 1) * scale * sum(values); the all-reduced total is what a TP step would agree on"""
        t = torch.tensor([float(sum(values)) * self.scale * (self.rank + 1)])
        dist.all_reduce(t)
        with open(f"{self.out_dir}/rank{self.rank}_{tag}.txt", "w") as f:
            f.write(str(float(t.item())))
        if tag == "boom":
            raise ValueError("argument error raised on every rank")
        return float(t.item())


def build(out_dir, scale=1.0):
    return Stub(out_dir, scale)
