"""Reduction hooks between the BA stages.  One process per GPU; `torch.distributed` with the
"nccl" backend is RCCL over xGMI on ROCm ("gloo" on CPU for the tests).  The only data that
crosses ranks is the reduced camera system [S | r] per damped solve and a few short vectors."""
from __future__ import annotations


class LocalComm:
    """Single-process: every reduction is the identity."""
    rank = 0
    world_size = 1

    def allreduce_sum(self, t):
        return t

    def allreduce_max(self, t):
        return t

    def all_gather_objects(self, obj):
        return [obj]


class DistComm:
    """torch.distributed process group (already initialised by the caller)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)

    def allreduce_sum(self, t):
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t

    def allreduce_max(self, t):
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self.group)
        return t

    def all_gather_objects(self, obj):
        out = [None] * self.world_size
        self._dist.all_gather_object(out, obj, group=self.group)
        return out
