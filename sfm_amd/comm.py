"""Reduction hooks between the BA stages.  One process per GPU.  The only data that crosses ranks is the
reduced camera system [S | r] per damped solve and a few short vectors.

RcclComm: the collectives run INSIDE libsfm_amd.so (sfm_comm_*: ncclAllReduce over xGMI enqueued on the handle's
stream between the stages, no host round trip; the trust-region loop's reduce hook is the C function
sfm_comm_reduce_hook).  DistComm: the same exchanges through `torch.distributed` ("nccl" = RCCL on ROCm; "gloo" on
CPU for the tests) from Python callbacks - what a host without the in-library communicator uses.
LocalComm: single process."""
from __future__ import annotations

import ctypes as C


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


class RcclComm:
    """Collectives inside the library.  Bootstrap: rank 0's 128-byte RCCL id reaches the other ranks through
    `exchange_id(bytes_or_None) -> bytes` (default: an already initialised torch.distributed group of any backend,
    used for these 128 bytes only); then sfm_comm_init_rank on every rank.  world_size 1 needs no exchange."""

    in_library = True

    def __init__(self, device=0, rank=None, world_size=None, exchange_id=None, group=None):
        from . import _lib
        self._lib = _lib
        self.h = _lib.get_handle(device)
        self.group = group
        if rank is None or world_size is None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank, world_size = dist.get_rank(group), dist.get_world_size(group)
            else:
                rank, world_size = 0, 1
        self.rank, self.world_size = int(rank), int(world_size)
        ident = (C.c_char * 128)()
        if self.rank == 0:
            self.h.call("sfm_comm_unique_id", C.cast(ident, C.c_void_p))
        if self.world_size > 1:
            if exchange_id is None:
                import torch
                import torch.distributed as dist
                dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
                t = torch.frombuffer(bytearray(ident.raw), dtype=torch.uint8).clone().to(dev)
                dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                raw = bytes(t.cpu().numpy().tobytes())
            else:
                raw = exchange_id(ident.raw if self.rank == 0 else None)
            ident = (C.c_char * 128).from_buffer_copy(raw)
        self.h.call("sfm_comm_init_rank", C.cast(ident, C.c_void_p), self.world_size, self.rank)
        self._owner = True

    def close(self):
        if getattr(self, "_owner", False):
            self.h.lib.sfm_comm_destroy(self.h._h)
            self._owner = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _reduce(self, t, op):
        import torch
        if not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous():
            raise ValueError("RcclComm reduces contiguous float64 CUDA tensors in place")
        self.h.call("sfm_comm_allreduce", C.c_void_p(t.data_ptr()), t.numel(), op)
        return t

    def allreduce_sum(self, t):
        return self._reduce(t, 0)

    def allreduce_max(self, t):
        return self._reduce(t, 1)

    def reduce_hook(self):
        """(sfm_reduce_fn, user pointer) for sfm_ba_trf_begin / sfm_ba_solve_pcg: the library's own C function."""
        fn = C.cast(self.h.lib.sfm_comm_reduce_hook, self._lib.REDUCE_FN)
        return fn, self.h._h

    def all_gather_objects(self, obj):
        if self.world_size == 1:
            return [obj]
        import torch.distributed as dist
        out = [None] * self.world_size
        dist.all_gather_object(out, obj, group=self.group)
        return out
