"""Process-group plumbing for slice-parallel runs (one process per GPU, `torch.distributed`).

The data path has no collective (slices are independent); the only cross-rank operations
are the barrier that brackets a timed region and the max-over-ranks of the elapsed time.  Backend "nccl" is RCCL
on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
import time


class Group:
    def __init__(self, backend: str | None = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend is None:
                import torch
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                import torch
                torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world)
            self.dist = dist
            self.backend = backend

    def barrier(self, sync_device=None):
        if self.dist is not None:
            self.dist.barrier()
        if sync_device is not None:
            sync_device()

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch
        dev = torch.device("cuda", self.local_rank) if self.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, fn, sync_device=None):
        """barrier + device sync, run fn, barrier + device sync; returns the max elapsed seconds over ranks."""
        self.barrier(sync_device)
        t0 = time.perf_counter()
        fn()
        self.barrier(sync_device)
        return self.max_over_ranks(time.perf_counter() - t0)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
