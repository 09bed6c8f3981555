"""Multi-GPU plumbing of the batch path: one process per GPU, frames sharded across ranks.

Images are independent streams (fresh estimator per plane, src/compression.rs:110 of the reference),
so a job of `total` frames is split into contiguous shards with NO data-path collective;
torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU) is only used for the barrier that
brackets a timed region, the MAX of the per-rank times and the gathering of per-rank results.
"""
import os


def env_rank():
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when launched alone."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(total, rank, world):
    """Contiguous shard [first, last) of `total` frames owned by `rank`: sizes differ by at most one."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


class Group:
    """Thin wrapper so single-process runs need no process group."""

    def __init__(self, backend=None, device=None):
        self.rank, self.world, self.local = env_rank()
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            kwargs = {}
            if backend == "nccl" and device is not None:
                kwargs["device_id"] = device
            dist.init_process_group(backend or "gloo", **kwargs)
            self.dist = dist
        self.device = device

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if not self.dist:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if not self.dist:
            return int(value)
        import torch

        t = torch.tensor([int(value)], dtype=torch.int64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def gather_objects(self, obj):
        """List of every rank's `obj` on every rank (small python objects only)."""
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()
            self.dist = None
