"""One process per GPU plumbing (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" on CPU).

Two ways to use several GPUs: independent replicas / ensemble members (BASELINE config 5: 64 UnCoupledModel
members, no RCCL in the data path; shard_members / aggregate_throughput below), and ONE simulation
slab-decomposed over the ranks (niwqg_amd/slab.py), whose all-to-alls go through the
process group made here.
"""
import os


class Group(object):
    """Process-group facade that also works with world size 1 (no torch.distributed needed)."""

    def __init__(self, backend=None, force=False):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = None
        if self.world > 1 or (force and "MASTER_ADDR" in os.environ):
            import torch
            import torch.distributed as dist
            if backend is None:
                # NIWQG_AMD_DIST_BACKEND=gloo: rehearse the multi-process path with several ranks on ONE GPU (RCCL
                # refuses two ranks per device); the slab transport then stages its buffers through the host
                backend = os.environ.get("NIWQG_AMD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            kw = {}
            if backend == "nccl":
                self.device = torch.device("cuda", self.local_rank)
                torch.cuda.set_device(self.device)
                kw["device_id"] = self.device
            if not dist.is_initialized():
                dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
            self.backend = backend

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def _tensor(self, values):
        import torch
        return torch.tensor(list(values), dtype=torch.float64, device=self.device if self.device is not None else "cpu")

    def max(self, value):
        """max over ranks of a python float (the bench's step time)"""
        if self.dist is None:
            return float(value)
        t = self._tensor([value])
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def sum(self, values):
        """element-wise sum over ranks of a list of floats (e.g. per-rank step counts, budget increments)"""
        if self.dist is None:
            return [float(v) for v in values]
        t = self._tensor(values)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t]

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()


def shard_members(n_members, rank, world):
    """Ensemble member ids owned by `rank`: contiguous blocks, sizes differing by at most one."""
    base, extra = divmod(n_members, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def aggregate_throughput(group, local_units, local_seconds):
    """Whole-job units/s = (units of all ranks) / (max over ranks of the timed region)."""
    total_units = group.sum([local_units])[0]
    slowest = group.max(local_seconds)
    return total_units / slowest, slowest
