"""Multi-process plumbing for bench.py (one process per GPU): rank discovery from
the torch.distributed.run environment, barrier, and the max-over-ranks reduction of
the elapsed time.  The data path itself never communicates: shards / replicas are
independent."""
import os


class Ranks:
    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist

            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend or "gloo", rank=self.rank, world_size=self.world, **kw)
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_over_ranks(self, seconds: float) -> float:
        if not self.dist:
            return seconds
        import torch

        t = torch.tensor([seconds], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def shard_of(self, n_units: int):
        """units (independent proofs) this rank owns: round-robin, no collective"""
        return list(range(self.rank, n_units, self.world))

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


def whole_job_rate(units_per_rank: int, world: int, steps: int, seconds: float) -> float:
    """value of the bench line: units processed by all ranks / max-over-ranks time"""
    return world * units_per_rank * steps / seconds
