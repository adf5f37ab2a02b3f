"""Multi-process plumbing of the shard-parallel prove (one process per GPU, launched by
torch.distributed.run): rank discovery, barrier, max-over-ranks timing, and the ONE exchange step of the
path — the all-gather of the 13-word shard headers between phase 1 and phase 2 (DESIGN.md "Multi-shard
proofs and multi-GPU"; SURVEY.md section 8e: a gather of shard commitments over RCCL/xGMI) — plus the
gather of the finished shard proofs for assembly.  Shards are owned round-robin: rank r proves shards
r, r + world, ...; nothing else of the data path communicates.

Backend "nccl" is RCCL on ROCm; "gloo" runs the same code on CPU tensors (tests/test_bench_contract.py)."""
import os

import numpy as np

HEADER_WORDS = 13   # main-trace Merkle root (8) + public values (5) = dvt_rv32_header_words()


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
        """units (shards of one execution, or independent proofs of a batch) this rank owns: round-robin"""
        return list(range(self.rank, n_units, self.world))

    # ------------------------------------------------------------------ the exchange step
    def exchange_headers(self, n_shards: int, local_headers):
        """all-gather of the shard headers.  local_headers: this rank's headers, in the order of shard_of(n_shards),
        each HEADER_WORDS uint32 words.  Returns [n_shards][HEADER_WORDS] uint32 in shard order, on every rank."""
        mine = self.shard_of(n_shards)
        assert len(local_headers) == len(mine)
        if not self.dist:
            return np.stack([np.asarray(h, np.uint32) for h in local_headers]) if mine else np.zeros((0, HEADER_WORDS), np.uint32)
        import torch

        rows = -(-n_shards // self.world)            # every rank sends the same shape; unused rows carry shard = -1
        send = torch.full((rows, 1 + HEADER_WORDS), -1, dtype=torch.int64)
        for k, (i, h) in enumerate(zip(mine, local_headers)):
            send[k, 0] = i
            send[k, 1:] = torch.from_numpy(np.asarray(h, np.uint32).astype(np.int64))
        dev = self.device if self.device is not None else "cpu"
        send = send.to(dev)
        got = [torch.empty_like(send) for _ in range(self.world)]
        self.dist.all_gather(got, send)
        out = np.zeros((n_shards, HEADER_WORDS), np.uint32)
        seen = 0
        for t in got:
            for row in t.cpu().numpy():
                if row[0] >= 0:
                    out[int(row[0])] = row[1:].astype(np.uint32)
                    seen += 1
        assert seen == n_shards, f"header exchange: {seen} of {n_shards} shards arrived"
        return out

    def gather_proofs(self, n_shards: int, local_proofs):
        """every rank's finished shard proofs (bytes, in the order of shard_of) -> the full list in shard order (on every
        rank; used outside the timed region to assemble and verify the whole proof)"""
        mine = self.shard_of(n_shards)
        assert len(local_proofs) == len(mine)
        if not self.dist:
            return list(local_proofs)
        got = [None] * self.world
        self.dist.all_gather_object(got, list(zip(mine, local_proofs)))
        out = [None] * n_shards
        for part in got:
            for i, b in part:
                out[i] = b
        assert all(x is not None for x in out)
        return out

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


def whole_job_rate(units_per_rank: int, world: int, steps: int, seconds: float) -> float:
    """value of the bench line: units processed by all ranks / max-over-ranks time"""
    return world * units_per_rank * steps / seconds
