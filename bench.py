#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: prover guest-cycles/second
(and proofs/hour) on MI355X.

A "step" = one pass of the GPU prover (K0 trace expansion .. K9 FRI queries) over
one shard of the synthetic DKG-like guest (tests/guests.py:bignum, ~2^21 RV32IM
cycles), with the shard's compact execution records already resident in HBM.
N > 1: every rank proves its own replica of the shard on its own GPU ("replicas
only" until multi-shard proofs land, DESIGN.md section Multi-GPU); no data-path
collective; value = cycles proven by all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--iters", type=int, default=900, help="bignum guest iterations (900 -> ~2.08M cycles, one 2^21-row shard)")
    ap.add_argument("--cpu-sample-iters", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    from dvt_circuits_amd import capi
    from tests import guests

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    def barrier():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    elf, want_pv = guests.bignum(args.iters)
    prover = capi.Prover('{"device": %d, "fri_queries": 100, "pow_bits": 16}' % local)
    pk, vk = prover.setup(elf)
    t_host = time.perf_counter()
    job, rep = prover.prepare(pk)
    t_host = time.perf_counter() - t_host
    cycles = int(rep["cycles"])

    # correctness outside the timed region: the proof this configuration produces must verify
    proof = prover.prove_job(pk, job)
    ok, ec, pv, why = capi.verify(vk, proof, 100, 16)
    assert ok and pv == want_pv and ec == 0, f"bench proof rejected: {why}"

    for _ in range(args.warmup):
        prover.prove_job(pk, job, want_bytes=False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        prover.prove_job(pk, job, want_bytes=False)
    prover.sync()
    barrier()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-stage split and roofline of the dominant stage, on a profiled handle (HIP events on the prover stream)
    prof = capi.Prover('{"device": %d, "fri_queries": 100, "pow_bits": 16, "profile": 1}' % local)
    ppk, _ = prof.setup(elf)
    pjob, _ = prof.prepare(ppk)
    prof.prove_job(ppk, pjob, want_bytes=False)
    prof.prove_job(ppk, pjob, want_bytes=False)
    stage = prof.stage_ms()
    shape = prof.shard_shape() if hasattr(prof, "shard_shape") else None
    prof.job_free(pjob)
    prof.pk_free(ppk)
    prof.close()

    out = {
        "metric": "SP1 prover cycles/sec + proofs/hour, finalization_prove at 1/2/4/8 MI355X",
        "value": world * cycles * args.steps / dt,
        "unit": "guest cycles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 (BabyBear, 31-bit modular)",
        "data": "synthetic",
        "config": {
            "workload": "finalization-like synthetic DKG guest (tests/guests.py:bignum, 384-bit multiply-accumulate), one shard, "
                        "BASELINE configs[1]; the reference's own guest ELF is prebuilt machine code and is not run",
            "guest_cycles_per_proof": cycles,
            "fri_queries": 100,
            "pow_bits": 16,
            "log_blowup": 1,
            "parallelism": "replicas x%d" % world,
            "proofs_per_hour": world * args.steps * 3600.0 / dt,
            "host_prepare_seconds": t_host,
        },
        "stage_ms": stage,
    }
    if rank == 0:
        print(json.dumps(out))
    prover.job_free(job)
    prover.pk_free(pk)
    prover.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
