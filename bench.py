#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: prover guest-cycles/second
(and proofs/hour) on MI355X.

A "step" = one pass of the GPU prover (K0 trace expansion .. K9 FRI queries) over
one shard of the synthetic DKG-like guest (tests/guests.py:bignum, ~2^21 RV32IM
cycles), with the shard's compact execution records already resident in HBM.
N > 1: every rank proves its own replica of the shard on its own GPU ("replicas
only" until multi-shard proofs land, DESIGN.md "Multi-GPU"); no data-path
collective; value = cycles proven by all ranks / max-over-ranks time.

Extra objects on the JSON line (rank 0 prints exactly one line):
  roofline      the dominant HBM-bound kernel family, K1 coset LDE: algorithmic bytes
                (12 B per trace element: read N, write 2N words per column) divided by
                the summed duration of its launches inside one prove, measured with HIP
                events on the prover's own stream; "alu_bound_exception" carries the
                Poseidon2 commitment kernels (K2/K3), which are integer-ALU bound
  cpu_baseline  the oracle's CPU prover (tests/_oracle_prover.py over oracle/*.c,
                OpenMP) timed on this box's host cores on two small shards; value =
                marginal cycles/s between them (fixed table costs cancel)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def cpu_baseline(small_iters, big_iters):
    """Oracle CPU prover on two bounded samples of the same workload (rank 0, N = 1 only)."""
    from dvt_circuits_amd import capi
    from tests import _oracle_prover, guests

    pts = []
    for it in (small_iters, big_iters):
        elf, _ = guests.bignum(it)
        chips, pubs, _ = capi.rv32_debug_traces(elf)
        cyc = capi.execute(elf)[1]["cycles"]
        t = time.perf_counter()
        gc = _oracle_prover.global_challenges(_oracle_prover.prep_root_of(chips), [_oracle_prover.main_root(chips) + [int(x) for x in pubs]])
        _oracle_prover.prove_shard("rv32", chips, pubs, 100, 4, perm_challenges=gc)
        pts.append((cyc, time.perf_counter() - t))
    (c0, t0), (c1, t1) = pts
    from tests import _orc

    cores = int(_orc.load().lib.orc_num_threads())  # OpenMP threads the oracle actually ran with
    marginal = (c1 - c0) / max(t1 - t0, 1e-9)
    return {
        "value": marginal,
        "unit": "guest cycles/s",
        "cores": cores,
        "kind": "port",
        "sample": "in-repo CPU restatement (oracle, C + OpenMP, not SP1): same guest at %d cycles (%.1f s) and %d cycles (%.1f s), "
                  "100 FRI queries, 4 PoW bits; value = marginal rate between the two (the 2^16-row byte table is a fixed cost); "
                  "whole-sample rate of the larger one = %.0f cycles/s" % (c0, t0, c1, t1, c1 / t1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--iters", type=int, default=900, help="bignum guest iterations (900 -> ~2.08M cycles, one 2^21-row shard)")
    ap.add_argument("--cpu-small", type=int, default=112)
    ap.add_argument("--cpu-big", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    from dvt_circuits_amd import capi
    from tests import guests

    from dvt_circuits_amd.dist_util import Ranks, whole_job_rate

    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    ranks = Ranks(backend="nccl", device=torch.device("cuda", local))   # nccl == RCCL on ROCm
    rank, world = ranks.rank, ranks.world
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    def barrier():
        ranks.barrier()
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
    dt = ranks.max_over_ranks(time.perf_counter() - t0)

    # end-to-end rate including host execution + PCIe upload (reported, never `value`)
    t1 = time.perf_counter()
    j2, _ = prover.prepare(pk)
    prover.prove_job(pk, j2, want_bytes=False)
    prover.sync()
    e2e = time.perf_counter() - t1
    prover.job_free(j2)

    # kernel-family timing on a profiled handle (HIP events on the prover stream, same shard)
    prof = capi.Prover('{"device": %d, "fri_queries": 100, "pow_bits": 16, "profile": 1}' % local)
    ppk, _ = prof.setup(elf)
    pjob, _ = prof.prepare(ppk)
    prof.prove_job(ppk, pjob, want_bytes=False)
    prof.prove_job(ppk, pjob, want_bytes=False)
    stage = prof.stage_ms()
    ks = prof.kernel_stats()
    prof.job_free(pjob)
    prof.pk_free(ppk)
    prof.close()
    lde_gbps = ks["lde_alg_bytes"] / (ks["lde_ms"] * 1e-3) / 1e9 if ks["lde_ms"] else 0.0

    out = {
        "metric": "SP1 prover cycles/sec + proofs/hour, finalization_prove at 1/2/4/8 MI355X",
        "value": whole_job_rate(cycles, world, args.steps, dt),
        "unit": "guest cycles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": "finalization-like synthetic DKG guest (tests/guests.py:bignum, 384-bit multiply-accumulate), one shard of "
                        "~2^21 RV32IM cycles, BASELINE configs[1]; the reference's own guest ELF is prebuilt machine code and is not run",
            "guest_cycles_per_proof": cycles,
            "fri_queries": 100,
            "pow_bits": 16,
            "log_blowup": 1,
            "parallelism": "replicas x%d" % world,
            "proofs_per_hour": world * args.steps * 3600.0 / dt,
            "end_to_end_cycles_per_s_incl_host_exec_and_pcie": cycles / e2e,
            "host_prepare_seconds": t_host,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "K1 coset LDE (ntt_strided_kernel<true> + lde_block_kernel + ntt_strided_kernel<false>), %d calls per proof" % ks["lde_calls"],
            "achieved": lde_gbps,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": lde_gbps / HBM_PEAK_GBPS,
            "traffic": None,
            "alg_bytes_per_proof": ks["lde_alg_bytes"],
            "ms_per_proof": ks["lde_ms"],
        },
        "alu_bound_exception": {
            "kernel": "K2+K3 Poseidon2 trace commitments (merkle_leaves_kernel + merkle_level_kernel + merkle_top_kernel)",
            "ms_per_proof": ks["merkle_ms"],
            "gperm_per_s": ks["merkle_perms"] / (ks["merkle_ms"] * 1e-3) / 1e9 if ks["merkle_ms"] else 0.0,
        },
        "stage_ms": stage,
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_small, args.cpu_big)
        print(json.dumps(out))
    prover.job_free(job)
    prover.pk_free(pk)
    prover.close()
    ranks.close()


if __name__ == "__main__":
    main()
