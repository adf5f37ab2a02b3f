#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: prover guest-cycles/second
(and proofs/hour) on MI355X.

A "step" = one proof of one execution of the finalization-shaped guest
(tests/guests.py:finalization_like) on the reference's examples/finalization_test.json
(kept as tests/golden/finalization_example.json), encoded exactly as the reference's
host encodes it (typed JSON -> CBOR -> one SP1Stdin buffer, src/main.rs:451-460):
K0 trace expansion .. K9 FRI queries over all its shards of 2^21 RV32IM cycles,
with the compact execution records already resident in HBM.
The execution has shards_per_gpu x N shards (weak scaling: N = 1 is the single-shard
configuration BASELINE.json quotes); shard i is proven on GPU i mod N.  The only
exchange is an all-gather of the 60-byte shard headers (main-trace root + public
values) between phase 1 (main commitments) and phase 2 (everything else), from
which every rank derives the common LogUp challenges.
value = guest cycles of the execution x steps / max-over-ranks time.

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


def workload_stdin(participants=0):
    """the reference's own example input (or, with --participants N, a synthetic N-participant one in the same format:
    tools/gen_dkg_input.py) through the reference's host-side encoding"""
    from dvt_circuits_amd import capi

    if participants:
        from tools import gen_dkg_input

        return capi.stdin_from_json("finalization", json.dumps(gen_dkg_input.finalization(participants, 2)).encode())
    with open(os.path.join(ROOT, "tests", "golden", "finalization_example.json"), "rb") as f:
        return capi.stdin_from_json("finalization", f.read())


def fit_iters(stdin_buf, total_shards, iters=0):
    """largest multiply-accumulate count whose execution still fits total_shards shards of 2^21 cycles"""
    from dvt_circuits_amd import capi
    from tests import guests

    if iters:
        return iters
    iters = 907 * total_shards
    while True:
        cycles = capi.execute(guests.finalization_like(iters, stdin_buf)[0], [stdin_buf])[1]["cycles"]
        over = cycles - (total_shards << 21)
        if over <= 0:
            return iters
        iters -= over // 2300 + 1


def cpu_baseline(small_iters, big_iters):
    """Oracle CPU prover on two bounded samples of the same workload (rank 0, N = 1 only)."""
    from dvt_circuits_amd import capi
    from tests import _oracle_prover, guests

    buf = workload_stdin()
    pts = []
    for it in (small_iters, big_iters):
        elf, _ = guests.finalization_like(it, buf)
        chips, pubs, _ = capi.rv32_debug_traces(elf, [buf])
        cyc = capi.execute(elf, [buf])[1]["cycles"]
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
    ap.add_argument("--iters", type=int, default=0, help="multiply-accumulate iterations of the guest (default 907 per shard - 3: ~2.096M cycles fill one 2^21-cycle shard)")
    ap.add_argument("--shards-per-gpu", type=int, default=1, help="weak scaling: the execution has shards_per_gpu x n_gpus shards")
    ap.add_argument("--cpu-small", type=int, default=112)
    ap.add_argument("--cpu-big", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--participants", type=int, default=0, help="0 = the reference's finalization example (BASELINE configs[1]); N = synthetic "
                    "N-participant input of the same format (tools/gen_dkg_input.py), e.g. 255")
    args = ap.parse_args()

    import torch

    from dvt_circuits_amd import capi
    from tests import guests

    from dvt_circuits_amd.dist_util import Ranks, whole_job_rate

    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # DVT_BENCH_REHEARSAL=1: rehearse the N > 1 control flow on a ONE-GPU box (every rank proves on cuda:0, collectives over
    # gloo on host tensors).  Numbers from such a run mean nothing; it exists so that the multi-rank path is exercised before
    # the driver's 8-GPU run.
    rehearsal = os.environ.get("DVT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    coll = torch.device("cpu") if rehearsal else torch.device("cuda", local)     # where collective operands live
    ranks = Ranks(backend="gloo" if rehearsal else "nccl", device=coll)   # nccl == RCCL on ROCm
    rank, world = ranks.rank, ranks.world
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    def barrier():
        ranks.barrier()
        torch.cuda.synchronize()

    # workload: one execution of (shards_per_gpu x world) shards of 2^21 cycles; rank r owns shards r, r+world, ...
    total_shards = args.shards_per_gpu * world
    stdin_buf = workload_stdin(args.participants)
    iters = fit_iters(stdin_buf, total_shards, args.iters)
    elf, want_pv = guests.finalization_like(iters, stdin_buf)
    prover = capi.Prover('{"device": %d, "fri_queries": 100, "pow_bits": 16, "log_shard_size": 21}' % local)
    pk, vk = prover.setup(elf)
    t_host = time.perf_counter()
    job, rep = prover.prepare(pk, [stdin_buf])
    t_host = time.perf_counter() - t_host
    cycles = int(rep["cycles"])
    n_shards = prover.job_shards(job)
    mine = ranks.shard_of(n_shards)

    def exchange_headers(local_headers):
        """the one exchange step of the path: all-gather of the 15-word shard headers (RCCL over xGMI)"""
        import numpy as np

        if world == 1:
            return np.stack(local_headers)
        h = torch.zeros((n_shards, 15), dtype=torch.int64, device=coll)
        for i, hd in zip(mine, local_headers):
            h[i] = torch.from_numpy(hd.astype(np.int64)).to(coll)
        ranks.dist.all_reduce(h)   # disjoint rows: sum == gather
        return h.cpu().numpy().astype(np.uint32)

    def prove_once(want_bytes):
        headers = exchange_headers([prover.commit_shard(pk, job, i) for i in mine])
        ch = capi.rv32_challenges(vk, headers)
        return [prover.prove_shard(pk, job, i, ch, want_bytes=want_bytes) for i in mine]

    # correctness outside the timed region: the proof this configuration produces must verify
    shard_proofs = prove_once(True)
    if world == 1:
        proof = prover.assemble(job, shard_proofs)
        ok, ec, pv, why = capi.verify(vk, proof, 100, 16)
        assert ok and pv == want_pv and ec == 0, f"bench proof rejected: {why}"
    else:
        import numpy as np

        lens = torch.zeros(n_shards, dtype=torch.int64, device=coll)
        for i, sp in zip(mine, shard_proofs):
            lens[i] = len(sp)
        ranks.dist.all_reduce(lens)
        mx = int(lens.max().item())
        buf = torch.zeros((n_shards, mx), dtype=torch.uint8, device=coll)
        for i, sp in zip(mine, shard_proofs):
            buf[i, : len(sp)] = torch.frombuffer(bytearray(sp), dtype=torch.uint8).to(coll)
        ranks.dist.all_reduce(buf)
        if rank == 0:
            allp = [bytes(buf[i, : int(lens[i].item())].cpu().numpy()) for i in range(n_shards)]
            ok, ec, pv, why = capi.verify(vk, prover.assemble(job, allp), 100, 16)
            assert ok and pv == want_pv and ec == 0, f"bench proof rejected: {why}"

    for _ in range(args.warmup):
        prove_once(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        prove_once(False)
    prover.sync()
    barrier()
    dt = ranks.max_over_ranks(time.perf_counter() - t0)

    # end-to-end rate including host execution + PCIe upload (reported, never `value`)
    e2e = None
    if world == 1:
        t1 = time.perf_counter()
        j2, _ = prover.prepare(pk, [stdin_buf])
        prover.prove_job(pk, j2, want_bytes=False)
        prover.sync()
        e2e = time.perf_counter() - t1
        prover.job_free(j2)

    # kernel-family timing on a profiled handle (HIP events on the prover stream, same shard)
    prof = capi.Prover('{"device": %d, "fri_queries": 100, "pow_bits": 16, "profile": 1}' % local)
    pelf, _ = guests.finalization_like(fit_iters(stdin_buf, 1), stdin_buf)           # one shard
    ppk, _ = prof.setup(pelf)
    pjob, _ = prof.prepare(ppk, [stdin_buf])
    prof.prove_job(ppk, pjob, want_bytes=False)
    prof.prove_job(ppk, pjob, want_bytes=False)
    stage = prof.stage_ms()
    ks = prof.kernel_stats()
    prof.job_free(pjob)
    prof.pk_free(ppk)
    prof.close()
    lde_gbps = ks["lde_alg_bytes"] / (ks["lde_ms"] * 1e-3) / 1e9 if ks["lde_ms"] else 0.0
    # compulsory streams of the whole shard, each counted once
    shard_alg = 36 * ks["cells_main"] + 36 * ks["cells_perm"] + 28 * ks["cells_quotient"] + 24 * ks["cells_prep"]
    step_s = dt / args.steps / max(len(mine), 1)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r1c_pmc_k1_traffic.json")))
    except OSError:
        pmc = {}

    out = {
        "metric": "SP1 prover cycles/sec + proofs/hour, finalization_prove at 1/2/4/8 MI355X",
        "value": cycles * args.steps / dt,
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
            "workload": ("synthetic %d-participant finalization input (tools/gen_dkg_input.py, reference format) through the reference's " % args.participants
                         if args.participants else
                         "BASELINE configs[1]: examples/finalization_test.json (tests/golden/finalization_example.json) through the reference's ") +
                        "JSON -> CBOR -> SP1Stdin encoding, proven by the finalization-shaped synthetic guest (tests/guests.py:finalization_like: "
                        "reads the buffer, 384-bit multiply-accumulate seeded by it), one shard of ~2^21 RV32IM cycles; the reference's own guest "
                        "ELF is prebuilt machine code and is not run",
            "stdin_bytes": len(stdin_buf),
            "participants": args.participants or 3,
            "guest_cycles_per_proof": cycles,
            "shards_per_proof": n_shards,
            "shards_per_gpu": args.shards_per_gpu,
            "fri_queries": 100,
            "pow_bits": 16,
            "log_blowup": 1,
            "parallelism": "one execution, shard i on GPU i mod %d; all-gather of 60-byte shard headers between the two phases" % world,
            "proofs_per_hour": args.steps * 3600.0 / dt,
            "end_to_end_cycles_per_s_incl_host_exec_and_pcie": (cycles / e2e) if e2e else None,
            "host_prepare_seconds": t_host,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "K1 coset LDE (ntt_strided_kernel<true> + lde_block_kernel + ntt_strided_kernel<false>), %d calls per proof" % ks["lde_calls"],
            "achieved": lde_gbps,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": lde_gbps / HBM_PEAK_GBPS,
            "traffic": pmc.get("k1_hbm_bytes_per_proof"),
            "traffic_source": "profiles/r1c_pmc_k1_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this command, "
                              "gfx950 FETCH_SIZE x2 correction applied where the guide prescribes it); bytes per proof, like alg_bytes_per_proof",
            "alg_bytes_per_proof": ks["lde_alg_bytes"],
            "ms_per_proof": ks["lde_ms"],
        },
        "alu_bound_exception": {
            "kernel": "K2+K3 Poseidon2 trace commitments (merkle_leaves_kernel + merkle_level_kernel + merkle_top_kernel)",
            "ms_per_proof": ks["merkle_ms"],
            "gperm_per_s": ks["merkle_perms"] / (ks["merkle_ms"] * 1e-3) / 1e9 if ks["merkle_ms"] else 0.0,
        },
        "stage_ms": stage,
        "whole_shard": {
            "alg_bytes": shard_alg,
            "formula": "36 M + 36 P + 28 Q + 24 Pre (field elements of the main / permutation / quotient / preprocessed traces)",
            "cells": {k: ks["cells_" + k] for k in ("main", "perm", "quotient", "prep")},
            "achieved_GBps": shard_alg / step_s / 1e9,
            "frac_of_hbm_peak": shard_alg / step_s / 1e9 / HBM_PEAK_GBPS,
        },
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_small, args.cpu_big)
        print(json.dumps(out), flush=True)
    prover.job_free(job)
    prover.pk_free(pk)
    prover.close()
    ranks.close()


if __name__ == "__main__":
    main()
