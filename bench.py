#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: prover guest-cycles/second
(and proofs/hour) on MI355X, measured over the WHOLE boundary call.

A "step" = one prove() of the reference's boundary (src/main.rs:449-478: execute the guest AND prove it):
the DKG-shaped guest (tests/guests.py:dkg_like — finalization-shaped: read the one stdin buffer, per
participant a SHA-256 commitment hash, signature-check and point-operation stand-ins (exact 384-bit
multiply-accumulate chains), an n^2 term, SHA-256 + COMMIT of the public values) runs on the reference's
examples/finalization_test.json (tests/golden/finalization_example.json), encoded exactly as the reference's
host encodes it (typed JSON -> CBOR -> one SP1Stdin buffer), and its execution of
shards_per_gpu x N shards of 2^21 RV32IM cycles is proven: host executor pipeline (one sequential fast pass
cutting shards + traced re-execution of the owned shards on host threads), upload of the 48-byte cycle records,
K0 trace expansion .. K9 FRI queries.  Nothing is resident in HBM when the timed region starts except the
proving key (client.setup, src/main.rs:462, is per program, not per proof).

N > 1 (torch.distributed.run, one process per GPU): shard i of the ONE execution is proven on GPU i mod N; every
rank runs the sequential fast pass (it is the Amdahl term, counted inside the timed region) and traces only its own
shards.  The only exchange is the all-gather of the 52-byte shard headers (main-trace root + public values)
between phase 1 and phase 2 (RCCL over xGMI), from which every rank derives the common LogUp challenges.
value = guest cycles of the execution x steps / max-over-ranks wall time.   scaling = "weak".

--batch B: B independent single-shard proofs (the example input with distinct gen_id each), round-robin over the
ranks, no collective on the data path (BASELINE configs[4], "replicas only").

Extra objects on the JSON line (rank 0 prints exactly one line):
  roofline      the dominant HBM-classified kernel family, K1 coset LDE: algorithmic bytes (12 B per trace element:
                read N, write 2N words per column) / summed duration of its launches inside one prove, HIP events on
                the prover's own stream; "alu_bound_exception" carries the Poseidon2 commitment kernels (K2/K3)
  reference_example  BASELINE configs[1] taken literally (tools/bench_reference_guest.py): one proof of the reference's example
                input through the re-stated finalization guest - ms per whole prove() call, proofs/hour with one and with eight
                prover handles on the GPU
  cpu_baseline  the oracle's CPU prover (tests/_oracle_prover.py over oracle/*.c, OpenMP) on two bounded samples of the
                same guest (smaller iteration constants), production parameters; value = marginal cycles/s
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
LOG_SHARD = 21


def example_json():
    with open(os.path.join(ROOT, "tests", "golden", "finalization_example.json"), "rb") as f:
        return json.load(f)


def workload_stdin(participants=0, instance=None):
    """the reference's own example input (or, with --participants N, a synthetic N-participant one in the same format:
    tools/gen_dkg_input.py) through the reference's host-side encoding.  instance = i: the example with
    gen_id = sha256("batch" || i)[:16] (distinct independent instances)."""
    from dvt_circuits_amd import capi

    if participants:
        from tools import gen_dkg_input

        doc = gen_dkg_input.finalization(participants, 2)
    else:
        doc = example_json()
    if instance is not None:
        doc["settings"]["gen_id"] = hashlib.sha256(b"batch" + str(instance).encode()).digest()[:16].hex()
    return capi.stdin_from_json("finalization", json.dumps(doc).encode())


SHA_PRECOMPILES = False   # --sha-precompiles
CURVE_PRECOMPILES = False  # --curve-precompiles: the per-key point operations are real G1 arithmetic through the BLS12381 precompile chips


def guest_kw():
    return dict(sha_precompiles=SHA_PRECOMPILES, curve_precompiles=CURVE_PRECOMPILES)


def fit_constants(stdin_buf, total_shards, sig=0):
    """iteration constants (sig, pt, pair) of the guest so that the execution fills total_shards shards of 2^21 cycles
    as closely as possible from below (pt = sig / 4: the reference's per-key work is a fraction of its pairing work)"""
    from dvt_circuits_amd import capi
    from tests import guests

    def cycles(s):
        c = (s, max(1, s // 4), 1)
        return capi.execute(guests.dkg_like("finalization", *c, **guest_kw()), [stdin_buf])[1]["cycles"], c

    if sig:
        return cycles(sig)[1]
    target = total_shards << LOG_SHARD
    c1, c2 = cycles(4)[0], cycles(8)[0]
    per = (c2 - c1) / 4.0
    s = max(1, int((target - c1) / per) + 4)
    while True:
        c, consts = cycles(s)
        if c <= target or s == 1:      # s == 1: the input alone overflows the target (n = 255 in one shard); smallest guest
            return consts
        s = max(1, s - max(1, int((c - target) / per)))


def cpu_baseline(stdin_buf, sizes):
    """Oracle CPU prover on two bounded samples of the same workload (rank 0, N = 1 only)."""
    from dvt_circuits_amd import capi
    from tests import _oracle_prover, _orc, guests

    pts = []
    for s in sizes:
        elf = guests.dkg_like("finalization", s, max(1, s // 4), 1)
        chips, pubs, _ = capi.rv32_debug_traces(elf, [stdin_buf])
        cyc = capi.execute(elf, [stdin_buf])[1]["cycles"]
        t = time.perf_counter()
        gc = _oracle_prover.global_challenges(_oracle_prover.prep_root_of(chips), [_oracle_prover.main_root(chips) + [int(x) for x in pubs]])
        _oracle_prover.prove_shard("rv32", chips, pubs, 100, 16, perm_challenges=gc)
        pts.append((cyc, time.perf_counter() - t))
        stage_s = {k: round(v, 3) for k, v in _oracle_prover.STAGE_SECONDS.items()}
    (c0, t0), (c1, t1) = pts
    cores = int(_orc.load().lib.orc_num_threads())  # OpenMP threads the oracle actually ran with
    return {
        "value": (c1 - c0) / max(t1 - t0, 1e-9),
        "unit": "guest cycles/s",
        "cores": cores,
        "kind": "port",
        "parity": "unpinned against stock SP1 (the reference's prover cannot be built here); the GPU proof bytes are checked against this port",
        "sample": "in-repo CPU restatement (oracle: every stage and the PoW grind in C + OpenMP, sequenced from Python; not SP1): the bench "
                  "guest at %d cycles (%.1f s) and %d cycles (%.1f s), 100 FRI queries, 16 PoW bits; value = marginal rate between the "
                  "two (the 2^16-row byte table is a fixed cost); whole-sample rate of the larger one = %.0f cycles/s" % (c0, t0, c1, t1, c1 / t1),
        "whole_sample_value": c1 / t1,
        # wall seconds of the stages of the larger sample (the stages of stage_ms, same names): where the CPU port spends its time
        "stage_s": stage_s,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shards-per-gpu", type=int, default=32, help="weak scaling: the execution has shards_per_gpu x n_gpus shards of 2^21 cycles")
    ap.add_argument("--sig-iters", type=int, default=0, help="guest iteration constant (0 = fitted to the shard count)")
    ap.add_argument("--participants", type=int, default=0, help="0 = the reference's finalization example (n = 3); N = synthetic N-participant "
                    "input of the same format (tools/gen_dkg_input.py), e.g. 255")
    ap.add_argument("--batch", type=int, default=0, help="B independent single-shard proofs round-robin over the ranks instead of one sharded execution")
    ap.add_argument("--batch-streams", type=int, default=3, help="--batch: prover handles (one host thread + HIP stream each) per GPU")
    ap.add_argument("--sha-precompiles", action="store_true", help="the guest hashes through SP1's SHA_EXTEND / SHA_COMPRESS precompile "
                    "syscalls (sha_extend / sha_compress chips) instead of RV32IM code; not the default workload")
    ap.add_argument("--curve-precompiles", action="store_true", help="the guest's per-key point operations are BLS12-381 G1 scalar multiplications "
                    "through SP1's BLS12381_ADD / _DOUBLE precompile calls (the bls_g1 chip) instead of the multiply-accumulate stand-in")
    ap.add_argument("--exec-threads", type=int, default=0)
    ap.add_argument("--cpu-sizes", type=int, nargs=2, default=[30, 95], help="guest iteration constants of the two CPU-baseline samples "
                    "(95 = a little over 1 M cycles: the fixed costs are then below a fifth of the larger sample)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-example", action="store_true", help="skip the extra object with BASELINE configs[1] taken literally")
    args = ap.parse_args()
    global SHA_PRECOMPILES, CURVE_PRECOMPILES
    SHA_PRECOMPILES = args.sha_precompiles
    CURVE_PRECOMPILES = args.curve_precompiles

    import torch

    from dvt_circuits_amd import capi
    from dvt_circuits_amd.dist_util import Ranks
    from tests import guests

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

    cfg = '{"device": %d, "fri_queries": 100, "pow_bits": 16, "log_shard_size": %d, "exec_threads": %d}' % (local, LOG_SHARD, args.exec_threads)
    prover = capi.Prover(cfg)
    stdin_buf = workload_stdin(args.participants)
    stdin_bytes = len(stdin_buf)
    extra = {}

    if args.batch:
        # ---------------------------------------------------------------- B independent proofs ("replicas only")
        consts = fit_constants(stdin_buf, 1, args.sig_iters)
        elf = guests.dkg_like("finalization", *consts, **guest_kw())
        pk, vk = prover.setup(elf)
        inputs = [workload_stdin(args.participants, instance=i) for i in range(args.batch)]
        mine = ranks.shard_of(args.batch)
        # correctness outside the timed region: this rank's first proof verifies with the public values the guest must commit
        if mine:
            proof, rep = prover.prove_core(pk, [inputs[mine[0]]])
            ok, ec, pv, why = capi.verify(vk, proof, 100, 16)
            assert ok and ec == 0 and pv == guests.dkg_like_expected(inputs[mine[0]], "finalization", *consts, curve_precompiles=CURVE_PRECOMPILES), f"bench proof rejected: {why}"
        # S prover handles on this GPU, one host thread each: the host-side execution / upload / proof download of one
        # proof overlaps the kernels of another (independent proofs share nothing; ctypes drops the GIL during the call)
        import threading

        lanes = [(prover, pk)]
        for _ in range(1, max(1, args.batch_streams)):
            h = capi.Prover(cfg)
            lanes.append((h, h.setup(elf)[0]))
        cyc_mine = 0

        def step():
            nonlocal cyc_mine
            done = [0] * len(lanes)

            def lane(k):
                h, hpk = lanes[k]
                for i in mine[k::len(lanes)]:
                    _, rep = h.prove_core(hpk, [inputs[i]])
                    done[k] += int(rep["cycles"])

            th = [threading.Thread(target=lane, args=(k,)) for k in range(len(lanes))]
            for t in th:
                t.start()
            for t in th:
                t.join()
            cyc_mine = sum(done)

        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        prover.sync()
        barrier()
        dt = ranks.max_over_ranks(time.perf_counter() - t0)
        t = torch.tensor([float(cyc_mine)], dtype=torch.float64, device=coll)
        if ranks.dist:
            ranks.dist.all_reduce(t)
        cycles = int(t.item())
        n_shards = args.batch
        workload = ("BASELINE configs[4] shape: batch of %d independent single-shard proofs of the finalization-shaped guest, each on the "
                    "reference's example input with its own gen_id, round-robin over %d rank(s), no collective" % (args.batch, world))
        extra["proofs_per_hour"] = args.batch * args.steps * 3600.0 / dt
        parallelism = "independent proofs, proof i on GPU i mod %d (replicas only), %d prover handles (host threads) per GPU" % (world, len(lanes))
        for h, hpk in lanes[1:]:
            h.pk_free(hpk)
            h.close()
    else:
        # ---------------------------------------------------------------- one execution, shard-parallel
        total_shards = args.shards_per_gpu * world
        consts = fit_constants(stdin_buf, total_shards, args.sig_iters)
        elf = guests.dkg_like("finalization", *consts, **guest_kw())
        pk, vk = prover.setup(elf)
        want_pv = guests.dkg_like_expected(stdin_buf, "finalization", *consts, curve_precompiles=CURVE_PRECOMPILES)
        state = {}

        def prove_once(want_bytes):
            job, rep = prover.prepare(pk, [stdin_buf], first=rank, stride=world)    # executor pipeline + upload + phase 1
            n = prover.job_shards(job)
            mine = ranks.shard_of(n)
            headers = ranks.exchange_headers(n, [prover.commit_shard(pk, job, i) for i in mine])   # the one exchange step
            ch = capi.rv32_challenges(vk, headers)
            proofs = [prover.prove_shard(pk, job, i, ch, want_bytes=want_bytes) for i in mine]     # phase 2
            state.update(cycles=int(rep["cycles"]), n_shards=n, exec_wait=prover.job_exec_wait(job))
            if want_bytes:
                allp = ranks.gather_proofs(n, proofs)
                full = prover.assemble(job, allp)
                prover.job_free(job)
                return full
            prover.job_free(job)
            return None

        # correctness outside the timed region: the proof this configuration produces must verify
        full = prove_once(True)
        if rank == 0:
            ok, ec, pv, why = capi.verify(vk, full, 100, 16)
            assert ok and pv == want_pv and ec == 0, f"bench proof rejected: {why}"
        del full
        for _ in range(args.warmup):
            prove_once(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            prove_once(False)
        prover.sync()
        barrier()
        dt = ranks.max_over_ranks(time.perf_counter() - t0)
        cycles, n_shards = state["cycles"], state["n_shards"]
        extra["executor_wait_seconds_per_step_rank0"] = state["exec_wait"]
        extra["proofs_per_hour"] = args.steps * 3600.0 / dt
        workload = (("synthetic %d-participant finalization input (tools/gen_dkg_input.py, reference format)" % args.participants if args.participants else
                     "BASELINE configs[1] input: examples/finalization_test.json (tests/golden/finalization_example.json)") +
                    " through the reference's JSON -> CBOR -> SP1Stdin encoding, executed AND proven (the whole prove() call) by the "
                    "finalization-shaped synthetic guest tests/guests.py:dkg_like%r: %d shards of 2^21 RV32IM cycles; the reference's own guest ELF is "
                    "prebuilt machine code and is not run (it would be 511 shards on this input, SURVEY.md App. B.3)" % (tuple(consts), n_shards))
        parallelism = "one execution, shard i on GPU i mod %d; all-gather of 52-byte shard headers between the two phases" % world

    # ---- secondary figures on one single-shard execution: resident prove rate, kernel-family timing (profile handle)
    #      (always on the reference's example input: an n = 255 input does not fit one shard)
    stdin_buf = workload_stdin(0)
    one = fit_constants(stdin_buf, 1)
    elf1 = guests.dkg_like("finalization", *one, **guest_kw())
    pk1, _ = prover.setup(elf1)
    job1, rep1 = prover.prepare(pk1, [stdin_buf])
    prover.prove_job(pk1, job1, want_bytes=False)
    prover.sync()
    t1 = time.perf_counter()
    for _ in range(3):
        prover.prove_job(pk1, job1, want_bytes=False)
    prover.sync()
    resident = 3 * int(rep1["cycles"]) / (time.perf_counter() - t1)
    prover.job_free(job1)
    prover.pk_free(pk1)
    # the host executor alone on this box's cores (the sequential fast pass is the Amdahl term of the N-GPU prove)
    extra["executor_fast_pass_cycles_per_s"] = capi.exec_rate(elf1, [stdin_buf], LOG_SHARD, False)
    extra["executor_trace_mode_cycles_per_s_per_thread"] = capi.exec_rate(elf1, [stdin_buf], LOG_SHARD, True)
    prof = capi.Prover('{"device": %d, "fri_queries": 100, "pow_bits": 16, "profile": 1}' % local)
    ppk, _ = prof.setup(elf1)
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
    pmc = {}
    for name in ("r3_pmc_k1_traffic.json", "r2_pmc_k1_traffic.json", "r1c_pmc_k1_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            pmc["_file"] = name
            break
        except OSError:
            pass

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
        "config": dict({
            "workload": workload,
            "measured_unit": "the whole boundary call: host execution of the guest + record upload + K0..K9 (reference src/main.rs:461-466)",
            "stdin_bytes": stdin_bytes,
            "participants": args.participants or 3,
            "sha_precompiles": bool(args.sha_precompiles),
            "curve_precompiles": bool(args.curve_precompiles),
            "guest_cycles_per_step": cycles,
            "shards_per_step": n_shards,
            "shards_per_gpu": args.shards_per_gpu if not args.batch else None,
            "fri_queries": 100,
            "pow_bits": 16,
            "log_blowup": 1,
            "parallelism": parallelism,
            "resident_single_shard_cycles_per_s": resident,
            "resident_note": "K0..K9 of one ~2^21-cycle shard (the reference's example input) whose cycle records are already in HBM (round 1's headline figure), for comparison",
        }, **extra),
        "roofline": {
            "bound": "hbm",
            "kernel": "K1 coset LDE (ntt_strided_v4_kernel<true> + lde_block2_kernel + ntt_strided_v4_kernel<false>), %d calls per proof" % ks["lde_calls"],
            "achieved": lde_gbps,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": lde_gbps / HBM_PEAK_GBPS,
            "traffic": pmc.get("k1_hbm_bytes_per_proof"),
            "traffic_source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 FETCH_SIZE x2 correction applied "
                              "where the guide prescribes it); bytes per single-shard proof, like alg_bytes_per_proof" % pmc.get("_file", "(none)"),
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
            "achieved_GBps_resident": shard_alg * resident / max(int(rep1["cycles"]), 1) / 1e9,
            "frac_of_hbm_peak_resident": shard_alg * resident / max(int(rep1["cycles"]), 1) / 1e9 / HBM_PEAK_GBPS,
        },
    }
    if rank == 0:
        if not args.no_reference_example and world == 1 and not args.batch:
            # BASELINE configs[1] taken literally (single shard, the reference's example input through the re-stated
            # finalization guest, tests/guests_finalization.py): whole-call latency, and proofs/hour with three handles
            from tools import bench_reference_guest

            out["reference_example"] = bench_reference_guest.measure(reps=5, handles=8)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(stdin_buf, args.cpu_sizes)
        print(json.dumps(out), flush=True)
    prover.pk_free(pk)
    prover.close()
    ranks.close()


if __name__ == "__main__":
    main()
