#!/usr/bin/env python3
"""Where the time of one proof of the re-stated finalization guest goes (tools/bench_reference_guest.py's workload): host
pipeline (execution, traces, upload, phase 1) against K0..K9 on the resident shard, and the prover's own stage clock
(HIP events, "profile": 1).  Prints one JSON line.

    python tools/profile_reference_guest.py [REPEATS]"""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    from dvt_circuits_amd import capi
    from tests import guests_finalization as gf

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    example = open(os.path.join(ROOT, "tests", "golden", "finalization_example.json"), "rb").read()
    buf = capi.stdin_from_json("finalization", example)
    elf = gf.finalization(nmax=8, kmax=8)
    out = {}
    for profile in (0, 1):
        p = capi.Prover('{"fri_queries": 100, "pow_bits": 16, "profile": %d}' % profile)
        pk, vk = p.setup(elf)
        p.prove_core(pk, [buf])
        t_prep = t_prove = 0.0
        for _ in range(reps):
            t = time.perf_counter()
            job, rep = p.prepare(pk, [buf])
            t_prep += time.perf_counter() - t
            t = time.perf_counter()
            p.prove_job(pk, job)
            t_prove += time.perf_counter() - t
            p.job_free(job)
        key = "profiled" if profile else "plain"
        out[key] = {"prepare_ms": 1000 * t_prep / reps, "prove_job_ms": 1000 * t_prove / reps}
        if profile:
            out[key]["stage_ms"] = p.stage_ms()
            out[key]["kernel_stats"] = p.kernel_stats()
        p.pk_free(pk)
        p.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
