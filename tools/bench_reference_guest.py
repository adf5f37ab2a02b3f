#!/usr/bin/env python3
"""BASELINE.json configs[1] as literally as this repository can run it: `finalization_prove on examples/finalization_test.json,
single shard, 1 x MI355X` — the re-stated finalization guest (tests/guests_finalization.py: the reference's guest minus the
pairings) proven on the reference's example input with production parameters (100 queries, 16 PoW bits).  Prints one JSON line:
cycles, rows of every chip, ms per proof (whole `prove()` call), proofs per hour.

    python tools/bench_reference_guest.py [REPEATS]"""
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
    p = capi.Prover('{"fri_queries": 100, "pow_bits": 16}')
    pk, vk = p.setup(elf)
    proof, rep = p.prove_core(pk, [buf])
    ok, ec, pv, why = capi.verify(vk, proof)
    assert ok and pv == gf.expected_public_values(json.loads(example)), why
    chips, pubs, n = capi.rv32_debug_traces(elf, [buf])
    t = time.perf_counter()
    for _ in range(reps):
        p.prove_core(pk, [buf])
    dt = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    capi.execute(elf, [buf])
    t_exec = time.perf_counter() - t
    print(json.dumps({"workload": "reference examples/finalization_test.json (n = 3, k = 2) through the re-stated finalization guest (no pairings)",
                      "guest_cycles": rep["cycles"], "shards": n, "public_values_bytes": len(pv), "proof_bytes": len(proof),
                      "ms_per_proof": 1000 * dt, "proofs_per_hour": 3600 / dt, "guest_cycles_per_s": rep["cycles"] / dt,
                      "execute_only_ms": 1000 * t_exec,
                      "chip_heights_by_id": {str(c["chip_id"]): [int(c["main"].shape[1]), int(c["main"].shape[0])] for c in chips}}))
    p.pk_free(pk)
    p.close()


if __name__ == "__main__":
    main()
