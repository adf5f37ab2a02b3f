#!/usr/bin/env python3
"""BASELINE.json configs[1] as literally as this repository can run it: `finalization_prove on examples/finalization_test.json,
single shard, 1 x MI355X` — the re-stated finalization guest (tests/guests_finalization.py: the reference's guest minus the
pairings) proven on the reference's example input with production parameters (100 queries, 16 PoW bits).  Prints one JSON line:
cycles, rows of every chip, ms per proof (whole `prove()` call), proofs per hour.

    python tools/bench_reference_guest.py [REPEATS] [HANDLES]

HANDLES > 1 adds a throughput figure: that many prover handles on the one GPU, each driven by its own host thread (what
bench.py --batch does for BASELINE configs[4]), so that one proof's host work (guest execution, uploads, proof download)
overlaps another's kernels."""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def measure(reps=5, handles=1):
    from dvt_circuits_amd import capi
    from tests import guests_finalization as gf

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
    batch = None
    if handles > 1:
        import threading

        provers = [capi.Prover('{"fri_queries": 100, "pow_bits": 16}') for _ in range(handles)]
        keys = [q.setup(elf)[0] for q in provers]
        for q, k in zip(provers, keys):
            q.prove_core(k, [buf])          # warm-up: buffers, tables
        out = [None] * handles

        def work(i):
            for _ in range(reps):
                out[i] = provers[i].prove_core(keys[i], [buf])[0]

        ts = [threading.Thread(target=work, args=(i,)) for i in range(handles)]
        t = time.perf_counter()
        for th in ts:
            th.start()
        for th in ts:
            th.join()
        dtb = (time.perf_counter() - t) / (reps * handles)
        assert all(o == proof for o in out), "a concurrent handle produced different proof bytes"
        batch = {"handles": handles, "ms_per_proof": 1000 * dtb, "proofs_per_hour": 3600 / dtb, "guest_cycles_per_s": rep["cycles"] / dtb}
        for q, k in zip(provers, keys):
            q.pk_free(k)
            q.close()
    result = ({"workload": "reference examples/finalization_test.json (n = 3, k = 2) through the re-stated finalization guest (no pairings)",
                      "guest_cycles": rep["cycles"], "shards": n, "public_values_bytes": len(pv), "proof_bytes": len(proof),
                      "ms_per_proof": 1000 * dt, "proofs_per_hour": 3600 / dt, "guest_cycles_per_s": rep["cycles"] / dt,
                      "execute_only_ms": 1000 * t_exec, "concurrent_handles": batch,
                      "chip_heights_by_id": {str(c["chip_id"]): [int(c["main"].shape[1]), int(c["main"].shape[0])] for c in chips}})
    p.pk_free(pk)
    p.close()
    return result


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    handles = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print(json.dumps(measure(reps, handles)))


if __name__ == "__main__":
    main()
