#!/usr/bin/env python3
"""GPU micro-benchmark of the field / curve precompile chips: proves the reference's evaluate_polynomial guest
(tests/guests_bls.py: decompression with the subgroup check + Horner, i.e. thousands of BLS12381_ADD / _DOUBLE / FP_MUL calls
on few RV32IM cycles) and prints what one precompile row costs next to a cpu row.

    python tools/bench_precompiles.py [K_POINTS] [N_IDS]"""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    from dvt_circuits_amd import capi
    from tests import guests_bls
    from tools import bls12_381 as bls

    k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    n_ids = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    pts = [bls.g1_compress(bls.E1.mul(bls.G1, 1000003 * (i + 1))) for i in range(k)]
    elf = guests_bls.horner(pts, list(range(1, n_ids + 1)), subgroup_check=True)
    p = capi.Prover('{"fri_queries": 100, "pow_bits": 16}')
    pk, vk = p.setup(elf)
    rc, rep, pv, err = capi.execute(elf)
    chips, pubs, n = capi.rv32_debug_traces(elf)
    rows = {c["chip_id"]: (c["main"].shape[1], c["main"].shape[0]) for c in chips}
    p.prove_core(pk)
    t = time.perf_counter()
    reps = 3
    for _ in range(reps):
        proof, _ = p.prove_core(pk)
    dt = (time.perf_counter() - t) / reps
    ok = capi.verify(vk, proof)[0]
    print(json.dumps({"guest": "horner k=%d ids=%d subgroup_check" % (k, n_ids), "cycles": rep["cycles"], "shards": n, "verified": ok,
                      "ms_per_proof": 1000 * dt, "chip_rows_by_id (padded height, width)": rows, "proof_bytes": len(proof)}))


if __name__ == "__main__":
    main()
