#!/usr/bin/env python3
"""Makes the proof fixtures of tests/test_verify_fuzz.py (run on a GPU box: proving needs the gfx950 device):
    python tools/make_proof_fixture.py OUT_DIR
writes OUT_DIR/proof_<name>.bin = u32 vk length | verifying key | proof container, for small guests proven with few FRI
queries (the fixtures only feed the parser / verifier fuzzing, they carry no security claim).  Copy them to tests/golden/."""
import os
import struct
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
Q, POW = 4, 4


def main(out_dir):
    from dvt_circuits_amd import capi
    from tests import guests

    os.makedirs(out_dir, exist_ok=True)
    p = capi.Prover('{"fri_queries": %d, "pow_bits": %d, "log_shard_size": 10}' % (Q, POW))
    for name, elf in (("commit", guests.commit_only(b"fuzz me!")), ("curve", guests.curve_ops()[0])):
        pk, vk = p.setup(elf)
        proof, rep = p.prove_core(pk)
        ok, ec, pv, why = capi.verify(vk, proof, Q, POW)
        assert ok, why
        with open(os.path.join(out_dir, f"proof_{name}.bin"), "wb") as f:
            f.write(struct.pack("<I", len(vk)) + vk + proof)
        print(name, "vk", len(vk), "proof", len(proof), "cycles", rep["cycles"])
        p.pk_free(pk)
    p.close()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
