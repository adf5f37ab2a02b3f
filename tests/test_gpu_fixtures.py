"""GPU-side guard of the CPU fuzz test's inputs: the committed proof fixtures (tests/golden/proof_*.bin) must still
verify and be what the current prover produces (same bytes), otherwise tests/test_verify_fuzz.py would be fuzzing
stale containers (it skips then)."""
import os
import struct

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["commit", "curve"])
def test_fixture_is_current(name):
    from dvt_circuits_amd import capi
    from tests import guests

    blob = open(os.path.join(ROOT, "tests", "golden", f"proof_{name}.bin"), "rb").read()
    (n,) = struct.unpack_from("<I", blob)
    vk, proof = blob[4:4 + n], blob[4 + n:]
    ok, ec, pv, why = capi.verify(vk, proof, 4, 4)
    assert ok, f"stale fixture ({why}): run tools/make_proof_fixture.py on a GPU box and copy gpurun_out/proof_{name}.bin to tests/golden/"
    elf = guests.commit_only(b"fuzz me!") if name == "commit" else guests.curve_ops()[0]
    p = capi.Prover('{"fri_queries": 4, "pow_bits": 4, "log_shard_size": 10}')
    pk, vk2 = p.setup(elf)
    proof2, _ = p.prove_core(pk)
    assert vk2 == vk and proof2 == proof, "the prover no longer reproduces the fixture: regenerate it"
    p.pk_free(pk)
    p.close()
