"""CPU tests of the host-only verifier on the committed GPU-made proofs (tests/golden/proof_*.bin, made by
tools/make_proof_fixture.py; kept current by tests/test_gpu_fixtures.py): dvt_verify needs no device, so acceptance,
rejection of every kind of tampering and the error contract are checked here too, not only in the `-m gpu` suite."""
import os
import struct

import numpy as np
import pytest

from dvt_circuits_amd import capi
from tests import guests

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q, POW = 4, 4
P = 2013265921


def _load(name):
    blob = open(os.path.join(ROOT, "tests", "golden", f"proof_{name}.bin"), "rb").read()
    (n,) = struct.unpack_from("<I", blob)
    vk, proof = blob[4:4 + n], blob[4 + n:]
    if not capi.verify(vk, proof, Q, POW)[0]:
        pytest.skip("stale fixture (the AIR or the proof format changed): regenerate with tools/make_proof_fixture.py on a GPU box")
    return vk, proof


def test_fixtures_verify_and_carry_the_guests_public_values():
    vk, proof = _load("commit")
    ok, ec, pv, why = capi.verify(vk, proof, Q, POW)
    assert ok and ec == 0 and pv == b"fuzz me!", why
    vk2, proof2 = _load("curve")
    ok, ec, pv, why = capi.verify(vk2, proof2, Q, POW)
    assert ok and ec == 0 and pv == guests.checksum(guests.curve_ops()[1]), why
    # a key of another program rejects the proof; so do other FRI parameters than the proof was made with
    assert not capi.verify(vk2, proof, Q, POW)[0] and not capi.verify(vk, proof2, Q, POW)[0]
    assert not capi.verify(vk, proof, Q + 1, POW)[0] and not capi.verify(vk, proof, Q, POW + 1)[0]


@pytest.mark.parametrize("name", ["commit", "curve"])
def test_every_tampered_word_is_rejected(name):
    vk, proof = _load(name)
    words = np.frombuffer(proof, np.uint32).copy()
    rng = np.random.default_rng(9)
    # the container head, then a spread over the shard proofs (field words + 1 mod p; counts / lengths + 1)
    for pos in list(range(0, 12)) + [int(x) for x in rng.integers(12, len(words), 120)]:
        w = words.copy()
        w[pos] = (int(w[pos]) + 1) % P if w[pos] < P else int(w[pos]) - 1
        ok, _, _, why = capi.verify(vk, w.tobytes(), Q, POW)
        assert not ok, f"{name}: tampered word {pos} accepted"
    # truncation and trailing data
    assert not capi.verify(vk, proof[:-4], Q, POW)[0] and not capi.verify(vk, proof + b"\0\0\0\0", Q, POW)[0]
    assert not capi.verify(vk, proof[:len(proof) // 2], Q, POW)[0]
    # the exit-code word is compared mod p with the proven value: ec + p must be refused, not reported
    w = words.copy()
    assert w[2] == 0
    w[2] = P
    ok, _, _, why = capi.verify(vk, w.tobytes(), Q, POW)
    assert not ok and "exit code" in why


def test_verifying_key_has_one_encoding():
    vk, proof = _load("commit")
    for pos in range(len(vk)):
        for bit in (0, 7):
            bad = bytearray(vk)
            bad[pos] ^= 1 << bit
            assert not capi.verify(bytes(bad), proof, Q, POW)[0], f"verifying-key byte {pos} is ignored"
