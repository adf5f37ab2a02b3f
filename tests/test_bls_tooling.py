"""CPU tests of the BLS12-381 test tooling (tools/bls12_381.py, tools/dkg_verify.py, tools/gen_dkg_input.py) against the
reference's own known answers: the signature KAT and its negative cases (crates/dkg/src/dkg_math.rs:258-278,
crypto/bls_common.rs:134-159), the Horner KAT (:281-297), the Lagrange KATs (:319-431), the id encoding
(crypto/bls_common.rs:42-47), and the reference's real finalization vectors, which its harness expects to be accepted
(tests/golden/README_host_inputs.json).  With those pinned, the synthetic generator's real-key inputs are checked by the
same restated verifier."""
import copy
import json
import os

import pytest

from tools import bls12_381 as B
from tools import dkg_verify, gen_dkg_input

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = bytes.fromhex


def test_signature_kat_and_negatives():
    B.ensure_ready()
    pk, sig = B.g1_decompress(B.KAT_PK), B.g2_decompress(B.KAT_SIG)
    assert B.bls_verify(pk, sig, B.KAT_MSG)
    assert not B.bls_verify(pk, sig, h("00"))
    wrong_pk = B.g1_decompress(h("98876a81fe982573ec5f986956bf9bf0bcb5349d95c3c8da0aefd05a49fea6215f59b0696f906547baed90ab245804e8"))
    assert not B.bls_verify(wrong_pk, sig, B.KAT_MSG)
    bad_sig = B.g2_decompress(h("999e7b24bee2587d687e8f358ed10627ef57ec54935bd7a500bbbb18a57e7aa21b800f8b1f487a980d7c93918fdbd8020b66ce9a9e5788a4826e610ac937d8"
                                "c2ce0ad9c0ee9a5732cf73052493e9a500cc5100a15bdbf9e5b79104db52dbf07c"))
    assert not B.bls_verify(pk, bad_sig, B.KAT_MSG)
    # hash-to-G2 lands in the r-torsion and is deterministic; a signature made here verifies
    hm = B.hash_to_g2(b"hello")
    assert B.E2.on_curve(hm) and B.E2.mul(hm, B.R) is None and hm != B.hash_to_g2(b"world")
    sk = 0x1234567890ABCDEF
    assert B.bls_verify(B.E1.mul(B.G1, sk), B.sign(sk, b"msg"), b"msg")


def test_encodings_round_trip_and_reject_garbage():
    for pt in (B.G1, B.E1.mul(B.G1, 7), None):
        assert B.g1_decompress(B.g1_compress(pt)) == pt
    for pt in (B.G2, B.E2.mul(B.G2, 11), None):
        assert B.g2_decompress(B.g2_compress(pt)) == pt
    for bad in (bytes(48), b"\x80" + bytes(47)):
        with pytest.raises(ValueError):
            B.g1_decompress(bad)                      # (the reference's to_g1_affine_slow errors on all-zero bytes, bls_common.rs:161-164)
    with pytest.raises(ValueError):
        B.g2_decompress(bytes(96))


def test_horner_and_lagrange_kats():
    pks = [B.g1_decompress(h(x)) for x in (
        "92cad77a95432bc1030d81b5465cb69be672c1dd0da752230bf8112f8449b03149e7fa208a6fae460a9f0a1d5bd175e9",
        "98876a81fe982573ec5f986956bf9bf0bcb5349d95c3c8da0aefd05a49fea6215f59b0696f906547baed90ab245804e8",
        "ad2c4e5b631fbded449ede4dca2d040b9c7eae58d1e73b3050486c1ba22c15a92d9ff13c05c356f974447e4fca84864a")]
    target = "af8e0095ecc662f65b95ce57e5bd2f8739ff93b0621a1ad53f5616538d1323ff40e6e9ddd7132298710974fe6fc0344e"
    assert B.g1_compress(dkg_verify.evaluate_polynomial(pks, 1)).hex() == target
    assert B.g1_compress(dkg_verify.evaluate_polynomial([pks[0]] * 3, 1)).hex() != target
    shares = [B.g1_decompress(h(x)) for x in (
        "8da434e68daef9af33e39ab727557a3cd86d7991cd6b545746bf92c8edec37012912cfa2292a21512bce9040a1c0e502",
        "a3cd061aab6013f7561978959482d79e9ca636392bc94d4bcad9cb6f90fe2cdf52100f211052f1570db0ca690b6a9903",
        "8cbfb6cb7af927cfe5fb17621df7036de539b7ff4aa0620cdc218d6b7fe7f2e714a96bdeddb2a0dc24867a90594427e1",
        "9892b390d9d3000c7bf04763006fbc617b7ba9c261fff35094aec3f43599f2c254ae667d9ba135747309b77cd02f1fbc",
        "b255c8a66fd1a13373537e8a4ba258f4990c141fc3c06daccda0711f5ebaffc092f0e5b0e4454e6344e2f97957be4017")]
    target = "a31d9a483703cd0da9873e5e76b4de5f7035d0a73d79b3be8667daa4fc7065a1bbb5bf77787fcf2a35bd327eecc4fa6b"
    assert B.g1_compress(dkg_verify.lagrange_interpolation(shares, [1, 2, 3, 4, 5])).hex() == target
    assert B.g1_compress(dkg_verify.lagrange_interpolation([shares[4]] + shares[:4], [5, 1, 2, 3, 4])).hex() == target      # out of order
    assert B.g1_compress(dkg_verify.lagrange_interpolation([shares[1], shares[0]] + shares[2:], [1, 2, 3, 4, 5])).hex() != target
    assert B.g1_compress(dkg_verify.lagrange_interpolation([shares[1]] * 5, [1, 2, 3, 4, 5])).hex() != target
    with pytest.raises(ValueError):
        dkg_verify.lagrange_interpolation(shares[:2], [3, 3])


@pytest.mark.parametrize("name", ["finalization_example.json", "finalization_no_auth_report1.json"])
def test_reference_finalization_vectors_are_accepted_and_mutations_rejected(name):
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", name)))
    assert dkg_verify.verify_finalization(doc) is None
    bad = copy.deepcopy(doc)
    bad["aggregate_pubkey"] = doc["generations"][0]["partial_pubkey"]
    assert "aggregate" in dkg_verify.verify_finalization(bad, check_signatures=False)
    bad = copy.deepcopy(doc)
    bad["generations"][1]["message_signature"] = doc["generations"][0]["message_signature"]
    assert "Invalid signature" in dkg_verify.verify_finalization(bad)
    bad = copy.deepcopy(doc)
    bad["generations"][0]["base_pubkeys"][0] = doc["generations"][1]["base_pubkeys"][0]
    assert "commitment hash" in dkg_verify.verify_finalization(bad, check_signatures=False)


def test_generator_emits_inputs_the_finalization_check_accepts():
    doc = gen_dkg_input.finalization(4, 3, real_keys=True)
    assert dkg_verify.verify_finalization(doc) is None
    # and the host encoder takes it like the reference's own files
    from dvt_circuits_amd import capi

    assert len(capi.stdin_from_json("finalization", json.dumps(doc).encode())) > 2000
    # pseudo-random "points" (real_keys=False) are what a DKG-verifying guest rejects
    assert dkg_verify.verify_finalization(gen_dkg_input.finalization(4, 3), check_signatures=False) is not None
