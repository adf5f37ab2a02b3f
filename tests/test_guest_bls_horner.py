"""CPU tests (no GPU): the reference's G1 work done INSIDE the proven machine through the field / curve precompiles
(VERDICT r2 item 1).  The guests of tests/guests_bls.py decompress the reference's compressed public keys, evaluate
`evaluate_polynomial` (reference crates/dkg/src/dkg_math.rs:160-174) / the point half of `lagrange_interpolation`
(:176-228) with double-and-add over the BLS12381_ADD / _DOUBLE precompiles, and commit the compressed result: the public
values must equal the reference's own known answers (dkg_math.rs:281-431) and tools/bls12_381.py."""
import pytest

from dvt_circuits_amd import capi
from tests import _orc, guests, guests_bls
from tests.test_rv32_exec_trace import check_traces
from tools import bls12_381 as bls
from tools import dkg_verify

# reference crates/dkg/src/dkg_math.rs:281-300 (test_evaluate_polynomial)
HORNER_PKS = ["92cad77a95432bc1030d81b5465cb69be672c1dd0da752230bf8112f8449b03149e7fa208a6fae460a9f0a1d5bd175e9",
              "98876a81fe982573ec5f986956bf9bf0bcb5349d95c3c8da0aefd05a49fea6215f59b0696f906547baed90ab245804e8",
              "ad2c4e5b631fbded449ede4dca2d040b9c7eae58d1e73b3050486c1ba22c15a92d9ff13c05c356f974447e4fca84864a"]
HORNER_TARGET = "af8e0095ecc662f65b95ce57e5bd2f8739ff93b0621a1ad53f5616538d1323ff40e6e9ddd7132298710974fe6fc0344e"
# reference crates/dkg/src/dkg_math.rs:322-346 (test_lagrange_interpolation)
LAGRANGE_PKS = ["8da434e68daef9af33e39ab727557a3cd86d7991cd6b545746bf92c8edec37012912cfa2292a21512bce9040a1c0e502",
                "a3cd061aab6013f7561978959482d79e9ca636392bc94d4bcad9cb6f90fe2cdf52100f211052f1570db0ca690b6a9903",
                "8cbfb6cb7af927cfe5fb17621df7036de539b7ff4aa0620cdc218d6b7fe7f2e714a96bdeddb2a0dc24867a90594427e1",
                "9892b390d9d3000c7bf04763006fbc617b7ba9c261fff35094aec3f43599f2c254ae667d9ba135747309b77cd02f1fbc",
                "b255c8a66fd1a13373537e8a4ba258f4990c141fc3c06daccda0711f5ebaffc092f0e5b0e4454e6344e2f97957be4017"]
LAGRANGE_TARGET = "a31d9a483703cd0da9873e5e76b4de5f7035d0a73d79b3be8667daa4fc7065a1bbb5bf77787fcf2a35bd327eecc4fa6b"


def expected_horner(pks, ids):
    cfs = [bls.g1_decompress(p) for p in pks]
    return b"".join(bls.g1_compress(dkg_verify.evaluate_polynomial(cfs, i)) for i in ids)


def test_horner_guest_reproduces_the_reference_kat():
    pks = [bytes.fromhex(h) for h in HORNER_PKS]
    elf = guests_bls.horner(pks, [1], subgroup_check=True)
    rc, rep, pv, err = capi.execute(elf)
    assert rc == 0 and rep["halted"] and not rep["unprovable"], err
    assert pv.hex() == HORNER_TARGET
    # reference :302-320 (test_evaluate_polynomial_bad_base_keys): three times the first key does NOT give the target
    rc, rep, pv, err = capi.execute(guests_bls.horner([pks[0]] * 3, [1], subgroup_check=False))
    assert rc == 0 and pv.hex() != HORNER_TARGET and pv == expected_horner([pks[0]] * 3, [1])


def test_horner_guest_on_many_ids_equals_the_python_tooling():
    pks = [bytes.fromhex(h) for h in HORNER_PKS]
    ids = [1, 2, 3, 7, 255, 0, 0x10001]
    rc, rep, pv, err = capi.execute(guests_bls.horner(pks, ids, subgroup_check=False))
    assert rc == 0, err
    assert pv == expected_horner(pks, ids)
    # the keys may also arrive as (private) stdin
    elf, buf = guests_bls.horner(pks, [4], subgroup_check=False, stdin=True)
    rc, rep, pv, err = capi.execute(elf, [buf])
    assert rc == 0 and pv == expected_horner(pks, [4]), err


def test_lincomb_guest_reproduces_the_reference_lagrange_kat():
    """r = sum_i [l_i(0)] Y_i with the Lagrange coefficients of ids 1..5 (Fr arithmetic on the host side of the test)"""
    pks = [bytes.fromhex(h) for h in LAGRANGE_PKS]
    xs = [1, 2, 3, 4, 5]
    R = bls.R
    a = 1
    for x in xs:
        a = a * x % R
    coef = []
    for i, xi in enumerate(xs):
        b = xi
        for j, xj in enumerate(xs):
            if j != i:
                b = b * (xj - xi) % R
        coef.append(a * pow(b, -1, R) % R)
    rc, rep, pv, err = capi.execute(guests_bls.lincomb(pks, coef))
    assert rc == 0, err
    assert pv.hex() == LAGRANGE_TARGET
    want = dkg_verify.lagrange_interpolation([bls.g1_decompress(p) for p in pks], xs)
    assert pv == bls.g1_compress(want)


def test_bad_points_make_the_guest_panic():
    good = bytes.fromhex(HORNER_PKS[0])
    # x with no square root of x^3 + 4; uncompressed form; x >= p
    not_on_curve = None
    x = 5
    while not_on_curve is None:
        if bls.fp_sqrt((x ** 3 + 4) % bls.P) is None:
            b = bytearray(x.to_bytes(48, "big"))
            b[0] |= 0x80
            not_on_curve = bytes(b)
        x += 1
    too_big = bytearray((bls.P + 1).to_bytes(48, "big"))
    too_big[0] |= 0x80
    for bad in (not_on_curve, bytes([good[0] & 0x7F]) + good[1:], bytes(too_big)):
        rc, rep, pv, err = capi.execute(guests_bls.horner([bad, good], [1], subgroup_check=False))
        assert rc == capi.DVT_ERR_GUEST and rep["halted"] and rep["exit_code"] == 1, err
    # a point of the curve outside the prime-order subgroup fails the subgroup check (as G1Affine::from_compressed does)
    x = 1
    while True:
        y = bls.fp_sqrt((x ** 3 + 4) % bls.P)
        if y is not None and bls.E1.mul((x, y), bls.R) is not None:
            break
        x += 1
    rogue = bls.g1_compress((x, y))
    rc, rep, pv, err = capi.execute(guests_bls.horner([rogue], [1], subgroup_check=True))
    assert rc == capi.DVT_ERR_GUEST and rep["exit_code"] == 1


def test_horner_guest_traces_satisfy_the_air_and_equal_the_model():
    import numpy as np

    from oracle import rv32_model

    air = _orc.air("rv32")
    pks = [bytes.fromhex(h) for h in HORNER_PKS[:2]]
    elf = guests_bls.horner(pks, [3], subgroup_check=False)
    rc, rep, pv, err = capi.execute(elf)
    assert rc == 0 and pv == expected_horner(pks, [3])
    chips, pubs = check_traces(air, elf, log_shard=13)
    run = rv32_model.Run(elf, (), 13)
    assert run.halted and run.cycles == rep["cycles"] and run.public_values == pv
    for pos in range(len(run.shards)):
        host, hpubs, _ = capi.rv32_debug_traces(elf, (), 13, pos)
        model, mpubs = rv32_model.traces(run, pos)
        assert (hpubs == mpubs).all() and [c["chip_id"] for c in host] == [c["chip_id"] for c in model]
        for h, m in zip(host, model):
            assert np.array_equal(h["main"], m["main"]), (pos, h["chip_id"])


def test_dkg_shaped_guest_with_real_curve_work():
    """dkg_like(curve_precompiles=True): the k point operations of every participant are G1 scalar multiplications and
    additions through the BLS12381 precompiles (Y := [id] Y + C_j, the Horner step of dkg_math.rs:160-174); the compressed
    accumulator joins the public values; its shards satisfy the AIR"""
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    example = open(os.path.join(root, "tests", "golden", "finalization_example.json"), "rb").read()
    buf = capi.stdin_from_json("finalization", example)
    elf = guests.dkg_like("finalization", curve_precompiles=True, sha_precompiles=True)
    rc, rep, pv, err = capi.execute(elf, [buf])
    want = guests.dkg_like_expected(buf, "finalization", curve_precompiles=True)
    assert rc == 0 and pv == want, err
    assert bls.g1_decompress(pv[-48:]) is not None          # the tail is a point of G1
    air = _orc.air("rv32")
    chips, pubs = check_traces(air, elf, [buf], log_shard=13)
