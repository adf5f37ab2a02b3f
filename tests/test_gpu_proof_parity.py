"""THE parity test: the proof bytes produced by the HIP prover must equal, bit for
bit, the bytes of the oracle's CPU prover (tests/_oracle_prover.py: oracle stage
restatements in C + a Python duplex challenger) on the same traces and parameters.
Covers K1-K9 end to end (and K0 + multi-shard chaining through the rv32 cases)."""
import numpy as np
import pytest

from tests import _oracle_prover, guests, toy_traces

pytestmark = pytest.mark.gpu
Q, POW = 6, 5


def first_diff(a: bytes, b: bytes):
    wa, wb = np.frombuffer(a, np.uint32), np.frombuffer(b, np.uint32)
    n = min(len(wa), len(wb))
    d = np.nonzero(wa[:n] != wb[:n])[0]
    return (int(d[0]) if len(d) else n), len(wa), len(wb)


def split_container(proof: bytes):
    """-> (exit_code, public value bytes, [shard proof bytes])"""
    w = np.frombuffer(proof, np.uint32)
    assert w[0] == 0x33435644
    n, ec, pvl = int(w[1]), int(w[2]), int(w[3])
    at = 4 + (pvl + 3) // 4
    pv = w[4:at].tobytes()[:pvl]
    shards = []
    for _ in range(n):
        ln = int(w[at])
        shards.append(w[at + 1:at + 1 + ln].tobytes())
        at += 1 + ln
    assert at == len(w)
    return ec, pv, shards


def oracle_prove_execution(elf, stdin, log_shard):
    """the oracle side end to end: the traces come from the oracle's own guest machine + row expansion
    (oracle/rv32_model.py), not from the product's executor"""
    from oracle import rv32_model

    run = rv32_model.Run(elf, stdin, log_shard)
    assert run.halted and not run.error
    shards = [rv32_model.traces(run, i) for i in range(len(run.shards))]
    prep_root = _oracle_prover.prep_root_of(shards[0][0])
    headers = [_oracle_prover.main_root(chips) + [int(x) for x in pubs] for chips, pubs in shards]
    gc = _oracle_prover.global_challenges(prep_root, headers)
    return [_oracle_prover.prove_shard("rv32", chips, pubs, Q, POW, perm_challenges=gc)[0] for chips, pubs in shards]


@pytest.mark.parametrize("shape", [(6, 4, 11), (3, 0, 1), (9, 11, 1500)])
def test_toy_proof_bytes_equal_oracle(shape):
    from dvt_circuits_amd import capi

    prep, main, pubs = toy_traces.build(*shape)
    p = capi.Prover('{"fri_queries": %d, "pow_bits": %d}' % (Q, POW))
    pk, vk = p.machine_setup("toy", prep)
    gpu_proof = p.machine_prove(pk, main, pubs)
    chips = [dict(chip_id=cid, main=m, prep=(prep[0][1] if cid == toy_traces.RANGE8 else np.zeros((0, m.shape[1]), np.uint32))) for cid, m in main]
    cpu_proof, prep_root = _oracle_prover.prove_shard("toy", chips, pubs, Q, POW)
    assert gpu_proof == cpu_proof, f"first differing word / lengths: {first_diff(gpu_proof, cpu_proof)}"
    assert capi.machine_verify(vk, cpu_proof, Q, POW)[0]
    p.pk_free(pk)
    p.close()


def _encshare_case():
    """the bad_encrypted_share-shaped guest on a synthetic n = 3 input: SHA-256 KDF, ChaCha20, divu, the n^2 loop"""
    import json

    from dvt_circuits_amd import capi
    from tools import gen_dkg_input

    buf = capi.stdin_from_json("bad-encrypted-share", json.dumps(gen_dkg_input.bad_encrypted_share(3, 2)).encode())
    return guests.dkg_like("encshare"), guests.dkg_like_expected(buf, "encshare"), [buf]


@pytest.mark.parametrize("log_shard,which", [(21, "bignum"), (8, "bignum"), (9, "shifts"), (10, "muldiv"), (16, "encshare"),
                                             (21, "sha_extend"), (9, "sha_extend"), (21, "sha256_precompiled"), (10, "sha256_precompiled"),
                                             (21, "field_ops"), (9, "field_ops"), (21, "curve_ops"), (8, "curve_ops"), (13, "horner"), (21, "u256_ops"), (7, "u256_ops")])
def test_rv32_proof_bytes_equal_oracle(log_shard, which):
    from dvt_circuits_amd import capi

    stdin = ()
    if which == "encshare":
        elf, want, stdin = _encshare_case()
    elif which == "horner":
        # the reference's evaluate_polynomial on two of its known-answer keys through the G1 / Fp precompiles
        from tests import guests_bls, test_guest_bls_horner as th

        pks = [bytes.fromhex(h) for h in th.HORNER_PKS[:2]]
        elf, want = guests_bls.horner(pks, [3], subgroup_check=False), th.expected_horner(pks, [3])
    else:
        elf, want = guests.bignum(2, limbs=3) if which == "bignum" else getattr(guests, which)()
    p = capi.Prover('{"fri_queries": %d, "pow_bits": %d, "log_shard_size": %d}' % (Q, POW, log_shard))
    pk, vk = p.setup(elf)
    proof, rep = p.prove_core(pk, stdin)
    ec, pv, gpu_shards = split_container(proof)
    assert ec == 0 and pv == (want if which in ("bignum", "encshare", "horner") else guests.checksum(want))
    cpu_shards = oracle_prove_execution(elf, stdin, log_shard)
    assert len(gpu_shards) == len(cpu_shards) and (len(gpu_shards) > 1) == (log_shard < 21)
    for i, (g, c) in enumerate(zip(gpu_shards, cpu_shards)):
        assert g == c, f"shard {i}: first differing word / lengths: {first_diff(g, c)}"
    assert capi.verify(vk, proof, Q, POW)[0]
    p.pk_free(pk)
    p.close()
