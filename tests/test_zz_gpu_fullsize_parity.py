"""Byte parity at PRODUCTION size (collected last on purpose: it is the slowest test and must not hide the others
behind `pytest -x`): one full shard of ~2^21 RV32IM cycles of the bench guest on the reference's example input, 100 FRI
queries, 16 proof-of-work bits — the GPU prover's shard proof must equal the oracle CPU prover's, byte for byte.
(at this size K4..K9 were only checked through properties.)"""
import os

import numpy as np
import pytest

from tests import _oracle_prover, guests

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_full_size_shard_bytes_equal_oracle():
    import bench
    from dvt_circuits_amd import capi

    buf = bench.workload_stdin()
    consts = bench.fit_constants(buf, 1)
    elf = guests.dkg_like("finalization", *consts)
    p = capi.Prover("{}")
    pk, vk = p.setup(elf)
    proof, rep = p.prove_core(pk, [buf])
    assert (1 << 21) - 16384 < rep["cycles"] <= 1 << 21
    ok, ec, pv, why = capi.verify(vk, proof)
    assert ok and pv == guests.dkg_like_expected(buf, "finalization", *consts), why
    w = np.frombuffer(proof, np.uint32)
    assert int(w[1]) == 1
    shard_gpu = w[4 + (len(pv) + 3) // 4 + 1:].tobytes()
    chips, pubs, n_shards = capi.rv32_debug_traces(elf, [buf])
    assert n_shards == 1 and max(c["main"].shape[1] for c in chips) == 1 << 21
    gc = _oracle_prover.global_challenges(_oracle_prover.prep_root_of(chips), [_oracle_prover.main_root(chips) + [int(x) for x in pubs]])
    shard_cpu, _ = _oracle_prover.prove_shard("rv32", chips, pubs, 100, 16, perm_challenges=gc)
    wa, wb = np.frombuffer(shard_gpu, np.uint32), np.frombuffer(shard_cpu, np.uint32)
    assert len(wa) == len(wb) and (wa == wb).all(), f"first differing word {int(np.nonzero(wa[:min(len(wa), len(wb))] != wb[:min(len(wa), len(wb))])[0][0]) if len(wa) == len(wb) else (len(wa), len(wb))}"
    p.pk_free(pk)
    p.close()


def _oracle_shards(elf, stdin, n_expected):
    """oracle CPU proofs of every shard of an execution at production parameters.  NOTE: at this size the oracle is fed the
    PRODUCT's host row expansion (capi.rv32_debug_traces); the independent Python model (oracle/rv32_model.py) covers the
    same guests cell by cell at <= 10^5 cycles (tests/test_rv32_model_parity.py, tests/test_gpu_k0_parity.py)."""
    from dvt_circuits_amd import capi

    shards = []
    for pos in range(n_expected):
        chips, pubs, n = capi.rv32_debug_traces(elf, stdin, 21, pos)
        assert n == n_expected
        shards.append((chips, pubs))
    headers = [_oracle_prover.main_root(chips) + [int(x) for x in pubs] for chips, pubs in shards]
    gc = _oracle_prover.global_challenges(_oracle_prover.prep_root_of(shards[0][0]), headers)
    return [_oracle_prover.prove_shard("rv32", chips, pubs, 100, 16, perm_challenges=gc)[0] for chips, pubs in shards]


def _gpu_shards(proof):
    from tests.test_gpu_proof_parity import split_container

    return split_container(proof)


def _same(a, b, what):
    wa, wb = np.frombuffer(a, np.uint32), np.frombuffer(b, np.uint32)
    n = min(len(wa), len(wb))
    d = np.nonzero(wa[:n] != wb[:n])[0]
    assert len(wa) == len(wb) and not len(d), f"{what}: first differing word {int(d[0]) if len(d) else n} (lengths {len(wa)} / {len(wb)})"


def test_two_full_shards_share_challenges_and_equal_oracle():
    """a TWO-shard execution at production parameters (2 x 2^21 cycles, 100 queries, 16 PoW bits): the
    LogUp challenges come from both shards' headers, the public values chain, the memory bus balances across the shards —
    every shard proof equals the oracle's, byte for byte"""
    import bench
    from dvt_circuits_amd import capi

    buf = bench.workload_stdin()
    consts = bench.fit_constants(buf, 2)
    elf = guests.dkg_like("finalization", *consts)
    p = capi.Prover("{}")
    pk, vk = p.setup(elf)
    proof, rep = p.prove_core(pk, [buf])
    assert (2 << 21) - 65536 < rep["cycles"] <= 2 << 21
    ok, ec, pv, why = capi.verify(vk, proof)
    assert ok and pv == guests.dkg_like_expected(buf, "finalization", *consts), why
    ec, pv2, gpu = _gpu_shards(proof)
    assert len(gpu) == 2
    cpu = _oracle_shards(elf, [buf], 2)
    for i in range(2):
        _same(gpu[i], cpu[i], f"shard {i}")
    p.pk_free(pk)
    p.close()


def test_full_size_shard_with_precompile_chips_equals_oracle():
    """the guest that hashes through SHA_EXTEND / SHA_COMPRESS and does its point operations through
    the BLS12381 precompiles, one full shard at production parameters: sha_extend, sha_compress and bls_g1 tables next to
    a 2^21-row cpu table, byte for byte against the oracle"""
    import bench
    from dvt_circuits_amd import capi

    buf = bench.workload_stdin()
    bench.SHA_PRECOMPILES, bench.CURVE_PRECOMPILES = True, True
    try:
        consts = bench.fit_constants(buf, 1)
    finally:
        bench.SHA_PRECOMPILES, bench.CURVE_PRECOMPILES = False, False
    elf = guests.dkg_like("finalization", *consts, sha_precompiles=True, curve_precompiles=True)
    p = capi.Prover("{}")
    pk, vk = p.setup(elf)
    proof, rep = p.prove_core(pk, [buf])
    assert (1 << 21) - 65536 < rep["cycles"] <= 1 << 21
    ok, ec, pv, why = capi.verify(vk, proof)
    assert ok and pv == guests.dkg_like_expected(buf, "finalization", *consts, curve_precompiles=True), why
    ec, pv2, gpu = _gpu_shards(proof)
    assert len(gpu) == 1
    chips, _, _ = capi.rv32_debug_traces(elf, [buf], 21, 0)
    assert {9 + 2, 7, 8} <= {c["chip_id"] for c in chips}          # bls_g1 (11), sha_extend (7), sha_compress (8) are present
    cpu = _oracle_shards(elf, [buf], 1)
    _same(gpu[0], cpu[0], "shard 0")
    p.pk_free(pk)
    p.close()
