"""THE parity test: the proof bytes produced by the HIP prover must equal, bit for
bit, the bytes of the oracle's CPU prover (tests/_oracle_prover.py: oracle stage
restatements in C + a Python duplex challenger) on the same traces and parameters.
Covers K1-K9 end to end (and K0 through the rv32 case)."""
import numpy as np
import pytest

from tests import _oracle_prover, guests, toy_traces

pytestmark = pytest.mark.gpu
Q, POW = 6, 5
CFG = '{"fri_queries": %d, "pow_bits": %d}' % (Q, POW)


def first_diff(a: bytes, b: bytes):
    wa, wb = np.frombuffer(a, np.uint32), np.frombuffer(b, np.uint32)
    n = min(len(wa), len(wb))
    d = np.nonzero(wa[:n] != wb[:n])[0]
    return (int(d[0]) if len(d) else n), len(wa), len(wb)


@pytest.mark.parametrize("shape", [(6, 4, 11), (3, 0, 1), (9, 11, 1500)])
def test_toy_proof_bytes_equal_oracle(shape):
    from dvt_circuits_amd import capi

    prep, main, pubs = toy_traces.build(*shape)
    p = capi.Prover(CFG)
    pk, vk = p.machine_setup("toy", prep)
    gpu_proof = p.machine_prove(pk, main, pubs)
    chips = [dict(chip_id=cid, main=m, prep=(prep[0][1] if cid == toy_traces.RANGE8 else np.zeros((0, m.shape[1]), np.uint32))) for cid, m in main]
    cpu_proof, prep_root = _oracle_prover.prove_shard("toy", chips, pubs, Q, POW)
    assert gpu_proof == cpu_proof, f"first differing word / lengths: {first_diff(gpu_proof, cpu_proof)}"
    assert capi.machine_verify(vk, cpu_proof, Q, POW)[0]
    p.pk_free(pk)
    p.close()


def test_rv32_proof_bytes_equal_oracle():
    from dvt_circuits_amd import capi

    elf, want = guests.bignum(2, limbs=3)
    p = capi.Prover(CFG)
    pk, vk = p.setup(elf)
    proof, rep = p.prove_core(pk)
    words = np.frombuffer(proof, np.uint32)
    body = 4 + (len(want) + 3) // 4
    shard_gpu = words[body + 1:].tobytes()
    chips, pubs = capi.rv32_debug_traces(elf)   # host traces == device traces (test_gpu_k0_parity)
    shard_cpu, _ = _oracle_prover.prove_shard("rv32", chips, pubs, Q, POW)
    assert shard_gpu == shard_cpu, f"first differing word / lengths: {first_diff(shard_gpu, shard_cpu)}"
    p.pk_free(pk)
    p.close()
