"""GPU tests of the generic STARK engine on the toy machine: a proof produced by
the HIP prover must be accepted by the verifier; any tampering must be rejected."""
import numpy as np
import pytest

from tests import toy_traces

pytestmark = pytest.mark.gpu
CFG = '{"fri_queries": 20, "pow_bits": 8}'


@pytest.fixture(scope="module")
def gpu():
    import torch
    from dvt_circuits_amd import capi

    assert torch.cuda.is_available()
    p = capi.Prover(CFG)
    yield p
    p.close()


@pytest.mark.parametrize("log_fib,log_pairs,real", [(6, 4, 11), (3, 0, 1), (10, 13, 5000), (14, 2, 0)])
def test_toy_prove_verify(gpu, log_fib, log_pairs, real):
    from dvt_circuits_amd import capi

    prep, main, pubs = toy_traces.build(log_fib, log_pairs, real)
    pk, vk = gpu.machine_setup("toy", prep)
    proof = gpu.machine_prove(pk, main, pubs)
    ok, why = capi.machine_verify(vk, proof, 20, 8)
    assert ok, why
    # determinism: same inputs -> same bytes
    assert gpu.machine_prove(pk, main, pubs) == proof
    # wrong config / public values / any flipped word must be rejected
    assert not capi.machine_verify(vk, proof, 21, 8)[0]
    words = np.frombuffer(proof, dtype=np.uint32).copy()
    rng = np.random.default_rng(1)
    for pos in list(rng.integers(1, len(words), 25)) + [1, 9, 17, 25, 26, 27]:
        w = words.copy()
        w[pos] = (int(w[pos]) + 1) % 2013265921
        ok2, _ = capi.machine_verify(vk, w.tobytes(), 20, 8)
        assert not ok2, f"tampered word {pos} accepted"
    gpu.pk_free(pk)


def test_toy_bad_witness_is_not_provable(gpu):
    from dvt_circuits_amd import capi

    prep, main, pubs = toy_traces.build(6, 4, 11)
    pk, vk = gpu.machine_setup("toy", prep)
    # (a) break an AIR constraint: c != a + b on one row
    bad = [(c, m.copy()) for c, m in main]
    bad[1][1][2, 5] = (int(bad[1][1][2, 5]) + 1) % 256
    with pytest.raises(capi.DvtError):
        proof = gpu.machine_prove(pk, bad, pubs)
        ok, why = capi.machine_verify(vk, proof, 20, 8)
        raise capi.DvtError(5, why) if not ok else AssertionError("bad trace accepted")
    # (b) break only the lookup balance: multiplicity off by one
    bad = [(c, m.copy()) for c, m in main]
    bad[0][1][0, 200] += 1
    proof = gpu.machine_prove(pk, bad, pubs)
    ok, why = capi.machine_verify(vk, proof, 20, 8)
    assert not ok and "cumulative" in why
    # (c) wrong public value
    p2 = pubs.copy()
    p2[2] = (int(p2[2]) + 1) % 256
    with pytest.raises(capi.DvtError):
        proof = gpu.machine_prove(pk, main, p2)
        ok, why = capi.machine_verify(vk, proof, 20, 8)
        raise capi.DvtError(5, why) if not ok else AssertionError("bad public value accepted")
    gpu.pk_free(pk)
