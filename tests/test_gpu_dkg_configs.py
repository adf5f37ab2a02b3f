"""GPU tests of the BASELINE.json configurations on the DKG-shaped guests (tests/guests.py:dkg_like; the reference's own
guest ELFs cannot be built here and its bundled one is prebuilt machine code):
  configs[2]  bad_encrypted_share-shaped input, synthetic n = 64 participants, 1 GPU
  configs[3]  finalization, synthetic n = 255 (the largest legal n), >= 8 shards of 2^21 cycles
  configs[4]  batch of independent proofs (small B here; bench.py --batch runs the big one)
Each proof must verify under production parameters with exactly the public values the Python model of the guest
computes from the same stdin bytes."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import guests

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    import torch
    from dvt_circuits_amd import capi

    assert torch.cuda.is_available()
    p = capi.Prover("{}")          # production parameters: 100 queries, 16 PoW bits, 2^21-cycle shards
    yield p
    p.close()


def test_config2_bad_encrypted_share_shaped_n64(gpu):
    from dvt_circuits_amd import capi
    from tools import gen_dkg_input

    doc = gen_dkg_input.bad_encrypted_share(64, 2)
    buf = capi.stdin_from_json("bad-encrypted-share", json.dumps(doc).encode())
    elf = guests.dkg_like("encshare")
    want = guests.dkg_like_expected(buf, "encshare")
    pk, vk = gpu.setup(elf)
    proof, rep = gpu.prove_core(pk, [buf])
    assert rep["cycles"] > 1 << 21                                  # two shards
    ok, ec, pv, why = capi.verify(vk, proof)
    assert ok and ec == 0 and pv == want, why
    # the first 192 public-value bytes are the ChaCha20-decrypted share message region: a different key changes them
    other = bytearray(buf)
    other[20] ^= 1
    proof2, _ = gpu.prove_core(pk, [bytes(other)])
    ok2, _, pv2, _ = capi.verify(vk, proof2)
    assert ok2 and pv2 == guests.dkg_like_expected(bytes(other), "encshare") and pv2[:192] != pv[:192]
    gpu.pk_free(pk)


def test_config3_finalization_n255_multi_shard(gpu):
    from dvt_circuits_amd import capi
    from tools import gen_dkg_input

    doc = gen_dkg_input.finalization(255, 2)
    buf = capi.stdin_from_json("finalization", json.dumps(doc).encode())
    assert len(buf) > 160_000                                        # the ~170 KB input of SURVEY.md section 8(a) a1
    elf = guests.dkg_like("finalization")
    want = guests.dkg_like_expected(buf, "finalization")
    assert len(want) == 32 * 255 + 144
    pk, vk = gpu.setup(elf)
    proof, rep = gpu.prove_core(pk, [buf])
    n_shards = int(np.frombuffer(proof, np.uint32)[1])
    assert n_shards >= 8 and n_shards == (rep["cycles"] + (1 << 21) - 1) >> 21
    ok, ec, pv, why = capi.verify(vk, proof)
    assert ok and ec == 0 and pv == want, why
    # shard-parallel path (what bench.py --gpus N runs): two "ranks" prepare disjoint halves of the shards of the same
    # execution; headers exchanged, common challenges, the assembled proof is byte-identical to the single-call one
    jobs = [gpu.prepare(pk, [buf], first=r, stride=2)[0] for r in range(2)]
    assert all(gpu.job_shards(j) == n_shards for j in jobs)
    headers = [gpu.commit_shard(pk, jobs[i % 2], i) for i in range(n_shards)]
    ch = capi.rv32_challenges(vk, headers)
    parts = [gpu.prove_shard(pk, jobs[i % 2], i, ch) for i in range(n_shards)]
    assert gpu.assemble(jobs[0], parts) == proof
    with pytest.raises(capi.DvtError):
        gpu.commit_shard(pk, jobs[0], 1)                             # shard 1 belongs to the other half
    for j in jobs:
        gpu.job_free(j)
    gpu.pk_free(pk)


def test_config4_batch_of_independent_proofs(gpu):
    """B independent proofs of distinct instances (the example input with its own gen_id each), one setup"""
    from dvt_circuits_amd import capi

    example = json.load(open(os.path.join(ROOT, "tests", "golden", "finalization_example.json")))
    elf = guests.dkg_like("finalization")
    pk, vk = gpu.setup(elf)
    seen = set()
    for i in range(3):
        doc = dict(example, settings=dict(example["settings"], gen_id=hashlib.sha256(b"batch%d" % i).digest()[:16].hex()))
        buf = capi.stdin_from_json("finalization", json.dumps(doc).encode())
        proof, rep = gpu.prove_core(pk, [buf])
        ok, ec, pv, why = capi.verify(vk, proof)
        assert ok and ec == 0 and pv == guests.dkg_like_expected(buf, "finalization"), why
        seen.add(pv)
    assert len(seen) == 3
    gpu.pk_free(pk)


def test_config4_concurrent_handles_reproduce_the_sequential_bytes(gpu):
    """bench.py --batch runs several prover handles of one GPU from their own host threads: every proof made that way is
    byte-identical to the one the single handle makes alone (handles share no mutable state)"""
    import threading

    from dvt_circuits_amd import capi

    example = json.load(open(os.path.join(ROOT, "tests", "golden", "finalization_example.json")))
    elf = guests.dkg_like("finalization")
    bufs = []
    for i in range(4):
        doc = dict(example, settings=dict(example["settings"], gen_id=hashlib.sha256(b"conc%d" % i).digest()[:16].hex()))
        bufs.append(capi.stdin_from_json("finalization", json.dumps(doc).encode()))
    pk, vk = gpu.setup(elf)
    want = [gpu.prove_core(pk, [b])[0] for b in bufs]
    gpu.pk_free(pk)
    handles = [capi.Prover("{}") for _ in range(2)]
    got, errs = [None] * len(bufs), []

    def lane(k):
        try:
            h = handles[k]
            hpk, _ = h.setup(elf)
            for rep in range(2):                                    # twice: pooled buffers are reused across proofs
                for i in range(k, len(bufs), 2):
                    got[i] = h.prove_core(hpk, [bufs[i]])[0]
            h.pk_free(hpk)
        except Exception as e:                                      # noqa: BLE001 - reported by the assert below
            errs.append(repr(e))

    th = [threading.Thread(target=lane, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for h in handles:
        h.close()
    assert not errs, errs
    assert got == want
    assert capi.verify(vk, got[3])[0]


def test_finalization_guest_with_sha_precompiles(gpu):
    """the finalization-shaped guest hashing through SHA_EXTEND / SHA_COMPRESS (as the reference's guests do through the
    patched sha2 crate): the proof verifies with the public values of the software-SHA form"""
    from dvt_circuits_amd import capi

    buf = capi.stdin_from_json("finalization", open(os.path.join(ROOT, "tests", "golden", "finalization_example.json"), "rb").read())
    elf = guests.dkg_like("finalization", 40, 10, 1, sha_precompiles=True)
    pk, vk = gpu.setup(elf)
    proof, rep = gpu.prove_core(pk, [buf])
    ok, ec, pv, why = capi.verify(vk, proof)
    assert ok and ec == 0 and pv == guests.dkg_like_expected(buf, "finalization", 40, 10, 1), why
    gpu.pk_free(pk)


def test_bench_batch_mode_line():
    """bench.py --batch (BASELINE configs[4], replicas only) end to end, small B"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["shards_per_step"] == 2
    assert "batch of 2 independent" in line["config"]["workload"] and line["roofline"]["frac"] > 0
