"""GPU parity tests (run with -m gpu on an MI355X): the HIP stage kernels,
called through the C-ABI, against the CPU oracle on the same seeded inputs —
bit-exact (integer work) — plus size-independent properties at full sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 2013265921


@pytest.fixture(scope="module")
def gpu():
    import torch
    from dvt_circuits_amd import capi

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    p = capi.Prover()
    yield p
    p.close()


def dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint32)


def test_representation_roundtrip(gpu):
    rng = np.random.default_rng(0)
    a = rng.integers(0, P, 100003, dtype=np.uint32)
    t = dev(a)
    gpu.to_internal(t)
    gpu.sync()
    m = host(t)
    assert ((m.astype(np.uint64) * pow(1 << 32, -1, P)) % P == a).all()
    gpu.from_internal(t)
    gpu.sync()
    assert (host(t) == a).all()


def test_poseidon2_permute_matches_oracle(gpu, oracle):
    rng = np.random.default_rng(1)
    s = rng.integers(0, P, (300, 16), dtype=np.uint32)
    s[0] = 0
    s[1] = P - 1
    t = dev(s)
    gpu.to_internal(t)
    gpu.poseidon2_permute(t)
    gpu.from_internal(t)
    gpu.sync()
    got = host(t).reshape(-1, 16)
    for i in range(0, 300, 7):
        assert (got[i] == oracle.permute(s[i])).all()


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 6, 10, 11, 12, 13, 14, 16])
@pytest.mark.parametrize("shift_mode", [0, 1, 2])
def test_coset_lde_matches_oracle(gpu, oracle, log_n, shift_mode):
    import torch

    if shift_mode and log_n in (1, 3, 11, 16):
        pytest.skip("covered by the other sizes")
    rng = np.random.default_rng(100 + log_n)
    width = 5 if log_n < 14 else 2
    n = 1 << log_n
    m = rng.integers(0, P, (width, n), dtype=np.uint32)
    shift = {0: 31, 1: 1, 2: pow(pow(31, (P - 1) >> (log_n + 1), P), -1, P)}[shift_mode]
    want = oracle.coset_lde(m, 1, shift)
    t_in = dev(m)
    t_out = torch.empty(width * 2 * n, dtype=torch.int32, device="cuda")
    gpu.to_internal(t_in)
    gpu.coset_lde(t_in, t_out, width, log_n, shift_mode)
    gpu.from_internal(t_out)
    gpu.sync()
    got = host(t_out).reshape(width, 2 * n)
    assert (got == want).all()


def test_coset_lde_edge_width_zero(gpu):
    import torch

    t = torch.empty(0, dtype=torch.int32, device="cuda")
    gpu.coset_lde(t, t, 0, 4)
    gpu.sync()


@pytest.mark.parametrize("log_n", [18, 21, 22])
def test_coset_lde_full_size_properties(gpu, oracle, log_n):
    """At BASELINE sizes the oracle is too slow for every element; check (a) one
    column fully against the oracle, (b) shift=1 reproduces the input at even
    indices (interpolation), (c) linearity across columns."""
    import torch

    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    a = rng.integers(0, P, n, dtype=np.uint32)
    b = rng.integers(0, P, n, dtype=np.uint32)
    s = ((a.astype(np.uint64) + b) % P).astype(np.uint32)
    m = np.stack([a, b, s])
    for mode in (0, 1):
        t_in = dev(m)
        t_out = torch.empty(3 * 2 * n, dtype=torch.int32, device="cuda")
        gpu.to_internal(t_in)
        gpu.coset_lde(t_in, t_out, 3, log_n, mode)
        gpu.from_internal(t_out)
        gpu.sync()
        got = host(t_out).reshape(3, 2 * n)
        assert (((got[0].astype(np.uint64) + got[1]) % P) == got[2]).all()
        if mode == 1:
            assert (got[:, ::2] == m).all()
        elif log_n <= 21:
            assert (got[0] == oracle.coset_lde(a[None, :], 1, 31)[0]).all()


def test_merkle_commit_matches_oracle(gpu, oracle):
    import torch

    rng = np.random.default_rng(7)
    shapes = [(3, 10), (9, 10), (17, 10), (2, 7), (8, 7), (1, 3), (5, 0)]
    mats = [rng.integers(0, P, (w, 1 << lh), dtype=np.uint32) for w, lh in shapes]
    want = oracle.merkle_commit(mats)
    ts = [dev(m) for m in mats]
    for t in ts:
        gpu.to_internal(t)
    dg = torch.empty(want.size, dtype=torch.int32, device="cuda")
    gpu.merkle_commit([(t, w, lh) for t, (w, lh) in zip(ts, shapes)], dg)
    gpu.from_internal(dg)
    gpu.sync()
    assert (host(dg).reshape(-1, 8) == want).all()


@pytest.mark.parametrize("shapes", [[(1, 0)], [(8, 1)], [(7, 5), (1, 5)], [(40, 12)]])
def test_merkle_commit_edge_shapes(gpu, oracle, shapes):
    import torch

    rng = np.random.default_rng(8)
    mats = [rng.integers(0, P, (w, 1 << lh), dtype=np.uint32) for w, lh in shapes]
    want = oracle.merkle_commit(mats)
    ts = [dev(m) for m in mats]
    for t in ts:
        gpu.to_internal(t)
    dg = torch.empty(want.size, dtype=torch.int32, device="cuda")
    gpu.merkle_commit([(t, w, lh) for t, (w, lh) in zip(ts, shapes)], dg)
    gpu.from_internal(dg)
    gpu.sync()
    assert (host(dg).reshape(-1, 8) == want).all()


def test_merkle_full_size_root_of_subtrees(gpu, oracle):
    """2^20 rows x 24 columns: the GPU root must equal the oracle's compression
    of the GPU's own level-10 digests (checksum of checksums), and 64 sampled
    leaves must equal the oracle's sponge of those rows."""
    import torch

    rng = np.random.default_rng(9)
    w, lh = 24, 20
    m = rng.integers(0, P, (w, 1 << lh), dtype=np.uint32)
    t = dev(m)
    gpu.to_internal(t)
    words = ((2 << lh) - 1) * 8
    dg = torch.empty(words, dtype=torch.int32, device="cuda")
    gpu.merkle_commit([(t, w, lh)], dg)
    gpu.from_internal(dg)
    gpu.sync()
    layers = host(dg).reshape(-1, 8)
    for r in rng.integers(0, 1 << lh, 64):
        assert (layers[r] == oracle.hash_slice(m[:, r])).all()
    off = sum((1 << lh) >> k for k in range(10))
    lvl = layers[off : off + (1 << (lh - 10))]
    cur = [x for x in lvl]
    while len(cur) > 1:
        cur = [oracle.compress(cur[i], cur[i + len(cur) // 2]) for i in range(len(cur) // 2)]
    assert (layers[-1] == cur[0]).all()
