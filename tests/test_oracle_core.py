"""CPU tests (no GPU): pin the C oracle against independent big-int Python
restatements and algebraic identities.  The reference holds no golden vector for
this path (parity unpinned), so these are the oracle's anchors."""
import hashlib

import numpy as np
import pytest

P = 2013265921
RC_TAG = b"dvt-amd/poseidon2-babybear-w16/rc"


def py_constants():
    need = 8 * 16 + 13
    out = []
    i = 0
    while len(out) < need:
        dg = hashlib.sha256(RC_TAG + i.to_bytes(4, "little")).digest()
        for k in range(4):
            out.append(int.from_bytes(dg[8 * k : 8 * k + 8], "little") % P)
        i += 1
    ext = [out[r * 16 : (r + 1) * 16] for r in range(8)]
    internal = out[128 : 128 + 13]
    diag = [d % P for d in (-2, 1, 2, 3, 4, -3, -4, 5, -5, 6, -6, 7, 8, -8, 9, -1)]
    return ext, internal, diag


def py_permute(s):
    ext, internal, diag = py_constants()
    s = [int(x) for x in s]

    def ext_layer(s):
        t = list(s)
        for c in range(4):
            x0, x1, x2, x3 = s[4 * c : 4 * c + 4]
            t[4 * c + 0] = (2 * x0 + 3 * x1 + x2 + x3) % P
            t[4 * c + 1] = (x0 + 2 * x1 + 3 * x2 + x3) % P
            t[4 * c + 2] = (x0 + x1 + 2 * x2 + 3 * x3) % P
            t[4 * c + 3] = (3 * x0 + x1 + x2 + 2 * x3) % P
        sums = [(t[k] + t[4 + k] + t[8 + k] + t[12 + k]) % P for k in range(4)]
        return [(t[i] + sums[i % 4]) % P for i in range(16)]

    def int_layer(s):
        tot = sum(s) % P
        return [(s[i] * diag[i] + tot) % P for i in range(16)]

    s = ext_layer(s)
    for r in range(4):
        s = [pow((s[i] + ext[r][i]) % P, 7, P) for i in range(16)]
        s = ext_layer(s)
    for r in range(13):
        s[0] = pow((s[0] + internal[r]) % P, 7, P)
        s = int_layer(s)
    for r in range(4, 8):
        s = [pow((s[i] + ext[r][i]) % P, 7, P) for i in range(16)]
        s = ext_layer(s)
    return s


def test_sha256_matches_hashlib(oracle):
    for n in [0, 1, 3, 55, 56, 63, 64, 65, 119, 120, 1000]:
        msg = bytes((i * 7 + n) & 0xFF for i in range(n))
        assert oracle.sha256(msg) == hashlib.sha256(msg).digest()


def test_poseidon2_constants_derivation(oracle):
    e, i, d = oracle.poseidon2_constants()
    pe, pi, pd = py_constants()
    assert e.tolist() == [x for r in pe for x in r]
    assert i.tolist() == pi
    assert d.tolist() == pd
    assert all(x < P for x in e.tolist() + i.tolist())


def test_internal_diagonal_satisfies_the_poseidon2_condition():
    """the product's small-integer internal diagonal (and, as a control of the checker, Plonky3's BabyBear-16 one) passes
    the Poseidon2 paper's test: the characteristic polynomial of M_I^i is irreducible for i = 1 .. 32; a diagonal with a
    repeated entry does not"""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import check_poseidon2_diag as chk
    import gen_poseidon2_rc

    assert sorted(gen_poseidon2_rc.INTERNAL_DIAG) == [-8, -6, -5, -4, -3, -2, -1, 1, 2, 3, 4, 5, 6, 7, 8, 9]
    assert py_constants()[2] == [d % P for d in gen_poseidon2_rc.INTERNAL_DIAG]
    assert chk.check(gen_poseidon2_rc.INTERNAL_DIAG)
    assert chk.check(chk.PLONKY3)
    assert not chk.check([1, 1] + list(range(2, 16)))


def test_poseidon2_permute_matches_python(oracle):
    rng = np.random.default_rng(1)
    for _ in range(4):
        s = rng.integers(0, P, 16, dtype=np.uint32)
        assert oracle.permute(s).tolist() == py_permute(s)
    z = np.zeros(16, np.uint32)
    assert oracle.permute(z).tolist() == py_permute(z)
    m = np.full(16, P - 1, np.uint32)
    assert oracle.permute(m).tolist() == py_permute(m)


def test_sponge_and_compress(oracle):
    rng = np.random.default_rng(2)
    assert oracle.hash_slice(np.zeros(0, np.uint32)).tolist() == [0] * 8
    for n in [1, 7, 8, 9, 16, 23, 100]:
        v = rng.integers(0, P, n, dtype=np.uint32)
        s = [0] * 16
        for i in range(0, n, 8):
            chunk = v[i : i + 8].tolist()
            s[: len(chunk)] = chunk
            s = py_permute(s)
        assert oracle.hash_slice(v).tolist() == s[:8]
    l = rng.integers(0, P, 8, dtype=np.uint32)
    r = rng.integers(0, P, 8, dtype=np.uint32)
    assert oracle.compress(l, r).tolist() == py_permute(l.tolist() + r.tolist())[:8]


def test_dft_matches_naive(oracle):
    rng = np.random.default_rng(3)
    for log_n in [0, 1, 2, 5, 7]:
        n = 1 << log_n
        a = rng.integers(0, P, n, dtype=np.uint32)
        w = pow(31, (P - 1) >> log_n, P)
        assert pow(w, n, P) == 1 and (n == 1 or pow(w, n // 2, P) == P - 1)
        naive = [sum(int(a[j]) * pow(w, j * k, P) for j in range(n)) % P for k in range(n)]
        f = oracle.dft(a)
        assert f.tolist() == naive
        assert oracle.dft(f, inverse=True).tolist() == a.tolist()


def test_coset_lde_is_polynomial_evaluation(oracle):
    rng = np.random.default_rng(4)
    log_n, width = 5, 3
    n = 1 << log_n
    m = rng.integers(0, P, (width, n), dtype=np.uint32)
    lde = oracle.coset_lde(m, added_bits=1, shift=31)
    assert lde.shape == (width, 2 * n)
    w2 = pow(31, (P - 1) >> (log_n + 1), P)
    for c in range(width):
        coeffs = oracle.dft(m[c], inverse=True).tolist()
        for j in [0, 1, 2, n - 1, n, 2 * n - 1]:
            x = 31 * pow(w2, j, P) % P
            v = sum(cf * pow(x, k, P) for k, cf in enumerate(coeffs)) % P
            assert int(lde[c, j]) == v
    # shift = 1 must reproduce the original evaluations at even indices
    lde1 = oracle.coset_lde(m, added_bits=1, shift=1)
    assert (lde1[:, ::2] == m).all()
    # linearity
    m2 = rng.integers(0, P, (width, n), dtype=np.uint32)
    s = ((m.astype(np.uint64) + m2) % P).astype(np.uint32)
    l2 = oracle.coset_lde(m2)
    assert (oracle.coset_lde(s) == ((lde.astype(np.uint64) + l2) % P).astype(np.uint32)).all()


def test_merkle_commit_structure(oracle):
    rng = np.random.default_rng(5)
    a = rng.integers(0, P, (3, 16), dtype=np.uint32)   # tall
    b = rng.integers(0, P, (9, 16), dtype=np.uint32)   # tall, wider than the rate
    c = rng.integers(0, P, (2, 4), dtype=np.uint32)    # short: injected at height 4
    d = rng.integers(0, P, (1, 1), dtype=np.uint32)    # height 1: injected at the root
    layers = oracle.merkle_commit([a, c, b, d])
    assert layers.shape == (31, 8)
    leaf = [oracle.hash_slice(np.concatenate([a[:, r], b[:, r]])) for r in range(16)]
    assert (layers[:16] == np.array(leaf)).all()
    l8 = [oracle.compress(leaf[i], leaf[i + 8]) for i in range(8)]
    assert (layers[16:24] == np.array(l8)).all()
    l4 = [oracle.compress(oracle.compress(l8[i], l8[i + 4]), oracle.hash_slice(c[:, i])) for i in range(4)]
    assert (layers[24:28] == np.array(l4)).all()
    l2 = [oracle.compress(l4[i], l4[i + 2]) for i in range(2)]
    root = oracle.compress(oracle.compress(l2[0], l2[1]), oracle.hash_slice(d[:, 0]))
    assert (layers[30] == root).all()
