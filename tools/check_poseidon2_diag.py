#!/usr/bin/env python3
"""Poseidon2 internal-layer check for BabyBear, width 16 (tools only, not on the product path).

The internal linear layer is y = M_I x with M_I = J + diag(d) (J = all ones): y_i = sum_j x_j + d_i x_i.  The Poseidon2
paper (Grassi, Khovratovich, Schofnegger, "Poseidon2: A Faster Version of the Poseidon Hash Function", section 5.3 and
its parameter script) asks, against invariant-subspace trails through the partial rounds, that for i = 1 .. 2t the
minimal polynomial of M_I^i has maximal degree t and is irreducible over F_p - equivalently (degree t) that the
characteristic polynomial of M_I^i is irreducible.  This script checks exactly that, in plain Python:

  * characteristic polynomial by Faddeev-LeVerrier (divisions by 1..16 are fine mod p),
  * irreducibility by Rabin's test: x^(p^16) = x mod f and gcd(x^(p^8) - x, f) = 1.

    python tools/check_poseidon2_diag.py                 # the diagonal the product uses + the Plonky3 one as a control
    python tools/check_poseidon2_diag.py --search 8      # list sets of 16 distinct integers in [-8, 8] that pass
"""
import argparse
import itertools
import sys

P = 2013265921
T = 16

# the BabyBear width-16 diagonal of Plonky3 (shift-friendly in Montgomery form), kept as the checker's control
PLONKY3 = ["-2", "1", "2", "1/2", "3", "4", "-1/2", "-3", "-4", "1/256", "1/4", "1/8", "1/134217728", "-1/256", "-1/16", "-1/134217728"]


def parse(v):
    v = str(v)
    if "/" in v:
        a, b = v.split("/")
        return int(a) * pow(int(b), P - 2, P) % P
    return int(v) % P


def matmul(a, b):
    n = len(a)
    return [[sum(a[i][k] * b[k][j] for k in range(n)) % P for j in range(n)] for i in range(n)]


def charpoly(m):
    """monic characteristic polynomial, coefficients low -> high (Faddeev-LeVerrier)"""
    n = len(m)
    c = [0] * (n + 1)
    c[n] = 1
    mk = [[0] * n for _ in range(n)]
    for k in range(1, n + 1):
        # M_k = m * M_{k-1} + c_{n-k+1} I
        mk = matmul(m, mk)
        for i in range(n):
            mk[i][i] = (mk[i][i] + c[n - k + 1]) % P
        tr = sum(sum(m[i][j] * mk[j][i] for j in range(n)) for i in range(n)) % P
        c[n - k] = (-tr) * pow(k, P - 2, P) % P
    return c


def pmod(a, f):
    a = a[:]
    n = len(f) - 1
    for i in range(len(a) - 1, n - 1, -1):
        q = a[i]
        if q:
            for j in range(n + 1):
                a[i - n + j] = (a[i - n + j] - q * f[j]) % P
    a = a[:n]
    return a + [0] * (n - len(a))


def pmul(a, b, f):
    r = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                r[i + j] = (r[i + j] + x * y) % P
    return pmod(r, f)


def ppow(a, e, f):
    r = [1] + [0] * (len(f) - 2)
    while e:
        if e & 1:
            r = pmul(r, a, f)
        a = pmul(a, a, f)
        e >>= 1
    return r


def pgcd(a, b):
    def strip(x):
        while x and x[-1] == 0:
            x = x[:-1]
        return x

    a, b = strip(a[:]), strip(b[:])
    while b:
        inv = pow(b[-1], P - 2, P)
        while len(a) >= len(b):
            q = a[-1] * inv % P
            sh = len(a) - len(b)
            for j, y in enumerate(b):
                a[sh + j] = (a[sh + j] - q * y) % P
            a = strip(a)
        a, b = b, a
    return a


def irreducible(f):
    """Rabin: f (monic, degree 16) is irreducible over F_p"""
    n = len(f) - 1
    x = [0, 1] + [0] * (n - 2)
    fr = x
    for k in range(1, n + 1):
        fr = ppow(fr, P, f)          # x^(p^k) mod f
        if k == n // 2:
            d = fr[:]
            d[1] = (d[1] - 1) % P
            if len(pgcd(f, d)) != 1:
                return False
    return fr == x


def check(diag, verbose=False):
    d = [parse(v) for v in diag]
    assert len(d) == T
    m = [[(1 + (d[i] if i == j else 0)) % P for j in range(T)] for i in range(T)]
    mi = m
    for i in range(1, 2 * T + 1):
        if not irreducible(charpoly(mi)):
            if verbose:
                print(f"  fails at power {i}")
            return False
        mi = matmul(m, mi)
    return True


def product_diag():
    """the diagonal the product and the oracle use (tools/gen_poseidon2_rc.py)"""
    import os

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gen_poseidon2_rc

    return gen_poseidon2_rc.INTERNAL_DIAG


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--search", type=int, default=0, help="search sets of 16 distinct integers in [-B, B]")
    ap.add_argument("--limit", type=int, default=3)
    a = ap.parse_args()
    if a.search:
        pool = [v for v in range(-a.search, a.search + 1)]
        found = 0
        # prefer small magnitudes: drop the candidates one at a time from the outside in
        for drop in itertools.combinations(sorted(pool, key=lambda v: -abs(v)), len(pool) - T):
            cand = [v for v in pool if v not in drop]
            if check(cand):
                print("passes:", cand)
                found += 1
                if found >= a.limit:
                    break
        sys.exit(0 if found else 1)
    ok_control = check(PLONKY3, True)
    print("control (Plonky3 BabyBear-16 diagonal):", "passes" if ok_control else "FAILS")
    diag = product_diag()
    ok = check(diag, True)
    print("product diagonal", diag, ":", "passes" if ok else "FAILS")
    sys.exit(0 if ok and ok_control else 1)
