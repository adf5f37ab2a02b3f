"""Trace generation for the toy machine (tools/airgen/toy.py) used by the engine tests."""
import numpy as np

RANGE8, FIB, PAIRS = 0, 1, 2


def build(log_fib=6, log_pairs=4, real_pairs=11, seed=0):
    rng = np.random.default_rng(seed)
    nf = 1 << log_fib
    fib = np.zeros((4, nf), np.uint32)
    a, b = 3, 7
    pubs = [a, b, 0]
    for r in range(nf):
        s = a + b
        fib[0, r], fib[1, r], fib[2, r], fib[3, r] = a, b, s & 255, s >> 8
        a, b = b, s & 255
    pubs[2] = int(fib[2, nf - 1])
    npairs = 1 << log_pairs
    pairs = np.zeros((4, npairs), np.uint32)
    for r in range(real_pairs):
        x, y = rng.integers(0, 16, 2)
        pairs[:, r] = (x, y, x * y, 1)
    mult = np.zeros(256, np.uint32)
    for v in fib[2]:
        mult[v] += 1
    for r in range(real_pairs):
        for v in pairs[:3, r]:
            mult[v] += 1
    prep = [(RANGE8, np.arange(256, dtype=np.uint32)[None, :])]
    main = [(RANGE8, mult[None, :]), (FIB, fib), (PAIRS, pairs)]
    return prep, main, np.array(pubs, np.uint32)
