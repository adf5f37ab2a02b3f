"""ctypes binding of the CPU oracle.  TEST INFRASTRUCTURE: imported only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libdvt_oracle.so")

P = 2013265921
u32p = C.POINTER(C.c_uint32)


class OrcMatrix(C.Structure):
    _fields_ = [("data", u32p), ("width", C.c_uint32), ("log_height", C.c_uint32)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def _ptr(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.orc_sha256.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        lib.orc_poseidon2_constants.argtypes = [u32p, u32p, u32p]
        lib.orc_poseidon2_permute.argtypes = [u32p]
        lib.orc_hash_slice.argtypes = [u32p, C.c_size_t, u32p]
        lib.orc_compress.argtypes = [u32p, u32p, u32p]
        lib.orc_dft.argtypes = [u32p, C.c_uint]
        lib.orc_idft.argtypes = [u32p, C.c_uint]
        lib.orc_coset_lde.argtypes = [u32p, u32p, C.c_uint32, C.c_uint, C.c_uint, C.c_uint32]
        lib.orc_merkle_commit.argtypes = [C.POINTER(OrcMatrix), C.c_size_t, u32p]
        lib.orc_merkle_digest_words.argtypes = [C.POINTER(OrcMatrix), C.c_size_t]
        lib.orc_merkle_digest_words.restype = C.c_size_t

    def sha256(self, b: bytes) -> bytes:
        out = C.create_string_buffer(32)
        self.lib.orc_sha256(b, len(b), out)
        return out.raw

    def poseidon2_constants(self):
        e = np.zeros(128, np.uint32)
        i = np.zeros(13, np.uint32)
        d = np.zeros(16, np.uint32)
        self.lib.orc_poseidon2_constants(_ptr(e), _ptr(i), _ptr(d))
        return e, i, d

    def permute(self, state):
        s = np.ascontiguousarray(state, dtype=np.uint32).copy()
        assert s.shape == (16,)
        self.lib.orc_poseidon2_permute(_ptr(s))
        return s

    def hash_slice(self, v):
        v = np.ascontiguousarray(v, dtype=np.uint32)
        out = np.zeros(8, np.uint32)
        self.lib.orc_hash_slice(_ptr(v), v.size, _ptr(out))
        return out

    def compress(self, l, r):
        l = np.ascontiguousarray(l, dtype=np.uint32)
        r = np.ascontiguousarray(r, dtype=np.uint32)
        out = np.zeros(8, np.uint32)
        self.lib.orc_compress(_ptr(l), _ptr(r), _ptr(out))
        return out

    def dft(self, a, inverse=False):
        a = np.ascontiguousarray(a, dtype=np.uint32).copy()
        log_n = int(a.size).bit_length() - 1
        assert 1 << log_n == a.size
        (self.lib.orc_idft if inverse else self.lib.orc_dft)(_ptr(a), log_n)
        return a

    def coset_lde(self, m, added_bits=1, shift=31):
        """m: [width][N] column-major (numpy row = one column)."""
        m = np.ascontiguousarray(m, dtype=np.uint32)
        w, n = m.shape
        log_n = int(n).bit_length() - 1
        assert 1 << log_n == n
        out = np.zeros((w, n << added_bits), np.uint32)
        self.lib.orc_coset_lde(_ptr(m), _ptr(out), w, log_n, added_bits, shift)
        return out

    def merkle_commit(self, mats):
        """mats: list of [width][height] uint32 arrays. Returns flat digest layers [(2H-1), 8]."""
        mats = [np.ascontiguousarray(m, dtype=np.uint32) for m in mats]
        arr = (OrcMatrix * len(mats))()
        for k, m in enumerate(mats):
            w, h = m.shape
            lh = int(h).bit_length() - 1
            assert 1 << lh == h
            arr[k] = OrcMatrix(_ptr(m), w, lh)
        words = self.lib.orc_merkle_digest_words(arr, len(mats))
        dg = np.zeros(words, np.uint32)
        self.lib.orc_merkle_commit(arr, len(mats), _ptr(dg))
        return dg.reshape(-1, 8)


_cached = None


def load() -> Oracle:
    global _cached
    if _cached is None:
        build()
        _cached = Oracle(C.CDLL(LIB))
    return _cached
