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


class ChipAir(C.Structure):
    _fields_ = [("name", C.c_char_p), ("main_w", C.c_uint32), ("prep_w", C.c_uint32), ("n_pub", C.c_uint32),
                ("n_constraints", C.c_uint32), ("n_interactions", C.c_uint32), ("max_arity", C.c_uint32),
                ("when", C.c_void_p), ("inter", C.c_void_p), ("constraints", C.c_void_p), ("interactions", C.c_void_p)]


class AirOracle:
    """Constraint checker + exact LogUp multiset + K4 restatement over the generated C AIRs."""

    def __init__(self, lib, machine: str):
        self.lib = lib
        lib.orc_machine.argtypes = [C.c_char_p, C.POINTER(C.c_uint)]
        lib.orc_machine.restype = C.POINTER(ChipAir)
        lib.orc_check_constraints.argtypes = [C.POINTER(ChipAir), u32p, u32p, C.c_uint32, u32p, C.POINTER(C.c_int), u32p]
        lib.orc_check_constraints.restype = C.c_size_t
        lib.orc_multiset_new.restype = C.c_void_p
        lib.orc_multiset_free.argtypes = [C.c_void_p]
        lib.orc_multiset_add_chip.argtypes = [C.c_void_p, C.POINTER(ChipAir), u32p, u32p, C.c_uint32, u32p]
        lib.orc_multiset_unbalanced.argtypes = [C.c_void_p, u32p, C.c_size_t]
        lib.orc_multiset_unbalanced.restype = C.c_size_t
        lib.orc_perm_trace.argtypes = [C.POINTER(ChipAir), u32p, u32p, C.c_uint32, u32p, u32p, u32p, u32p, u32p]
        n = C.c_uint()
        self.chips = lib.orc_machine(machine.encode(), C.byref(n))
        self.nchips = n.value
        assert self.nchips, machine

    def chip(self, cid):
        return self.chips[cid]

    @staticmethod
    def _arr(a):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        if a.size == 0:
            a = np.zeros(1, np.uint32)
        return a

    def check_constraints(self, cid, main, prep, pubs):
        main, prep, pubs = self._arr(main), self._arr(prep), self._arr(pubs)
        log_n = int(main.shape[1]).bit_length() - 1
        bc, br = C.c_int(-1), C.c_uint32(0)
        bad = self.lib.orc_check_constraints(C.byref(self.chips[cid]), _ptr(main), _ptr(prep), log_n, _ptr(pubs), C.byref(bc), C.byref(br))
        return bad, bc.value, br.value

    def logup_unbalanced(self, chips, pubs=None, extra=()):
        """chips: list of dict(chip_id, main, prep) with one `pubs`, or a list of (chips, pubs) groups
        (one per shard; the multiset is balanced across all of them).  Returns (count, first offending record)."""
        groups = [(chips, pubs)] if pubs is not None else chips
        ms = self.lib.orc_multiset_new()
        for gchips, gpubs in groups:
            gp = self._arr(gpubs)
            for ch in gchips:
                main, prep = self._arr(ch["main"]), self._arr(ch["prep"])
                log_n = int(ch["main"].shape[1]).bit_length() - 1
                self.lib.orc_multiset_add_chip(ms, C.byref(self.chips[ch["chip_id"]]), _ptr(main), _ptr(prep), log_n, _ptr(gp))
        self.lib.orc_multiset_add_tuple.argtypes = [C.c_void_p, C.c_uint32, u32p, C.c_uint32, C.c_int, C.c_uint32]
        for bus, vals, sign, mult in extra:   # explicit tuples: (bus, values, +1 send / -1 receive, multiplicity)
            v = self._arr(vals)
            self.lib.orc_multiset_add_tuple(ms, bus, _ptr(v), len(vals), sign, mult)
        out = np.zeros(64, np.uint32)
        n = self.lib.orc_multiset_unbalanced(ms, _ptr(out), out.size)
        self.lib.orc_multiset_free(ms)
        return n, out[: 3 + int(out[1])].tolist() if n else None

    def perm_trace(self, cid, main, prep, pubs, alpha, beta):
        main, prep, pubs = self._arr(main), self._arr(prep), self._arr(pubs)
        ch = self.chips[cid]
        n = main.shape[1]
        log_n = int(n).bit_length() - 1
        w = 4 * ((ch.n_interactions + 1) // 2)     # batches but the last, then phi (oracle/air_oracle.c)
        out = np.zeros((w, n), np.uint32)
        cs = np.zeros(4, np.uint32)
        a, b = self._arr(alpha), self._arr(beta)
        self.lib.orc_perm_trace(C.byref(ch), _ptr(main), _ptr(prep), log_n, _ptr(pubs), _ptr(a), _ptr(b), _ptr(out), _ptr(cs))
        return out, cs


def air(machine: str) -> AirOracle:
    load()
    return AirOracle(_cached.lib, machine)


_cached = None


def load() -> Oracle:
    global _cached
    if _cached is None:
        build()
        _cached = Oracle(C.CDLL(LIB))
    return _cached
