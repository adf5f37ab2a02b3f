"""ctypes binding of include/dvt_prover.h.  Loads the in-tree HIP library
`dvt_circuits_amd/libdvt_prover.so`; raises if it is missing — there is no
fallback path."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdvt_prover.so")
CSRC = os.path.join(HERE, "csrc")

DVT_OK, DVT_ERR_GUEST, DVT_ERR_INPUT, DVT_ERR_DEVICE, DVT_ERR_UNSUPPORTED, DVT_ERR_REJECTED = 0, 1, 2, 3, 4, 5
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


class DevMatrix(C.Structure):
    _fields_ = [("d_data", C.c_void_p), ("width", C.c_uint32), ("log_height", C.c_uint32)]


class HostTrace(C.Structure):
    _fields_ = [("chip_id", C.c_uint32), ("log_n", C.c_uint32), ("data", u32p)]


class DvtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dvt error {code}: {msg}")
        self.code = code
        self.msg = msg


def build():
    """Compile the HIP library for gfx950 (cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC])


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc); there is no CPU fallback")
    try:
        # PyTorch bundles its own HIP runtime; whichever runtime initialises the GPU first owns it, so
        # when torch is going to be used in this process (device-memory plumbing) it must be loaded first.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
    lib.dvt_abi_version.restype = u32
    lib.dvt_prover_create.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.dvt_prover_destroy.argtypes = [vp]
    lib.dvt_prover_destroy.restype = None
    lib.dvt_last_error.argtypes = [vp]
    lib.dvt_last_error.restype = C.c_char_p
    lib.dvt_free.argtypes = [vp]
    lib.dvt_free.restype = None
    lib.dvt_stream.argtypes = [vp]
    lib.dvt_stream.restype = vp
    lib.dvt_sync.argtypes = [vp]
    lib.dvt_dev_to_internal.argtypes = [vp, vp, sz]
    lib.dvt_dev_from_internal.argtypes = [vp, vp, sz]
    lib.dvt_stage_coset_lde.argtypes = [vp, vp, vp, vp, u32, u32, u32]
    lib.dvt_merkle_digest_words.argtypes = [C.POINTER(DevMatrix), sz]
    lib.dvt_merkle_digest_words.restype = sz
    lib.dvt_stage_merkle_commit.argtypes = [vp, C.POINTER(DevMatrix), sz, vp]
    lib.dvt_stage_poseidon2_permute.argtypes = [vp, vp, sz]
    lib.dvt_stage_fri_fold.argtypes = [vp, vp, vp, vp, u32p, u32]
    lib.dvt_machine_setup.argtypes = [vp, C.c_char_p, C.POINTER(HostTrace), sz, C.POINTER(vp), C.POINTER(u8p), C.POINTER(sz)]
    lib.dvt_pk_free.argtypes = [vp, vp]
    lib.dvt_pk_free.restype = None
    lib.dvt_machine_prove.argtypes = [vp, vp, C.POINTER(HostTrace), sz, u32p, sz, C.POINTER(u8p), C.POINTER(sz)]
    lib.dvt_machine_verify.argtypes = [C.c_char_p, sz, C.c_char_p, sz, u32, u32, C.POINTER(C.c_char_p)]
    lib.dvt_last_stage_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.dvt_setup.argtypes = [vp, C.c_char_p, sz, C.POINTER(vp), C.POINTER(u8p), C.POINTER(sz)]
    lib.dvt_execute.argtypes = [C.c_char_p, sz, C.POINTER(Buf), sz, C.c_uint64, C.POINTER(u8p), C.POINTER(sz), C.POINTER(Report), C.POINTER(C.c_char_p)]
    lib.dvt_execute_io.argtypes = [C.c_char_p, sz, C.POINTER(Buf), sz, C.c_uint64, C.POINTER(u8p), C.POINTER(sz), C.POINTER(u8p), C.POINTER(sz), C.POINTER(Report), C.POINTER(C.c_char_p)]
    lib.dvt_debug_exec_rate.argtypes = [C.c_char_p, sz, C.POINTER(Buf), sz, u32, C.c_int]
    lib.dvt_debug_exec_rate.restype = C.c_double
    lib.dvt_prove_core.argtypes = [vp, vp, C.POINTER(Buf), sz, C.POINTER(u8p), C.POINTER(sz), C.POINTER(Report)]
    lib.dvt_verify.argtypes = [C.c_char_p, sz, C.c_char_p, sz, u32, u32, C.POINTER(C.c_int32), C.POINTER(u8p), C.POINTER(sz), C.POINTER(C.c_char_p)]
    lib.dvt_rv32_prepare.argtypes = [vp, vp, C.POINTER(Buf), sz, C.POINTER(vp), C.POINTER(Report)]
    lib.dvt_rv32_prepare_part.argtypes = [vp, vp, C.POINTER(Buf), sz, sz, sz, C.POINTER(vp), C.POINTER(Report)]
    lib.dvt_rv32_header_words.restype = u32
    lib.dvt_rv32_job_exec_wait_seconds.argtypes = [vp]
    lib.dvt_rv32_job_exec_wait_seconds.restype = C.c_double
    lib.dvt_rv32_prove_job.argtypes = [vp, vp, vp, C.POINTER(u8p), C.POINTER(sz)]
    lib.dvt_job_free.argtypes = [vp, vp]
    lib.dvt_job_free.restype = None
    lib.dvt_rv32_debug_traces.argtypes = [C.c_char_p, sz, C.POINTER(Buf), sz, u32, u32, u32p, C.POINTER(u32p), C.POINTER(sz), C.POINTER(C.c_char_p)]
    lib.dvt_rv32_job_shards.argtypes = [vp]
    lib.dvt_rv32_job_shards.restype = sz
    lib.dvt_rv32_commit_shard.argtypes = [vp, vp, vp, sz, u32p]
    lib.dvt_rv32_challenges.argtypes = [C.c_char_p, sz, u32p, sz, u32p]
    lib.dvt_rv32_prove_shard.argtypes = [vp, vp, vp, sz, u32p, C.POINTER(u8p), C.POINTER(sz)]
    lib.dvt_rv32_assemble.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(sz), sz, C.POINTER(u8p), C.POINTER(sz)]
    lib.dvt_rv32_debug_device_traces.argtypes = [vp, vp, vp, sz, C.POINTER(u32p), C.POINTER(sz)]
    lib.dvt_stdin_from_json.argtypes = [C.c_char_p, C.c_char_p, sz, C.c_int, C.POINTER(u8p), C.POINTER(sz), C.POINTER(C.c_char_p)]
    _lib = lib
    return lib


class Buf(C.Structure):
    _fields_ = [("data", C.c_char_p), ("len", C.c_size_t)]


class Report(C.Structure):
    _fields_ = [("cycles", C.c_uint64), ("exit_code", C.c_int32), ("halted", C.c_uint32), ("unprovable", C.c_uint32)]


def _bufs(stdin):
    arr = (Buf * max(len(stdin), 1))()
    for i, b in enumerate(stdin):
        arr[i] = Buf(bytes(b), len(b))
    return arr


def _take_str(lib, p):
    s = p.value.decode() if p.value else ""
    if p.value:
        lib.dvt_free(C.cast(p, C.c_void_p))
    return s


def execute(elf: bytes, stdin=(), max_cycles=0):
    """Host-only emulation (reference src/main.rs:430-447).  Returns (rc, report dict, public_values, error text)."""
    lib = load()
    pv, n, rep, err = u8p(), C.c_size_t(), Report(), C.c_char_p()
    rc = lib.dvt_execute(elf, len(elf), _bufs(stdin), len(stdin), max_cycles, C.byref(pv), C.byref(n), C.byref(rep), C.byref(err))
    out = C.string_at(pv, n.value) if pv else b""
    if pv:
        lib.dvt_free(C.cast(pv, C.c_void_p))
    return rc, dict(cycles=rep.cycles, exit_code=rep.exit_code, halted=bool(rep.halted), unprovable=bool(rep.unprovable)), out, _take_str(lib, err)


def exec_rate(elf: bytes, stdin=(), log_shard=21, trace=False) -> float:
    """guest cycles / second of the host executor alone (fast or trace mode)"""
    return float(load().dvt_debug_exec_rate(elf, len(elf), _bufs(stdin), len(stdin), log_shard, int(trace)))


def execute_io(elf: bytes, stdin=(), max_cycles=0):
    """execute() that also returns what the guest wrote to fds other than 3: (rc, report, public_values, stdout, error text)."""
    lib = load()
    pv, n, so, m, rep, err = u8p(), C.c_size_t(), u8p(), C.c_size_t(), Report(), C.c_char_p()
    rc = lib.dvt_execute_io(elf, len(elf), _bufs(stdin), len(stdin), max_cycles, C.byref(pv), C.byref(n), C.byref(so), C.byref(m), C.byref(rep), C.byref(err))
    out = C.string_at(pv, n.value) if pv else b""
    sout = C.string_at(so, m.value) if so else b""
    for ptr in (pv, so):
        if ptr:
            lib.dvt_free(C.cast(ptr, C.c_void_p))
    return rc, dict(cycles=rep.cycles, exit_code=rep.exit_code, halted=bool(rep.halted), unprovable=bool(rep.unprovable)), out, sout, _take_str(lib, err)


def stdin_from_json(circuit_type: str, json_bytes: bytes, auth_commitment=False) -> bytes:
    """The one SP1Stdin buffer the reference's host writes for `circuit_type` (src/main.rs:448-459:
    typed serde_json parse -> serde_cbor -> stdin.write(&Vec<u8>)).  Raises DvtError(DVT_ERR_INPUT) where the
    reference's typed parse fails ("Failed to read input")."""
    lib = load()
    out, n, err = u8p(), C.c_size_t(), C.c_char_p()
    rc = lib.dvt_stdin_from_json(circuit_type.encode(), json_bytes, len(json_bytes), int(bool(auth_commitment)), C.byref(out), C.byref(n), C.byref(err))
    if rc:
        raise DvtError(rc, _take_str(lib, err))
    b = C.string_at(out, n.value)
    lib.dvt_free(C.cast(out, C.c_void_p))
    return b


def verify(vk: bytes, proof: bytes, fri_queries=100, pow_bits=16):
    """Host-only verification of a core proof.  Returns (ok, exit_code, public_values, reason)."""
    lib = load()
    ec, pv, n, why = C.c_int32(), u8p(), C.c_size_t(), C.c_char_p()
    rc = lib.dvt_verify(vk, len(vk), proof, len(proof), fri_queries, pow_bits, C.byref(ec), C.byref(pv), C.byref(n), C.byref(why))
    out = C.string_at(pv, n.value) if pv else b""
    if pv:
        lib.dvt_free(C.cast(pv, C.c_void_p))
    return rc == DVT_OK, ec.value, out, _take_str(lib, why)


def _parse_blob(w):
    nch = int(w[0])
    meta = w[1:1 + 4 * nch].reshape(nch, 4)
    at = 1 + 4 * nch
    npub = int(w[at])
    pubs = w[at + 1:at + 1 + npub].copy()
    at += 1 + npub
    chips = []
    for cid, lg, mw, pw in meta:
        h = 1 << int(lg)
        main = w[at:at + int(mw) * h].reshape(int(mw), h)
        at += int(mw) * h
        prep = w[at:at + int(pw) * h].reshape(int(pw), h)
        at += int(pw) * h
        chips.append(dict(chip_id=int(cid), log_n=int(lg), main=main, prep=prep))
    assert at == len(w)
    return chips, pubs


def rv32_debug_traces(elf: bytes, stdin=(), log_shard=0, shard=0):
    """Host-only: the traces the prover would commit for shard `shard` (0-based) of the run cut at
    2^log_shard cycles.  Returns (chips, pubs, n_shards); chips = list of dict(chip_id, log_n, main, prep)."""
    lib = load()
    blob, n, err, ns = u32p(), C.c_size_t(), C.c_char_p(), C.c_uint32()
    rc = lib.dvt_rv32_debug_traces(elf, len(elf), _bufs(stdin), len(stdin), log_shard, shard, C.byref(ns), C.byref(blob), C.byref(n), C.byref(err))
    if rc:
        raise DvtError(rc, _take_str(lib, err))
    w = np.ctypeslib.as_array(blob, shape=(n.value,)).copy()
    lib.dvt_free(C.cast(blob, C.c_void_p))
    chips, pubs = _parse_blob(w)
    return chips, pubs, ns.value


HEADER_WORDS = 13   # main-trace Merkle root (8) + public values (5): dvt_rv32_header_words()


def rv32_challenges(vk: bytes, headers):
    """Host-only: the LogUp challenges common to all shards, from their 13-word headers (in shard order)."""
    lib = load()
    h = np.ascontiguousarray(headers, dtype=np.uint32).reshape(-1, HEADER_WORDS)
    out = np.zeros(8, np.uint32)
    rc = lib.dvt_rv32_challenges(vk, len(vk), h.ctypes.data_as(u32p), h.shape[0], out.ctypes.data_as(u32p))
    if rc:
        raise DvtError(rc, "dvt_rv32_challenges")
    return out


def _traces(traces):
    """traces: list of (chip_id, ndarray [width][height] uint32 canonical)."""
    arr = (HostTrace * max(len(traces), 1))()
    keep = []
    for i, (cid, m) in enumerate(traces):
        m = np.ascontiguousarray(m, dtype=np.uint32)
        h = m.shape[1]
        lg = int(h).bit_length() - 1
        assert 1 << lg == h
        keep.append(m)
        arr[i] = HostTrace(cid, lg, m.ctypes.data_as(u32p))
    return arr, keep


def machine_verify(vk: bytes, proof: bytes, fri_queries=100, pow_bits=16):
    """Host-only verification.  Returns (ok, reason)."""
    lib = load()
    reason = C.c_char_p()
    rc = lib.dvt_machine_verify(vk, len(vk), proof, len(proof), fri_queries, pow_bits, C.byref(reason))
    why = reason.value.decode() if reason.value else ""
    if reason.value:
        lib.dvt_free(C.cast(reason, C.c_void_p))
    return rc == DVT_OK, why


class Prover:
    """Owns one dvt_prover handle (one GPU)."""

    def __init__(self, cfg: str = None):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.dvt_prover_create(cfg.encode() if cfg else None, C.byref(h))
        if rc:
            raise DvtError(rc, self.lib.dvt_last_error(None).decode())
        self.h = h

    def close(self):
        if self.h:
            self.lib.dvt_prover_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc:
            raise DvtError(rc, self.lib.dvt_last_error(self.h).decode())

    def sync(self):
        self.check(self.lib.dvt_sync(self.h))

    def stream_ptr(self):
        return self.lib.dvt_stream(self.h)

    # ---- stage-level helpers over torch int32 CUDA tensors (device memory plumbing only)
    def to_internal(self, t):
        self.check(self.lib.dvt_dev_to_internal(self.h, t.data_ptr(), t.numel()))

    def from_internal(self, t):
        self.check(self.lib.dvt_dev_from_internal(self.h, t.data_ptr(), t.numel()))

    def coset_lde(self, t_in, t_out, width, log_n, shift_mode=0, scratch=None):
        assert t_in.numel() == width << log_n and t_out.numel() == width << (log_n + 1)
        assert scratch is None or scratch.numel() >= width << log_n
        self.check(self.lib.dvt_stage_coset_lde(self.h, t_in.data_ptr(), scratch.data_ptr() if scratch is not None else None,
                                                t_out.data_ptr(), width, log_n, shift_mode))

    def merkle_commit(self, mats, t_digests):
        """mats: list of (tensor [width][height], width, log_height)."""
        arr = (DevMatrix * len(mats))()
        for i, (t, w, lh) in enumerate(mats):
            assert t.numel() == w << lh
            arr[i] = DevMatrix(t.data_ptr(), w, lh)
        need = self.lib.dvt_merkle_digest_words(arr, len(mats))
        assert t_digests.numel() >= need
        self.check(self.lib.dvt_stage_merkle_commit(self.h, arr, len(mats), t_digests.data_ptr()))
        return need

    def poseidon2_permute(self, t_states):
        assert t_states.numel() % 16 == 0
        self.check(self.lib.dvt_stage_poseidon2_permute(self.h, t_states.data_ptr(), t_states.numel() // 16))

    def fri_fold(self, t_v, t_out, beta, log_m, t_ro=None):
        assert t_v.numel() == 4 << log_m and t_out.numel() == 2 << log_m
        b = (C.c_uint32 * 4)(*[int(x) for x in beta])
        self.check(self.lib.dvt_stage_fri_fold(self.h, t_v.data_ptr(), t_out.data_ptr(), t_ro.data_ptr() if t_ro is not None else None, b, log_m))

    # ---- machine level
    def machine_setup(self, machine: str, prep):
        arr, keep = _traces(prep)
        pk, vk, n = C.c_void_p(), u8p(), C.c_size_t()
        self.check(self.lib.dvt_machine_setup(self.h, machine.encode(), arr, len(prep), C.byref(pk), C.byref(vk), C.byref(n)))
        vkb = bytes(bytearray(vk[: n.value]))
        self.lib.dvt_free(C.cast(vk, C.c_void_p))
        return pk, vkb

    def pk_free(self, pk):
        self.lib.dvt_pk_free(self.h, pk)

    def machine_prove(self, pk, main, pubs):
        arr, keep = _traces(main)
        pv = np.ascontiguousarray(pubs, dtype=np.uint32)
        out, n = u8p(), C.c_size_t()
        self.check(self.lib.dvt_machine_prove(self.h, pk, arr, len(main), pv.ctypes.data_as(u32p), pv.size, C.byref(out), C.byref(n)))
        b = C.string_at(out, n.value)
        self.lib.dvt_free(C.cast(out, C.c_void_p))
        return b

    # ---- the reference's boundary (src/main.rs:461-474)
    def setup(self, elf: bytes):
        pk, vk, n = C.c_void_p(), u8p(), C.c_size_t()
        self.check(self.lib.dvt_setup(self.h, elf, len(elf), C.byref(pk), C.byref(vk), C.byref(n)))
        vkb = C.string_at(vk, n.value)
        self.lib.dvt_free(C.cast(vk, C.c_void_p))
        return pk, vkb

    def prove_core(self, pk, stdin=()):
        out, n, rep = u8p(), C.c_size_t(), Report()
        self.check(self.lib.dvt_prove_core(self.h, pk, _bufs(stdin), len(stdin), C.byref(out), C.byref(n), C.byref(rep)))
        b = C.string_at(out, n.value)
        self.lib.dvt_free(C.cast(out, C.c_void_p))
        return b, dict(cycles=rep.cycles, exit_code=rep.exit_code, halted=bool(rep.halted))

    def prepare(self, pk, stdin=(), first=0, stride=1):
        """the executor pipeline (fast pass, traced re-execution of the owned shards first, first + stride, ...,
        upload, phase 1 on the GPU); returns (job handle, report)"""
        job, rep = C.c_void_p(), Report()
        self.check(self.lib.dvt_rv32_prepare_part(self.h, pk, _bufs(stdin), len(stdin), first, stride, C.byref(job), C.byref(rep)))
        return job, dict(cycles=rep.cycles, exit_code=rep.exit_code, halted=bool(rep.halted))

    def job_exec_wait(self, job):
        """seconds the GPU-side thread of that prepare waited for the host executor"""
        return float(self.lib.dvt_rv32_job_exec_wait_seconds(job))

    def prove_job(self, pk, job, want_bytes=True):
        """K0..K9 on a prepared, HBM-resident shard"""
        if not want_bytes:
            self.check(self.lib.dvt_rv32_prove_job(self.h, pk, job, None, None))
            return None
        out, n = u8p(), C.c_size_t()
        self.check(self.lib.dvt_rv32_prove_job(self.h, pk, job, C.byref(out), C.byref(n)))
        b = C.string_at(out, n.value)
        self.lib.dvt_free(C.cast(out, C.c_void_p))
        return b

    def debug_device_traces(self, pk, job, shard=0):
        """K0 on the device for one shard, traces downloaded (canonical): (chips, pubs)"""
        blob, n = u32p(), C.c_size_t()
        self.check(self.lib.dvt_rv32_debug_device_traces(self.h, pk, job, shard, C.byref(blob), C.byref(n)))
        w = np.ctypeslib.as_array(blob, shape=(n.value,)).copy()
        self.lib.dvt_free(C.cast(blob, C.c_void_p))
        return _parse_blob(w)

    # ---- shard-level API (multi-GPU: ranks own shards; the headers are the only thing exchanged)
    def job_shards(self, job):
        return int(self.lib.dvt_rv32_job_shards(job))

    def commit_shard(self, pk, job, shard):
        h = np.zeros(HEADER_WORDS, np.uint32)
        self.check(self.lib.dvt_rv32_commit_shard(self.h, pk, job, shard, h.ctypes.data_as(u32p)))
        return h

    def prove_shard(self, pk, job, shard, challenges, want_bytes=True):
        ch = np.ascontiguousarray(challenges, dtype=np.uint32)
        if not want_bytes:
            self.check(self.lib.dvt_rv32_prove_shard(self.h, pk, job, shard, ch.ctypes.data_as(u32p), None, None))
            return None
        out, n = u8p(), C.c_size_t()
        self.check(self.lib.dvt_rv32_prove_shard(self.h, pk, job, shard, ch.ctypes.data_as(u32p), C.byref(out), C.byref(n)))
        b = C.string_at(out, n.value)
        self.lib.dvt_free(C.cast(out, C.c_void_p))
        return b

    def assemble(self, job, shard_proofs):
        n = len(shard_proofs)
        arr = (C.c_char_p * n)(*shard_proofs)
        lens = (C.c_size_t * n)(*[len(x) for x in shard_proofs])
        out, m = u8p(), C.c_size_t()
        rc = self.lib.dvt_rv32_assemble(job, arr, lens, n, C.byref(out), C.byref(m))
        if rc:
            raise DvtError(rc, "dvt_rv32_assemble")
        b = C.string_at(out, m.value)
        self.lib.dvt_free(C.cast(out, C.c_void_p))
        return b

    def job_free(self, job):
        self.lib.dvt_job_free(self.h, job)

    def kernel_stats(self):
        """K1 / K2+K3 family totals of the last prove on a profile handle (HIP events on the prover stream)"""
        self.lib.dvt_last_kernel_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        out = (C.c_double * 9)()
        self.check(self.lib.dvt_last_kernel_stats(self.h, out))
        return dict(lde_ms=out[0], lde_alg_bytes=out[1], lde_calls=int(out[2]), merkle_ms=out[3], merkle_perms=out[4],
                    cells_main=out[5], cells_perm=out[6], cells_quotient=out[7], cells_prep=out[8])

    def stage_ms(self):
        out = (C.c_float * 6)()
        self.check(self.lib.dvt_last_stage_ms(self.h, out))
        return dict(zip(["commit_main", "permutation", "quotient", "openings", "fri", "total"], list(out)))
