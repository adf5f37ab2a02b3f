"""ctypes binding of include/dvt_prover.h.  Loads the in-tree HIP library
`dvt_circuits_amd/libdvt_prover.so`; raises if it is missing — there is no
fallback path."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdvt_prover.so")
CSRC = os.path.join(HERE, "csrc")

DVT_OK, DVT_ERR_GUEST, DVT_ERR_INPUT, DVT_ERR_DEVICE, DVT_ERR_UNSUPPORTED = 0, 1, 2, 3, 4
u32p = C.POINTER(C.c_uint32)


class DevMatrix(C.Structure):
    _fields_ = [("d_data", C.c_void_p), ("width", C.c_uint32), ("log_height", C.c_uint32)]


class DvtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dvt error {code}: {msg}")
        self.code = code


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
    lib.dvt_sync.argtypes = [vp, vp]
    lib.dvt_dev_to_internal.argtypes = [vp, vp, vp, sz]
    lib.dvt_dev_from_internal.argtypes = [vp, vp, vp, sz]
    lib.dvt_stage_coset_lde.argtypes = [vp, vp, vp, vp, u32, u32, u32]
    lib.dvt_merkle_digest_words.argtypes = [C.POINTER(DevMatrix), sz]
    lib.dvt_merkle_digest_words.restype = sz
    lib.dvt_stage_merkle_commit.argtypes = [vp, vp, C.POINTER(DevMatrix), sz, vp]
    lib.dvt_stage_poseidon2_permute.argtypes = [vp, vp, vp, sz]
    _lib = lib
    return lib


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

    def sync(self, stream=None):
        self.check(self.lib.dvt_sync(self.h, stream))

    # ---- stage-level helpers over torch int32 CUDA tensors (device memory plumbing only)
    def to_internal(self, t, stream=None):
        self.check(self.lib.dvt_dev_to_internal(self.h, stream, t.data_ptr(), t.numel()))

    def from_internal(self, t, stream=None):
        self.check(self.lib.dvt_dev_from_internal(self.h, stream, t.data_ptr(), t.numel()))

    def coset_lde(self, t_in, t_out, width, log_n, shift_mode=0, stream=None):
        assert t_in.numel() == width << log_n and t_out.numel() == width << (log_n + 1)
        self.check(self.lib.dvt_stage_coset_lde(self.h, stream, t_in.data_ptr(), t_out.data_ptr(), width, log_n, shift_mode))

    def merkle_commit(self, mats, t_digests, stream=None):
        """mats: list of (tensor [width][height], width, log_height)."""
        arr = (DevMatrix * len(mats))()
        for i, (t, w, lh) in enumerate(mats):
            assert t.numel() == w << lh
            arr[i] = DevMatrix(t.data_ptr(), w, lh)
        need = self.lib.dvt_merkle_digest_words(arr, len(mats))
        assert t_digests.numel() >= need
        self.check(self.lib.dvt_stage_merkle_commit(self.h, stream, arr, len(mats), t_digests.data_ptr()))
        return need

    def poseidon2_permute(self, t_states, stream=None):
        assert t_states.numel() % 16 == 0
        self.check(self.lib.dvt_stage_poseidon2_permute(self.h, stream, t_states.data_ptr(), t_states.numel() // 16))
