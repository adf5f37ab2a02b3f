"""CPU test: the C-ABI library loads and exports every symbol include/dvt_prover.h declares.
No compute call is made (no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dvt_prover.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dvt_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(prover_lib):
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(prover_lib, n), f"{n} declared in include/dvt_prover.h but not exported"
    assert prover_lib.dvt_abi_version() >= 2


def test_create_without_gpu_fails_loudly(prover_lib):
    import torch

    if torch.cuda.is_available():
        return
    h = ctypes.c_void_p()
    rc = prover_lib.dvt_prover_create(None, ctypes.byref(h))
    assert rc == 3 and not h.value
    assert b"no CPU fallback" in prover_lib.dvt_last_error(None)


def test_fp64_poseidon2_matches_integer_permutation_on_host():
    """the hashing kernels evaluate Poseidon2 with exact integers carried in doubles (csrc/poseidon2_f64.cuh);
    the same code on the host (IEEE doubles, fma) must reproduce the integer permutation word for word"""
    import ctypes as C

    from dvt_circuits_amd import capi

    lib = capi.load()
    lib.dvt_debug_p2_f64_selfcheck.restype = C.c_uint64
    lib.dvt_debug_p2_f64_selfcheck.argtypes = [C.c_uint32, C.c_uint32]
    assert lib.dvt_debug_p2_f64_selfcheck(20000, 1) == 0
