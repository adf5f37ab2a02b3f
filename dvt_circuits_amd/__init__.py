"""MI355X-native STARK prover behind metacraft-labs/dvt-circuits' prove() boundary.

The product is the C-ABI shared library `libdvt_prover.so` (hand-written HIP for
gfx950, sources under csrc/, header include/dvt_prover.h).  This package is the
thin host-side mirror used by the tests, bench.py and the CLI: ctypes bindings
(capi) and a ProverClient-shaped wrapper (client).  No CPU fallback exists: every
compute call needs the HIP library and a gfx950 device.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
