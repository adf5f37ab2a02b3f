#!/usr/bin/env python3
"""Per-stage milliseconds of one full 2^21-cycle shard of the bench guest (profile handle: HIP events on the prover stream)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from dvt_circuits_amd import capi  # noqa: E402
from tests import guests  # noqa: E402

buf = bench.workload_stdin()
elf = guests.dkg_like("finalization", *bench.fit_constants(buf, 1))
prof = capi.Prover('{"fri_queries": 100, "pow_bits": 16, "profile": 1}')
pk, _ = prof.setup(elf)
job, rep = prof.prepare(pk, [buf])
for _ in range(3):
    prof.prove_job(pk, job, want_bytes=False)
print(json.dumps(dict(cycles=rep["cycles"], stage_ms=prof.stage_ms(), kernels=prof.kernel_stats())))
