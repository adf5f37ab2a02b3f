"""GPU parity test of K0 (trace expansion kernel): the traces generated on the device must equal, word for word,
(a) the oracle's independent Python restatement of the guest machine and of the row expansion (oracle/rv32_model.py,
written from the AIR description, not from the product's sources) and (b) the product's own host-side expansion."""
import struct

import numpy as np
import pytest

from tests import guests

pytestmark = pytest.mark.gpu


# (field_ops / curve_ops / u256_ops: the rows of the precompile chips are built by k0_bigop_rows_kernel, one thread per call,
# with the byte-table lookups of those rows added to the shard's counts on the device)
@pytest.mark.parametrize("which,log_shard", [("arith", 21), ("bignum", 21), ("hint", 21), ("bignum", 8), ("subword", 21), ("shifts", 8), ("muldiv", 9),
                                             ("field_ops", 21), ("field_ops", 9), ("curve_ops", 21), ("curve_ops", 8), ("u256_ops", 21), ("u256_ops", 7)])
def test_k0_device_traces_equal_host_traces(which, log_shard):
    from dvt_circuits_amd import capi

    stdin = []
    if which == "arith":
        elf = guests.arith()[0]
    elif which in ("shifts", "muldiv", "field_ops", "curve_ops", "u256_ops"):
        elf = getattr(guests, which)()[0]
    elif which == "subword":
        elf = guests.subword()[0]
    elif which == "bignum":
        elf = guests.bignum(5, limbs=6)[0]
    else:
        elf, stdin = guests.hint_sum(), [struct.pack("<5I", 9, 8, 7, 6, 5)]
    from oracle import rv32_model

    p = capi.Prover('{"fri_queries": 8, "pow_bits": 4, "log_shard_size": %d}' % log_shard)
    pk, _ = p.setup(elf)
    job, rep = p.prepare(pk, stdin)
    n = p.job_shards(job)
    run = rv32_model.Run(elf, stdin, log_shard)
    assert n == len(run.shards) and rep["cycles"] == run.cycles
    assert n == capi.rv32_debug_traces(elf, stdin, log_shard, 0)[2]
    for shard in range(n):
        host, hpubs, _ = capi.rv32_debug_traces(elf, stdin, log_shard, shard)
        dev, dpubs = p.debug_device_traces(pk, job, shard)
        model, mpubs = rv32_model.traces(run, shard)
        assert (dpubs == mpubs).all() and len(model) == len(dev)
        for m, d in zip(model, dev):
            assert m["chip_id"] == d["chip_id"] and m["log_n"] == d["log_n"]
            diff = np.argwhere(m["main"] != d["main"])
            detail = [(int(c), int(r), int(m["main"][c, r]), int(d["main"][c, r])) for c, r in diff[:12]]
            assert diff.size == 0, f"shard {shard} chip {m['chip_id']}: {len(diff)} cells differ from the oracle model, first (col,row,model,dev): {detail}"
        assert (dpubs == hpubs).all() and len(host) == len(dev)
        for h, d in zip(host, dev):
            assert h["chip_id"] == d["chip_id"] and h["log_n"] == d["log_n"]
            diff = np.argwhere(h["main"] != d["main"])
            detail = [(int(c), int(r), int(h["main"][c, r]), int(d["main"][c, r])) for c, r in diff[:12]]
            assert diff.size == 0, f"shard {shard} chip {h['chip_id']}: {len(diff)} mismatches, first (col,row,host,dev): {detail}"
    p.job_free(job)
    p.pk_free(pk)
    p.close()
