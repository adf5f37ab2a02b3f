"""CPU tests: the product's executor and host trace expansion (csrc/rv32_exec.hip, csrc/rv32.h, reached through the
C-ABI debug hook) against the oracle's INDEPENDENT Python restatement of the guest machine and of K0
(oracle/rv32_model.py, written from the AIR description, not from the product's sources): same cycle count, exit code,
public values and stdout; every chip's trace matrix equal cell by cell, shard by shard; and the model's own traces
satisfy the generated constraint checker and balance the LogUp multiset."""
import hashlib
import json
import struct

import numpy as np
import pytest

from dvt_circuits_amd import capi
from oracle import rv32_model
from tests import _orc, guests


def cases():
    from tools import gen_dkg_input

    enc = capi.stdin_from_json("bad-encrypted-share", json.dumps(gen_dkg_input.bad_encrypted_share(3, 2)).encode())
    return {
        "arith": (guests.arith()[0], (), 21),
        "arith_sharded": (guests.arith()[0], (), 10),
        "subword": (guests.subword()[0], (), 21),
        "shifts": (guests.shifts()[0], (), 11),
        "muldiv": (guests.muldiv()[0], (), 12),
        "bignum": (guests.bignum(2, limbs=3)[0], (), 9),
        "hint": (guests.hint_sum(), [struct.pack("<7I", *range(3, 10))], 21),
        "commit_only": (guests.commit_only(b"public-values"[:12]), (), 21),
        "sha_extend": (guests.sha_extend()[0], (), 21),
        "sha_extend_sharded": (guests.sha_extend(3)[0], (), 10),
        "sha256_precompiled": (guests.sha256_precompiled(bytes(range(150)))[0], (), 21),
        "sha256_precompiled_sharded": (guests.sha256_precompiled(bytes(range(150)))[0], (), 10),
        "encshare_n3": (guests.dkg_like("encshare"), [enc], 15),
    }


@pytest.mark.parametrize("name", list(cases()))
def test_product_execution_and_traces_equal_the_model(name):
    elf, stdin, log_shard = cases()[name]
    run = rv32_model.Run(elf, stdin, log_shard)
    rc, rep, pv, out, err = capi.execute_io(elf, stdin)
    assert rc == 0 and run.error == "" and run.halted, (err, run.error)
    assert rep["cycles"] == run.cycles and rep["exit_code"] == run.exit_code == 0
    assert pv == run.public_values and out == run.stdout
    dg = hashlib.sha256(pv).digest()
    assert [run.committed.get(k) for k in range(8)] == list(struct.unpack("<8I", dg))
    n = len(run.shards)
    for pos in range(n):
        host, hpubs, hn = capi.rv32_debug_traces(elf, stdin, log_shard, pos)
        model, mpubs = rv32_model.traces(run, pos)
        assert hn == n and (hpubs == mpubs).all()
        assert [c["chip_id"] for c in host] == [c["chip_id"] for c in model]
        for h, m in zip(host, model):
            for part in ("main", "prep"):
                assert h[part].shape == m[part].shape, (pos, h["chip_id"], part, h[part].shape, m[part].shape)
                diff = np.argwhere(h[part] != m[part])
                detail = [(int(c), int(r), int(h[part][c, r]), int(m[part][c, r])) for c, r in diff[:8]]
                assert diff.size == 0, f"shard {pos} chip {h['chip_id']} {part}: {len(diff)} cells differ, first (col,row,product,model): {detail}"


def test_model_traces_satisfy_the_air():
    """the model is checked too: its matrices satisfy every generated constraint and balance the LogUp multiset with the
    verifier-side public-value tuples"""
    air = _orc.air("rv32")
    elf, stdin, log_shard = cases()["arith_sharded"]
    run = rv32_model.Run(elf, stdin, log_shard)
    groups = []
    for pos in range(len(run.shards)):
        chips, pubs = rv32_model.traces(run, pos)
        for ch in chips:
            bad, bc, br = air.check_constraints(ch["chip_id"], ch["main"], ch["prep"], pubs)
            assert bad == 0, (pos, ch["chip_id"], bc, br)
        groups.append((chips, pubs))
    dg = hashlib.sha256(run.public_values).digest()
    extra = [(5, [0x10, 0, 0, 0, k, 0, 0, 0] + list(dg[4 * k:4 * k + 4]) + [0, 0], -1, 1) for k in range(8)]
    assert air.logup_unbalanced(groups, extra=extra)[0] == 0


def test_model_and_product_agree_on_traps():
    for elf, frag in ((guests.traps(), "misaligned"), (guests.exit_with(3), "")):
        run = rv32_model.Run(elf)
        rc, rep, pv, err = capi.execute(elf)
        if frag:
            assert rc == capi.DVT_ERR_GUEST and frag in err and frag in run.error
        else:
            assert rep["exit_code"] == run.exit_code == 3


def test_hand_written_family_checks_agree_with_the_generated_checker():
    """ADD / LW / MUL / branch rows: the hand-written statements (oracle/rv32_handcheck.py, not generated from
    tools/airgen) accept the real trace, and every single-cell change of a cell a statement reads is rejected by BOTH the
    hand-written check and the generated constraint checker (or, for the generated side, by the LogUp balance)"""
    from oracle import rv32_handcheck
    from tools.airgen import rv32 as air_desc

    air = _orc.air("rv32")
    names = next(c for c in air_desc.build().chips if c.name == "cpu").main_names
    col = {n: i for i, n in enumerate(names)}
    elf, _ = guests.arith(commit=False)
    chips, pubs, _ = capi.rv32_debug_traces(elf)
    cpu = next(c for c in chips if air.chip(c["chip_id"]).name == b"cpu")
    main = cpu["main"]
    base = rv32_handcheck.check_all(names, main, int(pubs[3]))
    assert all(n > 0 and bad == 0 for n, bad in base.values()), base
    run = rv32_model.Run(elf)
    mchips, mpubs = rv32_model.traces(run, 0)
    mcpu = next(c for c in mchips if c["chip_id"] == cpu["chip_id"])
    assert all(bad == 0 for _, bad in rv32_handcheck.check_all(names, mcpu["main"], int(mpubs[3])).values())
    targets = {
        "add": ("is_add", [f"{x}[{i}]" for x in "abc" for i in range(4)] + [f"u[{i}]" for i in range(4)] + ["next_pc"]),
        "lw": ("is_lw", [f"a[{i}]" for i in range(4)] + [f"b[{i}]" for i in range(4)] + [f"u[{i}]" for i in range(24)] + ["next_pc"]),
        "mul": ("is_mul", [f"{x}[{i}]" for x in "abc" for i in range(4)] + [f"u[{i}]" for i in range(11)]),
        "branch": ("is_brge", [f"{x}[{i}]" for x in "bc" for i in range(4)] + [f"u[{i}]" for i in (0, 1, 2, 3, 10, 19, 20)] + ["next_pc", "aux"]),
    }
    wordof = lambda x, r: sum(int(main[col[f"{x}[{i}]"], r]) << (8 * i) for i in range(4))
    # rows whose operands make every listed cell matter (a product with a zero factor, or a comparison decided by the top
    # byte, stays TRUE under some single-cell changes): 0x12345678 * 0x9ABCDEF0, and the taken 5 >= 3 (operands differ in byte 0 only)
    want_ops = {"mul": (0x12345678, 0x9ABCDEF0), "branch": (5, 3)}
    for fam, (flag, cells) in targets.items():
        cand = [int(r) for r in np.nonzero(main[col[flag]] == 1)[0] if fam not in want_ops or (wordof("b", r), wordof("c", r)) == want_ops[fam]]
        row = cand[0]
        for cell in cells:
            m = main.copy()
            m[col[cell], row] = (int(m[col[cell], row]) + 1) % rv32_model.P
            hand_bad = rv32_handcheck.check_all(names, m, int(pubs[3]))[fam][1]
            gen_bad = air.check_constraints(cpu["chip_id"], m, cpu["prep"], pubs)[0] or \
                air.logup_unbalanced([dict(cpu, main=m) if ch is cpu else ch for ch in chips], pubs)[0]
            assert hand_bad > 0, f"{fam}: hand-written check misses a change of {cell}"
            assert gen_bad > 0, f"{fam}: generated AIR misses a change of {cell}"
