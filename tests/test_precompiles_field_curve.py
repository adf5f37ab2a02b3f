"""CPU tests (no GPU) of the field / curve precompiles: SP1's BLS12381_FP_*, BLS12381_FP2_*,
BLS12381_ADD / _DOUBLE and SECP256K1_ADD / _DOUBLE syscalls, the calls the reference's current guests make through the
patched bls12_381 / secp256k1 crates (reference crates/dkg/Cargo.toml:24-25; scalar multiplications of
crates/dkg/src/dkg_math.rs:160-174, ECDSA of crates/dkg/src/crypto/secp256k1_keys.rs:51-64).
Executor semantics against plain-Python big-integer arithmetic, the product's rows against the generated checker, the
LogUp multiset and the independent Python model, the calls that must trap, and per-cell soundness of every chip."""
import os
import re

import numpy as np
import pytest

from dvt_circuits_amd import capi
from oracle import rv32_model
from tests import _orc, guests
from tests.test_rv32_exec_trace import check_traces, pv_extra

P = 2013265921
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def air():
    return _orc.air("rv32")


def _names(chip):
    names = {}
    for line in open(os.path.join(ROOT, "dvt_circuits_amd", "csrc", "gen", "rv32_cols.h")):
        m = re.match(r"#define RV32_%s_(\w+) (\d+)" % chip, line)
        if m and m.group(1) not in ("MAIN_W", "PREP_W") and not m.group(1).startswith("P_"):
            names[int(m.group(2))] = m.group(1)
    return names


def test_python_curve_helpers_reproduce_known_multiples():
    """the expected values of the guests come from ec_add / ec_double: pin those on 2G of both curves (public constants)"""
    x2, y2 = guests.ec_double(guests.SECP_G, guests.SECP_P)
    assert x2 == 0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5
    assert y2 == 0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A
    from tools import bls12_381 as bls

    g = bls.G1_GEN if hasattr(bls, "G1_GEN") else None
    for k, pt in ((2, guests.ec_double(guests.BLS_G1, guests.BLS_P)),):
        assert (pt[1] * pt[1] - pt[0] ** 3 - 4) % guests.BLS_P == 0
    g3 = guests.ec_add(guests.ec_double(guests.BLS_G1, guests.BLS_P), guests.BLS_G1, guests.BLS_P)
    assert (g3[1] * g3[1] - g3[0] ** 3 - 4) % guests.BLS_P == 0 and (guests.BLS_G1[1] ** 2 - guests.BLS_G1[0] ** 3 - 4) % guests.BLS_P == 0
    assert (guests.SECP_G[1] ** 2 - guests.SECP_G[0] ** 3 - 7) % guests.SECP_P == 0


GUESTS = {"field": guests.field_ops, "curve": guests.curve_ops, "u256": guests.u256_ops}


@pytest.mark.parametrize("which,log_shard", [("field", 0), ("field", 9), ("curve", 0), ("curve", 8), ("u256", 0), ("u256", 7)])
def test_guest_runs_and_every_shard_satisfies_the_air(air, which, log_shard):
    elf, want = GUESTS[which]()
    rc, rep, pv, out, err = capi.execute_io(elf)
    assert rc == 0 and rep["halted"] and not rep["unprovable"], err
    assert out == want and pv == guests.checksum(want)
    chips, pubs = check_traces(air, elf, log_shard=log_shard)
    if log_shard == 0:
        present = {air.chip(c["chip_id"]).name.decode() for c in chips}
        assert {"field": {"fp_op", "fp2_op"}, "curve": {"bls_g1", "secp_k1"}, "u256": {"u256_mul"}}[which] <= present


@pytest.mark.parametrize("which,log_shard", [("field", 21), ("field", 9), ("curve", 21), ("curve", 8), ("u256", 21), ("u256", 7)])
def test_product_rows_equal_the_independent_model(which, log_shard):
    elf, want = GUESTS[which]()
    run = rv32_model.Run(elf, (), log_shard)
    rc, rep, pv, out, err = capi.execute_io(elf)
    assert rc == 0 and run.error == "" and run.halted, (err, run.error)
    assert rep["cycles"] == run.cycles and pv == run.public_values and out == run.stdout == want
    for pos in range(len(run.shards)):
        host, hpubs, hn = capi.rv32_debug_traces(elf, (), log_shard, pos)
        model, mpubs = rv32_model.traces(run, pos)
        assert hn == len(run.shards) and (hpubs == mpubs).all()
        assert [c["chip_id"] for c in host] == [c["chip_id"] for c in model]
        for h, m in zip(host, model):
            for part in ("main", "prep"):
                assert h[part].shape == m[part].shape, (pos, h["chip_id"], part)
                diff = np.argwhere(h[part] != m[part])
                detail = [(int(c), int(r), int(h[part][c, r]), int(m[part][c, r])) for c, r in diff[:8]]
                assert diff.size == 0, f"shard {pos} chip {h['chip_id']} {part}: {len(diff)} cells differ, first (col,row,product,model): {detail}"


def test_invalid_calls_trap():
    for bad, frag in (("misaligned", "misaligned"), ("low", "out of range"), ("high", "out of range")):
        rc, rep, _, err = capi.execute(guests.field_ops(bad=bad)[0])
        assert rc == capi.DVT_ERR_GUEST and frag in err, (bad, err)
        assert frag in rv32_model.Run(guests.field_ops(bad=bad)[0]).error
    rc, rep, _, err = capi.execute(guests.u256_ops(bad="misaligned")[0])
    assert rc == capi.DVT_ERR_GUEST and "misaligned" in err
    for bad, frag in (("equal", "equal abscissae"), ("unreduced", "not reduced"), ("a1", "a1 != 0")):
        rc, rep, _, err = capi.execute(guests.curve_ops(bad=bad)[0])
        assert rc == capi.DVT_ERR_GUEST and frag in err, (bad, err)
        assert frag in rv32_model.Run(guests.curve_ops(bad=bad)[0]).error
    # a precompile code nobody implements has no receiver on the sys bus: the executor traps, nothing can prove it
    from tools.rvasm import Asm

    a = Asm()
    a.li("a0", 0x300000); a.li("a1", 0); a.li("t0", 0x00010199); a.ecall(); a.halt(0)
    rc, rep, _, err = capi.execute(a.elf())
    assert rc == capi.DVT_ERR_GUEST and "unknown syscall" in err


def _free_cells(air, chips, pubs, extra, chip, names, row):
    main, free = chip["main"], set()
    for c in range(main.shape[0]):
        m = main.copy()
        m[c, row] = (int(m[c, row]) + 1) % P
        if air.check_constraints(chip["chip_id"], m, chip["prep"], pubs)[0]:
            continue
        if air.logup_unbalanced([dict(chip, main=m) if ch is chip else ch for ch in chips], pubs, extra=extra)[0]:
            continue
        free.add(names[c])
    return free


@pytest.mark.parametrize("chip_name,guest", [("fp_op", "field"), ("fp2_op", "field"), ("bls_g1", "curve"), ("secp_k1", "curve"), ("u256_mul", "u256")])
def test_every_cell_of_a_row_is_pinned(air, chip_name, guest):
    """single-cell changes on one row per operation of the chip: every one breaks a constraint or the LogUp balance,
    except cells the operation does not read (the second operand and its access columns on DOUBLE rows, the inequality
    witnesses of the groups that are not used)"""
    elf, want = GUESTS[guest]()
    chips, pubs, _ = capi.rv32_debug_traces(elf)
    extra = pv_extra(guests.checksum(want))
    assert air.logup_unbalanced(chips, pubs, extra=extra)[0] == 0
    chip = next(c for c in chips if air.chip(c["chip_id"]).name == chip_name.encode())
    names = _names(chip_name.upper())
    col = {n: i for i, n in names.items()}
    main = chip["main"]
    if chip_name == "u256_mul":
        # an odd modulus, the modulus 0 (= 2^256) and an even one: every cell pinned but the inverse witnesses of m's groups
        # (any solution of sum m_g z_g = 1 will do) and, for m = 0, the comparison witnesses nothing reads
        for row in (0, 3, 4):
            free = _free_cells(air, chips, pubs, extra, chip, names, row)
            allowed = {n for n in names.values() if n.startswith("mz_")}
            if main[col["m_zero"], row] == 1:
                allowed |= {n for n in names.values() if n.startswith("rlt_")}
            assert free <= allowed, (row, sorted(free - allowed)[:20])
        return
    ops = ["is_add", "is_sub", "is_mul"] if chip_name.startswith("fp") else ["is_add", "is_dbl"]
    for op in ops:
        row = next(r for r in range(main.shape[1]) if main[col[op], r] == 1)
        free = _free_cells(air, chips, pubs, extra, chip, names, row)
        allowed = set()
        if op == "is_dbl":
            # DOUBLE reads no second point: its cells, their access columns and checks are idle (qp is pinned to 0)
            allowed |= {n for n in names.values() if re.match(r"^(x2|y2|mq_(sh|ts|same|lo|hi)|x2lt_[fd]|y2lt_[fd]|xne_z)_\d+$", n)}
        if op == "is_add" and not chip_name.startswith("fp"):
            # x1 != x2: any solution of sum (x1_g - x2_g) z_g = 1 will do, so a z cell is free where its group's difference is 0
            allowed |= {n for n in names.values() if n.startswith("xne_z_")}
        bad = free - allowed
        assert not bad, (chip_name, op, sorted(bad)[:20])
        if op == "is_add" and not chip_name.startswith("fp"):
            assert len({n for n in free if n.startswith("xne_z_")}) < sum(1 for n in names.values() if n.startswith("xne_z_")), "the used inverse must be pinned"


def test_add_with_equal_abscissae_has_no_witness(air):
    """P + P through ADD would leave the slope unconstrained (0 * lambda = 0): the inequality witness makes such a row
    unsatisfiable.  Forge: take an ADD row, overwrite the second point with the first, recompute nothing else: whatever the
    other cells hold, sum (x1_g - x2_g) z_g = 1 cannot hold."""
    elf, want = guests.curve_ops()
    chips, pubs, _ = capi.rv32_debug_traces(elf)
    chip = next(c for c in chips if air.chip(c["chip_id"]).name == b"bls_g1")
    col = {n: i for i, n in _names("BLS_G1").items()}
    main = chip["main"].copy()
    row = next(r for r in range(main.shape[1]) if main[col["is_add"], r] == 1)
    for i in range(48):
        main[col[f"x2_{i}"], row] = main[col[f"x1_{i}"], row]
    from tools.airgen import rv32 as airdef

    cdef = next(c for c in airdef.build().chips if c.name == "bls_g1")
    # the inequality constraint is the one that reads every xne_z column: find it by perturbing a z cell on the honest row
    honest = chip["main"]
    zcols = [col[n] for n in col if n.startswith("xne_z_")]
    used = next(c for c in zcols if honest[c, row] != 0)
    m2 = honest.copy()
    m2[used, row] = (int(m2[used, row]) + 1) % P
    bad_honest, which, _ = air.check_constraints(chip["chip_id"], m2, chip["prep"], pubs)
    assert bad_honest == 1
    # with x2 = x1 that constraint fails for EVERY choice of the z cells (it reads 0 * z = 1)
    rng = np.random.default_rng(5)
    for _ in range(4):
        m3 = main.copy()
        for c in zcols:
            m3[c, row] = int(rng.integers(0, P))
        n_bad, first, r = air.check_constraints(chip["chip_id"], m3, chip["prep"], pubs)
        assert n_bad >= 1
        out = np.zeros(cdef.main_width, np.uint32)
    m4 = main.copy()
    _, first, r = air.check_constraints(chip["chip_id"], m4, chip["prep"], pubs)
    assert r == row
