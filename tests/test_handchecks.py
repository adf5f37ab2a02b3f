"""CPU tests: every chip has at least one statement that does NOT come from tools/airgen.
oracle/rv32_handcheck.py states, over the integers and from the ISA / FIPS 180-4 / the curve equations, what the result
cells of a row must hold given its operand cells.  The statements must accept the real traces of guests that exercise
every family, and reject changes of the result cells they read — the same changes the generated checker (or the LogUp
multiset) rejects, so the two descriptions of the AIR agree on them."""
import numpy as np
import pytest

from dvt_circuits_amd import capi
from oracle import rv32_handcheck as hc
from tests import _orc, guests
from tools.airgen import rv32 as airdef

P = 2013265921

GUESTS = {"arith": lambda: guests.arith()[0], "subword": lambda: guests.subword()[0], "shifts": lambda: guests.shifts()[0],
          "muldiv": lambda: guests.muldiv()[0], "sha": lambda: guests.sha256_precompiled(bytes(range(70)))[0],
          "field": lambda: guests.field_ops()[0], "curve": lambda: guests.curve_ops()[0], "u256": lambda: guests.u256_ops()[0]}
# chip -> (guest, result cells whose change the statement must notice)
CASES = {
    "cpu": ("subword", ["a[0]", "a[3]", "u[9]", "u[12]", "next_pc"]),
    "shift": ("shifts", ["a[0]", "a[2]"]),
    "muldiv": ("muldiv", ["a[0]", "a[3]"]),
    "sha_extend": ("sha", ["nw[0]", "nw[3]", "w15[1]"]),
    "sha_compress": ("sha", ["ab[0]", "eb[31]", "mv[2]"]),
    "mem_init": ("arith", ["ab[0]", "ab[2]"]),
    "fp_op": ("field", ["r[0]", "r[47]"]),
    "fp2_op": ("field", ["r0[5]", "r1[40]"]),
    "bls_g1": ("curve", ["x3[0]", "y3[47]"]),
    "secp_k1": ("curve", ["x3[31]", "y3[0]"]),
    "u256_mul": ("u256", ["r[0]", "r[31]", "m_zero"]),
}


@pytest.fixture(scope="module")
def machine():
    return {c.name: c for c in airdef.build().chips}


@pytest.fixture(scope="module")
def traces():
    air = _orc.air("rv32")
    out = {}
    for g, mk in GUESTS.items():
        chips, pubs, _ = capi.rv32_debug_traces(mk())
        out[g] = ({air.chip(c["chip_id"]).name.decode(): c for c in chips}, pubs)
    return out


@pytest.mark.parametrize("chip", list(CASES))
def test_statement_accepts_real_rows_and_rejects_changed_results(machine, traces, chip):
    guest, cells = CASES[chip]
    chips, pubs = traces[guest]
    names = machine[chip].main_names
    main = chips[chip]["main"]
    rows, ok = hc.check_chip(chip, names, main, pubs)
    assert rows.sum() > 0 and bool((ok | ~rows).all()), (chip, int((rows & ~ok).sum()))
    rng = np.random.default_rng(1)
    real = np.nonzero(rows)[0]
    for cell in cells:
        c = names.index(cell)
        hit = 0
        for r in rng.choice(real, min(6 if chip != "cpu" else 60, len(real)), replace=False):
            m = main.copy()
            m[c, r] = (int(m[c, r]) + 1) % P
            rows2, ok2 = hc.check_chip(chip, names, m, pubs)
            hit += int(not bool((ok2 | ~rows2).all()))
        assert hit > 0, f"{chip}: no statement noticed a change of {cell}"


def test_every_family_of_the_cpu_chip_has_rows_under_a_statement(machine, traces):
    """the cpu chip's statements together cover ADD/SUB-less families: add, lw, mul, branches, sub-word accesses, ecall"""
    names = machine["cpu"].main_names
    col = {n: i for i, n in enumerate(names)}
    seen = set()
    for g in ("arith", "subword"):
        chips, pubs = traces[g]
        main = chips["cpu"]["main"]
        for fam, fn in (("add", lambda: hc.check_add(names, main)), ("lw", lambda: hc.check_lw(names, main, int(pubs[3]))),
                        ("mul", lambda: hc.check_mul(names, main)), ("branch", lambda: hc.check_branches(names, main)),
                        ("subword", lambda: hc.check_subword(names, main)), ("ecall", lambda: hc.check_ecall(names, main, pubs))):
            rows, ok = fn()
            assert bool((ok | ~rows).all()), (g, fam)
            if rows.sum():
                seen.add(fam)
    assert seen == {"add", "lw", "mul", "branch", "subword", "ecall"}
