"""CPU soundness net: TWO-cell forgeries.  Both attacks the round-1 advisor demonstrated changed a
PAIR of cells (two cancelling flags on a padding row; an address and a gap that wrap around p together), a class the
single-cell mutation tests cannot see.  For every chip, on real traces, random pairs of cells — in one row and in two
adjacent rows — are changed by the deltas such attacks use (+-1, +-256, a flag pair +1 / -1, the two bytes that add p to
a little-endian word) and the result must be rejected by the generated constraint checker or by the exact LogUp
multiset, unless BOTH cells are cells that already escape on their own (cells the row's instruction family does not
read) or the pair is a documented re-encoding of one constrained EXPRESSION.

The LogUp side is evaluated with a small interpreter over the AIR description (tools/airgen) on the touched rows only:
a forgery changes the multiset exactly when the tuples its rows send / receive change."""
import collections
import os

import numpy as np
import pytest

from dvt_circuits_amd import capi
from tests import _orc, guests
from tools.airgen import emit, rv32 as airdef

P = 2013265921
TRIALS = int(os.environ.get("DVT_PAIR_TRIALS", "500"))


@pytest.fixture(scope="module")
def air():
    return _orc.air("rv32")


@pytest.fixture(scope="module")
def machine():
    return airdef.build()


class RowEval:
    """evaluates a chip's interaction tuples on single rows (local + next) from the AIR description"""

    def __init__(self, chip):
        self.chip = chip
        roots = []
        for it in chip.interactions:
            roots.append(it.mult)
            roots += it.vals
        self.order = emit.topo(roots)

    def tuples(self, main, prep, pubs, r):
        n = main.shape[1]
        val = {}
        for e in self.order:
            if e.op == "const":
                v = e.args[0]
            elif e.op == "var":
                kind, idx, rot = e.args
                v = int(pubs[idx]) if kind == "pub" else int((main if kind == "main" else prep)[idx, (r + rot) % n])
            elif e.op == "neg":
                v = -val[e.args[0].id]
            else:
                x, y = val[e.args[0].id], val[e.args[1].id]
                v = x + y if e.op == "add" else x - y if e.op == "sub" else x * y
            val[e.id] = v % P
        out = collections.Counter()
        for it in self.chip.interactions:
            m = val[it.mult.id]
            if m == 0:
                continue
            m = m - P if m > P // 2 else m
            out[(it.bus, tuple(val[v.id] for v in it.vals))] += it.sign * m
        return out


def _rows_multiset(ev, main, prep, pubs, rows):
    tot = collections.Counter()
    for r in rows:
        tot.update(ev.tuples(main, prep, pubs, r))
    return {k: v for k, v in tot.items() if v}


def _caught(air, ev, chip, pubs, honest, forged, rows):
    if air.check_constraints(chip["chip_id"], forged, chip["prep"], pubs)[0]:
        return True
    n = honest.shape[1]
    touched = sorted({(r + d) % n for r in rows for d in (-1, 0)})       # a row's tuples may read the next row
    return _rows_multiset(ev, forged, chip["prep"], pubs, touched) != _rows_multiset(ev, honest, chip["prep"], pubs, touched)


def _word_groups(names):
    """column indices of the little-endian 4-byte words of a chip: name[0..3]"""
    groups = collections.defaultdict(dict)
    for i, n in enumerate(names):
        if n.endswith("]") and "[" in n:
            base, k = n[:-1].split("[")
            groups[base][int(k)] = i
    return [[g[k] for k in range(4)] for g in groups.values() if sorted(g) == [0, 1, 2, 3]]


def _hunt(air, machine, chips, pubs, chip_name, rng, allowed_pairs=(), extra_rows=(), pair_cols=None):
    cdef = next(c for c in machine.chips if c.name == chip_name)
    chip = next(c for c in chips if air.chip(c["chip_id"]).name == chip_name.encode())
    ev = RowEval(cdef)
    honest = chip["main"]
    W, n = honest.shape
    names = cdef.main_names
    flag_cols = [i for i, nm in enumerate(names) if nm.startswith("is_") or nm in ("rd_en", "imm_c", "sys_m")]
    words = _word_groups(names)
    real_col = names.index("is_real") if "is_real" in names else None
    n_real = int(honest[real_col].sum()) if real_col is not None else int(sum(honest[c] for c in flag_cols if names[c].startswith("is_")).sum())
    n_real = max(1, min(n_real, n))
    # rows to attack: a spread of real rows, the last real row, the first padding row (where the flag-pair attack lived)
    rows = sorted(set(int(x) for x in rng.integers(0, n_real, 12)) | {0, n_real - 1, min(n_real, n - 1)} | set(extra_rows))
    # cells that escape on their own on each of those rows (unread cells of the row's family)
    free1 = {}
    for r in rows:
        free1[r] = set()
        for c in range(W):
            m = honest.copy()
            m[c, r] = (int(m[c, r]) + 1) % P
            if not _caught(air, ev, chip, pubs, honest, m, [r]):
                free1[r].add(c)
    escapes = []
    deltas = [1, P - 1, 256, P - 256, 2, 0x78]
    for t in range(TRIALS):
        r1 = rows[int(rng.integers(0, len(rows)))]
        r2 = r1 if t % 3 else (r1 + 1) % n
        kind = t % 5
        if kind == 0 and len(flag_cols) >= 2:                     # two flags, +1 / -1
            c1, c2 = (int(x) for x in rng.choice(flag_cols, 2, replace=False))
            d1, d2 = 1, P - 1
        elif kind == 1 and words:                                 # byte 0 + 1 and byte 3 + 0x78: the word grows by p
            g = words[int(rng.integers(0, len(words)))]
            c1, c2, d1, d2 = g[0], g[3], 1, 0x78
            r2 = r1
        elif pair_cols is not None:                               # (self-test of the hunter: pairs inside a given column set)
            c1, c2 = (int(x) for x in rng.choice(pair_cols, 2, replace=False))
            r1 = r2 = extra_rows[0]
            d1, d2 = deltas[int(rng.integers(0, len(deltas)))], deltas[int(rng.integers(0, len(deltas)))]
        else:
            c1, c2 = int(rng.integers(0, W)), int(rng.integers(0, W))
            d1, d2 = deltas[int(rng.integers(0, len(deltas)))], deltas[int(rng.integers(0, len(deltas)))]
        if (c1, r1) == (c2, r2):
            continue
        m = honest.copy()
        m[c1, r1] = (int(m[c1, r1]) + d1) % P
        m[c2, r2] = (int(m[c2, r2]) + d2) % P
        if _caught(air, ev, chip, pubs, honest, m, [r1, r2]):
            continue
        f1 = c1 in free1.get(r1, ()) or r1 not in free1
        f2 = c2 in free1.get(r2, ()) or r2 not in free1
        if r2 not in free1:                                       # the neighbour row was not classified: classify the one cell
            m1 = honest.copy()
            m1[c2, r2] = (int(m1[c2, r2]) + d2) % P
            f2 = not _caught(air, ev, chip, pubs, honest, m1, [r2])
        if f1 and f2:
            continue                                              # two cells nothing reads on those rows
        if (names[c1], names[c2]) in allowed_pairs or (names[c2], names[c1]) in allowed_pairs:
            continue
        escapes.append((names[c1], r1, d1, names[c2], r2, d2, f1, f2))
    return escapes


def _traces(elf, stdin=()):
    chips, pubs, n = capi.rv32_debug_traces(elf, stdin)
    assert n == 1
    return chips, pubs


def test_cpu_chip_pairs(air, machine):
    chips, pubs = _traces(guests.arith(commit=True)[0])
    # on COMMIT / precompile rows ("sys rows") the port's address is pinned as an EXPRESSION of u[0..3], u[21..23]: the
    # cells can trade value among themselves (tests/test_precompiles_sha.py documents the same for single cells)
    expr_cells = ["u[0]", "u[1]", "u[2]", "u[3]", "u[21]", "u[22]", "u[23]"]
    allowed = {(a, b) for a in expr_cells for b in expr_cells}
    cpu = next(c for c in chips if air.chip(c["chip_id"]).name == b"cpu")
    names = next(c for c in machine.chips if c.name == "cpu").main_names
    sys_row = int(np.nonzero(cpu["main"][names.index("sys_m")])[0][0])
    esc = _hunt(air, machine, chips, pubs, "cpu", np.random.default_rng(11), allowed, extra_rows=[sys_row])
    assert not esc, esc[:10]


def test_the_hunter_finds_a_known_two_cell_freedom(air, machine):
    """not vacuous: on a COMMIT row u[0] and u[22] enter ONE pinned expression (u0 + ... - (u21 + 2 u22 + 3 u23)), so
    (u[0] + 2, u[22] + 1) is a two-cell change that nothing rejects although each cell alone is pinned — the hunter must
    report it when the pair is not whitelisted"""
    chips, pubs = _traces(guests.arith(commit=True)[0])
    cpu = next(c for c in chips if air.chip(c["chip_id"]).name == b"cpu")
    names = next(c for c in machine.chips if c.name == "cpu").main_names
    sys_row = int(np.nonzero(cpu["main"][names.index("sys_m")])[0][0])
    cols = [names.index(n) for n in ("u[0]", "u[22]", "u[23]")]
    esc = _hunt(air, machine, chips, pubs, "cpu", np.random.default_rng(3), (), extra_rows=[sys_row], pair_cols=cols)
    assert any({e[0], e[3]} <= {"u[0]", "u[22]", "u[23]"} for e in esc), "the hunter missed the documented expression-level freedom"


@pytest.mark.parametrize("chip_name,guest", [("shift", "shifts"), ("muldiv", "muldiv"), ("mem_init", "subword")])
def test_small_chip_pairs(air, machine, chip_name, guest):
    chips, pubs = _traces(getattr(guests, guest)()[0])
    esc = _hunt(air, machine, chips, pubs, chip_name, np.random.default_rng(12))
    assert not esc, esc[:10]


@pytest.mark.parametrize("chip_name", ["sha_extend", "sha_compress"])
def test_sha_chip_pairs(air, machine, chip_name):
    chips, pubs = _traces(guests.sha256_precompiled(bytes(range(70)))[0])
    esc = _hunt(air, machine, chips, pubs, chip_name, np.random.default_rng(13))
    assert not esc, esc[:10]


@pytest.mark.parametrize("chip_name,guest", [("fp_op", "field_ops"), ("fp2_op", "field_ops"), ("bls_g1", "curve_ops"), ("secp_k1", "curve_ops"),
                                             ("u256_mul", "u256_ops")])
def test_precompile_chip_pairs(air, machine, chip_name, guest):
    chips, pubs = _traces(getattr(guests, guest)()[0])
    # the inequality witness of ADD: sum_g (x1_g - x2_g) z_g = 1 — two z cells of groups that differ can trade value
    nz = 16 if chip_name == "bls_g1" else 11
    allowed = {(f"xne_z[{i}]", f"xne_z[{j}]") for i in range(nz) for j in range(nz)}
    allowed |= {(f"mz[{i}]", f"mz[{j}]") for i in range(11) for j in range(11)}       # u256_mul: sum_g m_g z_g = 1 likewise
    esc = _hunt(air, machine, chips, pubs, chip_name, np.random.default_rng(14), allowed)
    assert not esc, esc[:10]
