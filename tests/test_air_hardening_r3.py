"""CPU tests (no GPU) of the round-2 advisor's findings on the rv32 AIR:
  * halting is bound to a HALT row: next_pc = HALT_PC = 2^30 is a value no JAL / JALR / branch / sequential row can
    produce, the verifier requires it of the last shard (it used to accept next_pc = 0, which `jalr x0, 0(x0)` reaches);
  * the registers live at REG_BASE + r = 0x38800000 + r of the memory argument, above every address a load, store or
    precompile can form, so no guest access aliases a register;
  * on JALR rows the byte-offset cells are one-hot as on memory rows, so the ADDR lookup bounds the target's top byte;
  * JAL / branch targets outside the text are BAD_PC in the program table (the executor traps there)."""
import os
import re

import numpy as np
import pytest

from dvt_circuits_amd import capi
from tests import _orc, guests
from tools.rvasm import Asm

P = 2013265921
HALT_PC, REG_BASE, BAD_PC = 1 << 30, 0x38800000, 1
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


def _chip(air, chips, name):
    return next(c for c in chips if air.chip(c["chip_id"]).name == name)


def test_constants_agree_between_the_air_and_the_model():
    from oracle import rv32_model
    from tools.airgen import rv32

    assert (rv32.HALT_PC, rv32.REG_BASE, rv32.BAD_PC) == (HALT_PC, REG_BASE, BAD_PC)
    assert (rv32_model.HALT_PC, rv32_model.REG_BASE, rv32_model.BAD_PC) == (HALT_PC, REG_BASE, BAD_PC)
    # nothing but a HALT row reaches HALT_PC: guest addresses, static targets and JALR targets are below REG_BASE < HALT_PC,
    # the JALR "target" 0 - 1 is p - 1
    assert REG_BASE == (rv32.ADDR_TOP_BYTE << 24) + (1 << 23) and REG_BASE + 32 <= HALT_PC < P - 1


def test_the_last_shard_ends_in_a_halt_row_with_the_sentinel_pc(air):
    elf, _ = guests.arith(commit=False)
    chips, pubs, _ = capi.rv32_debug_traces(elf)
    assert pubs[1] == HALT_PC
    cpu = _chip(air, chips, b"cpu")
    col = {n: i for i, n in _names("CPU").items()}
    main = cpu["main"]
    n_real = int(sum(main[col[n]] for n in col if n.startswith("is_")).sum())
    last = n_real - 1
    assert main[col["is_ecall"], last] == 1 and main[col["u_4"], last] == 1 and main[col["next_pc"], last] == HALT_PC
    # the old convention (next_pc = 0 after HALT) no longer satisfies the ECALL constraint
    m = main.copy()
    m[col["next_pc"], last] = 0
    p0 = pubs.copy()
    p0[1] = 0
    assert air.check_constraints(cpu["chip_id"], m, cpu["prep"], p0)[0] > 0


def test_a_jump_cannot_stand_in_for_halt(air):
    """the advisor's attack: end the last shard on a JALR (a `ret` with ra = 0 reaches pc 0) and claim `halted, exit code 0`.
    With next_pc = HALT_PC required of the last shard, the forged last row would have to be a JALR row with
    next_pc = HALT_PC: every way of writing that breaks a constraint or a byte lookup."""
    elf, _ = guests.arith(commit=False)
    chips, pubs, _ = capi.rv32_debug_traces(elf)
    cpu = _chip(air, chips, b"cpu")
    col = {n: i for i, n in _names("CPU").items()}
    main = cpu["main"]
    row = next(r for r in range(main.shape[1]) if main[col["is_jalr"], r] == 1)

    def forged(edit):
        m = main.copy()
        m[:, row + 1:] = 0                      # the JALR row becomes the last real row
        edit(m)
        p2 = pubs.copy()
        p2[1] = HALT_PC
        bad = air.check_constraints(cpu["chip_id"], m, cpu["prep"], p2)[0]
        unb = air.logup_unbalanced([dict(cpu, main=m) if ch is cpu else ch for ch in chips], p2)[0]
        return bad, unb

    # (truncating alone leaves exactly one violation: the row's honest next_pc is not the claimed public value)
    assert forged(lambda m: None)[0] == 1

    def set_next(m):
        m[col["next_pc"], row] = HALT_PC
    bad, _ = forged(set_next)
    assert bad > 0, "next_pc = HALT_PC on a JALR row must contradict next_pc = sum - low bit"

    def sum_bytes(m):                           # ... so the forger rewrites the sum bytes to spell HALT_PC = 00 00 00 40
        set_next(m)
        for i, v in enumerate((0, 0, 0, 0x40)):
            m[col[f"u_{i}"], row] = v
        m[col["u_8"], row] = 0
    bad, _ = forged(sum_bytes)
    assert bad > 0, "the adder ties the sum bytes to rs1 + offset"

    def field_valued(m):                        # ... or puts the whole value into byte 0 (a field element, not a byte)
        set_next(m)
        s = sum(int(m[col[f"u_{i}"], row]) << (8 * i) for i in range(4))
        m[col["u_0"], row] = (int(m[col["u_0"], row]) + HALT_PC - s + int(m[col["u_8"], row])) % P
    bad, unb = forged(field_valued)
    assert bad > 0 or unb > 0


def test_jalr_offset_cells_are_one_hot_so_the_target_bound_holds(air):
    """on JALR rows u[21..23] were free, so the entry (s0 & 3) + 4 of the ADDR lookup (top byte >= 0x38)
    passed.  They are one-hot now: the value 4 + (s0 & 3) cannot be written."""
    elf, _ = guests.arith(commit=False)
    chips, pubs, _ = capi.rv32_debug_traces(elf)
    cpu = _chip(air, chips, b"cpu")
    col = {n: i for i, n in _names("CPU").items()}
    main = cpu["main"]
    row = next(r for r in range(main.shape[1]) if main[col["is_jalr"], r] == 1)
    assert air.check_constraints(cpu["chip_id"], main, cpu["prep"], pubs)[0] == 0
    s0 = int(main[col["u_0"], row])
    for cells in ((4 + (s0 & 3), 0, 0), (1, 0, 1), (0, 2, 0), (2, 1, 0)):   # o1 + 2 o2 + 3 o3 = 4 + (s0 & 3) in several ways
        m = main.copy()
        for k, v in enumerate(cells):
            m[col[f"u_{21 + k}"], row] = v
        assert air.check_constraints(cpu["chip_id"], m, cpu["prep"], pubs)[0] > 0, cells


def test_registers_sit_above_every_guest_address(air):
    """every register-port tuple of the memory bus carries REG_BASE + r; the memory port of a load / store carries an
    address below REG_BASE; the mem_init table lists the 32 registers last, at 0x38800000 + r"""
    from tools.airgen import rv32 as airdef

    cpu_def = next(c for c in airdef.build().chips if c.name == "cpu")

    def ev(e, env):
        if e.op == "const":
            return e.args[0]
        if e.op == "var":
            return env.get(e.args, 0)
        v = [ev(a, env) for a in e.args]
        return {"add": lambda: v[0] + v[1], "sub": lambda: v[0] - v[1], "mul": lambda: v[0] * v[1], "neg": lambda: -v[0]}[e.op]() % P

    idx = {n: i for i, n in enumerate(cpu_def.main_names)}
    mem = [it for it in cpu_def.interactions if it.bus == "mem"]
    assert len(mem) == 8                                   # three register ports + the memory port, receive + send each
    for reg in ("rs2", "rs1", "rd"):
        hits = 0
        for it in mem:
            a0 = ev(it.vals[0], {})
            a7 = ev(it.vals[0], {("main", idx[reg], 0): 7})
            if a7 - a0 == 7:
                assert a0 == REG_BASE, reg
                hits += 1
        assert hits == 2, reg
    elf, _ = guests.subword()
    chips, pubs, _ = capi.rv32_debug_traces(elf)
    mi = _chip(air, chips, b"mem_init")
    col = {n: i for i, n in _names("MEM_INIT").items()}
    m = mi["main"]
    n_real = int(m[col["is_real"]].sum())
    addrs = [sum(int(m[col[f"ab_{i}"], r]) << (8 * i) for i in range(4)) for r in range(n_real)]
    assert addrs[-32:] == [REG_BASE + r for r in range(32)] and all(a < REG_BASE for a in addrs[:-32])
    assert all(m[col["is_img"], r] == 1 for r in range(n_real - 32, n_real))


def test_low_addresses_and_null_pointers_trap_in_the_executor():
    for build in (lambda a: a.lw("a5", "zero", 8), lambda a: a.sw("ra", "zero", 0), lambda a: a.sb("ra", "zero", 31)):
        a = Asm()
        build(a)
        a.halt(0)
        rc, rep, _, err = capi.execute(a.elf())
        assert rc == capi.DVT_ERR_GUEST and "out of range" in err, err
    a = Asm()
    a.li("a0", 0); a.li("a1", 0); a.li("t0", guests.SYS_SHA_EXTEND); a.ecall(); a.halt(0)
    rc, rep, _, err = capi.execute(a.elf())
    assert rc == capi.DVT_ERR_GUEST and "out of range" in err
    a = Asm()
    w = a.dword("w", [0] * 64)
    a.li("a0", w); a.li("a1", 16); a.li("t0", guests.SYS_SHA_COMPRESS); a.ecall(); a.halt(0)
    rc, rep, _, err = capi.execute(a.elf())
    assert rc == capi.DVT_ERR_GUEST and "out of range" in err


def test_jump_to_address_zero_traps_and_is_not_a_halt():
    a = Asm()
    a.li("a0", 0)
    a.jalr("zero", "zero", 0)          # `ret` with ra = 0
    rc, rep, _, err = capi.execute(a.elf())
    assert rc == capi.DVT_ERR_GUEST and not rep["halted"] and "pc outside text" in err


def test_static_targets_outside_the_text_are_bad_pc_in_the_program_table(air):
    """a branch that is never taken / a JAL that is never reached may point outside the text: the program table then
    holds BAD_PC (odd: no row's pc; not HALT_PC), in the product and in the model alike"""
    from oracle import rv32_model

    a = Asm()
    a.li("a0", 0)
    a.beq("a0", "ra", "cont")          # taken (ra = 0): skips the two wild instructions
    a.jal("zero", a.text_base + 0x8002)  # outside the text (and misaligned)
    a.bne("a0", "zero", a.text_base + 0x800)
    a.label("cont")
    a.halt(0)
    elf = a.elf()
    rc, rep, _, err = capi.execute(elf)
    assert rc == 0, err
    chips, pubs, _ = capi.rv32_debug_traces(elf)
    prog = _chip(air, chips, b"program")
    pcol = {}
    for line in open(os.path.join(ROOT, "dvt_circuits_amd", "csrc", "gen", "rv32_cols.h")):
        m = re.match(r"#define RV32_PROGRAM_P_(\w+) (\d+)", line)
        if m:
            pcol[m.group(1)] = int(m.group(2))
    prep = prog["prep"]
    wild = [int(prep[pcol["aux"], r]) for r in range(prep.shape[1]) if prep[pcol["is_jal"], r] == 1 or prep[pcol["is_bne"], r] == 1]
    assert BAD_PC in wild and all(t == BAD_PC or a.text_base <= t < a.pc() for t in wild)
    run = rv32_model.Run(elf)
    mchips, mpubs = rv32_model.traces(run, 0)
    mprog = next(c for c in mchips if c["chip_id"] == prog["chip_id"])
    assert np.array_equal(mprog["prep"], prep)
