"""CPU tests (no GPU) of the SHA-256 precompiles (SP1 syscalls SHA_EXTEND 0x00_30_01_05 and
SHA_COMPRESS 0x00_01_01_06, the calls the reference's guests make through the patched `sha2` crate,
crates/dkg/Cargo.toml:22): executor semantics against hashlib, the chips' traces against the generated checker and the
LogUp multiset, per-cell soundness of both chips and of the cpu chip's precompile row, and the calls that must trap."""
import os
import re
import struct

import pytest

from dvt_circuits_amd import capi
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


def test_schedule_matches_hashlib_through_a_software_compression():
    """sha_schedule_py (what the guest's expected bytes come from) is the SHA-256 message schedule: compressing one
    padded block with it reproduces hashlib"""
    import hashlib

    msg = b"dvt-circuits sha_extend"
    block = msg + b"\x80" + b"\0" * (55 - len(msg)) + struct.pack(">Q", 8 * len(msg))
    w = guests.sha_schedule_py(struct.unpack(">16I", block))
    k = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
         0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
         0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
         0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
         0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
         0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
    h = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
    rotr = lambda v, n: ((v >> n) | (v << (32 - n))) & 0xFFFFFFFF
    a, b, c, d, e, f, g, hh = h
    for i in range(64):
        t1 = (hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + k[i] + w[i]) & 0xFFFFFFFF
        t2 = ((rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c))) & 0xFFFFFFFF
        a, b, c, d, e, f, g, hh = (t1 + t2) & 0xFFFFFFFF, a, b, c, (d + t1) & 0xFFFFFFFF, e, f, g
    out = struct.pack(">8I", *[(x + y) & 0xFFFFFFFF for x, y in zip(h, (a, b, c, d, e, f, g, hh))])
    assert out == hashlib.sha256(msg).digest()


@pytest.mark.parametrize("blocks,log_shard", [(1, 0), (2, 0), (3, 10)])
def test_guest_runs_and_every_shard_satisfies_the_air(air, blocks, log_shard):
    elf, want = guests.sha_extend(blocks)
    rc, rep, pv, out, err = capi.execute_io(elf)
    assert rc == 0 and rep["halted"] and not rep["unprovable"], err
    assert out == want and pv == guests.checksum(want)
    chips, pubs = check_traces(air, elf, log_shard=log_shard)
    if log_shard == 0:
        ext = next(c for c in chips if air.chip(c["chip_id"]).name == b"sha_extend")
        col = {n: i for i, n in _names("SHA_EXTEND").items()}
        assert int(ext["main"][col["is_real"]].sum()) == 64 * blocks and int(ext["main"][col["is_first"]].sum()) == blocks


def test_shards_without_a_call_have_no_sha_extend_chip(air):
    elf, _ = guests.arith()
    chips, _, _ = capi.rv32_debug_traces(elf)
    assert not any(air.chip(c["chip_id"]).name == b"sha_extend" for c in chips)


def test_invalid_calls_trap():
    for kw, frag in ((dict(a1=4), "a1 != 0"), (dict(ptr_off=2), "misaligned"), (dict(ptr_off=0x38000000), "out of range")):
        rc, rep, _, err = capi.execute(guests.sha_extend(1, **kw)[0])
        assert rc != 0 and frag in err, (kw, err)


def test_every_cell_of_the_chip_is_pinned(air):
    """single-cell changes on a load row, the row where the window is full for the first time, a compute row and the last
    row of a call: every one breaks a constraint or the LogUp balance, except cells the row does not use"""
    elf, want = guests.sha_extend(2)
    chips, pubs, _ = capi.rv32_debug_traces(elf)
    extra = pv_extra(guests.checksum(want))
    assert air.logup_unbalanced(chips, pubs, extra=extra)[0] == 0
    ext = next(c for c in chips if air.chip(c["chip_id"]).name == b"sha_extend")
    names = _names("SHA_EXTEND")
    main = ext["main"]
    for row in (0, 3, 15, 16, 40, 63, 64 + 20, 127):
        free = set()
        for c in range(main.shape[0]):
            m = main.copy()
            m[c, row] = (int(m[c, row]) + 1) % P
            if air.check_constraints(ext["chip_id"], m, ext["prep"], pubs)[0]:
                continue
            if air.logup_unbalanced([dict(ext, main=m) if ch is ext else ch for ch in chips], pubs, extra=extra)[0]:
                continue
            free.add(names[c])
        j = row % 64
        # free cells: the carries of a load row (the addition is not checked there), j_inv where it is multiplied by zero,
        # and window words of the call's first row that never reach a computation (they are shifted out before row 16)
        allowed = set()
        if j < 16:
            allowed |= {f"cy_{i}" for i in range(4)}
        if j == 63:
            allowed |= {"j_inv"}
        if j == 0:
            allowed |= {f"w{k}_{i}" for k in (0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15) for i in range(4)}
        assert free <= allowed, (row, sorted(free - allowed))


def test_the_cpu_row_of_a_precompile_call_is_pinned(air):
    """on the ECALL row of the call: the code, a0, a1, clk and shard cells that reach the sys bus, the port that reads a1
    and the precompile flag are pinned"""
    elf, want = guests.sha_extend(1)
    chips, pubs, _ = capi.rv32_debug_traces(elf)
    extra = pv_extra(guests.checksum(want))
    cpu = next(c for c in chips if air.chip(c["chip_id"]).name == b"cpu")
    names = _names("CPU")
    col = {n: i for i, n in names.items()}
    main = cpu["main"]
    rows = [r for r in range(main.shape[1]) if main[col["sys_m"], r] == 1 and main[col["u_1"], r] == 1]
    assert len(rows) == 1
    row, free = rows[0], set()
    for c in range(main.shape[0]):
        m = main.copy()
        m[c, row] = (int(m[c, row]) + 1) % P
        if air.check_constraints(cpu["chip_id"], m, cpu["prep"], pubs)[0]:
            continue
        if air.logup_unbalanced([dict(cpu, main=m) if ch is cpu else ch for ch in chips], pubs, extra=extra)[0]:
            continue
        free.add(names[c])
    # u[2] = 1 / (code byte 1 - 1) and u[5] .. are inverses multiplied by zero or unused decodings; u[22], u[23] only enter the
    # address expression together with u[0] (any of them can absorb a change of another: the EXPRESSION is pinned)
    assert free <= {"u_2", "u_24", "u_25"}, sorted(free)


# ------------------------------------------------------------------------------------------------ SHA_COMPRESS
@pytest.mark.parametrize("msg,log_shard", [(b"abc", 0), (bytes(range(200)), 0), (bytes(range(130)), 10)])
def test_sha256_through_both_precompiles_equals_hashlib_and_satisfies_the_air(air, msg, log_shard):
    elf, want = guests.sha256_precompiled(msg)            # want = hashlib.sha256(msg) as eight words
    rc, rep, pv, out, err = capi.execute_io(elf)
    assert rc == 0 and rep["halted"] and not rep["unprovable"], err
    assert out == want and pv == guests.checksum(want)
    chips, pubs = check_traces(air, elf, log_shard=log_shard)
    if log_shard == 0:
        blocks = (len(msg) + 9 + 63) // 64
        cmp_ = next(c for c in chips if air.chip(c["chip_id"]).name == b"sha_compress")
        col = {n: i for i, n in _names("SHA_COMPRESS").items()}
        assert int(cmp_["main"][col["is_real"]].sum()) == 80 * blocks and int(cmp_["main"][col["is_first"]].sum()) == blocks


def test_invalid_compress_calls_trap():
    for kw, frag in ((dict(h_off=2), "misaligned"), (dict(w_off=1), "misaligned"), (dict(h_off=0x38000000), "out of range")):
        rc, rep, _, err = capi.execute(guests.sha256_precompiled(b"abc", **kw)[0])
        assert rc != 0 and frag in err, (kw, err)
    # state inside the schedule array: the same word would be accessed twice at one timestamp
    elf, _ = guests.sha256_precompiled(b"abc")
    from tools.rvasm import Asm

    a = Asm()
    w = a.dword("w", [0] * 80)
    for ptr_w, ptr_h in ((w, w + 64), (w + 16, w)):
        a2 = Asm()
        w2 = a2.dword("w", [0] * 80)
        a2.li("a0", w2 + (ptr_w - w)); a2.li("a1", w2 + (ptr_h - w)); a2.li("t0", guests.SYS_SHA_COMPRESS); a2.ecall(); a2.halt(0)
        rc, rep, _, err = capi.execute(a2.elf())
        assert rc != 0 and "overlap" in err, err


def test_every_cell_of_the_compress_chip_is_pinned(air):
    """single-cell changes on a state-load row, round rows and a write-back row of the second call of a two-block hash:
    every one breaks a constraint or the LogUp balance, except cells the row does not use"""
    elf, want = guests.sha256_precompiled(bytes(range(70)))
    chips, pubs, _ = capi.rv32_debug_traces(elf)
    extra = pv_extra(guests.checksum(want))
    assert air.logup_unbalanced(chips, pubs, extra=extra)[0] == 0
    cmp_ = next(c for c in chips if air.chip(c["chip_id"]).name == b"sha_compress")
    names = _names("SHA_COMPRESS")
    main = cmp_["main"]
    carries = {f"c{t}_{i}" for t in "ea" for i in range(6)}
    for row in (0, 5, 8, 9, 40, 71, 72, 79, 80 + 3, 80 + 30, 80 + 76):
        free = set()
        for c in range(main.shape[0]):
            m = main.copy()
            m[c, row] = (int(m[c, row]) + 1) % P
            if air.check_constraints(cmp_["chip_id"], m, cmp_["prep"], pubs)[0]:
                continue
            if air.logup_unbalanced([dict(cmp_, main=m) if ch is cmp_ else ch for ch in chips], pubs, extra=extra)[0]:
                continue
            free.add(names[c])
        g = (row % 80) // 8
        allowed = set()
        if g == 0 or g == 9:
            allowed |= carries                      # no addition on these rows
        if g != 9:
            allowed |= {"cf_0", "cf_1"}
        if row % 80 == 0:
            allowed |= {"h_0", "h_1"}               # the variables of a call's first row are shifted out before any use
        assert free <= allowed, (row, sorted(free - allowed))


def test_dkg_shaped_guest_hashing_through_the_precompiles(air):
    """the finalization-shaped guest with every SHA-256 block done by SHA_EXTEND + SHA_COMPRESS (the patched `sha2` crate's
    calls) commits the same public values as its software-SHA form in fewer cycles, and its shards satisfy the AIR"""
    import json

    example = open(os.path.join(ROOT, "tests", "golden", "finalization_example.json"), "rb").read()
    buf = capi.stdin_from_json("finalization", example)
    want = guests.dkg_like_expected(buf, "finalization")
    soft = capi.execute(guests.dkg_like("finalization"), [buf])
    elf = guests.dkg_like("finalization", sha_precompiles=True)
    rc, rep, pv, err = capi.execute(elf, [buf])
    assert rc == 0 and pv == want == soft[2], err
    assert rep["cycles"] < soft[1]["cycles"] // 2
    check_traces(air, elf, [buf], log_shard=13)
