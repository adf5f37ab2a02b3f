"""ORACLE — TEST INFRASTRUCTURE ONLY.

Hand-written constraint checks of four instruction families of the rv32 cpu chip — ADD, LW, MUL, and the branches —
stated directly from DESIGN.md section 5 (what each family must prove) and evaluated over the integers with numpy.
They do NOT come from tools/airgen: the generated checker (oracle/gen/air_rv32.c) and the product's kernels are both
emitted from the same AIR description, so a wrong constraint there would be wrong identically on both sides; these
checks are the second opinion.  tests/test_rv32_model_parity.py requires them to accept real
traces and to reject every single-cell change of the cells a family's statement reads.

Each check returns a boolean array over the family's rows: True = the row satisfies the statement.
"""
import numpy as np

P = 2013265921


def _cols(names, main):
    idx = {n: i for i, n in enumerate(names)}

    def col(name):
        return main[idx[name]].astype(np.int64)

    def vec(name, n=4):
        return [main[idx[f"{name}[{i}]"]].astype(np.int64) for i in range(n)]

    return col, vec


def word(bs):
    return sum(b << (8 * i) for i, b in enumerate(bs))


def is_byte(x):
    return (x >= 0) & (x < 256)


def check_add(names, main):
    """ADD / ADDI: a = b + c mod 2^32 with byte limbs and boolean carries in u[0..3]; falls through to pc + 4"""
    col, vec = _cols(names, main)
    a, b, c, u = vec("a"), vec("b"), vec("c"), vec("u", 4)
    rows = col("is_add") == 1
    ok = np.ones_like(rows)
    carry = 0
    for i in range(4):
        ok &= (u[i] == 0) | (u[i] == 1)
        ok &= b[i] + c[i] + carry == a[i] + 256 * u[i]
        carry = u[i]
    ok &= (word(a) == (word(b) + word(c)) % (1 << 32)) | ~(is_byte(a[0]) & is_byte(a[1]) & is_byte(a[2]) & is_byte(a[3]))
    ok &= col("next_pc") == col("pc") + 4
    imm = vec("imm")
    immc = col("imm_c") == 1
    for i in range(4):
        ok &= ~immc | (c[i] == imm[i])
    return rows, ok


def check_lw(names, main, shard):
    """LW: address = b + off (bytes in u[0..3], carries u[4..7]) word aligned; the memory word is unchanged (u[9..12] =
    u[13..16]) and equals a; the previous access (shard u[19], clk u[17]) is strictly earlier than (shard, clk + 2) with the
    gap in u[8] + 2^16 u[18]"""
    col, vec = _cols(names, main)
    a, b, off, u = vec("a"), vec("b"), vec("imm"), vec("u", 26)      # (the immediate field of a load is its address offset)
    rows = col("is_lw") == 1
    ok = np.ones_like(rows)
    carry = 0
    for i in range(4):
        ok &= (u[4 + i] == 0) | (u[4 + i] == 1)
        ok &= b[i] + off[i] + carry == u[i] + 256 * u[4 + i]
        carry = u[4 + i]
    ok &= (u[21] == 0) & (u[22] == 0) & (u[23] == 0)                 # no byte offset
    ok &= u[0] % 4 == 0
    for i in range(4):
        ok &= (u[9 + i] == u[13 + i]) & (a[i] == u[9 + i])
    clk = col("clk")
    same = u[20]
    ok &= (same == 0) | (same == 1)
    gap = u[8] + 65536 * u[18]
    ok &= np.where(same == 1, (u[19] == shard) & (clk + 2 - u[17] - 1 == gap), shard - u[19] - 1 == gap)
    ok &= (gap >= 0) & (u[8] < 65536) & (u[18] < 256)
    ok &= col("next_pc") == col("pc") + 4
    return rows, ok


def check_mul(names, main):
    """MUL / MULHU: the 64-bit product of b and c byte by byte; the result half is a, the other half u[0..3], carries out of
    bytes 0..6 in u[4..10]"""
    col, vec = _cols(names, main)
    a, b, c, u = vec("a"), vec("b"), vec("c"), vec("u", 11)
    is_mul, is_hu = col("is_mul") == 1, col("is_mulhu") == 1
    rows = is_mul | is_hu
    ok = np.ones_like(rows)
    carry = 0
    for k in range(8):
        terms = sum(b[i] * c[k - i] for i in range(4) if 0 <= k - i < 4)
        out = np.where(is_mul, a[k] if k < 4 else u[k - 4], u[k] if k < 4 else a[k - 4])
        cy = u[4 + k] if k < 7 else 0
        ok &= terms + carry == out + 256 * cy
        carry = cy
    prod = word(b).astype(np.uint64) * word(c).astype(np.uint64)             # < 2^64: exact in uint64
    wa = word(a).astype(np.uint64)
    ok &= np.where(is_mul, wa == prod % np.uint64(1 << 32), wa == prod >> np.uint64(32)) | ~rows
    ok &= col("next_pc") == col("pc") + 4
    return rows, ok


def check_branches(names, main):
    """BEQ BNE BLT BGE BLTU BGEU: next_pc = target (aux) when the condition on (b, c) holds, else pc + 4; the comparison
    witness: u[0..3] flags the most significant differing byte (sign bits flipped for the signed forms), u[10] / u[20] hold
    that byte of b / c, u[19] = u[10] < u[20]"""
    col, vec = _cols(names, main)
    b, c, u = vec("b"), vec("c"), vec("u", 26)
    # BLT / BLTU share the selector is_brlt, BGE / BGEU is_brge; the value column cmp_signed tells them apart
    signed = col("cmp_signed") == 1
    lt_sel, ge_sel = col("is_brlt") == 1, col("is_brge") == 1
    fam = {"beq": col("is_beq") == 1, "bne": col("is_bne") == 1, "blt": lt_sel & signed, "bltu": lt_sel & ~signed,
           "bge": ge_sel & signed, "bgeu": ge_sel & ~signed}
    rows = np.zeros_like(fam["beq"])
    for v in fam.values():
        rows |= v
    bw, cw = word(b), word(c)
    sb = np.where(bw >= 1 << 31, bw - (1 << 32), bw)
    sc = np.where(cw >= 1 << 31, cw - (1 << 32), cw)
    taken = (fam["beq"] & (bw == cw)) | (fam["bne"] & (bw != cw)) | (fam["blt"] & (sb < sc)) | (fam["bge"] & (sb >= sc)) \
        | (fam["bltu"] & (bw < cw)) | (fam["bgeu"] & (bw >= cw))
    ok = col("next_pc") == np.where(taken, col("aux"), col("pc") + 4)
    bb = b[:3] + [np.where(signed, b[3] ^ 0x80, b[3])]
    cc = c[:3] + [np.where(signed, c[3] ^ 0x80, c[3])]
    nflag = sum(u[i] for i in range(4))
    ok &= (nflag == 0) | (nflag == 1)
    for i in range(4):
        ok &= (u[i] == 0) | (u[i] == 1)
        above_equal = np.ones_like(rows)
        for j in range(i + 1, 4):
            above_equal &= bb[j] == cc[j]
        ok &= (u[i] == 0) | (above_equal & (bb[i] != cc[i]))
    ok &= (nflag == 1) | (bw == cw)
    ok &= u[10] == sum(u[i] * bb[i] for i in range(4))
    ok &= u[20] == sum(u[i] * cc[i] for i in range(4))
    ok &= u[19] == (u[10] < u[20])
    return rows, ok


def check_all(names, main, shard):
    """-> dict family -> (number of rows of the family, number of rows violating its statement)"""
    res = {}
    for fam, (rows, ok) in (("add", check_add(names, main)), ("lw", check_lw(names, main, shard)), ("mul", check_mul(names, main)),
                            ("branch", check_branches(names, main))):
        res[fam] = (int(rows.sum()), int((rows & ~ok).sum()))
    return res
