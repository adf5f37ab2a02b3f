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


# ------------------------------------------------------------------------------------------------------------------
# Round 3: one hand-written statement per remaining family / chip.  Functional statements over the
# integers: "the cells that carry the result hold f(the cells that carry the operands)", with f written here from the
# ISA / FIPS 180-4 / the curve equations — not from tools/airgen.
M32 = 0xFFFFFFFF
HALT_PC = 1 << 30
REG_BASE = 0x38800000
BLS_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
SECP_P = (1 << 256) - (1 << 32) - 977


def _sx(v, bits):
    return v - (1 << bits) if v >> (bits - 1) else v


def check_subword(names, main):
    """LB LBU LH LHU SB SH: address = b + off (sum bytes u[0..3]); the access moves the aligned word (before u[13..16],
    after u[9..12]); loads leave it unchanged and return the addressed byte / halfword, sign- or zero-extended; stores patch
    exactly the addressed byte / halfword with the low bytes of c; halfword accesses are even; falls through to pc + 4"""
    col, vec = _cols(names, main)
    a, b, c, off, u = vec("a"), vec("b"), vec("c"), vec("imm"), vec("u", 26)
    fam = {k: col("is_" + k) == 1 for k in ("lb", "lbu", "lh", "lhu", "sb", "sh")}
    rows = np.zeros_like(fam["lb"])
    for v in fam.values():
        rows |= v
    ok = np.ones_like(rows)
    n = main.shape[1]
    for r in np.nonzero(rows)[0]:
        g = lambda vecs: sum(int(x[r]) << (8 * i) for i, x in enumerate(vecs))
        addr = (g(b) + g(off)) & M32
        good = g(u[0:4]) == addr
        o = addr & 3
        good &= [int(u[21][r]), int(u[22][r]), int(u[23][r])] == [int(o == 1), int(o == 2), int(o == 3)]
        before, after, av, cv = g(u[13:17]), g(u[9:13]), g(a), g(c)
        sh = 8 * o
        if fam["lb"][r] or fam["lbu"][r]:
            byte = (before >> sh) & 0xFF
            want = _sx(byte, 8) & M32 if fam["lb"][r] else byte
            good &= after == before and av == want
        elif fam["lh"][r] or fam["lhu"][r]:
            half = (before >> sh) & 0xFFFF
            want = _sx(half, 16) & M32 if fam["lh"][r] else half
            good &= o % 2 == 0 and after == before and av == want
        elif fam["sb"][r]:
            good &= after == (before & ~(0xFF << sh) & M32) | ((cv & 0xFF) << sh)
        else:
            good &= o % 2 == 0 and after == (before & ~(0xFFFF << sh) & M32) | ((cv & 0xFFFF) << sh)
        good &= int(col("next_pc")[r]) == int(col("pc")[r]) + 4
        ok[r] = bool(good)
    return rows, ok


def check_ecall(names, main, pubs):
    """ECALL rows: b = t0 (the id), c = a0.  HALT (id 0): next_pc = HALT_PC, exit code c = the public value, below 2^24;
    every other call continues at pc + 4.  Only HINT_LEN (0xF0) changes t0 (a is advice), all others leave a = b.
    COMMIT (0x10) and precompile calls (byte 1 of the id = 1) are "sys rows": sys_m = 1 and the memory port reads register
    a1 (word REG_BASE + 11) without changing it."""
    col, vec = _cols(names, main)
    a, b, c, u = vec("a"), vec("b"), vec("c"), vec("u", 26)
    rows = col("is_ecall") == 1
    ok = np.ones_like(rows)
    for r in np.nonzero(rows)[0]:
        g = lambda vecs: sum(int(x[r]) << (8 * i) for i, x in enumerate(vecs))
        sid, a0 = g(b), g(c)
        pc, nxt = int(col("pc")[r]), int(col("next_pc")[r])
        good = True
        if sid == 0:
            good &= nxt == HALT_PC and a0 < 1 << 24 and a0 == int(pubs[2])
        else:
            good &= nxt == pc + 4
        if sid != 0xF0:
            good &= g(a) == sid
        is_sys = sid == 0x10 or int(b[1][r]) == 1
        good &= int(col("sys_m")[r]) == int(is_sys)
        if is_sys:
            addr = (int(u[0][r]) + 256 * int(u[1][r]) + 65536 * int(u[2][r]) + (1 << 24) * int(u[3][r]) - (int(u[21][r]) + 2 * int(u[22][r]) + 3 * int(u[23][r]))) % P
            good &= addr == REG_BASE + 11 and g(u[9:13]) == g(u[13:17])
        ok[r] = bool(good)
    return rows, ok


def check_shift(names, main):
    """shift chip: a = b << / >> / >>a (c mod 32)"""
    col, vec = _cols(names, main)
    a, b, c = vec("a"), vec("b"), vec("c")
    rows = col("is_real") == 1
    ok = np.ones_like(rows)
    for r in np.nonzero(rows)[0]:
        g = lambda vecs: sum(int(x[r]) << (8 * i) for i, x in enumerate(vecs))
        bv, s = g(b), g(c) & 31
        want = (bv << s) & M32 if col("is_sll")[r] else bv >> s if col("is_srl")[r] else (_sx(bv, 32) >> s) & M32
        ok[r] = g(a) == want and int(col("is_sll")[r]) + int(col("is_srl")[r]) + int(col("is_sra")[r]) == 1
    return rows, ok


def check_muldiv(names, main):
    """muldiv chip: MULH MULHSU DIV DIVU REM REMU as the RISC-V manual defines them (division by zero, signed overflow)"""
    col, vec = _cols(names, main)
    a, b, c = vec("a"), vec("b"), vec("c")
    rows = col("is_real") == 1
    ok = np.ones_like(rows)

    def tdiv(x, y):
        q = abs(x) // abs(y)
        return -q if (x < 0) != (y < 0) else q

    for r in np.nonzero(rows)[0]:
        g = lambda vecs: sum(int(x[r]) << (8 * i) for i, x in enumerate(vecs))
        bv, cv = g(b), g(c)
        sb, sc = _sx(bv, 32), _sx(cv, 32)
        op = [k for k in ("mulh", "mulhsu", "div", "divu", "rem", "remu") if col("is_" + k)[r] == 1]
        if len(op) != 1:
            ok[r] = False
            continue
        op = op[0]
        if op == "mulh":
            want = ((sb * sc) >> 32) & M32
        elif op == "mulhsu":
            want = ((sb * cv) >> 32) & M32
        elif op == "divu":
            want = M32 if cv == 0 else bv // cv
        elif op == "remu":
            want = bv if cv == 0 else bv % cv
        elif cv == 0:
            want = M32 if op == "div" else bv
        elif bv == 0x80000000 and cv == M32:
            want = bv if op == "div" else 0
        else:
            q = tdiv(sb, sc)
            want = (q if op == "div" else sb - q * sc) & M32
        ok[r] = g(a) == want
    return rows, ok


def _rotr(v, n):
    return ((v >> n) | (v << (32 - n))) & M32


def check_sha_extend(names, main):
    """sha_extend chip, rows j >= 16 of a call (is_load = 0): the word written is s1(w[j-2]) + w[j-7] + s0(w[j-15]) + w[j-16]
    (FIPS 180-4 section 6.2.2 step 1) with the window W[k] = w[j-16+k]; on rows j < 16 the word read leaves memory
    unchanged; between the rows of a call the window moves by one word"""
    col, vec = _cols(names, main)
    W = [vec(f"w{k}") for k in range(16)]
    nw, old = vec("nw"), vec("old")
    rows = col("is_real") == 1
    ok = np.ones_like(rows)
    n = main.shape[1]
    for r in np.nonzero(rows)[0]:
        g = lambda vecs, rr=r: sum(int(x[rr]) << (8 * i) for i, x in enumerate(vecs))
        good = True
        if col("is_load")[r] == 1:
            good &= g(nw) == g(old)
        else:
            x, y = g(W[1]), g(W[14])
            s0 = _rotr(x, 7) ^ _rotr(x, 18) ^ (x >> 3)
            s1 = _rotr(y, 17) ^ _rotr(y, 19) ^ (y >> 10)
            good &= g(nw) == (s1 + g(W[9]) + s0 + g(W[0])) & M32
        if col("is_last")[r] == 0 and r + 1 < n:
            for k in range(15):
                good &= g(W[k], r + 1) == g(W[k + 1])
            good &= g(W[15], r + 1) == g(nw)
            good &= int(col("j")[r + 1]) == int(col("j")[r]) + 1
        ok[r] = bool(good)
    return rows, ok


SHA_K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]


def check_sha_compress(names, main):
    """sha_compress chip: on a round row (groups 1..8, round i = 8 (g - 1) + o) the next row's a and e are
    T1 + T2 and d + T1 with T1 = h + Sigma1(e) + Ch(e,f,g) + K[i] + w[i], T2 = Sigma0(a) + Maj(a,b,c) (FIPS 180-4 6.2.2 step 3),
    w[i] being the word the row reads; on every row of a call but the last b..d and f..h are the previous a..c and e..g; a
    write-back row stores old + h"""
    col, vec = _cols(names, main)
    bits = {k: [main[names.index(f"{k}b[{i}]")].astype(np.int64) for i in range(32)] for k in "abcefg"}
    d, h = vec("d", 2), vec("h", 2)
    mv, mo = vec("mv"), vec("mo")
    oc, gr = vec("oc", 8), vec("gr", 10)
    rows = col("is_real") == 1
    ok = np.ones_like(rows)
    n = main.shape[1]
    for r in np.nonzero(rows)[0]:
        w32 = lambda k, rr=r: sum(int(bits[k][i][rr]) << i for i in range(32))
        hw = lambda v, rr=r: int(v[0][rr]) | (int(v[1][rr]) << 16)
        g8 = lambda vecs, rr=r: sum(int(x[rr]) << (8 * i) for i, x in enumerate(vecs))
        g = [int(x[r]) for x in gr].index(1)
        o = [int(x[r]) for x in oc].index(1)
        a_, b_, c_, e_, f_, g_ = (w32(k) for k in "abcefg")
        good = True
        last = g == 9 and o == 7
        if not last and r + 1 < n:
            good &= w32("b", r + 1) == a_ and w32("c", r + 1) == b_ and hw(d, r + 1) == c_
            good &= w32("f", r + 1) == e_ and w32("g", r + 1) == f_ and hw(h, r + 1) == g_
            if 1 <= g <= 8:
                i = 8 * (g - 1) + o
                t1 = (hw(h) + (_rotr(e_, 6) ^ _rotr(e_, 11) ^ _rotr(e_, 25)) + ((e_ & f_) ^ (~e_ & g_ & M32)) + SHA_K[i] + g8(mv)) & M32
                t2 = ((_rotr(a_, 2) ^ _rotr(a_, 13) ^ _rotr(a_, 22)) + ((a_ & b_) ^ (a_ & c_) ^ (b_ & c_))) & M32
                good &= w32("a", r + 1) == (t1 + t2) & M32 and w32("e", r + 1) == (hw(d) + t1) & M32
                good &= g8(mv) == g8(mo)
            elif g == 0:
                good &= w32("a", r + 1) == g8(mv) and w32("e", r + 1) == hw(d) and g8(mv) == g8(mo)
        if g == 9:
            good &= g8(mv) == (g8(mo) + hw(h)) & M32
        ok[r] = bool(good)
    return rows, ok


def check_mem_init(names, main):
    """mem_init: addresses (four bytes) strictly increase over the real rows, the gap to the previous address is the row's
    four d bytes (+ 1), the registers' words REG_BASE + r come last"""
    col, vec = _cols(names, main)
    ab, dd = vec("ab"), vec("d")
    rows = col("is_real") == 1
    ok = np.ones_like(rows)
    idx = np.nonzero(rows)[0]
    addr = [sum(int(x[r]) << (8 * i) for i, x in enumerate(ab)) for r in idx]
    for k, r in enumerate(idx):
        good = all(0 <= int(x[r]) < 256 for x in ab) and addr[k] < 0x39000000
        if k:
            gap = sum(int(x[r]) << (8 * i) for i, x in enumerate(dd))
            good &= addr[k] > addr[k - 1] and r == idx[k - 1] + 1 and addr[k] - addr[k - 1] - 1 == gap
        ok[r] = bool(good)
    if len(addr) >= 32:
        for k in range(32):
            ok[idx[len(idx) - 32 + k]] &= addr[len(idx) - 32 + k] == REG_BASE + k
    return rows, ok


def check_field_op(names, main, fp2=False):
    """fp_op / fp2_op: the result cells hold x op y in Fp (Fp2 = Fp[u] / (u^2 + 1)), reduced below p"""
    col, _ = _cols(names, main)
    rows = col("is_real") == 1
    ok = np.ones_like(rows)
    val = lambda nm, r: sum(int(main[names.index(f"{nm}[{i}]"), r]) << (8 * i) for i in range(48))
    for r in np.nonzero(rows)[0]:
        op = [k for k in ("add", "sub", "mul") if col("is_" + k)[r] == 1]
        if len(op) != 1:
            ok[r] = False
            continue
        f = {"add": lambda x, y: (x + y) % BLS_P, "sub": lambda x, y: (x - y) % BLS_P, "mul": lambda x, y: x * y % BLS_P}[op[0]]
        if not fp2:
            ok[r] = val("r", r) == f(val("x", r), val("y", r))
        else:
            x0, x1, y0, y1 = val("x0", r), val("x1", r), val("y0", r), val("y1", r)
            want = ((x0 * y0 - x1 * y1) % BLS_P, (x0 * y1 + x1 * y0) % BLS_P) if op[0] == "mul" else (f(x0, y0), f(x1, y1))
            ok[r] = (val("r0", r), val("r1", r)) == want
    return rows, ok


def check_curve_op(names, main, L, p):
    """bls_g1 / secp_k1: (x3, y3) = (x1, y1) + (x2, y2) resp. 2 (x1, y1) on y^2 = x^3 + b by the chord-and-tangent formulas,
    all coordinates reduced; ADD needs x1 != x2"""
    col, _ = _cols(names, main)
    rows = col("is_real") == 1
    ok = np.ones_like(rows)
    val = lambda nm, r: sum(int(main[names.index(f"{nm}[{i}]"), r]) << (8 * i) for i in range(L))
    for r in np.nonzero(rows)[0]:
        x1, y1, x3, y3 = val("x1", r), val("y1", r), val("x3", r), val("y3", r)
        good = max(x1, y1, x3, y3) < p and int(col("is_add")[r]) + int(col("is_dbl")[r]) == 1
        if col("is_add")[r] == 1:
            x2, y2 = val("x2", r), val("y2", r)
            good &= max(x2, y2) < p and x1 != x2
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p if x1 != x2 else 0
        else:
            x2 = x1
            good &= y1 != 0
            lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p if y1 else 0
        good &= x3 == (lam * lam - x1 - x2) % p and y3 == (lam * (x1 - x3) - y1) % p
        ok[r] = bool(good)
    return rows, ok


def check_u256_mul(names, main):
    """u256_mul: r = x * y mod m, m = 0 standing for 2^256; r below the modulus"""
    col, _ = _cols(names, main)
    rows = col("is_real") == 1
    ok = np.ones_like(rows)
    val = lambda nm, r: sum(int(main[names.index(f"{nm}[{i}]"), r]) << (8 * i) for i in range(32))
    for r in np.nonzero(rows)[0]:
        m = val("m", r) or 1 << 256
        ok[r] = val("r", r) == val("x", r) * val("y", r) % m and int(col("m_zero")[r]) == int(val("m", r) == 0)
    return rows, ok


def check_chip(name, names, main, pubs):
    """-> (rows of the statement, rows satisfying it) for the hand-written statement(s) of chip `name`, or None"""
    shard = int(pubs[3])
    if name == "cpu":
        out_rows, out_ok = None, None
        for rows, ok in (check_add(names, main), check_lw(names, main, shard), check_mul(names, main), check_branches(names, main),
                         check_subword(names, main), check_ecall(names, main, pubs)):
            out_rows = rows if out_rows is None else out_rows | rows
            out_ok = (ok | ~rows) if out_ok is None else out_ok & (ok | ~rows)
        return out_rows, out_ok
    table = {"shift": check_shift, "muldiv": check_muldiv, "sha_extend": check_sha_extend, "sha_compress": check_sha_compress,
             "mem_init": check_mem_init, "fp_op": check_field_op, "fp2_op": lambda n_, m_: check_field_op(n_, m_, fp2=True),
             "bls_g1": lambda n_, m_: check_curve_op(n_, m_, 48, BLS_P), "secp_k1": lambda n_, m_: check_curve_op(n_, m_, 32, SECP_P),
             "u256_mul": check_u256_mul}
    return table[name](names, main) if name in table else None
