"""RV32IM guest programs assembled with tools/rvasm.py for the executor / prover tests
and the benchmark.  `bignum` is the synthetic "n-participant DKG-like" workload:
multi-limb multiply-accumulate over 384-bit operands, whose instruction mix
(add / sltu / lw / mul / mulhu / sw / branches) follows the histogram measured on
the reference's finalization guest."""
import struct

from tools.rvasm import Asm, SYS_COMMIT, SYS_HINT_LEN, SYS_HINT_READ, SYS_WRITE

M32 = 0xFFFFFFFF
HEAP = 0x00400000


def _write_pv(a, ptr_reg, nbytes):
    """commit nbytes/4 words starting at ptr_reg as public values (one COMMIT ecall per word)"""
    assert nbytes % 4 == 0
    lab = f"commit_{len(a.items)}"
    a.mv("s8", ptr_reg)
    a.li("s9", nbytes // 4)
    a.label(lab)
    a.lw("a0", "s8", 0)
    a.li("t0", SYS_COMMIT)
    a.ecall()
    a.addi("s8", "s8", 4)
    a.addi("s9", "s9", -1)
    a.bne("s9", "zero", lab)


def arith():
    """Every provable instruction on a few operand pairs; results go to public values."""
    pairs = [(0, 0), (1, M32), (0x80000000, 1), (0x7FFFFFFF, 0x80000000), (0x12345678, 0x9ABCDEF0), (M32, M32), (5, 3), (3, 5)]
    a = Asm()
    out = a.dword("out", [0] * (len(pairs) * 16))
    a.li("s0", out)
    exp = []
    sx = lambda v: v - (1 << 32) if v >> 31 else v
    for x, y in pairs:
        a.li("a3", x)
        a.li("a4", y)
        ops = [("add", (x + y) & M32), ("sub", (x - y) & M32), ("and_", x & y), ("or_", x | y), ("xor", x ^ y),
               ("slt", int(sx(x) < sx(y))), ("sltu", int(x < y)), ("mul", (x * y) & M32), ("mulhu", (x * y) >> 32)]
        for name, want in ops:
            getattr(a, name)("a5", "a3", "a4")
            a.sw("a5", "s0", 0)
            a.addi("s0", "s0", 4)
            exp.append(want)
        imm = (y & 0x7FF) - (0x400 if y & 1 else 0)
        for name, want in [("addi", (x + imm) & M32), ("slti", int(sx(x) < imm)), ("sltiu", int(x < (imm & M32))),
                           ("xori", x ^ (imm & M32)), ("ori", x | (imm & M32)), ("andi", x & (imm & M32))]:
            getattr(a, name)("a5", "a3", imm)
            a.sw("a5", "s0", 0)
            a.addi("s0", "s0", 4)
            exp.append(want)
        # branches: record which are taken as a bitmask
        a.li("a5", 0)
        for bit, (name, taken) in enumerate([("beq", x == y), ("bne", x != y), ("blt", sx(x) < sx(y)), ("bge", sx(x) >= sx(y)),
                                             ("bltu", x < y), ("bgeu", x >= y)]):
            lab = f"t{len(exp)}_{bit}"
            getattr(a, name)("a3", "a4", lab)
            a.j(lab + "_n")
            a.label(lab)
            a.ori("a5", "a5", 1 << bit)
            a.label(lab + "_n")
        a.sw("a5", "s0", 0)
        a.addi("s0", "s0", 4)
        exp.append(sum(1 << b for b, t in enumerate([x == y, x != y, sx(x) < sx(y), sx(x) >= sx(y), x < y, x >= y]) if t))
    # call / return, load back, auipc
    a.call("fn")
    a.j("after")
    a.label("fn")
    a.lw("a5", "s0", -4)
    a.addi("a5", "a5", 1)
    a.ret()
    a.label("after")
    a.sw("a5", "s0", 0)
    exp.append((exp[-1] + 1) & M32)
    a.li("s1", out)
    _write_pv(a, "s1", 4 * len(exp))
    a.halt(0)
    return a.elf(), b"".join(struct.pack("<I", v) for v in exp)


def _bignum_consts(limbs):
    A = [(0x9E3779B9 * (i + 1)) & M32 for i in range(limbs)]
    B = [(0x85EBCA6B * (i + 3) + 7) & M32 for i in range(limbs)]
    return A, B


def _bignum_loop(a, pa, pb, pt, iters, limbs):
    """emit: t += a * b (schoolbook, `limbs` x `limbs` words) repeated `iters` times, then commit t and halt"""
    a.li("s0", iters)
    a.label("outer")
    a.li("s1", pb)                 # &b[i]
    a.li("s2", pt)                 # &t[i]
    a.li("s3", limbs)              # i counter
    a.label("row")
    a.lw("a3", "s1", 0)            # b[i]
    a.li("s4", pa)                 # &a[j]
    a.mv("s5", "s2")               # &t[i+j]
    a.li("s6", limbs)              # j counter
    a.li("a6", 0)                  # carry
    a.label("col")
    a.lw("a4", "s4", 0)            # a[j]
    a.mul("a5", "a4", "a3")        # lo
    a.mulhu("a7", "a4", "a3")      # hi
    a.lw("t1", "s5", 0)            # t[i+j]
    a.add("t1", "t1", "a5")
    a.sltu("t2", "t1", "a5")       # c1
    a.add("t1", "t1", "a6")
    a.sltu("t3", "t1", "a6")       # c2
    a.sw("t1", "s5", 0)
    a.add("a6", "a7", "t2")
    a.add("a6", "a6", "t3")        # carry = hi + c1 + c2
    a.addi("s4", "s4", 4)
    a.addi("s5", "s5", 4)
    a.addi("s6", "s6", -1)
    a.bne("s6", "zero", "col")
    a.lw("t1", "s5", 0)            # propagate the final carry into t[i+limbs]
    a.add("t1", "t1", "a6")
    a.sw("t1", "s5", 0)
    a.addi("s1", "s1", 4)
    a.addi("s2", "s2", 4)
    a.addi("s3", "s3", -1)
    a.bne("s3", "zero", "row")
    a.addi("s0", "s0", -1)
    a.bne("s0", "zero", "outer")
    a.li("s1", pt)
    _write_pv(a, "s1", 4 * 2 * limbs)
    a.halt(0)


def _bignum_expected(A, B, iters, limbs):
    """the same word-level algorithm in Python (the carry out of t[i+limbs] is dropped, as in the guest)"""
    t = [0] * (2 * limbs + 1)
    for _ in range(iters):
        for i in range(limbs):
            carry = 0
            for j in range(limbs):
                p = A[j] * B[i]
                lo, hi = p & M32, p >> 32
                v = (t[i + j] + lo) & M32
                c1 = int(v < lo)
                v2 = (v + carry) & M32
                c2 = int(v2 < carry)
                t[i + j] = v2
                carry = (hi + c1 + c2) & M32
            t[i + limbs] = (t[i + limbs] + carry) & M32
    return b"".join(struct.pack("<I", t[i]) for i in range(2 * limbs))


def bignum(iters, limbs=12):
    """t += a * b (schoolbook, `limbs` x `limbs` words) repeated `iters` times; returns (elf, expected public values)."""
    A, B = _bignum_consts(limbs)
    a = Asm()
    pa, pb = a.dword("a", A), a.dword("b", B)
    pt = a.dword("t", [0] * (2 * limbs + 1))
    _bignum_loop(a, pa, pb, pt, iters, limbs)
    return a.elf(), _bignum_expected(A, B, iters, limbs)


def finalization_like(iters, stdin_buf: bytes, limbs=12):
    """The benchmark workload in the shape of the reference's finalization guest (crates/finalization_prove/src/main.rs:
    read ONE stdin buffer, do multi-limb field arithmetic seeded by it, commit the results): reads the buffer through
    HINT_LEN / HINT_READ, folds every input word into the operands (a[k mod limbs] += w_k, b[(k + limbs/2) mod limbs] ^= w_k),
    then runs the 384-bit multiply-accumulate of `bignum`.  Returns (elf, expected public values for `stdin_buf`)."""
    A, B = _bignum_consts(limbs)
    a = Asm()
    pa, pb = a.dword("a", A), a.dword("b", B)
    pt = a.dword("t", [0] * (2 * limbs + 1))
    a.li("t0", SYS_HINT_LEN)
    a.ecall()                      # t0 <- length
    a.mv("s1", "t0")
    a.li("a0", HEAP)
    a.mv("a1", "s1")
    a.li("t0", SYS_HINT_READ)
    a.ecall()
    a.li("s2", HEAP)
    a.add("s3", "s2", "s1")        # end of the input
    a.li("s4", pa)                 # cursor in a
    a.li("s5", pa + 4 * limbs)
    a.li("s6", pb + 4 * (limbs // 2))   # cursor in b
    a.li("s7", pb + 4 * limbs)
    a.beq("s2", "s3", "folded")
    a.label("fold")
    a.lw("a4", "s2", 0)
    a.lw("a5", "s4", 0)
    a.add("a5", "a5", "a4")
    a.sw("a5", "s4", 0)
    a.lw("a5", "s6", 0)
    a.xor("a5", "a5", "a4")
    a.sw("a5", "s6", 0)
    a.addi("s4", "s4", 4)
    a.bne("s4", "s5", "fa")
    a.li("s4", pa)
    a.label("fa")
    a.addi("s6", "s6", 4)
    a.bne("s6", "s7", "fb")
    a.li("s6", pb)
    a.label("fb")
    a.addi("s2", "s2", 4)
    a.bltu("s2", "s3", "fold")
    a.label("folded")
    _bignum_loop(a, pa, pb, pt, iters, limbs)
    padded = stdin_buf + b"\0" * (-len(stdin_buf) % 4)
    for k, (w,) in enumerate(struct.iter_unpack("<I", padded)):
        A[k % limbs] = (A[k % limbs] + w) & M32
        B[(k + limbs // 2) % limbs] ^= w
    return a.elf(), _bignum_expected(A, B, iters, limbs)


def hint_sum():
    """Reads one stdin buffer through HINT_LEN / HINT_READ and commits the sum of its words."""
    a = Asm()
    res = a.dword("res", [0])
    a.li("t0", SYS_HINT_LEN)
    a.ecall()                      # t0 <- length
    a.mv("s1", "t0")
    a.li("a0", HEAP)
    a.mv("a1", "s1")
    a.li("t0", SYS_HINT_READ)
    a.ecall()
    a.li("s2", HEAP)
    a.add("s3", "s2", "s1")        # end
    a.li("a5", 0)
    a.beq("s2", "s3", "done")
    a.label("loop")
    a.lw("a4", "s2", 0)
    a.add("a5", "a5", "a4")
    a.addi("s2", "s2", 4)
    a.bltu("s2", "s3", "loop")
    a.label("done")
    a.li("s4", res)
    a.sw("a5", "s4", 0)
    _write_pv(a, "s4", 4)
    a.halt(0)
    return a.elf()


def exit_with(code):
    a = Asm()
    a.li("a3", 7)
    a.addi("a3", "a3", 1)
    a.halt(code)
    return a.elf()


def uses_unprovable():
    """retires an instruction that executes but that no chip proves (FENCE; every RV32IM computational
    instruction has a chip)"""
    a = Asm()
    a.li("a3", 77)
    a.word(0x0000000F)      # fence
    a.halt(0)
    return a.elf()


def muldiv():
    """MULH / MULHSU / DIV / DIVU / REM / REMU over sign-critical operands, including RISC-V's special cases:
    division by zero (quotient all ones, remainder = dividend) and the signed overflow -2^31 / -1"""
    vals = [0, 1, 2, 7, 0x7FFFFFFF, 0x80000000, 0x80000001, 0xFFFFFFFF, 0xFFFFFFFE, 0x12345678, 0xDEADBEEF, 0x00010000, 0xFFFF0001, 100, 0xFFFFFF9C]
    sx = lambda v: v - (1 << 32) if v >> 31 else v

    def tdiv(x, y):     # C-style truncating division
        q = abs(x) // abs(y)
        return -q if (x < 0) != (y < 0) else q

    def ref(op, b, c):
        if op == "mulh":
            return ((sx(b) * sx(c)) >> 32) & M32
        if op == "mulhsu":
            return ((sx(b) * c) >> 32) & M32
        if op == "divu":
            return M32 if c == 0 else b // c
        if op == "remu":
            return b if c == 0 else b % c
        if c == 0:
            return M32 if op == "div" else b
        if b == 0x80000000 and c == 0xFFFFFFFF:
            return b if op == "div" else 0
        q = tdiv(sx(b), sx(c))
        return (q if op == "div" else sx(b) - q * sx(c)) & M32

    ops = ["mulh", "mulhsu", "div", "divu", "rem", "remu"]
    pairs = [(b, c) for b in vals for c in vals]
    a = Asm()
    out = a.dword("out", [0] * (len(pairs) * len(ops) + 4))
    a.li("s0", out)
    exp = []
    for b, c in pairs:
        a.li("a3", b)
        a.li("a4", c)
        for op in ops:
            getattr(a, op)("a5", "a3", "a4")
            a.sw("a5", "s0", 0)
            a.addi("s0", "s0", 4)
            exp.append(ref(op, b, c))
    a.li("s1", out)
    _write_pv(a, "s1", 4 * len(exp))
    a.halt(0)
    return a.elf(), b"".join(struct.pack("<I", v) for v in exp)


def shifts():
    """SLL/SRL/SRA (register and immediate forms) over sign-critical values and every class of shift amount"""
    vals = [0x80000001, 0x7FFFFFFF, 0x12345678, 0xFFFFFFFF, 0x00000001, 0xF0F0F0F0]
    amounts = [0, 1, 7, 8, 9, 15, 16, 24, 31, 32 + 3]     # (the last one checks that only the low 5 bits count)
    a = Asm()
    out = a.dword("out", [0] * (len(vals) * len(amounts) * 6 + 4))
    a.li("s0", out)
    exp = []
    sx = lambda v: v - (1 << 32) if v >> 31 else v
    for v in vals:
        a.li("a3", v)
        for sh in amounts:
            a.li("a4", sh)
            k = sh & 31
            res = [("sll", (v << k) & M32), ("srl", v >> k), ("sra", (sx(v) >> k) & M32)]
            for name, want in res:
                getattr(a, name)("a5", "a3", "a4")
                a.sw("a5", "s0", 0)
                a.addi("s0", "s0", 4)
                exp.append(want)
            if sh < 32:
                for name, want in (("slli", (v << k) & M32), ("srli", v >> k), ("srai", (sx(v) >> k) & M32)):
                    getattr(a, name)("a5", "a3", sh)
                    a.sw("a5", "s0", 0)
                    a.addi("s0", "s0", 4)
                    exp.append(want)
    a.li("s1", out)
    _write_pv(a, "s1", 4 * len(exp))
    a.halt(0)
    return a.elf(), b"".join(struct.pack("<I", v) for v in exp)


def traps():
    a = Asm()
    a.li("a3", 0x1000)
    a.lw("a4", "a3", 1)            # misaligned
    a.halt(0)
    return a.elf()


def subword():
    """LB/LBU/LH/LHU at every offset of sign-critical words, SB/SH patches; results are committed."""
    words = [0x80FF7F01, 0x12B4E6F8, 0x00000080]
    a = Asm()
    src = a.dword("src", words)
    dst = a.dword("dst", [0x11223344, 0x55667788, 0x99AABBCC])
    out = a.dword("out", [0] * 64)
    a.li("s0", out)
    a.li("s1", src)
    exp = []
    sx = lambda v, b: (v - (1 << b)) & M32 if v >> (b - 1) else v
    for wi, w in enumerate(words):
        for o in range(4):
            byte = (w >> (8 * o)) & 0xFF
            for name, want in (("lb", sx(byte, 8)), ("lbu", byte)):
                getattr(a, name)("a5", "s1", 4 * wi + o)
                a.sw("a5", "s0", 0)
                a.addi("s0", "s0", 4)
                exp.append(want)
        for o in (0, 2):
            half = (w >> (8 * o)) & 0xFFFF
            for name, want in (("lh", sx(half, 16)), ("lhu", half)):
                getattr(a, name)("a5", "s1", 4 * wi + o)
                a.sw("a5", "s0", 0)
                a.addi("s0", "s0", 4)
                exp.append(want)
    a.li("s2", dst)
    a.li("a3", 0xDEADBEEF)
    mem = [0x11223344, 0x55667788, 0x99AABBCC]
    for (op, off, nbytes) in (("sb", 1, 1), ("sb", 3, 1), ("sh", 4, 2), ("sh", 6, 2), ("sb", 8, 1), ("sh", 10, 2)):
        getattr(a, op)("a3", "s2", off)
        wi, o = off // 4, off % 4
        mask = ((1 << (8 * nbytes)) - 1) << (8 * o)
        mem[wi] = (mem[wi] & ~mask & M32) | ((0xDEADBEEF << (8 * o)) & mask)
    for wi in range(3):
        a.lw("a5", "s2", 4 * wi)
        a.sw("a5", "s0", 0)
        a.addi("s0", "s0", 4)
        exp.append(mem[wi])
    a.li("s1", out)
    _write_pv(a, "s1", 4 * len(exp))
    a.halt(0)
    return a.elf(), b"".join(struct.pack("<I", v) for v in exp)
