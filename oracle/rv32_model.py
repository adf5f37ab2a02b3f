"""ORACLE — TEST INFRASTRUCTURE ONLY (imported by tests/ alone; never by the product path).

An independent restatement, in plain Python, of the two pieces of the path that the C oracle does not cover:
  * the guest machine: ELF32 loader, RV32IM decoder and interpreter with SP1's syscall ABI
    (what stands behind reference src/main.rs:439-442 `client.execute(elf,&stdin).run()`), and
  * K0, the expansion of an execution into the per-chip trace matrices of the rv32 machine
    (SURVEY.md section 8(a) row K0), written from the AIR's own description (tools/airgen/rv32.py: which witness a
    constraint names) and DESIGN.md section 5 — NOT from the product's csrc/rv32.h / rv32_exec.hip, which it is used to
    check: tests compare the product's host expansion AND the device kernel's output with these matrices cell by
    cell, and feed the oracle CPU prover from them.

Only column NAMES are taken from the AIR description (tools.airgen.rv32.build()); every value is computed here.
Slow by design (pure Python): use on guests of at most ~10^5 cycles.
"""
import struct

import numpy as np

P = 2013265921
M32 = 0xFFFFFFFF
ADDR_LIMIT = 0x38000000
REG_BASE = ADDR_LIMIT + (1 << 23)   # the registers are words REG_BASE + 0..31 of the memory argument (above every guest address)
HALT_PC = 1 << 30            # next_pc of a HALT row: no other row can produce it
BAD_PC = 1                   # program-table target of a JAL / branch whose static target lies outside the text
B_AND, B_OR, B_XOR, B_LTU, B_MSB, B_RANGE, B_U16, B_ADDR = 1, 2, 3, 4, 5, 6, 7, 8
SHA_K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
ALU_CODES = dict(sll=1, srl=2, sra=3, mulh=4, mulhsu=5, div=6, divu=7, rem=8, remu=9)
# field / curve precompiles (SP1's syscall numbers as best recalled [EXTERNAL, unverified]): code -> (chip, operation,
# words of the operand at a0, words of the operand at a1)
BLS_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
SECP_P = (1 << 256) - (1 << 32) - 977
BIG_OPS = {0x00010120: ("fp_op", "add", 12, 12), 0x00010121: ("fp_op", "sub", 12, 12), 0x00010122: ("fp_op", "mul", 12, 12),
           0x00010123: ("fp2_op", "add", 24, 24), 0x00010124: ("fp2_op", "sub", 24, 24), 0x00010125: ("fp2_op", "mul", 24, 24),
           0x0001011E: ("bls_g1", "add", 24, 24), 0x0000011F: ("bls_g1", "dbl", 24, 0),
           0x0001010A: ("secp_k1", "add", 16, 16), 0x0000010B: ("secp_k1", "dbl", 16, 0),
           0x0001011D: ("u256_mul", "mul", 8, 16)}


def words_to_int(ws):
    return sum(w << (32 * i) for i, w in enumerate(ws))


def int_to_words(v, n):
    return [(v >> (32 * i)) & M32 for i in range(n)]


def big_op(chip, op, a, b):
    """-> (result words, slope or None) of one call, or raises Trap: the guest machine's semantics of the precompiles.
    Field operands may be any 384-bit numbers (the result is reduced); curve coordinates must be reduced."""
    if chip == "u256_mul":          # x * y mod m, the modulus behind y in memory; m = 0 stands for 2^256
        m = words_to_int(b[8:]) or 1 << 256
        return int_to_words(words_to_int(a) * words_to_int(b[:8]) % m, 8), None
    if chip in ("fp_op", "fp2_op"):
        f = {"add": lambda x, y: (x + y) % BLS_P, "sub": lambda x, y: (x - y) % BLS_P, "mul": lambda x, y: x * y % BLS_P}[op]
        if chip == "fp_op":
            return int_to_words(f(words_to_int(a), words_to_int(b)), 12), None
        x0, x1, y0, y1 = words_to_int(a[:12]), words_to_int(a[12:]), words_to_int(b[:12]), words_to_int(b[12:])
        if op == "mul":
            r0, r1 = (x0 * y0 - x1 * y1) % BLS_P, (x0 * y1 + x1 * y0) % BLS_P
        else:
            r0, r1 = f(x0, y0), f(x1, y1)
        return int_to_words(r0, 12) + int_to_words(r1, 12), None
    p, w = (BLS_P, 12) if chip == "bls_g1" else (SECP_P, 8)
    x1, y1 = words_to_int(a[:w]), words_to_int(a[w:])
    if x1 >= p or y1 >= p:
        raise Trap("curve precompile: coordinate of p not reduced")
    if op == "add":
        x2, y2 = words_to_int(b[:w]), words_to_int(b[w:])
        if x2 >= p or y2 >= p:
            raise Trap("curve precompile: coordinate of q not reduced")
        if x1 == x2:
            raise Trap("ADD precompile with equal abscissae (p = q or p = -q: the guest must handle those)")
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    else:
        if y1 == 0:
            raise Trap("DOUBLE precompile of a point with y = 0")
        x2 = x1
        lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    y3 = (lam * (x1 - x3) - y1) % p
    return int_to_words(x3, w) + int_to_words(y3, w), lam


def sx(v, bits=32):
    return v - (1 << bits) if v >> (bits - 1) else v


def inv(x):
    return pow(x % P, P - 2, P)


# ---------------------------------------------------------------------------------------------- decoding
class Ins:
    __slots__ = ("pc", "word", "kind", "rd", "rs1", "rs2", "imm", "off", "tgt", "tgt_col", "alu", "imm_form", "ok")

    def __init__(self, pc, word):
        self.pc, self.word, self.kind, self.rd, self.rs1, self.rs2 = pc, word, None, 0, 0, 0
        self.imm = self.off = self.tgt = self.tgt_col = self.alu = 0
        self.imm_form, self.ok = False, True


def decode(word, pc):
    """kind in: lui (also auipc, constant folded), jal, jalr, beq.., lb lbu lh lhu lw, sb sh sw, add sub and or xor slt
    sltu mul mulhu, alu (shifts, mulh/mulhsu/div/divu/rem/remu -> other chips), ecall; ok = False: no chip"""
    i = Ins(pc, word)
    op, rd, f3, rs1, rs2, f7 = word & 0x7F, (word >> 7) & 31, (word >> 12) & 7, (word >> 15) & 31, (word >> 20) & 31, word >> 25
    imm_i = sx(word >> 20, 12) & M32
    imm_s = sx(((word >> 25) << 5) | ((word >> 7) & 31), 12) & M32
    imm_b = sx((((word >> 31) & 1) << 12) | (((word >> 7) & 1) << 11) | (((word >> 25) & 63) << 5) | (((word >> 8) & 15) << 1), 13) & M32
    imm_j = sx((((word >> 31) & 1) << 20) | (((word >> 12) & 255) << 12) | (((word >> 20) & 1) << 11) | (((word >> 21) & 1023) << 1), 21) & M32
    if op == 0x37:
        i.kind, i.rd, i.imm = "lui", rd, word & 0xFFFFF000
    elif op == 0x17:
        i.kind, i.rd, i.imm = "lui", rd, (pc + (word & 0xFFFFF000)) & M32
    elif op == 0x6F:
        i.kind, i.rd, i.tgt = "jal", rd, (pc + imm_j) & M32          # (the link value pc + 4 is constrained by the AIR, not tabulated)
    elif op == 0x67 and f3 == 0:
        i.kind, i.rd, i.rs1, i.off = "jalr", rd, rs1, imm_i
    elif op == 0x63 and f3 in (0, 1, 4, 5, 6, 7):
        i.kind = {0: "beq", 1: "bne", 4: "blt", 5: "bge", 6: "bltu", 7: "bgeu"}[f3]
        i.rs1, i.rs2, i.tgt = rs1, rs2, (pc + imm_b) & M32
    elif op == 0x03 and f3 in (0, 1, 2, 4, 5):
        i.kind, i.rd, i.rs1, i.off = {0: "lb", 1: "lh", 2: "lw", 4: "lbu", 5: "lhu"}[f3], rd, rs1, imm_i
    elif op == 0x23 and f3 in (0, 1, 2):
        i.kind, i.rs1, i.rs2, i.off = {0: "sb", 1: "sh", 2: "sw"}[f3], rs1, rs2, imm_s
    elif op == 0x13:
        i.rd, i.rs1, i.imm_form = rd, rs1, True
        if f3 in (0, 2, 3, 4, 6, 7):
            i.kind, i.imm = {0: "add", 2: "slt", 3: "sltu", 4: "xor", 6: "or", 7: "and"}[f3], imm_i
        elif f3 == 1 and f7 == 0:
            i.kind, i.alu, i.imm = "alu", "sll", rs2
        elif f3 == 5 and f7 in (0, 0x20):
            i.kind, i.alu, i.imm = "alu", "srl" if f7 == 0 else "sra", rs2
        else:
            i.ok = False
    elif op == 0x33:
        i.rd, i.rs1, i.rs2 = rd, rs1, rs2
        table = {(0, 0): "add", (0x20, 0): "sub", (0, 2): "slt", (0, 3): "sltu", (0, 4): "xor", (0, 6): "or", (0, 7): "and",
                 (1, 0): "mul", (1, 3): "mulhu"}
        alus = {(0, 1): "sll", (0, 5): "srl", (0x20, 5): "sra", (1, 1): "mulh", (1, 2): "mulhsu", (1, 4): "div", (1, 5): "divu",
                (1, 6): "rem", (1, 7): "remu"}
        if (f7, f3) in table:
            i.kind = table[(f7, f3)]
        elif (f7, f3) in alus:
            i.kind, i.alu = "alu", alus[(f7, f3)]
        else:
            i.ok = False
    elif word == 0x73:
        i.kind, i.rd, i.rs1, i.rs2 = "ecall", 5, 5, 10
    else:
        i.ok = False
    return i


def alu_result(name, b, c):
    s = c & 31
    if name == "sll":
        return (b << s) & M32
    if name == "srl":
        return b >> s
    if name == "sra":
        return (sx(b) >> s) & M32
    if name == "mulh":
        return ((sx(b) * sx(c)) >> 32) & M32
    if name == "mulhsu":
        return ((sx(b) * c) >> 32) & M32
    if name == "divu":
        return M32 if c == 0 else b // c
    if name == "remu":
        return b if c == 0 else b % c
    if c == 0:
        return M32 if name == "div" else b
    if b == 0x80000000 and c == M32:
        return b if name == "div" else 0
    q = abs(sx(b)) // abs(sx(c))
    if (sx(b) < 0) != (sx(c) < 0):
        q = -q
    return (q if name == "div" else sx(b) - q * sx(c)) & M32


# ---------------------------------------------------------------------------------------------- the machine
class Trap(Exception):
    pass


class Row:
    """one retired instruction with everything the cpu chip's row needs"""
    __slots__ = ("ins", "a", "b", "c", "next_pc", "pa", "pb", "pc_", "mem", "maddr", "m_prev", "m_val")


class Run:
    def __init__(self, elf: bytes, stdin=(), log_shard=21, max_cycles=1 << 32):
        self.log_shard = log_shard
        self._load(elf)
        self.stdin = [bytes(b) for b in stdin]
        self.next_input = 0
        self.public_values, self.stdout = b"", b""
        self.committed = {}
        self.exit_code, self.halted, self.error, self.cycles = -1, False, "", 0
        self.unsupported = False
        # state: value and (shard, clk) of the last access per register / memory word
        self.reg = [0] * 32
        self.reg_t = [(0, 0)] * 32
        self.mem = dict(self.image_mem)         # addr -> value
        self.mem_t = {}                         # addr -> (shard, clk) of the last access
        self.first_val = {}                     # non-image words: value at their first access
        self.hinted = set()
        self.shards = []
        try:
            self._run(max_cycles)
        except Trap as e:
            self.error = str(e)

    # ---- ELF
    def _load(self, elf):
        if len(elf) < 52 or elf[:4] != b"\x7fELF" or elf[4] != 1 or elf[5] != 1:
            raise ValueError("not an ELF32 little-endian file")
        (machine,) = struct.unpack_from("<H", elf, 18)
        if machine != 243:
            raise ValueError("not RISC-V")
        self.entry, phoff = struct.unpack_from("<II", elf, 24)
        phentsize, phnum = struct.unpack_from("<HH", elf, 42)
        self.image_mem, self.text = {}, {}
        self.text_base = None
        for k in range(phnum):
            typ, off, vaddr, _, filesz, memsz, flags, _ = struct.unpack_from("<8I", elf, phoff + k * phentsize)
            if typ != 1:
                continue
            blob = elf[off:off + filesz] + b"\0" * (memsz - filesz)
            blob += b"\0" * (-len(blob) % 4)
            for j in range(0, memsz, 4):
                self.image_mem[vaddr + j] = struct.unpack_from("<I", blob, j)[0]
            if flags & 1:
                self.text_base = vaddr
                self.n_instr = (filesz + 3) // 4
                for j in range(self.n_instr):
                    self.text[vaddr + 4 * j] = decode(self.image_mem[vaddr + 4 * j], vaddr + 4 * j)
        # the target COLUMN of a JAL / branch: the target itself when it is an instruction of the text, BAD_PC otherwise
        # (the machine traps there; a raw 32-bit value could alias a valid pc mod p)
        for i in self.text.values():
            if i.kind in ("jal", "beq", "bne", "blt", "bge", "bltu", "bgeu"):
                i.tgt_col = i.tgt if i.tgt in self.text else BAD_PC
        # the program table lists the instructions that have a chip, in address order
        self.provable = [i for _, i in sorted(self.text.items()) if i.ok]

    # ---- memory argument bookkeeping
    def _touch_reg(self, r, t):
        prev = self.reg_t[r]
        self.reg_t[r] = t
        return prev

    def _mem_word(self, addr):
        if addr not in self.mem:
            self.mem[addr] = 0
        if addr not in self.image_mem and addr not in self.first_val:
            self.first_val[addr] = self.mem[addr]
        return self.mem[addr]

    def _run(self, max_cycles):
        pc, shard, i_in = self.entry, 1, 0
        size = 1 << self.log_shard
        cur = dict(index=1, start_pc=pc, rows=[], alu=[], sha_ext=[], sha_cmp=[], big=[])
        while True:
            if self.cycles >= max_cycles:
                raise Trap("cycle limit reached before HALT")
            if i_in == size:
                cur["next_pc"] = pc
                self.shards.append(cur)
                shard, i_in = shard + 1, 0
                cur = dict(index=shard, start_pc=pc, rows=[], alu=[], sha_ext=[], sha_cmp=[], big=[])
            ins = self.text.get(pc)
            if ins is None:
                raise Trap("pc outside text at pc 0x%x" % pc)
            clk = 4 * (i_in + 1)
            if not ins.ok:
                if ins.word & 0x7F != 0x0F:
                    raise Trap("illegal instruction at pc 0x%x" % pc)
                self.unsupported = True
                pc, i_in, self.cycles = pc + 4, i_in + 1, self.cycles + 1
                continue
            row = Row()
            row.ins, row.pa, row.pb, row.pc_, row.mem = ins, None, None, None, None
            k = ins.kind
            # ports in time order: rs2 at clk, rs1 at clk + 1, memory at clk + 2, rd at clk + 3
            reads_rs2 = (k in ("beq", "bne", "blt", "bge", "bltu", "bgeu", "sb", "sh", "sw", "ecall")
                         or (not ins.imm_form and k in ("add", "sub", "and", "or", "xor", "slt", "sltu", "mul", "mulhu", "alu")))
            reads_rs1 = k not in ("lui", "jal")
            b = c = a = 0
            if reads_rs2:
                c = self.reg[ins.rs2]
                row.pc_ = self._touch_reg(ins.rs2, (shard, clk))
            if ins.imm_form:
                c = ins.imm
            if reads_rs1:
                b = self.reg[ins.rs1]
                row.pb = self._touch_reg(ins.rs1, (shard, clk + 1))
            nxt = (pc + 4) & M32
            if k == "add":
                a = (b + c) & M32
            elif k == "sub":
                a = (b - c) & M32
            elif k == "and":
                a = b & c
            elif k == "or":
                a = b | c
            elif k == "xor":
                a = b ^ c
            elif k == "slt":
                a = int(sx(b) < sx(c))
            elif k == "sltu":
                a = int(b < c)
            elif k == "mul":
                a = (b * c) & M32
            elif k == "mulhu":
                a = (b * c) >> 32
            elif k == "lui":
                a = ins.imm
            elif k == "jal":
                a, nxt = (pc + 4) & M32, ins.tgt
            elif k == "jalr":
                t = (b + ins.off) & M32
                if t >= ADDR_LIMIT:
                    raise Trap("jump target out of range at pc 0x%x" % pc)
                a, nxt = (pc + 4) & M32, t & ~1
            elif k in ("beq", "bne", "blt", "bge", "bltu", "bgeu"):
                taken = {"beq": b == c, "bne": b != c, "blt": sx(b) < sx(c), "bge": sx(b) >= sx(c), "bltu": b < c, "bgeu": b >= c}[k]
                if taken:
                    nxt = ins.tgt
            elif k == "alu":
                a = alu_result(ins.alu, b, c)
                cur["alu"].append((ins.alu, a, b, c))
            elif k in ("lb", "lbu", "lh", "lhu", "lw", "sb", "sh", "sw"):
                addr = (b + ins.off) & M32
                if addr < 32 or addr >= ADDR_LIMIT:
                    raise Trap("memory access out of range at pc 0x%x" % pc)
                if (k in ("lw", "sw") and addr & 3) or (k in ("lh", "lhu", "sh") and addr & 1):
                    raise Trap("misaligned %s access at pc 0x%x" % ("word" if k in ("lw", "sw") else "halfword", pc))
                wa, sh8 = addr & ~3, 8 * (addr & 3)
                before = self._mem_word(wa)
                after = before
                if k == "lw":
                    a = before
                elif k == "lb":
                    a = sx((before >> sh8) & 0xFF, 8) & M32
                elif k == "lbu":
                    a = (before >> sh8) & 0xFF
                elif k == "lh":
                    a = sx((before >> sh8) & 0xFFFF, 16) & M32
                elif k == "lhu":
                    a = (before >> sh8) & 0xFFFF
                elif k == "sw":
                    after = c
                elif k == "sb":
                    after = (before & ~(0xFF << sh8) & M32) | ((c & 0xFF) << sh8)
                else:
                    after = (before & ~(0xFFFF << sh8) & M32) | ((c & 0xFFFF) << sh8)
                self.mem[wa] = after
                row.mem = self.mem_t.get(wa, (0, 0))
                self.mem_t[wa] = (shard, clk + 2)
                row.maddr, row.m_prev, row.m_val = addr, before, after
            elif k == "ecall":
                a = b
                a1, a2 = self.reg[11], self.reg[12]
                if b == 0x00:
                    if c >> 24:
                        raise Trap("HALT with an exit code of 2^24 or more at pc 0x%x" % pc)
                    self.halted, self.exit_code, nxt = True, sx(c), HALT_PC
                elif b == 0x02:
                    if a1 + a2 > ADDR_LIMIT:
                        raise Trap("WRITE buffer out of range at pc 0x%x" % pc)
                    data = bytes((self.mem.get((a1 + j) & ~3, 0) >> (8 * ((a1 + j) & 3))) & 0xFF if a1 + j >= 32 else 0 for j in range(a2))
                    if c == 3:
                        self.public_values += data
                    else:
                        self.stdout += data
                elif b == 0x10:
                    row.mem = self._touch_reg(11, (shard, clk + 2))
                    row.maddr, row.m_prev, row.m_val = 11, a1, a1
                    if c < 8:
                        self.committed[c] = a1
                elif b == 0x00300105:
                    # SHA_EXTEND(a0 = w): w[16..63] of the SHA-256 message schedule in place; a1 (read through the port like
                    # COMMIT's) must be 0; every word of the array is accessed once, at (shard, clk + 2)
                    if a1 != 0:
                        raise Trap("SHA_EXTEND with a1 != 0 at pc 0x%x" % pc)
                    if c % 4 or c < 32 or c + 256 > ADDR_LIMIT:
                        raise Trap("SHA_EXTEND pointer misaligned or out of range at pc 0x%x" % pc)
                    row.mem = self._touch_reg(11, (shard, clk + 2))
                    row.maddr, row.m_prev, row.m_val = 11, a1, a1
                    rotr = lambda v, n: ((v >> n) | (v << (32 - n))) & M32
                    w, old, prev = [], [], []
                    for kk in range(64):
                        wa = c + 4 * kk
                        before = self._mem_word(wa)
                        if kk < 16:
                            w.append(before)
                        else:
                            x, y = w[kk - 15], w[kk - 2]
                            s0 = rotr(x, 7) ^ rotr(x, 18) ^ (x >> 3)
                            s1 = rotr(y, 17) ^ rotr(y, 19) ^ (y >> 10)
                            w.append((s1 + w[kk - 7] + s0 + w[kk - 16]) & M32)
                        old.append(before)
                        prev.append(self.mem_t.get(wa, (0, 0)))
                        self.mem[wa] = w[kk]
                        self.mem_t[wa] = (shard, clk + 2)
                    cur["sha_ext"].append(dict(clk=clk, ptr=c, w=w, old=old, prev=prev))
                elif b == 0x00010106:
                    # SHA_COMPRESS(a0 = w, a1 = state): 64 rounds of the SHA-256 compression function over the schedule
                    # words, state += result in place; all reads at (shard, clk + 2), the eight write-backs at clk + 3
                    if c % 4 or c < 32 or c + 256 > ADDR_LIMIT or a1 % 4 or a1 < 32 or a1 + 32 > ADDR_LIMIT:
                        raise Trap("SHA_COMPRESS pointer misaligned or out of range at pc 0x%x" % pc)
                    if a1 + 32 > c and c + 256 > a1:
                        raise Trap("SHA_COMPRESS arrays overlap at pc 0x%x" % pc)
                    row.mem = self._touch_reg(11, (shard, clk + 2))
                    row.maddr, row.m_prev, row.m_val = 11, a1, a1
                    rotr = lambda v, n: ((v >> n) | (v << (32 - n))) & M32
                    hs, hprev, ws, wprev = [], [], [], []
                    for kk in range(8):
                        hs.append(self._mem_word(a1 + 4 * kk))
                        hprev.append(self.mem_t.get(a1 + 4 * kk, (0, 0)))
                        self.mem_t[a1 + 4 * kk] = (shard, clk + 2)
                    for kk in range(64):
                        ws.append(self._mem_word(c + 4 * kk))
                        wprev.append(self.mem_t.get(c + 4 * kk, (0, 0)))
                        self.mem_t[c + 4 * kk] = (shard, clk + 2)
                    v = list(hs)
                    for i in range(64):
                        S1 = rotr(v[4], 6) ^ rotr(v[4], 11) ^ rotr(v[4], 25)
                        chv = (v[4] & v[5]) ^ (~v[4] & v[6] & M32)
                        t1 = (v[7] + S1 + chv + SHA_K[i] + ws[i]) & M32
                        S0 = rotr(v[0], 2) ^ rotr(v[0], 13) ^ rotr(v[0], 22)
                        mj = (v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2])
                        v = [(t1 + S0 + mj) & M32, v[0], v[1], v[2], (v[3] + t1) & M32, v[4], v[5], v[6]]
                    for kk in range(8):
                        self.mem[a1 + 4 * kk] = (hs[kk] + v[kk]) & M32
                        self.mem_t[a1 + 4 * kk] = (shard, clk + 3)
                    cur["sha_cmp"].append(dict(clk=clk, w_ptr=c, h_ptr=a1, w=ws, hs=hs, wprev=wprev, hprev=hprev))
                elif b == 0x1A:
                    pass
                elif b == 0xF0:
                    a = len(self.stdin[self.next_input]) if self.next_input < len(self.stdin) else 0
                elif b == 0xF1:
                    if self.next_input >= len(self.stdin):
                        raise Trap("HINT_READ with no input left at pc 0x%x" % pc)
                    buf = self.stdin[self.next_input]
                    self.next_input += 1
                    if a1 != len(buf):
                        raise Trap("HINT_READ length mismatch at pc 0x%x" % pc)
                    if c % 4 or c < 32 or c + a1 > ADDR_LIMIT:
                        raise Trap("HINT_READ pointer misaligned or out of range at pc 0x%x" % pc)
                    padded = buf + b"\0" * (-len(buf) % 4)
                    for j in range(0, len(buf), 4):
                        if (c + j) in self.image_mem or (c + j) in self.mem_t:
                            raise Trap("HINT_READ into the program image or into memory that was already accessed at pc 0x%x" % pc)
                        self.mem[c + j] = struct.unpack_from("<I", padded, j)[0]
                elif b in BIG_OPS:
                    # field / curve precompiles: a0 -> operand replaced by the result (read and written at clk + 3),
                    # a1 -> second operand (read at clk + 2; 0 for DOUBLE)
                    chip_, op_, wa, wb = BIG_OPS[b]
                    if c % 4 or c < 32 or c + 4 * wa > ADDR_LIMIT or (wb and (a1 % 4 or a1 < 32 or a1 + 4 * wb > ADDR_LIMIT)):
                        raise Trap("precompile operand pointer misaligned or out of range at pc 0x%x" % pc)
                    if not wb and a1 != 0:
                        raise Trap("DOUBLE precompile with a1 != 0 at pc 0x%x" % pc)
                    va = [self.mem.get(c + 4 * j, 0) for j in range(wa)]
                    vb = [self.mem.get(a1 + 4 * j, 0) for j in range(wb)]
                    try:
                        res, lam = big_op(chip_, op_, va, vb)
                    except Trap as e:
                        raise Trap("%s at pc 0x%x" % (e, pc))
                    row.mem = self._touch_reg(11, (shard, clk + 2))
                    row.maddr, row.m_prev, row.m_val = 11, a1, a1
                    bprev, aprev = [], []
                    for j in range(wb):
                        self._mem_word(a1 + 4 * j)
                        bprev.append(self.mem_t.get(a1 + 4 * j, (0, 0)))
                        self.mem_t[a1 + 4 * j] = (shard, clk + 2)
                    for j in range(wa):
                        self._mem_word(c + 4 * j)
                        aprev.append(self.mem_t.get(c + 4 * j, (0, 0)))
                        self.mem[c + 4 * j] = res[j]
                        self.mem_t[c + 4 * j] = (shard, clk + 3)
                    cur["big"].append(dict(chip=chip_, op=op_, clk=clk, a_ptr=c, b_ptr=a1, a=va, b=vb, r=res, lam=lam, aprev=aprev, bprev=bprev))
                else:
                    raise Trap("unknown syscall at pc 0x%x" % pc)
            else:
                raise AssertionError(k)
            if ins.rd != 0:
                row.pa = (self.reg[ins.rd],) + self._touch_reg(ins.rd, (shard, clk + 3))
                self.reg[ins.rd] = a
            row.a, row.b, row.c, row.next_pc = a, b, c, nxt
            cur["rows"].append(row)
            pc, i_in, self.cycles = nxt, i_in + 1, self.cycles + 1
            if self.halted:
                cur["next_pc"] = HALT_PC
                self.shards.append(cur)
                return

    # ---- the sorted initial / final table of the last shard
    def mem_rows(self):
        rows = []
        for r in range(32):
            t = self.reg_t[r]
            rows.append((REG_BASE + r, 0, self.reg[r] if t != (0, 0) else 0, t, 1))
        for addr, v in self.image_mem.items():
            t = self.mem_t.get(addr, (0, 0))
            rows.append((addr, v, self.mem[addr] if t != (0, 0) else v, t, 1))
        for addr, v in self.first_val.items():
            rows.append((addr, v, self.mem[addr], self.mem_t[addr], 0))
        rows.sort()
        return rows


# ---------------------------------------------------------------------------------------------- K0: trace matrices
def _chips():
    from tools.airgen import rv32

    m = rv32.build()
    return {c.name: (i, c) for i, c in enumerate(m.chips)}, rv32


def byts(w):
    return [(w >> (8 * i)) & 0xFF for i in range(4)]


class Lookups:
    def __init__(self):
        self.byte = np.zeros((8, 65536), np.int64)

    def add(self, op, b, c=0):
        self.byte[op - 1, (b << 8) | c if op != B_U16 else b] += 1


def _col_setter(chip, mat, row):
    idx = {n: i for i, n in enumerate(chip.main_names)}

    def put(name, v):
        mat[idx[name], row] = v % P

    def putv(name, vals):
        for i, v in enumerate(vals):
            mat[idx[f"{name}[{i}]"], row] = v % P

    return put, putv


def log2ceil(n):
    l = 0
    while (1 << l) < n:
        l += 1
    return l


def traces(run: Run, pos: int):
    """-> (list of dict(chip_id, log_n, main [w][N], prep [w][N]) sorted by chip id, public values) for shard `pos`"""
    chips, air = _chips()
    sh = run.shards[pos]
    shard = sh["index"]
    last = pos + 1 == len(run.shards)
    lk = Lookups()
    prog_mult = {}
    out = {}
    FLAGS = air.FLAGS

    def gap(prev, now_clk):
        """(same-shard flag, 16-bit low limb, 8-bit high limb) of the timestamp gap to the previous access"""
        psh, pclk = prev
        d = now_clk - pclk - 1 if psh == shard else shard - psh - 1
        assert 0 <= d < 1 << 24
        return int(psh == shard), d & 0xFFFF, d >> 16

    # ---- cpu
    cid, chip = chips["cpu"]
    n = 1 << log2ceil(len(sh["rows"]))
    cpu = np.zeros((chip.main_width, n), np.int64)
    # selector column of each instruction form; forms that share one carry a value column beside it
    fam_of = dict(add="is_add", sub="is_sub", slt="is_set", sltu="is_set", mul="is_mul", mulhu="is_mulhu", lui="is_lui", jal="is_jal", jalr="is_jalr",
                  beq="is_beq", bne="is_bne", blt="is_brlt", bge="is_brge", bltu="is_brlt", bgeu="is_brge", lw="is_lw", sw="is_sw", ecall="is_ecall",
                  lb="is_lb", lbu="is_lbu", lh="is_lh", lhu="is_lhu", sb="is_sb", sh="is_sh", alu="is_alu")
    fam_of.update({"and": "is_bit", "or": "is_bit", "xor": "is_bit"})
    bit_op_of = {"and": B_AND, "or": B_OR, "xor": B_XOR}
    signed_forms = ("slt", "blt", "bge")
    for r, row in enumerate(sh["rows"]):
        ins, a, b, c = row.ins, row.a, row.b, row.c
        put, putv = _col_setter(chip, cpu, r)
        clk = 4 * (r + 1)
        k = ins.kind
        put("clk", clk); put("pc", ins.pc); put("next_pc", row.next_pc)
        put("rd", ins.rd); put("rs1", ins.rs1); put("rs2", ins.rs2)
        putv("imm", byts(ins.imm | ins.off))          # one immediate field: operand / constant or address offset, never both
        put("aux", ins.tgt_col + (ALU_CODES[ins.alu] if k == "alu" else 0))
        put("bit_op", bit_op_of.get(k, 0)); put("cmp_signed", int(k in signed_forms))
        putv("a", byts(a)); putv("b", byts(b)); putv("c", byts(c))
        put(fam_of[k], 1)
        put("rd_en", int(ins.rd != 0)); put("imm_c", int(ins.imm_form))
        prog_mult[ins.pc] = prog_mult.get(ins.pc, 0) + 1
        hi = dict(pa=0, pb=0, pc=0, m=0)
        for port, prev, t in (("pc", row.pc_, clk), ("pb", row.pb, clk + 1)):
            if prev is not None:
                same, lo, h = gap(prev, t)
                put(f"{port}_ts", prev[1]); put(f"{port}_sh", prev[0]); put(f"{port}_same", same); put(f"{port}_lo", lo)
                hi[port] = h
                lk.add(B_U16, lo)
        if row.pa is not None:
            same, lo, h = gap(row.pa[1:], clk + 3)
            putv("pa_prev", byts(row.pa[0]))
            put("pa_ts", row.pa[2]); put("pa_sh", row.pa[1]); put("pa_same", same); put("pa_lo", lo)
            hi["pa"] = h
            lk.add(B_U16, lo)
        U = [0] * air.UNION_W

        def mem_port(prev_val, new_val, prev_t):
            same, lo, h = gap(prev_t, clk + 2)
            U[9:13], U[13:17] = byts(new_val), byts(prev_val)
            U[17], U[8], U[18], U[19], U[20] = prev_t[1], lo, h, prev_t[0], same
            hi["m"] = h
            lk.add(B_U16, lo)

        if k in ("add", "sub"):
            x, cy = (b if k == "add" else a), 0
            for i in range(4):
                cy = (byts(x)[i] + byts(c)[i] + cy) >> 8
                U[i] = cy
        elif k in ("and", "or", "xor"):
            for i in range(4):
                U[11 + i], U[4 + i], U[19 + i] = byts(a)[i], byts(b)[i], byts(c)[i]
                lk.add({"and": B_AND, "or": B_OR, "xor": B_XOR}[k], byts(b)[i], byts(c)[i])
        elif k in ("slt", "sltu", "beq", "bne", "blt", "bge", "bltu", "bgeu"):
            signed = k in ("slt", "blt", "bge")
            bb, cc = byts(b), byts(c)
            if signed:
                U[24], U[25], U[9], U[21] = bb[3], bb[3] >> 7, cc[3], cc[3] >> 7
                lk.add(B_MSB, bb[3]); lk.add(B_MSB, cc[3])
                bb[3] ^= 0x80; cc[3] ^= 0x80
            top = next((i for i in (3, 2, 1, 0) if bb[i] != cc[i]), None)
            bc, ccv = (bb[top], cc[top]) if top is not None else (0, 0)
            if top is not None:
                U[top] = 1
                U[4] = inv(bc - ccv)
            U[10], U[20], U[19] = bc, ccv, int(bc < ccv)
            lk.add(B_LTU, bc, ccv)
        elif k in ("mul", "mulhu"):
            prod = b * c
            pb8 = [(prod >> (8 * i)) & 0xFF for i in range(8)]
            carries, acc = [], 0
            for kk in range(8):
                acc += sum(byts(b)[i] * byts(c)[kk - i] for i in range(4) if 0 <= kk - i < 4)
                assert acc & 0xFF == pb8[kk]
                acc >>= 8
                carries.append(acc)
            other = pb8[4:] if k == "mul" else pb8[:4]
            U[0:4] = other
            U[4:11] = carries[:7]
            for v in carries[:7]:
                lk.add(B_U16, v)
            lk.add(B_RANGE, other[1], other[2]); lk.add(B_RANGE, other[0], other[3])     # (the address adder's two lookups)
        elif k in ("lw", "sw", "lb", "lbu", "lh", "lhu", "sb", "sh", "jalr"):
            total, cy = (b + ins.off) & M32, 0
            for i in range(4):
                t = byts(b)[i] + byts(ins.off)[i] + cy
                U[i], cy = t & 0xFF, t >> 8
                U[4 + i] = cy
            s = byts(total)
            # byte ranges of the sum, top byte below 0x38 and byte offset s0 & 3 (one-hot in u[21..23]) in two lookups
            lk.add(B_RANGE, s[1], s[2]); lk.add(B_ADDR, s[0], s[3])
            if total & 3:
                U[20 + (total & 3)] = 1
            if k == "jalr":
                U[8] = total & 1
                U[19], U[10], U[20] = 1, byts(a)[3], 0x78          # the link value's top byte is below 0x78 (comparator slot)
                lk.add(B_LTU, byts(a)[3], 0x78)
            else:
                mem_port(row.m_prev, row.m_val, row.mem)
                if k in ("lb", "lh"):
                    sb = byts(a)[0] if k == "lb" else byts(a)[1]
                    U[24], U[25] = sb, sb >> 7
                    lk.add(B_MSB, sb)
        elif k == "jal":
            U[19], U[10], U[20] = 1, byts(a)[3], 0x78
            lk.add(B_LTU, byts(a)[3], 0x78)
        elif k == "ecall":
            sid = byts(b)[0] + 256 * byts(b)[1] + 65536 * byts(b)[2] + (byts(b)[3] << 22)   # the id as the AIR compares it
            U[24] = int(sid == 0xF0)                               # HINT_LEN: the one call whose return value is advice
            U[25] = 0 if sid == 0xF0 else inv(sid - 0xF0)
            U[4] = int(sid == 0)
            U[5] = inv(sid) if sid else 0
            is_commit = int(sid == 0x10)
            U[6] = is_commit
            U[7] = 0 if is_commit else inv(sid - 0x10)
            # byte 1 of the code = 1: the call has a precompile chip; COMMIT and precompile rows read a1 through the memory
            # port and send (t0 bytes, a0 bytes, a1 bytes, u_clk, u_sh) on the sys bus
            is_pre = int(byts(b)[1] == 1)
            U[1] = is_pre
            U[2] = 0 if is_pre else inv(byts(b)[1] - 1)
            put("sys_m", int(is_commit or is_pre))
            if is_commit or is_pre:
                U[3], U[21] = (clk, shard) if is_pre else (0, 0)
                # the port's address expression u0 + 256 u1 + 65536 u2 + 2^24 u3 - (u21 + 2 u22 + 3 u23) is register a1's word
                U[0] = (REG_BASE + 11 - 256 * U[1] - 65536 * U[2] - (1 << 24) * U[3] + U[21]) % P
                mem_port(row.m_prev, row.m_val, row.mem)
        putv("u", U)
        put("pb_hi", hi["pb"]); put("pc_hi", hi["pc"]); put("pa_hi", hi["pa"])
        lk.add(B_RANGE, hi["pb"], hi["pc"]); lk.add(B_RANGE, hi["pa"], hi["m"])
        if k in ("add", "sub", "mul", "mulhu", "ecall", "jal", "jalr"):
            lk.add(B_RANGE, byts(a)[0], byts(a)[1]); lk.add(B_RANGE, byts(a)[2], byts(a)[3])
    out["cpu"] = cpu

    # ---- shift
    ev = [e for e in sh["alu"] if e[0] in ("sll", "srl", "sra")]
    if ev:
        cid, chip = chips["shift"]
        mat = np.zeros((chip.main_width, 1 << log2ceil(len(ev))), np.int64)
        for r, (name, a, b, c) in enumerate(ev):
            put, putv = _col_setter(chip, mat, r)
            s5 = c & 31
            q, rb = s5 >> 3, s5 & 7
            m, mi = 1 << rb, 1 << (8 - rb)
            left = name == "sll"
            sgn = (b >> 31) if name == "sra" else 0
            put("is_real", 1); put("is_" + name, 1); put("sh", s5); put("sgn", sgn)
            qv = [0, 0, 0]
            if q:
                qv[q - 1] = 1
            putv("q", qv)
            putv("r", [int(i == rb) for i in range(8)])
            putv("a", byts(a)); putv("b", byts(b)); putv("c", byts(c))
            lo, hi4 = [], []
            for i in range(4):
                if left:
                    pr = byts(b)[i] * m
                    lo.append(pr & 0xFF); hi4.append(pr >> 8)
                else:
                    hi4.append(byts(b)[i] >> rb); lo.append(byts(b)[i] & (m - 1))
            t = [(lo[i] + (hi4[i - 1] if i else 0)) if left else (hi4[i] + (lo[i + 1] * mi if i < 3 else sgn * (256 - mi))) for i in range(4)]
            putv("lo", lo); putv("hi", hi4); putv("t", t)
            lk.add(B_AND, byts(c)[0], 31)
            lk.add(B_RANGE, lo[0], lo[1]); lk.add(B_RANGE, lo[2], lo[3]); lk.add(B_RANGE, hi4[0], hi4[1]); lk.add(B_RANGE, hi4[2], hi4[3])
            if name == "sra":
                lk.add(B_MSB, byts(b)[3])
            if not left:
                for i in range(4):
                    lk.add(B_LTU, lo[i], m)
        out["shift"] = mat

    # ---- muldiv
    ev = [e for e in sh["alu"] if e[0] not in ("sll", "srl", "sra")]
    if ev:
        cid, chip = chips["muldiv"]
        mat = np.zeros((chip.main_width, 1 << log2ceil(len(ev))), np.int64)
        for r, (name, a, b, c) in enumerate(ev):
            put, putv = _col_setter(chip, mat, r)
            is_mul, is_sdr = name in ("mulh", "mulhsu"), name in ("div", "rem")
            is_dr = not is_mul
            c0, ovf = is_dr and c == 0, is_sdr and b == 0x80000000 and c == M32
            if is_mul:
                q, rem = b, 0
            elif c0:
                q, rem = M32, b
            elif ovf:
                q, rem = b, 0
            elif is_sdr:
                q, rem = alu_result("div", b, c), alu_result("rem", b, c)
            else:
                q, rem = b // c, b % c
            mx, my, mr, mb = q >> 31, c >> 31, rem >> 31, b >> 31
            sxx = mx if (is_mul or is_sdr) else 0
            syy = my if (name == "mulh" or is_sdr) else 0
            srr, sbb = (mr, mb) if is_sdr else (0, 0)
            put("is_real", 1); put("is_" + name, 1)
            putv("a", byts(a)); putv("b", byts(b)); putv("c", byts(c)); putv("q", byts(q)); putv("r", byts(rem))
            for nm, v in (("mx", mx), ("my", my), ("mr", mr), ("mb", mb), ("sx", sxx), ("sy", syy), ("sr", srr), ("sb", sbb)):
                put(nm, v)
            for v in (q, c, rem, b):
                lk.add(B_MSB, byts(v)[3])
            prod = q * c
            pb8 = [(prod >> (8 * i)) & 0xFF for i in range(8)]
            carries, acc = [], 0
            for kk in range(8):
                acc += sum(byts(q)[i] * byts(c)[kk - i] for i in range(4) if 0 <= kk - i < 4)
                acc >>= 8
                carries.append(acc)
                lk.add(B_U16, acc)
            putv("prod", pb8); putv("mcy", carries)
            for kk in range(4):
                lk.add(B_RANGE, pb8[2 * kk], pb8[2 * kk + 1])
            h, bw, borrow = [], [], 0
            for i in range(4):
                t = pb8[4 + i] - sxx * byts(c)[i] - syy * byts(q)[i] - borrow
                borrow = 0
                while t < 0:
                    t += 256
                    borrow += 1
                h.append(t); bw.append(borrow)
            putv("h", h); putv("bw", bw)
            lk.add(B_RANGE, h[0], h[1]); lk.add(B_RANGE, h[2], h[3])
            lk.add(B_RANGE, byts(q)[0], byts(q)[1]); lk.add(B_RANGE, byts(q)[2], byts(q)[3])
            lk.add(B_RANGE, byts(rem)[0], byts(rem)[1]); lk.add(B_RANGE, byts(rem)[2], byts(rem)[3])
            dl0 = dl1 = 0
            ea = 1
            eb = 0
            if is_dr:
                put("is_c0", int(c0)); put("is_ovf", int(ovf))
                csum = sum(byts(c))
                put("cinv", inv(csum) if csum else 0)
                if not ovf:
                    Pl = [pb8[0] | pb8[1] << 8, pb8[2] | pb8[3] << 8, h[0] | h[1] << 8, h[2] | h[3] << 8]
                    Rl = [rem & 0xFFFF, rem >> 16, 65535 * srr, 65535 * srr]
                    cy, dcy = 0, []
                    for kk in range(4):
                        cy = (Pl[kk] + Rl[kk] + cy) >> 16
                        dcy.append(cy)
                    putv("dcy", dcy)
                if not c0:
                    sc_, sr_ = 1 - 2 * syy, 1 - 2 * srr
                    t0 = sc_ * (c & 0xFFFF) - sr_ * (rem & 0xFFFF) - 1
                    e0 = -(t0 // 65536)
                    dl0 = t0 + 65536 * e0
                    dl1 = sc_ * (c >> 16) - sr_ * (rem >> 16) + 65536 * (syy - srr) - e0
                    ea, eb = (e0 + 1) & 1, (e0 + 1) >> 1
            put("ea", ea); put("eb", eb)
            putv("dl", [dl0, dl1])
            lk.add(B_U16, dl0); lk.add(B_U16, dl1)
        out["muldiv"] = mat

    # ---- mem_init (last shard only)
    if last:
        cid, chip = chips["mem_init"]
        rows = run.mem_rows()
        mat = np.zeros((chip.main_width, 1 << log2ceil(len(rows))), np.int64)
        prev = None
        for r, (addr, v, f, t, is_img) in enumerate(rows):
            put, putv = _col_setter(chip, mat, r)
            d = addr - prev - 1 if prev is not None else 0
            putv("ab", byts(addr)); putv("v", byts(v)); putv("f", byts(f)); putv("d", byts(d))
            put("fts", t[1]); put("fsh", t[0]); put("is_img", is_img); put("is_real", 1)
            for w, top in ((addr, (ADDR_LIMIT >> 24) + 1), (d, (ADDR_LIMIT >> 24) + 1)):      # (+ 1: the registers at REG_BASE)
                lk.add(B_RANGE, byts(w)[0], byts(w)[1]); lk.add(B_RANGE, byts(w)[2], byts(w)[3]); lk.add(B_LTU, byts(w)[3], top)
            if not is_img:
                lk.add(B_RANGE, byts(v)[0], byts(v)[1]); lk.add(B_RANGE, byts(v)[2], byts(v)[3])
            prev = addr
        out["mem_init"] = mat

    # ---- sha_extend: 64 rows per call (rows 0..15 read w[j] into the window, rows 16..63 compute and write w[j])
    if sh["sha_ext"]:
        cid, chip = chips["sha_extend"]
        mat = np.zeros((chip.main_width, 1 << log2ceil(64 * len(sh["sha_ext"]))), np.int64)
        rotr = lambda v, n: ((v >> n) | (v << (32 - n))) & M32
        for e, ev in enumerate(sh["sha_ext"]):
            w = ev["w"]
            for j in range(64):
                put, putv = _col_setter(chip, mat, 64 * e + j)
                put("is_real", 1); put("is_first", int(j == 0)); put("is_last", int(j == 63))
                put("is_load", int(j < 16)); put("is_e", int(j == 15)); put("j", j)
                put("j_inv", inv(j - 63) if j != 63 else 0)
                put("clk", ev["clk"]); putv("p", byts(ev["ptr"]))
                win = [w[j - 16 + k] if j - 16 + k >= 0 else 0 for k in range(16)]
                for k in range(16):
                    putv(f"w{k}", byts(win[k]))
                x, y = win[1], win[14]
                putv("xb", [(x >> t) & 1 for t in range(32)]); putv("yb", [(y >> t) & 1 for t in range(32)])
                s0 = rotr(x, 7) ^ rotr(x, 18) ^ (x >> 3)
                s1 = rotr(y, 17) ^ rotr(y, 19) ^ (y >> 10)
                putv("s0", [s0 & 0xFFFF, s0 >> 16]); putv("s1", [s1 & 0xFFFF, s1 >> 16])
                putv("nw", byts(w[j])); putv("old", byts(ev["old"][j]))
                if j >= 16:
                    lo = (win[0] & 0xFFFF) + (s0 & 0xFFFF) + (win[9] & 0xFFFF) + (s1 & 0xFFFF)
                    hi_ = (win[0] >> 16) + (s0 >> 16) + (win[9] >> 16) + (s1 >> 16) + (lo >> 16)
                    assert ((hi_ & 0xFFFF) << 16 | (lo & 0xFFFF)) == w[j]
                    putv("cy", [(lo >> 16) & 1, lo >> 17, (hi_ >> 16) & 1, hi_ >> 17])
                    lk.add(B_RANGE, byts(w[j])[0], byts(w[j])[1]); lk.add(B_RANGE, byts(w[j])[2], byts(w[j])[3])
                psh, pts = ev["prev"][j]
                d = ev["clk"] + 2 - pts - 1 if psh == sh["index"] else sh["index"] - psh - 1
                put("m_sh", psh); put("m_ts", pts); put("m_same", int(psh == sh["index"]))
                put("m_lo", d & 0xFFFF); put("m_hi", d >> 16)
                lk.add(B_U16, d & 0xFFFF); lk.add(B_RANGE, d >> 16, 0)
                if j == 0:
                    lk.add(B_ADDR, byts(ev["ptr"])[0], byts(ev["ptr"])[3])
        out["sha_extend"] = mat

    # ---- sha_compress: 80 rows per call (group 0 loads the state, groups 1..8 are the rounds, group 9 writes back)
    if sh["sha_cmp"]:
        cid, chip = chips["sha_compress"]
        mat = np.zeros((chip.main_width, 1 << log2ceil(80 * len(sh["sha_cmp"]))), np.int64)
        rotr = lambda v, n: ((v >> n) | (v << (32 - n))) & M32
        bits = lambda v: [(v >> t) & 1 for t in range(32)]
        halves = lambda v: [v & 0xFFFF, v >> 16]
        for e, ev in enumerate(sh["sha_cmp"]):
            v = [0] * 8                                   # a..h at the start of the row
            for j in range(80):
                g, o = divmod(j, 8)
                put, putv = _col_setter(chip, mat, 80 * e + j)
                put("is_real", 1); put("is_first", int(j == 0)); put("is_last", int(j == 79))
                putv("oc", [int(t == o) for t in range(8)]); putv("gr", [int(t == g) for t in range(10)])
                put("clk", ev["clk"]); putv("wp", byts(ev["w_ptr"])); putv("hp", byts(ev["h_ptr"]))
                putv("ab", bits(v[0])); putv("bb", bits(v[1])); putv("cb", bits(v[2]))
                putv("eb", bits(v[4])); putv("fb", bits(v[5])); putv("gb", bits(v[6]))
                putv("d", halves(v[3])); putv("h", halves(v[7]))
                S1 = rotr(v[4], 6) ^ rotr(v[4], 11) ^ rotr(v[4], 25)
                S0 = rotr(v[0], 2) ^ rotr(v[0], 13) ^ rotr(v[0], 22)
                mj = (v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2])
                chv = (v[4] & v[5]) ^ (~v[4] & v[6] & M32)
                putv("s1", halves(S1)); putv("s0", halves(S0)); putv("mj", halves(mj))
                ts = ev["clk"] + 2
                if g == 0:
                    addr, before, prev = ev["h_ptr"] + 4 * (7 - o), ev["hs"][7 - o], ev["hprev"][7 - o]
                    after, nxt_a, nxt_e = before, before, v[3]
                elif g <= 8:
                    i = 8 * (g - 1) + o
                    addr, before, prev = ev["w_ptr"] + 4 * i, ev["w"][i], ev["wprev"][i]
                    after = before
                    terms = [v[7], S1, chv, SHA_K[i], before]
                    lo_t, hi_t = sum(t & 0xFFFF for t in terms), sum(t >> 16 for t in terms)
                    e_lo = lo_t + (v[3] & 0xFFFF); e_hi = hi_t + (v[3] >> 16) + (e_lo >> 16)
                    a_lo = lo_t + (S0 & 0xFFFF) + (mj & 0xFFFF); a_hi = hi_t + (S0 >> 16) + (mj >> 16) + (a_lo >> 16)
                    putv("ce", [(e_lo >> 16 >> t) & 1 for t in range(3)] + [(e_hi >> 16 >> t) & 1 for t in range(3)])
                    putv("ca", [(a_lo >> 16 >> t) & 1 for t in range(3)] + [(a_hi >> 16 >> t) & 1 for t in range(3)])
                    nxt_e = (e_lo & 0xFFFF) | ((e_hi & 0xFFFF) << 16)
                    nxt_a = (a_lo & 0xFFFF) | ((a_hi & 0xFFFF) << 16)
                    t1 = sum(terms) & M32
                    assert nxt_e == (v[3] + t1) & M32 and nxt_a == (t1 + S0 + mj) & M32
                else:
                    addr, before, prev = ev["h_ptr"] + 4 * (7 - o), ev["hs"][7 - o], (sh["index"], ev["clk"] + 2)
                    after, ts = (before + v[7]) & M32, ev["clk"] + 3
                    lo = (before & 0xFFFF) + (v[7] & 0xFFFF)
                    putv("cf", [lo >> 16, ((before >> 16) + (v[7] >> 16) + (lo >> 16)) >> 16])
                    lk.add(B_RANGE, byts(after)[0], byts(after)[1]); lk.add(B_RANGE, byts(after)[2], byts(after)[3])
                    nxt_a, nxt_e = 0, v[3]
                put("maddr", addr); putv("mv", byts(after)); putv("mo", byts(before))
                psh, pts = prev
                dgap = ts - pts - 1 if psh == sh["index"] else sh["index"] - psh - 1
                put("m_sh", psh); put("m_ts", pts); put("m_same", int(psh == sh["index"]))
                put("m_lo", dgap & 0xFFFF); put("m_hi", dgap >> 16)
                lk.add(B_U16, dgap & 0xFFFF); lk.add(B_RANGE, dgap >> 16, 0)
                if j == 0:
                    lk.add(B_ADDR, byts(ev["w_ptr"])[0], byts(ev["w_ptr"])[3]); lk.add(B_ADDR, byts(ev["h_ptr"])[0], byts(ev["h_ptr"])[3])
                v = [nxt_a, v[0], v[1], v[2], nxt_e, v[4], v[5], v[6]]
        out["sha_compress"] = mat

    # ---- field / curve precompile chips: one row per call.  The big-integer identities are restated here limb by limb
    # (convolutions of byte vectors); only the carry OFFSETS (constants the AIR picked) are read from its description.
    def bytes_of(v, n):
        return [(v >> (8 * i)) & 0xFF for i in range(n)]

    def conv(x, y):
        out = [0] * (len(x) + len(y) - 1)
        for i, xi in enumerate(x):
            if xi:
                for j, yj in enumerate(y):
                    out[i + j] += xi * yj
        return out

    def lin(K, *terms):
        """sum of coef * limb vector (padded to K coefficients)"""
        out = [0] * K
        for coef, v in terms:
            for i, x in enumerate(v):
                out[i] += coef * x
        return out

    def carry_cells(put, name, rel, coefs):
        """carries W_k = (c_k + W_(k-1)) / 256 of an identity whose coefficients are `coefs`; cell = W_k + offset, low 16
        bits and (when the AIR gave the identity a top-bit column) the bit above"""
        W, wide = 0, rel.w_top is not None
        assert len(coefs) == rel.K
        for k in range(rel.K - 1):
            t = coefs[k] + W
            assert t % 256 == 0
            W = t // 256
            cell = W + rel.w_off[k]
            assert 0 <= cell < (1 << 17 if wide else 1 << 16)
            put(f"{name}_w[{k}]", cell & 0xFFFF)
            if wide:
                put(f"{name}_wb[{k}]", cell >> 16)
            lk.add(B_U16, cell & 0xFFFF)
        assert coefs[-1] + W == 0

    def range_pairs(vals):
        for i in range(0, len(vals) - 1, 2):
            lk.add(B_RANGE, vals[i], vals[i + 1])
        if len(vals) % 2:
            lk.add(B_RANGE, vals[-1], 0)

    def lt_cells(put, putv, name, v, p, L):
        """v < p on 3-byte groups: flag of the most significant group that differs, p_g - v_g - 1 there"""
        vb, pb = bytes_of(v, L), bytes_of(p, L)
        G = (L + 2) // 3
        grp = lambda bs, g: sum(x << (8 * t) for t, x in enumerate(bs[3 * g:3 * g + 3]))
        g = next(g for g in reversed(range(G)) if grp(vb, g) != grp(pb, g))
        assert grp(vb, g) < grp(pb, g)
        d = grp(pb, g) - grp(vb, g) - 1
        put(f"{name}_f[{g}]", 1)
        putv(f"{name}_d", bytes_of(d, 3))
        lk.add(B_RANGE, d & 0xFF, (d >> 8) & 0xFF); lk.add(B_RANGE, d >> 16, 0)

    def mem_cells(put, name, prevs, ts):
        his = []
        for k_, prev in enumerate(prevs):
            same, lo, h = gap(prev, ts)
            put(f"{name}_sh[{k_}]", prev[0]); put(f"{name}_ts[{k_}]", prev[1]); put(f"{name}_same[{k_}]", same)
            put(f"{name}_lo[{k_}]", lo); put(f"{name}_hi[{k_}]", h)
            lk.add(B_U16, lo)
            his.append(h)
        range_pairs(his)

    for cname_ in ("fp_op", "fp2_op", "bls_g1", "secp_k1", "u256_mul"):
        evs = [e for e in sh["big"] if e["chip"] == cname_]
        if not evs:
            continue
        cid, chip = chips[cname_]
        rels = {r.name: r for r in chip.poly_rels}
        mat = np.zeros((chip.main_width, 1 << log2ceil(len(evs))), np.int64)
        for r_, ev in enumerate(evs):
            put, putv = _col_setter(chip, mat, r_)
            op = ev["op"]
            put("is_real", 1); put("clk", ev["clk"])
            if cname_ == "u256_mul":
                L = 32
                putv("xp", byts(ev["a_ptr"])); putv("yp", byts(ev["b_ptr"]))
                lk.add(B_ADDR, byts(ev["a_ptr"])[0], byts(ev["a_ptr"])[3]); lk.add(B_ADDR, byts(ev["b_ptr"])[0], byts(ev["b_ptr"])[3])
                mem_cells(put, "my", ev["bprev"], ev["clk"] + 2)
                mem_cells(put, "mx", ev["aprev"], ev["clk"] + 3)
                xv, yv, mv, rv = words_to_int(ev["a"]), words_to_int(ev["b"][:8]), words_to_int(ev["b"][8:]), words_to_int(ev["r"])
                putv("x", bytes_of(xv, L)); putv("y", bytes_of(yv, L)); putv("m", bytes_of(mv, L)); putv("r", bytes_of(rv, L))
                range_pairs(bytes_of(rv, L))
                mod = mv or 1 << 256
                assert (xv * yv - rv) % mod == 0
                qv = bytes_of((xv * yv - rv) // mod, L + 1)
                putv("q", qv)
                range_pairs(qv)
                put("m_zero", int(mv == 0))
                if mv:
                    grp = lambda v, g: (v >> (24 * g)) & 0xFFFFFF
                    g = next(g for g in range(11) if grp(mv, g))
                    put(f"mz[{g}]", inv(grp(mv, g)))
                    lt_cells(put, putv, "rlt", rv, mv, L)
                M33 = bytes_of(mv, L) + [int(mv == 0)]
                co = lin(2 * L + 1, (1, conv(bytes_of(xv, L), bytes_of(yv, L))), (-1, bytes_of(rv, L)), (-1, conv(qv, M33)))
                carry_cells(put, "rel", rels["rel"], co)
                continue
            put("is_" + op, 1)
            if cname_ in ("fp_op", "fp2_op"):
                L, p_ = 48, BLS_P
                Pl = bytes_of(p_, L)
                putv("xp", byts(ev["a_ptr"])); putv("yp", byts(ev["b_ptr"]))
                lk.add(B_ADDR, byts(ev["a_ptr"])[0], byts(ev["a_ptr"])[3]); lk.add(B_ADDR, byts(ev["b_ptr"])[0], byts(ev["b_ptr"])[3])
                mem_cells(put, "my", ev["bprev"], ev["clk"] + 2)
                mem_cells(put, "mx", ev["aprev"], ev["clk"] + 3)
                parts = 1 if cname_ == "fp_op" else 2
                xs = [words_to_int(ev["a"][12 * i:12 * i + 12]) for i in range(parts)]
                ys = [words_to_int(ev["b"][12 * i:12 * i + 12]) for i in range(parts)]
                rs = [words_to_int(ev["r"][12 * i:12 * i + 12]) for i in range(parts)]
                sfx = [""] if parts == 1 else ["0", "1"]
                for i in range(parts):
                    putv("x" + sfx[i], bytes_of(xs[i], L)); putv("y" + sfx[i], bytes_of(ys[i], L)); putv("r" + sfx[i], bytes_of(rs[i], L))
                    range_pairs(bytes_of(rs[i], L))
                for i in range(parts):
                    xb, yb, rb = bytes_of(xs[i], L), bytes_of(ys[i], L), bytes_of(rs[i], L)
                    K = 2 * L
                    if op == "mul" and parts == 1:
                        V, co = xs[0] * ys[0] - rs[0], lin(K, (1, conv(xb, yb)), (-1, rb))
                    elif op == "mul" and i == 0:     # x0 y0 - x1 y1 + 2^388 p - r0
                        off = bytes_of(1 << 388, L + 1)
                        V = xs[0] * ys[0] - xs[1] * ys[1] + (p_ << 388) - rs[0]
                        co = lin(K, (1, conv(bytes_of(xs[0], L), bytes_of(ys[0], L))), (-1, conv(bytes_of(xs[1], L), bytes_of(ys[1], L))), (1, conv(off, Pl)), (-1, rb))
                    elif op == "mul":                # x0 y1 + x1 y0 - r1
                        V = xs[0] * ys[1] + xs[1] * ys[0] - rs[1]
                        co = lin(K, (1, conv(bytes_of(xs[0], L), bytes_of(ys[1], L))), (1, conv(bytes_of(xs[1], L), bytes_of(ys[0], L))), (-1, rb))
                    elif op == "add":
                        V, co = xs[i] + ys[i] - rs[i], lin(K, (1, xb), (1, yb), (-1, rb))
                    else:                            # x - y + 10 p - r
                        V, co = xs[i] - ys[i] + 10 * p_ - rs[i], lin(K, (1, xb), (-1, yb), (10, Pl), (-1, rb))
                    assert V % p_ == 0 and V >= 0
                    qv = bytes_of(V // p_, L + 1)
                    putv("q" + sfx[i], qv)
                    range_pairs(qv)
                    co = [a_ - b_ for a_, b_ in zip(co, lin(K, (1, conv(qv, Pl))))]
                    lt_cells(put, putv, ("r" + sfx[i] + "lt"), rs[i], p_, L)
                    carry_cells(put, "rel" + sfx[i], rels["rel" + sfx[i]], co)
            else:
                L, p_ = (48, BLS_P) if cname_ == "bls_g1" else (32, SECP_P)
                Wn, K, M_ = L // 4, 2 * L, 1 << (8 * L)
                Pl = bytes_of(p_, L)
                add = op == "add"
                putv("pp", byts(ev["a_ptr"])); putv("qp", byts(ev["b_ptr"]))
                lk.add(B_ADDR, byts(ev["a_ptr"])[0], byts(ev["a_ptr"])[3])
                if add:
                    lk.add(B_ADDR, byts(ev["b_ptr"])[0], byts(ev["b_ptr"])[3])
                    mem_cells(put, "mq", ev["bprev"], ev["clk"] + 2)
                mem_cells(put, "mp", ev["aprev"], ev["clk"] + 3)
                x1, y1 = words_to_int(ev["a"][:Wn]), words_to_int(ev["a"][Wn:])
                x2, y2 = (words_to_int(ev["b"][:Wn]), words_to_int(ev["b"][Wn:])) if add else (0, 0)
                x3, y3, lam = words_to_int(ev["r"][:Wn]), words_to_int(ev["r"][Wn:]), ev["lam"]
                B = lambda v: bytes_of(v, L)
                for nm_, v in (("x1", x1), ("y1", y1), ("x2", x2), ("y2", y2), ("lam", lam), ("x3", x3), ("y3", y3)):
                    putv(nm_, B(v))
                for v in (lam, x3, y3):
                    range_pairs(B(v))
                # slope:  ADD  lam (x2 - x1) - (y2 - y1) + 4 M p,   DOUBLE  2 lam y1 - 3 x1^2 + 4 M p
                offp = conv(bytes_of(4 * M_, L + 1), Pl)
                if add:
                    V1 = lam * (x2 - x1) - (y2 - y1) + 4 * M_ * p_
                    c1 = lin(K, (1, conv(B(lam), B(x2))), (-1, conv(B(lam), B(x1))), (-1, B(y2)), (1, B(y1)), (1, offp))
                else:
                    V1 = 2 * lam * y1 - 3 * x1 * x1 + 4 * M_ * p_
                    c1 = lin(K, (2, conv(B(lam), B(y1))), (-3, conv(B(x1), B(x1))), (1, offp))
                xs_ = x2 if add else x1
                V2 = lam * lam - x1 - xs_ - x3 + 4 * p_
                c2 = lin(K, (1, conv(B(lam), B(lam))), (-1, B(x1)), (-1, B(xs_)), (-1, B(x3)), (4, Pl))
                V3 = lam * (x1 - x3) - y1 - y3 + 2 * M_ * p_
                c3 = lin(K, (1, conv(B(lam), B(x1))), (-1, conv(B(lam), B(x3))), (-1, B(y1)), (-1, B(y3)), (1, conv(bytes_of(2 * M_, L + 1), Pl)))
                qvs = []
                for i, (V, co) in enumerate(((V1, c1), (V2, c2), (V3, c3)), start=1):
                    assert V % p_ == 0 and V >= 0
                    qv = bytes_of(V // p_, L + 1)
                    putv(f"q{i}", qv)
                    qvs.append((qv, co))
                for qv, _ in qvs:
                    range_pairs(qv)
                checks = [("x1lt", x1), ("y1lt", y1)] + ([("x2lt", x2), ("y2lt", y2)] if add else []) + [("x3lt", x3), ("y3lt", y3)]
                for nm_, v in checks:
                    lt_cells(put, putv, nm_, v, p_, L)
                if add:      # x1 != x2: the inverse of one differing 3-byte group
                    G = (L + 2) // 3
                    grp = lambda v, g: (v >> (24 * g)) & 0xFFFFFF
                    g = next(g for g in range(G) if grp(x1, g) != grp(x2, g))
                    put(f"xne_z[{g}]", inv(grp(x1, g) - grp(x2, g)))
                for i, (qv, co) in enumerate(qvs, start=1):
                    co = [a_ - b_ for a_, b_ in zip(co, lin(K, (1, conv(qv, Pl))))]
                    carry_cells(put, f"rel{i}", rels[f"rel{i}"], co)
        out[cname_] = mat

    # ---- preprocessed chips and their multiplicity columns
    cidp, chipp = chips["program"]
    np_rows = 1 << log2ceil(max(len(run.provable), 1))
    prep = np.zeros((chipp.prep_width, np_rows), np.int64)
    pidx = {nm: i for i, nm in enumerate(chipp.prep_names)}
    mult = np.zeros((1, np_rows), np.int64)
    for r in range(np_rows):
        ins = run.provable[r] if r < len(run.provable) else run.provable[0]
        k = ins.kind
        vals = dict(pc=ins.pc, rd=ins.rd, rs1=ins.rs1, rs2=ins.rs2, aux=ins.tgt_col + (ALU_CODES[ins.alu] if k == "alu" else 0))
        for i in range(4):
            vals[f"imm[{i}]"] = byts(ins.imm | ins.off)[i]
        vals["bit_op"], vals["cmp_signed"] = bit_op_of.get(k, 0), int(k in signed_forms)
        reads_rs2 = (k in ("beq", "bne", "blt", "bge", "bltu", "bgeu", "sb", "sh", "sw", "ecall")
                     or (not ins.imm_form and k in ("add", "sub", "and", "or", "xor", "slt", "sltu", "mul", "mulhu", "alu")))
        fl = {f: 0 for f in FLAGS}
        fl[fam_of[k]] = 1
        fl["rd_en"], fl["rs1_en"], fl["rs2_en"], fl["imm_c"] = int(ins.rd != 0), int(k not in ("lui", "jal")), int(reads_rs2), int(ins.imm_form)
        vals.update(fl)
        for nm, v in vals.items():
            prep[pidx[nm], r] = v
        if r < len(run.provable):
            mult[0, r] = prog_mult.get(ins.pc, 0)
    out["program"] = (mult, prep)
    cidb, chipb = chips["byte"]
    bprep = np.zeros((chipb.prep_width, 65536), np.int64)
    rr = np.arange(65536)
    bb, cc = rr >> 8, rr & 255
    for nm, v in (("b", bb), ("c", cc), ("and", bb & cc), ("or", bb | cc), ("xor", bb ^ cc), ("ltu", (bb < cc).astype(np.int64)), ("msb", bb >> 7),
                  ("addr", (bb & 3) + 4 * (cc >= (ADDR_LIMIT >> 24)).astype(np.int64))):
        bprep[chipb.prep_names.index(nm)] = v
    out["byte"] = (lk.byte, bprep)
    cidi, chipi = chips["mem_image"]
    img = sorted([(REG_BASE + r_, 0) for r_ in range(32)] + list(run.image_mem.items()))
    ni = 1 << log2ceil(len(img))
    iprep = np.zeros((chipi.prep_width, ni), np.int64)
    for r, (addr, v) in enumerate(img):
        iprep[chipi.prep_names.index("addr"), r] = addr
        for i in range(4):
            iprep[chipi.prep_names.index(f"v[{i}]"), r] = byts(v)[i]
        iprep[chipi.prep_names.index("is_real"), r] = 1
    out["mem_image"] = (np.zeros((1, ni), np.int64), iprep)

    result = []
    for name, (cid, chip) in sorted(chips.items(), key=lambda kv: kv[1][0]):
        if name not in out:
            continue
        v = out[name]
        main, prep_m = v if isinstance(v, tuple) else (v, np.zeros((0, v.shape[1]), np.int64))
        result.append(dict(chip_id=cid, log_n=int(main.shape[1]).bit_length() - 1, main=(main % P).astype(np.uint32), prep=(prep_m % P).astype(np.uint32)))
    pubs = np.array([sh["start_pc"] % P, sh["next_pc"] % P, (run.exit_code % P) if last else 0, shard, int(last)], np.uint32)
    return result, pubs
