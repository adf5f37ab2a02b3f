"""A tiny RV32IM assembler + ELF32 writer.  The image has no RISC-V toolchain
(SURVEY.md section 0.3), and the reference's only guest binary is prebuilt
machine code that must not be run, so test and benchmark guests are assembled
here from Python.

    a = Asm()
    a.li("t0", 5); a.label("loop"); a.addi("t0", "t0", -1); a.bne("t0", "zero", "loop")
    a.halt(0)
    elf = a.elf()
"""
import struct

ABI = ("zero ra sp gp tp t0 t1 t2 s0 s1 a0 a1 a2 a3 a4 a5 a6 a7 s2 s3 s4 s5 s6 s7 s8 s9 s10 s11 t3 t4 t5 t6").split()
REG = {n: i for i, n in enumerate(ABI)}
REG.update({f"x{i}": i for i in range(32)})
REG["fp"] = 8

SYS_HALT, SYS_WRITE, SYS_COMMIT, SYS_HINT_LEN, SYS_HINT_READ = 0x00, 0x02, 0x10, 0xF0, 0xF1
TEXT_BASE = 0x00200800
DATA_BASE = 0x00300000


def r(x):
    return x if isinstance(x, int) else REG[x]


class Asm:
    def __init__(self, text_base=TEXT_BASE, data_base=DATA_BASE):
        self.text_base, self.data_base = text_base, data_base
        self.items = []      # ("w", word) | ("b", kind, args) needing label resolution
        self.labels = {}
        self.data = []       # words
        self.data_labels = {}

    # ---- layout
    def pc(self):
        return self.text_base + 4 * len(self.items)

    def label(self, name):
        assert name not in self.labels, name
        self.labels[name] = self.pc()

    def word(self, w):
        self.items.append(("w", w & 0xFFFFFFFF))

    def dword(self, name, values):
        """reserve initialised data words; returns the byte address"""
        addr = self.data_base + 4 * len(self.data)
        self.data_labels[name] = addr
        self.data += [v & 0xFFFFFFFF for v in values]
        return addr

    # ---- encoders
    def _r(self, f7, f3, op, rd, rs1, rs2):
        self.word((f7 << 25) | (r(rs2) << 20) | (r(rs1) << 15) | (f3 << 12) | (r(rd) << 7) | op)

    def _i(self, f3, op, rd, rs1, imm):
        assert -2048 <= imm < 2048, imm
        self.word(((imm & 0xFFF) << 20) | (r(rs1) << 15) | (f3 << 12) | (r(rd) << 7) | op)

    def _s(self, f3, rs1, rs2, imm):
        assert -2048 <= imm < 2048, imm
        imm &= 0xFFF
        self.word(((imm >> 5) << 25) | (r(rs2) << 20) | (r(rs1) << 15) | (f3 << 12) | ((imm & 31) << 7) | 0x23)

    def _b(self, f3, rs1, rs2, target):
        self.items.append(("b", f3, r(rs1), r(rs2), target))

    # ALU
    def add(s, d, a, b): s._r(0, 0, 0x33, d, a, b)
    def sub(s, d, a, b): s._r(0x20, 0, 0x33, d, a, b)
    def sll(s, d, a, b): s._r(0, 1, 0x33, d, a, b)
    def slt(s, d, a, b): s._r(0, 2, 0x33, d, a, b)
    def sltu(s, d, a, b): s._r(0, 3, 0x33, d, a, b)
    def xor(s, d, a, b): s._r(0, 4, 0x33, d, a, b)
    def srl(s, d, a, b): s._r(0, 5, 0x33, d, a, b)
    def sra(s, d, a, b): s._r(0x20, 5, 0x33, d, a, b)
    def or_(s, d, a, b): s._r(0, 6, 0x33, d, a, b)
    def and_(s, d, a, b): s._r(0, 7, 0x33, d, a, b)
    def mul(s, d, a, b): s._r(1, 0, 0x33, d, a, b)
    def mulh(s, d, a, b): s._r(1, 1, 0x33, d, a, b)
    def mulhsu(s, d, a, b): s._r(1, 2, 0x33, d, a, b)
    def mulhu(s, d, a, b): s._r(1, 3, 0x33, d, a, b)
    def div(s, d, a, b): s._r(1, 4, 0x33, d, a, b)
    def divu(s, d, a, b): s._r(1, 5, 0x33, d, a, b)
    def rem(s, d, a, b): s._r(1, 6, 0x33, d, a, b)
    def remu(s, d, a, b): s._r(1, 7, 0x33, d, a, b)
    def addi(s, d, a, i): s._i(0, 0x13, d, a, i)
    def slti(s, d, a, i): s._i(2, 0x13, d, a, i)
    def sltiu(s, d, a, i): s._i(3, 0x13, d, a, i)
    def xori(s, d, a, i): s._i(4, 0x13, d, a, i)
    def ori(s, d, a, i): s._i(6, 0x13, d, a, i)
    def andi(s, d, a, i): s._i(7, 0x13, d, a, i)
    def slli(s, d, a, sh): s._i(1, 0x13, d, a, sh & 31)
    def srli(s, d, a, sh): s._i(5, 0x13, d, a, sh & 31)
    def srai(s, d, a, sh): s._i(5, 0x13, d, a, (sh & 31) | 0x400)
    # memory
    def lw(s, d, base, off=0): s._i(2, 0x03, d, base, off)
    def lb(s, d, base, off=0): s._i(0, 0x03, d, base, off)
    def lbu(s, d, base, off=0): s._i(4, 0x03, d, base, off)
    def lh(s, d, base, off=0): s._i(1, 0x03, d, base, off)
    def lhu(s, d, base, off=0): s._i(5, 0x03, d, base, off)
    def sw(s, src, base, off=0): s._s(2, base, src, off)
    def sb(s, src, base, off=0): s._s(0, base, src, off)
    def sh(s, src, base, off=0): s._s(1, base, src, off)
    # control
    def beq(s, a, b, t): s._b(0, a, b, t)
    def bne(s, a, b, t): s._b(1, a, b, t)
    def blt(s, a, b, t): s._b(4, a, b, t)
    def bge(s, a, b, t): s._b(5, a, b, t)
    def bltu(s, a, b, t): s._b(6, a, b, t)
    def bgeu(s, a, b, t): s._b(7, a, b, t)
    def jal(s, d, target): s.items.append(("j", r(d), target))
    def jalr(s, d, base, off=0): s._i(0, 0x67, d, base, off)
    def lui(s, d, imm20): s.word(((imm20 & 0xFFFFF) << 12) | (r(d) << 7) | 0x37)
    def auipc(s, d, imm20): s.word(((imm20 & 0xFFFFF) << 12) | (r(d) << 7) | 0x17)
    def ecall(s): s.word(0x73)
    # pseudo
    def nop(s): s.addi("zero", "zero", 0)
    def mv(s, d, a): s.addi(d, a, 0)
    def j(s, target): s.jal("zero", target)
    def call(s, target): s.jal("ra", target)
    def ret(s): s.jalr("zero", "ra", 0)

    def li(self, d, v):
        v &= 0xFFFFFFFF
        lo = v & 0xFFF
        if lo >= 0x800:
            lo -= 0x1000
        hi = ((v - lo) >> 12) & 0xFFFFF
        if hi:
            self.lui(d, hi)
            if lo:
                self.addi(d, d, lo)
        else:
            self.addi(d, "zero", lo)

    def halt(self, code=None):
        if code is not None:
            self.li("a0", code)
        self.li("t0", SYS_HALT)
        self.ecall()

    # ---- output
    def words(self):
        out = []
        for i, it in enumerate(self.items):
            pc = self.text_base + 4 * i
            if it[0] == "w":
                out.append(it[1])
            elif it[0] == "b":
                _, f3, rs1, rs2, tgt = it
                off = (self.labels[tgt] if isinstance(tgt, str) else tgt) - pc
                assert -4096 <= off < 4096 and off % 2 == 0, (tgt, off)
                o = off & 0x1FFF
                out.append(((o >> 12) << 31) | (((o >> 5) & 0x3F) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12)
                           | (((o >> 1) & 0xF) << 8) | (((o >> 11) & 1) << 7) | 0x63)
            else:
                _, rd, tgt = it
                off = (self.labels[tgt] if isinstance(tgt, str) else tgt) - pc
                assert -(1 << 20) <= off < (1 << 20) and off % 2 == 0
                o = off & 0x1FFFFF
                out.append(((o >> 20) << 31) | (((o >> 1) & 0x3FF) << 21) | (((o >> 11) & 1) << 20) | (((o >> 12) & 0xFF) << 12)
                           | (rd << 7) | 0x6F)
        return out

    def elf(self, entry=None):
        text = b"".join(struct.pack("<I", w) for w in self.words())
        data = b"".join(struct.pack("<I", w) for w in self.data)
        segs = [(self.text_base, text, 5)]
        if data:
            segs.append((self.data_base, data, 6))
        ehsize, phsize = 52, 32
        off = ehsize + phsize * len(segs)
        ph, blobs = b"", b""
        for vaddr, blob, flags in segs:
            ph += struct.pack("<IIIIIIII", 1, off + len(blobs), vaddr, vaddr, len(blob), len(blob), flags, 4)
            blobs += blob
        ent = self.text_base if entry is None else (self.labels[entry] if isinstance(entry, str) else entry)
        eh = b"\x7fELF" + bytes([1, 1, 1, 0]) + bytes(8) + struct.pack("<HHIIIIIHHHHHH", 2, 243, 1, ent, ehsize, 0, 0, ehsize, phsize, len(segs), 40, 0, 0)
        return eh + ph + blobs
