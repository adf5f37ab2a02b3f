"""BLS12-381 G1 guest code over the field / curve precompiles, assembled with tools/rvasm.py: point decompression
(zcash encoding, square root by the Fp multiplication precompile), double-and-add scalar multiplication with the
cases the affine ADD / DOUBLE precompiles do not cover (infinity, equal and opposite points), compression, and the
reference's Horner evaluation of a G1 polynomial (reference crates/dkg/src/dkg_math.rs:160-174 `evaluate_polynomial`,
called per participant id by `agg_coefficients`, :230-248).  This is the curve work the reference's current guests do
through the SP1-patched bls12_381 crate (reference crates/dkg/Cargo.toml:25).

Memory formats: a field element is 12 little-endian words (SP1's precompile layout); a POINT is 25 words
x[12] || y[12] || is_infinity; a compressed point is 48 bytes, big-endian x with the three flag bits in byte 0.
Register discipline (no general stack frames): level-0 code keeps its loop state in memory variables, scalar_mul owns
s0-s3, the point routines s4-s7, leaf routines s8-s11 and the temporaries; `ra` is pushed on a small stack."""
import struct

from tools.rvasm import Asm, SYS_WRITE

M32 = 0xFFFFFFFF
BLS_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
BLS_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
SYS_BLS12381_ADD, SYS_BLS12381_DOUBLE = 0x0001011E, 0x0000011F
SYS_BLS12381_FP_ADD, SYS_BLS12381_FP_SUB, SYS_BLS12381_FP_MUL = 0x00010120, 0x00010121, 0x00010122
POINT_WORDS = 25


def words_of(v, n):
    return [(v >> (32 * i)) & M32 for i in range(n)]


class G1Lib:
    """emits the subroutines once per program; `a` must jump over them (emit() is called before `main`)"""

    def __init__(self, a: Asm):
        self.a = a
        self.one = a.dword("bls_one", words_of(1, 12))
        self.zero = a.dword("bls_zero", words_of(0, 12))
        self.four = a.dword("bls_four", words_of(4, 12))
        self.p = a.dword("bls_p", words_of(BLS_P, 12))
        self.half = a.dword("bls_half_p", words_of((BLS_P - 1) // 2, 12))
        self.exp_sqrt = a.dword("bls_exp_sqrt", words_of((BLS_P + 1) // 4, 12))
        self.order = a.dword("bls_order", words_of(BLS_R, 8))
        self.t0 = a.dword("bls_t0", [0] * 12)
        self.t1 = a.dword("bls_t1", [0] * 12)
        self.stack = a.dword("bls_stack", [0] * 16) + 64
        self.tmp_pt = a.dword("bls_tmp_pt", [0] * POINT_WORDS)

    # ---- small macros
    def push_ra(self):
        a = self.a
        a.addi("sp", "sp", -4)
        a.sw("ra", "sp", 0)

    def pop_ret(self):
        a = self.a
        a.lw("ra", "sp", 0)
        a.addi("sp", "sp", 4)
        a.ret()

    def sys(self, code):
        self.a.li("t0", code)
        self.a.ecall()

    def copy_words(self, dst, src, n, tmp="t1"):
        """registers dst / src hold addresses; straight-line"""
        for i in range(n):
            self.a.lw(tmp, src, 4 * i)
            self.a.sw(tmp, dst, 4 * i)

    def bswap(self, dst, src, t1="t2", t2="t3"):
        a = self.a
        a.slli(dst, src, 24)
        a.srli(t1, src, 24)
        a.or_(dst, dst, t1)
        a.srli(t1, src, 8)
        a.li(t2, 0xFF00)
        a.and_(t1, t1, t2)
        a.or_(dst, dst, t1)
        a.slli(t1, src, 8)
        a.li(t2, 0xFF0000)
        a.and_(t1, t1, t2)
        a.or_(dst, dst, t1)

    def fail(self):
        """the reference's guests panic on a bad point (reference crates/dkg/src/dkg_math.rs:24-31): exit code 1"""
        self.a.halt(1)

    def emit(self):
        a = self.a
        # ---- cmp12(a0, a1) -> a0 = 0 equal, 1 first greater, 2 first smaller (12-word unsigned numbers)
        a.label("cmp12")
        a.li("t4", 12)
        a.label("cmp12_l")
        a.addi("t4", "t4", -1)
        a.slli("t1", "t4", 2)
        a.add("t2", "a0", "t1")
        a.add("t3", "a1", "t1")
        a.lw("t2", "t2", 0)
        a.lw("t3", "t3", 0)
        a.bltu("t3", "t2", "cmp12_gt")
        a.bltu("t2", "t3", "cmp12_lt")
        a.bne("t4", "zero", "cmp12_l")
        a.li("a0", 0)
        a.ret()
        a.label("cmp12_gt")
        a.li("a0", 1)
        a.ret()
        a.label("cmp12_lt")
        a.li("a0", 2)
        a.ret()
        # ---- fp_pow(a0 = out, a1 = base, a2 = 12-word exponent): out = base^e, square-and-multiply from the top bit
        a.label("fp_pow")
        a.mv("s8", "a0")
        a.mv("s9", "a1")
        a.mv("s10", "a2")
        a.li("t4", self.one)
        self.copy_words("s8", "t4", 12)
        a.li("s11", 12)
        a.label("fp_pow_w")
        a.addi("s11", "s11", -1)
        a.slli("t1", "s11", 2)
        a.add("t1", "s10", "t1")
        a.lw("t5", "t1", 0)
        a.li("t6", 32)
        a.label("fp_pow_b")
        a.mv("a0", "s8")
        a.mv("a1", "s8")
        self.sys(SYS_BLS12381_FP_MUL)
        a.srli("t1", "t5", 31)
        a.slli("t5", "t5", 1)
        a.beq("t1", "zero", "fp_pow_s")
        a.mv("a0", "s8")
        a.mv("a1", "s9")
        self.sys(SYS_BLS12381_FP_MUL)
        a.label("fp_pow_s")
        a.addi("t6", "t6", -1)
        a.bne("t6", "zero", "fp_pow_b")
        a.bne("s11", "zero", "fp_pow_w")
        a.ret()
        # ---- g1_double(a0 = R): R := 2 R
        a.label("g1_double")
        a.lw("t1", "a0", 96)
        a.bne("t1", "zero", "g1_double_r")
        a.li("a1", 0)
        self.sys(SYS_BLS12381_DOUBLE)
        a.label("g1_double_r")
        a.ret()
        # ---- g1_add(a0 = R, a1 = Q): R := R + Q, every case
        a.label("g1_add")
        self.push_ra()
        a.mv("s4", "a0")
        a.mv("s5", "a1")
        a.lw("t1", "s5", 96)
        a.bne("t1", "zero", "g1_add_r")            # Q = 0
        a.lw("t1", "s4", 96)
        a.beq("t1", "zero", "g1_add_f")
        self.copy_words("s4", "s5", POINT_WORDS)    # R = 0: R := Q
        a.j("g1_add_r")
        a.label("g1_add_f")
        a.mv("a0", "s4")
        a.mv("a1", "s5")
        a.call("cmp12")
        a.bne("a0", "zero", "g1_add_g")
        a.addi("a0", "s4", 48)                      # equal abscissae: the same point (double) or opposite points (infinity)
        a.addi("a1", "s5", 48)
        a.call("cmp12")
        a.bne("a0", "zero", "g1_add_o")
        a.mv("a0", "s4")
        a.li("a1", 0)
        self.sys(SYS_BLS12381_DOUBLE)
        a.j("g1_add_r")
        a.label("g1_add_o")
        a.li("t1", 1)
        a.sw("t1", "s4", 96)
        a.j("g1_add_r")
        a.label("g1_add_g")
        a.mv("a0", "s4")
        a.mv("a1", "s5")
        self.sys(SYS_BLS12381_ADD)
        a.label("g1_add_r")
        self.pop_ret()
        # ---- scalar_mul(a0 = R, a1 = P, a2 = scalar words, a3 = number of words): R := [scalar] P  (R and P distinct)
        a.label("scalar_mul")
        self.push_ra()
        a.mv("s0", "a0")
        a.mv("s1", "a1")
        a.mv("s2", "a2")
        a.mv("s3", "a3")
        a.li("t1", 1)
        a.sw("t1", "s0", 96)                        # R = infinity
        a.label("smul_w")
        a.addi("s3", "s3", -1)
        a.slli("t1", "s3", 2)
        a.add("t1", "s2", "t1")
        a.lw("s6", "t1", 0)                          # the current scalar word (s6 / s7 are not used by g1_add / g1_double)
        a.li("s7", 32)
        a.label("smul_b")
        a.mv("a0", "s0")
        a.call("g1_double")
        a.srli("t1", "s6", 31)
        a.slli("s6", "s6", 1)
        a.beq("t1", "zero", "smul_s")
        a.mv("a0", "s0")
        a.mv("a1", "s1")
        a.call("g1_add")
        a.label("smul_s")
        a.addi("s7", "s7", -1)
        a.bne("s7", "zero", "smul_b")
        a.bne("s3", "zero", "smul_w")
        self.pop_ret()
        # ---- g1_decompress(a0 = out point, a1 = 48 compressed bytes, word aligned); exits with code 1 on a bad encoding
        a.label("g1_decompress")
        self.push_ra()
        a.mv("s4", "a0")
        a.mv("s5", "a1")
        a.lbu("t1", "s5", 0)
        a.andi("t2", "t1", 0x80)
        a.beq("t2", "zero", "g1_dec_bad")           # only the compressed form is accepted
        a.andi("t2", "t1", 0x40)
        a.beq("t2", "zero", "g1_dec_fin")
        for i in range(24):                          # the point at infinity
            a.sw("zero", "s4", 4 * i)
        a.li("t1", 1)
        a.sw("t1", "s4", 96)
        a.j("g1_dec_r")
        a.label("g1_dec_fin")
        a.srli("s6", "t1", 5)
        a.andi("s6", "s6", 1)                        # sign flag: y is the larger root
        for j in range(12):                          # big-endian bytes -> little-endian words
            a.lw("t4", "s5", 4 * (11 - j))
            self.bswap("t5", "t4")
            if j == 11:
                a.li("t2", 0x1FFFFFFF)
                a.and_("t5", "t5", "t2")
            a.sw("t5", "s4", 4 * j)
        a.sw("zero", "s4", 96)
        a.mv("a0", "s4")
        a.li("a1", self.p)
        a.call("cmp12")
        a.addi("a0", "a0", -2)
        a.bne("a0", "zero", "g1_dec_bad")           # x must be reduced
        a.li("s7", self.t0)                          # t0 = x^3 + 4
        self.copy_words("s7", "s4", 12)
        for _ in range(2):
            a.mv("a0", "s7")
            a.mv("a1", "s4")
            self.sys(SYS_BLS12381_FP_MUL)
        a.mv("a0", "s7")
        a.li("a1", self.four)
        self.sys(SYS_BLS12381_FP_ADD)
        a.addi("a0", "s4", 48)                       # y = t0^((p + 1) / 4)
        a.mv("a1", "s7")
        a.li("a2", self.exp_sqrt)
        a.call("fp_pow")
        a.li("t4", self.t1)                          # t1 = y^2 must equal t0
        a.addi("t5", "s4", 48)
        self.copy_words("t4", "t5", 12)
        a.li("a0", self.t1)
        a.li("a1", self.t1)
        self.sys(SYS_BLS12381_FP_MUL)
        a.li("a0", self.t1)
        a.li("a1", self.t0)
        a.call("cmp12")
        a.bne("a0", "zero", "g1_dec_bad")           # x^3 + 4 is not a square: not on the curve
        a.addi("a0", "s4", 48)
        a.li("a1", self.half)
        a.call("cmp12")
        a.addi("a0", "a0", -1)
        a.sltiu("a0", "a0", 1)                       # a0 = (y > (p - 1) / 2)
        a.beq("a0", "s6", "g1_dec_r")
        a.li("t4", self.t1)                          # y := p - y
        a.li("t5", self.zero)
        self.copy_words("t4", "t5", 12)
        a.li("a0", self.t1)
        a.addi("a1", "s4", 48)
        self.sys(SYS_BLS12381_FP_SUB)
        a.addi("t4", "s4", 48)
        a.li("t5", self.t1)
        self.copy_words("t4", "t5", 12)
        a.label("g1_dec_r")
        self.pop_ret()
        a.label("g1_dec_bad")
        self.fail()
        # ---- g1_compress(a0 = 48-byte output, a1 = point)
        a.label("g1_compress")
        self.push_ra()
        a.mv("s4", "a0")
        a.mv("s5", "a1")
        a.lw("t1", "s5", 96)
        a.beq("t1", "zero", "g1_cmp_fin")
        for i in range(12):
            a.sw("zero", "s4", 4 * i)
        a.li("t1", 0xC0)
        a.sw("t1", "s4", 0)
        a.j("g1_cmp_r")
        a.label("g1_cmp_fin")
        for j in range(12):
            a.lw("t4", "s5", 4 * (11 - j))
            self.bswap("t5", "t4")
            a.sw("t5", "s4", 4 * j)
        a.addi("a0", "s5", 48)
        a.li("a1", self.half)
        a.call("cmp12")
        a.addi("a0", "a0", -1)
        a.sltiu("a0", "a0", 1)
        a.slli("a0", "a0", 5)
        a.ori("a0", "a0", 0x80)
        a.lw("t1", "s4", 0)
        a.or_("t1", "t1", "a0")
        a.sw("t1", "s4", 0)
        a.label("g1_cmp_r")
        self.pop_ret()
        # ---- g1_check_subgroup(a0 = point): [r] P must be the point at infinity (bls12_381's from_compressed does this check)
        a.label("g1_check_subgroup")
        self.push_ra()
        a.mv("a1", "a0")
        a.li("a0", self.tmp_pt)
        a.li("a2", self.order)
        a.li("a3", 8)
        a.call("scalar_mul")
        a.li("t1", self.tmp_pt)
        a.lw("t1", "t1", 96)
        a.bne("t1", "zero", "g1_sub_ok")
        self.fail()
        a.label("g1_sub_ok")
        self.pop_ret()


def horner(coeffs, ids, subgroup_check=True, stdin=False):
    """The reference's evaluate_polynomial (crates/dkg/src/dkg_math.rs:160-174) for every id, as a guest:
    decompress the k coefficient points (48-byte compressed G1, from the data segment, or with stdin=True from ONE
    stdin buffer of k * 48 bytes), y = cfs[k-1]; for i = k-2 .. 0: y = [id] y + cfs[i]; the compressed results (48 bytes
    per id, in order) are the public values.  ids are u32 (the reference's participant ids, crates/dkg/src/crypto/
    bls_common.rs:42-47).  Returns the ELF (stdin=True: (elf, the buffer))."""
    from tests.guests import _write_pv

    k = len(coeffs)
    assert k >= 1 and all(len(c) == 48 for c in coeffs)
    blob = b"".join(coeffs)
    a = Asm()
    lib = G1Lib(a)
    # (hinted input becomes initial memory: it must land outside the program image)
    src = 0x00400000 if stdin else a.dword("cfs_c", [w for (w,) in struct.iter_unpack("<I", blob)])
    pts = a.dword("cfs", [0] * (POINT_WORDS * k))
    idw = a.dword("ids", list(ids))
    y = a.dword("hy", [0] * POINT_WORDS)
    tmp = a.dword("htmp", [0] * POINT_WORDS)
    out = a.dword("hout", [0] * (12 * len(ids)))
    var = a.dword("hvars", [0] * 4)
    V_I, V_J = var, var + 4
    a.j("main")
    lib.emit()
    a.label("main")
    a.li("sp", lib.stack)
    if stdin:
        from tools.rvasm import SYS_HINT_LEN, SYS_HINT_READ

        a.li("t0", SYS_HINT_LEN)
        a.ecall()
        a.li("t1", 48 * k)
        a.beq("t0", "t1", "len_ok")
        a.halt(1)
        a.label("len_ok")
        a.li("a0", src)
        a.li("a1", 48 * k)
        a.li("t0", SYS_HINT_READ)
        a.ecall()
    # decompress (and check) every coefficient
    a.li("t1", 0)
    a.li("t2", V_I)
    a.sw("t1", "t2", 0)
    a.label("dec")
    a.li("t2", V_I)
    a.lw("t1", "t2", 0)
    a.li("t3", 4 * POINT_WORDS)
    a.mul("t3", "t1", "t3")
    a.li("a0", pts)
    a.add("a0", "a0", "t3")
    a.li("t3", 48)
    a.mul("t3", "t1", "t3")
    a.li("a1", src)
    a.add("a1", "a1", "t3")
    a.call("g1_decompress")
    if subgroup_check:
        a.li("t2", V_I)
        a.lw("t1", "t2", 0)
        a.li("t3", 4 * POINT_WORDS)
        a.mul("t3", "t1", "t3")
        a.li("a0", pts)
        a.add("a0", "a0", "t3")
        a.call("g1_check_subgroup")
    a.li("t2", V_I)
    a.lw("t1", "t2", 0)
    a.addi("t1", "t1", 1)
    a.sw("t1", "t2", 0)
    a.li("t3", k)
    a.bne("t1", "t3", "dec")
    # per id: Horner
    a.li("t1", 0)
    a.li("t2", V_J)
    a.sw("t1", "t2", 0)
    a.label("per_id")
    a.li("t4", y)
    a.li("t5", pts + 4 * POINT_WORDS * (k - 1))
    lib.copy_words("t4", "t5", POINT_WORDS)
    if k > 1:
        a.li("t1", k - 1)
        a.li("t2", V_I)
        a.sw("t1", "t2", 0)
        a.label("hstep")                             # i = V_I - 1:  y = [id] y + cfs[i]
        a.li("t2", V_J)
        a.lw("t1", "t2", 0)
        a.slli("t1", "t1", 2)
        a.li("a2", idw)
        a.add("a2", "a2", "t1")
        a.li("a0", tmp)
        a.li("a1", y)
        a.li("a3", 1)
        a.call("scalar_mul")
        a.li("t2", V_I)
        a.lw("t1", "t2", 0)
        a.addi("t1", "t1", -1)
        a.sw("t1", "t2", 0)
        a.li("t3", 4 * POINT_WORDS)
        a.mul("t3", "t1", "t3")
        a.li("a1", pts)
        a.add("a1", "a1", "t3")
        a.li("a0", tmp)
        a.call("g1_add")
        a.li("t4", y)
        a.li("t5", tmp)
        lib.copy_words("t4", "t5", POINT_WORDS)
        a.li("t2", V_I)
        a.lw("t1", "t2", 0)
        a.bne("t1", "zero", "hstep")
    a.li("t2", V_J)
    a.lw("t1", "t2", 0)
    a.li("t3", 48)
    a.mul("t3", "t1", "t3")
    a.li("a0", out)
    a.add("a0", "a0", "t3")
    a.li("a1", y)
    a.call("g1_compress")
    a.li("t2", V_J)
    a.lw("t1", "t2", 0)
    a.addi("t1", "t1", 1)
    a.sw("t1", "t2", 0)
    a.li("t3", len(ids))
    a.bne("t1", "t3", "per_id")
    a.li("s1", out)
    _write_pv(a, "s1", 48 * len(ids))
    a.halt(0)
    return (a.elf(), blob) if stdin else a.elf()


def lincomb(points, scalars, subgroup_check=False):
    """sum_i [s_i] P_i for compressed G1 points and 256-bit scalars (8 little-endian words each): the G1 half of the
    reference's lagrange_interpolation (crates/dkg/src/dkg_math.rs:176-228: r = sum_i [l_i(0)] Y_i; the Fr coefficients
    l_i(0) are computed by the caller).  The compressed result (48 bytes) is the public value."""
    from tests.guests import _write_pv

    n = len(points)
    assert n == len(scalars) and n >= 1
    a = Asm()
    lib = G1Lib(a)
    src = a.dword("pts_c", [w for (w,) in struct.iter_unpack("<I", b"".join(points))])
    sc = a.dword("scalars", [w for s in scalars for w in words_of(s, 8)])
    pt = a.dword("pt", [0] * POINT_WORDS)
    term = a.dword("term", [0] * POINT_WORDS)
    acc = a.dword("acc", [0] * 24 + [1])
    out = a.dword("out", [0] * 12)
    var = a.dword("vars", [0] * 2)
    a.j("main")
    lib.emit()
    a.label("main")
    a.li("sp", lib.stack)
    a.li("t2", var)
    a.sw("zero", "t2", 0)
    a.label("term_l")
    a.li("t2", var)
    a.lw("t1", "t2", 0)
    a.li("t3", 48)
    a.mul("t3", "t1", "t3")
    a.li("a1", src)
    a.add("a1", "a1", "t3")
    a.li("a0", pt)
    a.call("g1_decompress")
    if subgroup_check:
        a.li("a0", pt)
        a.call("g1_check_subgroup")
    a.li("t2", var)
    a.lw("t1", "t2", 0)
    a.slli("t1", "t1", 5)
    a.li("a2", sc)
    a.add("a2", "a2", "t1")
    a.li("a0", term)
    a.li("a1", pt)
    a.li("a3", 8)
    a.call("scalar_mul")
    a.li("a0", acc)
    a.li("a1", term)
    a.call("g1_add")
    a.li("t2", var)
    a.lw("t1", "t2", 0)
    a.addi("t1", "t1", 1)
    a.sw("t1", "t2", 0)
    a.li("t3", n)
    a.bne("t1", "t3", "term_l")
    a.li("a0", out)
    a.li("a1", acc)
    a.call("g1_compress")
    a.li("s1", out)
    _write_pv(a, "s1", 48)
    a.halt(0)
    return a.elf()
