"""RV32IM guest programs assembled with tools/rvasm.py for the executor / prover tests
and the benchmark.  `bignum` is the synthetic "n-participant DKG-like" workload:
multi-limb multiply-accumulate over 384-bit operands, whose instruction mix
(add / sltu / lw / mul / mulhu / sw / branches) follows the histogram measured on
the reference's finalization guest."""
import struct

from tools.rvasm import Asm, SYS_COMMIT, SYS_HINT_LEN, SYS_HINT_READ, SYS_WRITE

M32 = 0xFFFFFFFF
HEAP = 0x00400000


SHA_K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
SHA_H0 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


def sha_padding(nbytes):
    """the SHA-256 padding of an nbytes-long message (nbytes % 4 == 0) as little-endian memory words"""
    assert nbytes % 4 == 0
    pad = b"\x80" + b"\0" * ((55 - nbytes) % 64) + (8 * nbytes).to_bytes(8, "big")
    assert (nbytes + len(pad)) % 64 == 0
    return [w for (w,) in struct.iter_unpack("<I", pad)]


def emit_sha256(a):
    """SHA-256 compression in RV32IM: label `sha256_blocks`(a0 = message, already padded, a1 = number of 64-byte blocks,
    a2 = the eight state words, updated in place).  Clobbers every t, a and s register except ra / sp.  Emitted once
    per program; its tables live in the data segment.  With a.sha_precompiles = True the message schedule and the 64 rounds
    of every block are SP1's SHA_EXTEND / SHA_COMPRESS precompile calls (what the patched `sha2` crate of the reference's
    guests does, crates/dkg/Cargo.toml:22) instead of ~5 000 RV32IM cycles."""
    if "sha256_blocks" in a.labels or getattr(a, "_sha_pending", False):
        return
    a._sha_pending = True
    ktab = a.dword("sha_k", SHA_K)
    wtab = a.dword("sha_w", [0] * 64)
    st = ["a3", "a4", "a5", "a6", "a7", "t3", "t4", "t5"]   # a b c d e f g h

    def rotr(dst, src, n, tmp):
        a.srli(dst, src, n)
        a.slli(tmp, src, 32 - n)
        a.or_(dst, dst, tmp)

    a.j("sha256_end")
    a.label("sha256_blocks")
    a.mv("s0", "a0")
    a.mv("s1", "a1")
    a.mv("s2", "a2")
    a.li("s3", wtab)
    a.li("s4", ktab)
    a.label("sha_block")
    # message schedule: W[0..15] = big-endian words of the block
    a.li("s5", 0)
    a.li("s6", 64)
    a.label("sha_ld")
    a.add("t6", "s0", "s5")
    a.lbu("t0", "t6", 0)
    a.slli("t0", "t0", 24)
    a.lbu("t1", "t6", 1)
    a.slli("t1", "t1", 16)
    a.or_("t0", "t0", "t1")
    a.lbu("t1", "t6", 2)
    a.slli("t1", "t1", 8)
    a.or_("t0", "t0", "t1")
    a.lbu("t1", "t6", 3)
    a.or_("t0", "t0", "t1")
    a.add("t6", "s3", "s5")
    a.sw("t0", "t6", 0)
    a.addi("s5", "s5", 4)
    a.bne("s5", "s6", "sha_ld")
    if getattr(a, "sha_precompiles", False):
        a.mv("a0", "s3")
        a.li("a1", 0)
        a.li("t0", SYS_SHA_EXTEND)
        a.ecall()
        a.mv("a0", "s3")
        a.mv("a1", "s2")
        a.li("t0", SYS_SHA_COMPRESS)
        a.ecall()
        a.addi("s0", "s0", 64)
        a.addi("s1", "s1", -1)
        a.bne("s1", "zero", "sha_block")
        a.ret()
        a.label("sha256_end")
        return
    a.li("s6", 256)
    a.label("sha_ext")          # W[i] = W[i-16] + s0(W[i-15]) + W[i-7] + s1(W[i-2])
    a.add("t6", "s3", "s5")
    a.lw("t0", "t6", -60)
    rotr("t1", "t0", 7, "t2")
    rotr("s7", "t0", 18, "t2")
    a.xor("t1", "t1", "s7")
    a.srli("s7", "t0", 3)
    a.xor("t1", "t1", "s7")
    a.lw("t0", "t6", -8)
    rotr("s7", "t0", 17, "t2")
    rotr("s8", "t0", 19, "t2")
    a.xor("s7", "s7", "s8")
    a.srli("s8", "t0", 10)
    a.xor("s7", "s7", "s8")
    a.add("t1", "t1", "s7")
    a.lw("t0", "t6", -64)
    a.add("t1", "t1", "t0")
    a.lw("t0", "t6", -28)
    a.add("t1", "t1", "t0")
    a.sw("t1", "t6", 0)
    a.addi("s5", "s5", 4)
    a.bne("s5", "s6", "sha_ext")
    for k, r_ in enumerate(st):
        a.lw(r_, "s2", 4 * k)
    A, B, Cc, D, E, F, G, H = st
    a.li("s5", 0)
    a.label("sha_round")
    rotr("t0", E, 6, "t2")
    rotr("t1", E, 11, "t2")
    a.xor("t0", "t0", "t1")
    rotr("t1", E, 25, "t2")
    a.xor("t0", "t0", "t1")        # S1
    a.and_("t1", E, F)
    a.xori("t2", E, -1)
    a.and_("t2", "t2", G)
    a.xor("t1", "t1", "t2")        # ch
    a.add("t0", "t0", "t1")
    a.add("t0", "t0", H)
    a.add("t6", "s4", "s5")
    a.lw("t1", "t6", 0)
    a.add("t0", "t0", "t1")
    a.add("t6", "s3", "s5")
    a.lw("t1", "t6", 0)
    a.add("t0", "t0", "t1")        # T1
    rotr("t1", A, 2, "t2")
    rotr("t6", A, 13, "t2")
    a.xor("t1", "t1", "t6")
    rotr("t6", A, 22, "t2")
    a.xor("t1", "t1", "t6")        # S0
    a.and_("t2", A, B)
    a.and_("t6", A, Cc)
    a.xor("t2", "t2", "t6")
    a.and_("t6", B, Cc)
    a.xor("t2", "t2", "t6")        # maj
    a.add("t1", "t1", "t2")        # T2
    a.mv(H, G)
    a.mv(G, F)
    a.mv(F, E)
    a.add(E, D, "t0")
    a.mv(D, Cc)
    a.mv(Cc, B)
    a.mv(B, A)
    a.add(A, "t0", "t1")
    a.addi("s5", "s5", 4)
    a.bne("s5", "s6", "sha_round")
    for k, r_ in enumerate(st):
        a.lw("t0", "s2", 4 * k)
        a.add("t0", "t0", r_)
        a.sw("t0", "s2", 4 * k)
    a.addi("s0", "s0", 64)
    a.addi("s1", "s1", -1)
    a.bne("s1", "zero", "sha_block")
    a.ret()
    a.label("sha256_end")


def _bswap(a, dst, src, t1, t2):
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


def _write_pv(a, ptr_reg, nbytes):
    """Commit the nbytes at ptr_reg as the guest's public values the way an SP1 guest does (sp1_zkvm::io::commit + the
    runtime's exit path; reference crates/finalization_prove/src/main.rs:26-32): WRITE them to fd 3,
    hash them with SHA-256 and COMMIT(k, word k) the eight digest words (little-endian words of the digest bytes)."""
    assert nbytes % 4 == 0
    uid = len(a.items)
    buf = a.dword(f"pvbuf_{uid}", [0] * (nbytes // 4) + sha_padding(nbytes))
    state = a.dword(f"pvstate_{uid}", SHA_H0)
    emit_sha256(a)
    a.mv("s8", ptr_reg)
    a.li("s9", buf)
    a.li("s10", buf + nbytes)
    if nbytes:
        a.label(f"pvcopy_{uid}")
        a.lw("t1", "s8", 0)
        a.sw("t1", "s9", 0)
        a.addi("s8", "s8", 4)
        a.addi("s9", "s9", 4)
        a.bne("s9", "s10", f"pvcopy_{uid}")
    a.li("a0", 3)
    a.li("a1", buf)
    a.li("a2", nbytes)
    a.li("t0", SYS_WRITE)
    a.ecall()
    a.li("a0", buf)
    a.li("a1", (nbytes + 4 * len(sha_padding(nbytes))) // 64)
    a.li("a2", state)
    a.call("sha256_blocks")
    a.li("s8", state)
    for k in range(8):
        a.lw("t1", "s8", 4 * k)
        _bswap(a, "a1", "t1", "t2", "t6")
        a.li("a0", k)
        a.li("t0", SYS_COMMIT)
        a.ecall()


def checksum(results: bytes) -> bytes:
    """public values of the result-heavy test guests: the 32-bit sum of their result words (the results themselves go to
    fd 1, so that the SHA-256 epilogue stays one block long)"""
    return struct.pack("<I", sum(w for (w,) in struct.iter_unpack("<I", results)) & M32)


def _finish(a, base, nbytes):
    """WRITE the nbytes of results at `base` to fd 1, commit their word sum as the public values, HALT(0)"""
    assert nbytes % 4 == 0 and nbytes
    uid = len(a.items)
    acc = a.dword(f"sum_{uid}", [0, 0])
    a.li("a0", 1)
    a.li("a1", base)
    a.li("a2", nbytes)
    a.li("t0", SYS_WRITE)
    a.ecall()
    a.li("s8", base)
    a.li("s9", base + nbytes)
    a.li("a5", 0)
    a.label(f"sum_{uid}")
    a.lw("a4", "s8", 0)
    a.add("a5", "a5", "a4")
    a.addi("s8", "s8", 4)
    a.bne("s8", "s9", f"sum_{uid}")
    a.li("s1", acc)
    a.sw("a5", "s1", 0)
    _write_pv(a, "s1", 4)
    a.halt(0)


def arith(commit=True):
    """Every provable instruction on a few operand pairs.  Returns (elf, expected result bytes): the results go to fd 1 and
    their checksum() to the public values (commit=False: they stay in memory and the guest halts without the SHA-256 /
    COMMIT epilogue — a short trace for the per-cell soundness tests)."""
    pairs = [(0, 0), (1, M32), (0x80000000, 1), (0x7FFFFFFF, 0x80000000), (0x12345678, 0x9ABCDEF0), (M32, M32), (5, 3), (3, 5)]
    a = Asm()
    out = a.dword("out", [0] * (len(pairs) * 16 + 4))
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
    if commit:
        _finish(a, out, 4 * len(exp))
    else:
        a.halt(0)
    return a.elf(), b"".join(struct.pack("<I", v) for v in exp)


SYS_SHA_EXTEND = 0x00300105


def sha_schedule_py(w16):
    """w[0..63] of the SHA-256 message schedule (what the SHA_EXTEND precompile computes in place)"""
    rotr = lambda v, n: ((v >> n) | (v << (32 - n))) & M32
    w = list(w16)
    for i in range(16, 64):
        s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)
        s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10)
        w.append((s1 + w[i - 7] + s0 + w[i - 16]) & M32)
    return w


def sha_extend(blocks=2, a1=0, ptr_off=0):
    """SP1's SHA_EXTEND precompile (syscall 0x00_30_01_05, a0 = pointer to 64 words, a1 = 0) on `blocks` arrays: the second
    array's first 16 words are the LAST 16 words the first call produced (memory written by one call is read by the next).
    Returns (elf, expected result bytes): the 64 x blocks words go to fd 1, their checksum() to the public values.
    a1 / ptr_off make the call invalid (it traps)."""
    first = [(0x6A09E667 * (i + 1) + 0x1F83D9AB * (i * i)) & M32 for i in range(16)]
    a = Asm()
    buf = a.dword("w", first + [0] * (64 * blocks - 16 + 4))
    exp = []
    cur = first
    for b in range(blocks):
        base = buf + 256 * b
        if b:
            # copy the previous array's last 16 words to the start of this one
            a.li("s1", base - 64)
            a.li("s2", base)
            for i in range(16):
                a.lw("a5", "s1", 4 * i)
                a.sw("a5", "s2", 4 * i)
        a.li("a0", base + ptr_off)
        a.li("a1", a1)
        a.li("t0", SYS_SHA_EXTEND)
        a.ecall()
        w = sha_schedule_py(cur)
        exp += w
        cur = w[48:]
    _finish(a, buf, 4 * len(exp))
    return a.elf(), b"".join(struct.pack("<I", v) for v in exp)


SYS_SHA_COMPRESS = 0x00010106


def sha256_precompiled(msg: bytes = bytes(range(150)), w_off=0, h_off=0):
    """SHA-256 of `msg` with SP1's two SHA precompiles, the way the reference's guests hash through the patched `sha2`
    crate (crates/dkg/Cargo.toml:22): per 64-byte block, SHA_EXTEND(w) then SHA_COMPRESS(w, state).  The padded message
    sits in the data segment as big-endian words.  Returns (elf, expected result bytes = the eight state words): they go to
    fd 1, their checksum() to the public values.  w_off / h_off move the pointers (invalid calls trap)."""
    import hashlib

    padded = msg + b"\x80" + b"\0" * ((55 - len(msg)) % 64) + struct.pack(">Q", 8 * len(msg))
    words = list(struct.unpack(">%dI" % (len(padded) // 4), padded))
    a = Asm()
    src = a.dword("msg", words + [0])
    w = a.dword("w", [0] * 68)
    st = a.dword("state", [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19, 0, 0, 0, 0])
    a.li("s3", src)
    a.li("s4", len(words) // 16)
    a.label("block")
    a.li("s2", w)
    for i in range(16):
        a.lw("a5", "s3", 4 * i)
        a.sw("a5", "s2", 4 * i)
    a.li("a0", w + w_off)
    a.li("a1", 0)
    a.li("t0", SYS_SHA_EXTEND)
    a.ecall()
    a.li("a0", w + w_off)
    a.li("a1", st + h_off)
    a.li("t0", SYS_SHA_COMPRESS)
    a.ecall()
    a.addi("s3", "s3", 64)
    a.addi("s4", "s4", -1)
    a.bne("s4", "zero", "block")
    _finish(a, st, 32)
    return a.elf(), b"".join(struct.pack("<I", v) for v in struct.unpack(">8I", hashlib.sha256(msg).digest()))


def commit_only(payload=b""):
    """The SP1 exit path alone: WRITE `payload` (a multiple of 4 bytes, from the data segment) to fd 3, hash it, COMMIT the
    eight digest words, HALT(0)."""
    assert len(payload) % 4 == 0
    a = Asm()
    src = a.dword("payload", [w for (w,) in struct.iter_unpack("<I", payload)] + [0])
    a.li("s1", src)
    _write_pv(a, "s1", len(payload))
    a.halt(0)
    return a.elf()


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
    _finish(a, out, 4 * len(exp))
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
    _finish(a, out, 4 * len(exp))
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
    _finish(a, out, 4 * len(exp))
    return a.elf(), b"".join(struct.pack("<I", v) for v in exp)


# ====================================================================== n-participant DKG-shaped guests
# Shape of the reference's guests (the ELFs themselves cannot be built here):
#   finalization (crates/finalization_prove/src/main.rs:8-32 -> crates/dkg/src/verification.rs:262-331): read ONE stdin
#   buffer (u64 length + CBOR), per generation a SHA-256 commitment hash and a signature check, n*k point operations,
#   an n^2 Lagrange term, then commit every base hash and the aggregate key.
#   bad_encrypted_share (crates/bad_encrypted_share_prove/src/main.rs:339-403): SHA-256 KDF, ChaCha20 decryption of
#   the share message (3 blocks for the 178-byte layout of :150-176), then the per-hash / per-key checks.
# Here the elliptic-curve work is stood in for by exact multi-limb multiply-accumulate chains (12 x 12 limbs = a
# 384-bit field-sized product, 4 x 4 for the scalar-sized n^2 term); SHA-256 and ChaCha20 are the real algorithms,
# n and k are parsed from the CBOR settings map of the real input.  Cycles(n, k) = n (c_sha + (SIG + k PT) c_12)
# + n^2 PAIR c_4 + c_exit(n), the reference's n (2 pairings + SHA) + n k scalar-mul + n^2.
DKG_L, DKG_L2 = 12, 4
DKG_A = [(0x9E3779B9 * (i + 1)) & M32 for i in range(DKG_L)]
DKG_B = [(0x85EBCA6B * (i + 3) + 7) & M32 for i in range(DKG_L)]
DKG_A2 = [(0xC2B2AE35 * (i + 5) + 1) & M32 for i in range(DKG_L2)]
DKG_B2 = [(0x27D4EB2F * (i + 2) + 3) & M32 for i in range(DKG_L2)]
PV_MAX_WORDS = (192 + 32 * 255 + 4 * (2 * DKG_L + 2) + 4 * (2 * DKG_L2 + 2) + 48 + 128) // 4
DKG_NC = 4      # curve_precompiles mode: the per-key point operations cycle through [2] G, [3] G, [4] G, [5] G


def _emit_mulacc(a):
    """mulacc(a0 = t, a1 = x, a2 = y, a3 = L): t += x * y exactly (t has 2L + 2 words: every carry is propagated)"""
    a.label("mulacc")
    a.mv("s7", "a3")
    a.mv("s1", "a2")
    a.mv("s2", "a0")
    a.mv("s3", "s7")
    a.label("ma_row")
    a.lw("t4", "s1", 0)
    a.mv("s4", "a1")
    a.mv("s5", "s2")
    a.mv("s6", "s7")
    a.li("a6", 0)
    a.label("ma_col")
    a.lw("a4", "s4", 0)
    a.mul("a5", "a4", "t4")
    a.mulhu("a7", "a4", "t4")
    a.lw("t1", "s5", 0)
    a.add("t1", "t1", "a5")
    a.sltu("t2", "t1", "a5")
    a.add("t1", "t1", "a6")
    a.sltu("t3", "t1", "a6")
    a.sw("t1", "s5", 0)
    a.add("a6", "a7", "t2")
    a.add("a6", "a6", "t3")
    a.addi("s4", "s4", 4)
    a.addi("s5", "s5", 4)
    a.addi("s6", "s6", -1)
    a.bne("s6", "zero", "ma_col")
    a.label("ma_carry")
    a.lw("t1", "s5", 0)
    a.add("t1", "t1", "a6")
    a.sltu("a6", "t1", "a6")
    a.sw("t1", "s5", 0)
    a.addi("s5", "s5", 4)
    a.bne("a6", "zero", "ma_carry")
    a.addi("s1", "s1", 4)
    a.addi("s2", "s2", 4)
    a.addi("s3", "s3", -1)
    a.bne("s3", "zero", "ma_row")
    a.ret()


def _emit_chacha(a):
    """chacha20_block(a0 = 16-word input state, a1 = 16-word output): RFC 8439 block function (10 double rounds + feed-forward)"""
    regs = ["a3", "a4", "a5", "a6", "a7", "t3", "t4", "t5", "t6", "s0", "s1", "s2", "s3", "s4", "s5", "s6"]

    def rotl(x, n):
        a.slli("t0", x, n)
        a.srli(x, x, 32 - n)
        a.or_(x, x, "t0")

    def qr(i, j, k, l):
        A, B, Cc, D = regs[i], regs[j], regs[k], regs[l]
        a.add(A, A, B); a.xor(D, D, A); rotl(D, 16)
        a.add(Cc, Cc, D); a.xor(B, B, Cc); rotl(B, 12)
        a.add(A, A, B); a.xor(D, D, A); rotl(D, 8)
        a.add(Cc, Cc, D); a.xor(B, B, Cc); rotl(B, 7)

    a.label("chacha20_block")
    for i, r_ in enumerate(regs):
        a.lw(r_, "a0", 4 * i)
    a.li("s7", 10)
    a.label("cc_round")
    qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
    qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    a.addi("s7", "s7", -1)
    a.bne("s7", "zero", "cc_round")
    for i, r_ in enumerate(regs):
        a.lw("t0", "a0", 4 * i)
        a.add(r_, r_, "t0")
        a.sw(r_, "a1", 4 * i)
    a.ret()


def chacha20_block_py(state):
    x = list(state)

    def rotl(v, n):
        return ((v << n) | (v >> (32 - n))) & M32

    def qr(i, j, k, l):
        x[i] = (x[i] + x[j]) & M32; x[l] = rotl(x[l] ^ x[i], 16)
        x[k] = (x[k] + x[l]) & M32; x[j] = rotl(x[j] ^ x[k], 12)
        x[i] = (x[i] + x[j]) & M32; x[l] = rotl(x[l] ^ x[i], 8)
        x[k] = (x[k] + x[l]) & M32; x[j] = rotl(x[j] ^ x[k], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(x[i] + state[i]) & M32 for i in range(16)]


def dkg_like(kind="finalization", sig_iters=2, pt_iters=1, pair_iters=1, sha_precompiles=False, curve_precompiles=False):
    """The n-participant DKG-shaped guest (see the block comment above).  kind = "finalization" | "encshare".  The ELF
    depends only on the iteration constants; n and k come from the stdin buffer at run time.  Returns the ELF;
    dkg_like_expected() is the same computation in Python.
    curve_precompiles=True: the k point operations of every participant are REAL BLS12-381 G1 arithmetic through SP1's
    BLS12381_ADD / _DOUBLE precompiles (what the reference's guests do through the patched bls12_381 crate, reference
    crates/dkg/Cargo.toml:25): Y := [i + 1] Y + C_j for j < k, the Horner step of reference crates/dkg/src/dkg_math.rs:160-174
    with the participant's id as the scalar; the compressed Y is appended to the public values.  (The signature check
    stays the multiply-accumulate stand-in: pairings need the Fp12 tower in guest code.)"""
    import hashlib  # noqa: F401  (the model below uses it; imported here to fail early if missing)

    assert kind in ("finalization", "encshare")
    L, L2 = DKG_L, DKG_L2
    a = Asm()
    a.sha_precompiles = sha_precompiles
    lib = None
    if curve_precompiles:
        from tests import guests_bls
        from tools import bls12_381 as bls

        lib = guests_bls.G1Lib(a)
        pw = lambda pt: guests_bls.words_of(pt[0], 12) + guests_bls.words_of(pt[1], 12) + [0]
        g_y = a.dword("g1_y", pw(bls.G1))
        g_tmp = a.dword("g1_tmp", [0] * 25)
        g_c = a.dword("g1_c", [w for m in range(2, 2 + DKG_NC) for w in pw(bls.E1.mul(bls.G1, m))])
        g_id = a.dword("g1_id", [0])
        g_out = a.dword("g1_out", [0] * 12)
    pa, pb = a.dword("x", DKG_A), a.dword("y", DKG_B)
    pt = a.dword("t", [0] * (2 * L + 2))
    pa2, pb2 = a.dword("x2", DKG_A2), a.dword("y2", DKG_B2)
    pt2 = a.dword("t2", [0] * (2 * L2 + 2))
    V = a.dword("vars", [0] * 12)
    V_LEN, V_N, V_K, V_I, V_J, V_S, V_R, V_PV, V_OFF = (V + 4 * i for i in range(9))
    chunk = a.dword("chunk", [0] * 16 + sha_padding(64))
    state = a.dword("hstate", [0] * 8)
    h0 = a.dword("h0", SHA_H0)
    cc_in = a.dword("cc_in", [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + [0] * 12)
    cc_out = a.dword("cc_out", [0] * 16)
    a.dword("align", [0] * ((-len(a.data)) % 16))        # the run-time padding loop below wants pvbuf 64-byte aligned
    pvbuf = a.dword("pvbuf", [0] * PV_MAX_WORDS)
    assert (pvbuf - a.data_base) % 64 == 0 and a.data_base % 64 == 0

    def load_var(reg, addr):
        a.li(reg, addr)
        a.lw(reg, reg, 0)

    def store_var(reg, addr, tmp="t6"):
        a.li(tmp, addr)
        a.sw(reg, tmp, 0)

    def copy_words(dst, src, n, tmp="t1"):
        """static addresses, n words, straight-line"""
        a.li("t5", src)
        a.li("t6", dst)
        for i in range(n):
            a.lw(tmp, "t5", 4 * i)
            a.sw(tmp, "t6", 4 * i)

    def sha_of_chunk():
        """hstate = SHA-256 state after hashing the 64 bytes in `chunk` (+ its static padding block)"""
        copy_words(state, h0, 8)
        a.li("a0", chunk)
        a.li("a1", 2)
        a.li("a2", state)
        a.call("sha256_blocks")

    a.j("main")
    emit_sha256(a)
    _emit_mulacc(a)
    if kind == "encshare":
        _emit_chacha(a)
    if lib:
        lib.emit()
    a.label("main")
    if lib:
        a.li("sp", lib.stack)
    # ---- read the one stdin buffer
    a.li("t0", SYS_HINT_LEN)
    a.ecall()
    a.mv("s1", "t0")
    store_var("s1", V_LEN)
    a.li("a0", HEAP)
    a.mv("a1", "s1")
    a.li("t0", SYS_HINT_READ)
    a.ecall()
    # ---- settings: scan the CBOR for map(3) + the text key "n" (0xa3 0x61 0x6e: the settings map {n, k, gen_id}; 0xa3
    # occurs in no hex string or key name), then "k" follows
    a.li("s2", HEAP + 7)
    a.label("scan")
    a.addi("s2", "s2", 1)
    a.lbu("t1", "s2", 0)
    a.lbu("t2", "s2", 1)
    a.lbu("t3", "s2", 2)
    a.addi("t1", "t1", -0xA3)
    a.addi("t2", "t2", -0x61)
    a.addi("t3", "t3", -0x6E)
    a.or_("t1", "t1", "t2")
    a.or_("t1", "t1", "t3")
    a.bne("t1", "zero", "scan")
    a.addi("s2", "s2", 2)              # s2 -> the 'n' byte
    for var, nxt in ((V_N, "got_n"), (V_K, "got_k")):
        a.lbu("t1", "s2", 1)           # s2 -> the key's second byte; the value follows (uint < 24 inline, else 0x18 + byte)
        a.addi("s2", "s2", 2)
        a.addi("t2", "t1", -0x18)
        a.bne("t2", "zero", nxt)
        a.lbu("t1", "s2", 0)
        a.addi("s2", "s2", 1)
        a.label(nxt)
        store_var("t1", var)
        a.addi("s2", "s2", 1)          # skip the 0x61 of the next key ("k")
    # ---- S = (LEN - 72) / n : stride of the per-participant 64-byte chunks
    load_var("t1", V_LEN)
    load_var("t2", V_N)
    a.addi("t1", "t1", -72)
    a.divu("t1", "t1", "t2")
    store_var("t1", V_S)
    a.li("t1", pvbuf)
    store_var("t1", V_PV)
    if kind == "encshare":
        # KDF: key = SHA-256(buf[8:72]) (stands for the hash of the ECDH point, main.rs:16-30,344), nonce = its first 12 bytes
        copy_words(chunk, HEAP + 8, 16)
        sha_of_chunk()
        a.li("s8", state)
        a.li("s9", cc_in)
        for k in range(8):
            a.lw("t1", "s8", 4 * k)
            _bswap(a, "t3", "t1", "t2", "t6")    # ChaCha words = little-endian words of the digest BYTES
            a.sw("t3", "s9", 16 + 4 * k)
            if k < 3:
                a.sw("t3", "s9", 52 + 4 * k)
        for blk in range(3):                     # decrypt buf[72 : 72 + 192] in place, append the plaintext to the public values
            a.li("s9", cc_in)
            a.li("t1", blk)
            a.sw("t1", "s9", 48)
            a.li("a0", cc_in)
            a.li("a1", cc_out)
            a.call("chacha20_block")
            a.li("s8", cc_out)
            a.li("s9", HEAP + 72 + 64 * blk)
            load_var("s10", V_PV)
            for w in range(16):
                a.lw("t1", "s8", 4 * w)
                a.lw("t2", "s9", 4 * w)
                a.xor("t1", "t1", "t2")
                a.sw("t1", "s9", 4 * w)
                a.sw("t1", "s10", 4 * w)
            a.addi("s10", "s10", 64)
            store_var("s10", V_PV)
    # ---- per participant
    a.li("t1", 0)
    store_var("t1", V_I)
    a.label("part")
    load_var("t1", V_I)
    load_var("t2", V_S)
    a.mul("t1", "t1", "t2")
    a.andi("t1", "t1", -4)
    a.li("t2", HEAP + 8)
    a.add("s8", "t1", "t2")                      # chunk_i = buf[8 + (i S & ~3) : +64]
    a.li("s9", chunk)
    for w in range(16):
        a.lw("t1", "s8", 4 * w)
        a.sw("t1", "s9", 4 * w)
    sha_of_chunk()                               # the commitment hash of generation i
    a.li("s8", state)
    load_var("s10", V_PV)
    a.li("s9", pb)
    for k in range(8):
        a.lw("t1", "s8", 4 * k)
        a.lw("t2", "s9", 4 * k)
        a.xor("t2", "t2", "t1")
        a.sw("t2", "s9", 4 * k)                  # y ^= digest words
        _bswap(a, "t3", "t1", "t2", "t6")
        a.sw("t3", "s10", 4 * k)                 # public values: the digest bytes (as the guest commits each base_hash)
    a.addi("s10", "s10", 32)
    store_var("s10", V_PV)
    # signature check + k point operations: (SIG + k PT) x { t += x * y ; x = t mod 2^384 }
    load_var("t1", V_K)
    a.li("t2", 0 if lib else pt_iters)
    a.mul("t1", "t1", "t2")
    a.li("t2", sig_iters)
    a.add("t1", "t1", "t2")
    store_var("t1", V_R)
    a.label("big")
    a.li("a0", pt); a.li("a1", pa); a.li("a2", pb); a.li("a3", L)
    a.call("mulacc")
    copy_words(pa, pt, L)
    load_var("t1", V_R)
    a.addi("t1", "t1", -1)
    store_var("t1", V_R)
    a.bne("t1", "zero", "big")
    if lib:
        # the k point operations of participant i on the curve: Y := [i + 1] Y + C_(j mod NC), j < k
        load_var("t1", V_I)
        a.addi("t1", "t1", 1)
        store_var("t1", g_id)
        a.li("t1", 0)
        store_var("t1", V_R)
        a.label("ptop")
        a.li("a0", g_tmp); a.li("a1", g_y); a.li("a2", g_id); a.li("a3", 1)
        a.call("scalar_mul")
        load_var("t1", V_R)
        a.andi("t1", "t1", DKG_NC - 1)
        a.li("t2", 100)
        a.mul("t1", "t1", "t2")
        a.li("a1", g_c)
        a.add("a1", "a1", "t1")
        a.li("a0", g_tmp)
        a.call("g1_add")
        a.li("t4", g_y)
        a.li("t5", g_tmp)
        lib.copy_words("t4", "t5", 25)
        load_var("t1", V_R)
        a.addi("t1", "t1", 1)
        store_var("t1", V_R)
        load_var("t2", V_K)
        a.bne("t1", "t2", "ptop")
    # the n^2 term: for every j: PAIR x { t2 += x2 * y2 ; x2 = t2 mod 2^128 } with y2[0] = digest word 0 ^ i, y2[1] = j
    a.li("s8", state)
    a.lw("t1", "s8", 0)
    load_var("t2", V_I)
    a.xor("t1", "t1", "t2")
    a.li("s9", pb2)
    a.sw("t1", "s9", 0)
    a.li("t1", 0)
    store_var("t1", V_J)
    a.label("pair")
    load_var("t1", V_J)
    a.li("s9", pb2)
    a.sw("t1", "s9", 4)
    for _ in range(pair_iters):
        a.li("a0", pt2); a.li("a1", pa2); a.li("a2", pb2); a.li("a3", L2)
        a.call("mulacc")
        copy_words(pa2, pt2, L2)
    load_var("t1", V_J)
    a.addi("t1", "t1", 1)
    store_var("t1", V_J)
    load_var("t2", V_N)
    a.bne("t1", "t2", "pair")
    load_var("t1", V_I)
    a.addi("t1", "t1", 1)
    store_var("t1", V_I)
    load_var("t2", V_N)
    a.bne("t1", "t2", "part")
    # ---- public values tail: t and t2 (the "aggregate key"), then SHA-256 padding written at run time (n is an input)
    load_var("s10", V_PV)
    for src, nw in ((pt, 2 * L + 2), (pt2, 2 * L2 + 2)):
        a.li("s8", src)
        for w in range(nw):
            a.lw("t1", "s8", 4 * w)
            a.sw("t1", "s10", 4 * w)
        a.addi("s10", "s10", 4 * nw)
    if lib:
        store_var("s10", V_PV)
        a.li("a0", g_out); a.li("a1", g_y)
        a.call("g1_compress")
        load_var("s10", V_PV)
        a.li("s8", g_out)
        for w in range(12):
            a.lw("t1", "s8", 4 * w)
            a.sw("t1", "s10", 4 * w)
        a.addi("s10", "s10", 48)
    a.li("s9", pvbuf)
    a.sub("s11", "s10", "s9")                    # T = public-value bytes
    a.li("a0", 3)
    a.li("a1", pvbuf)
    a.mv("a2", "s11")
    a.li("t0", SYS_WRITE)
    a.ecall()
    a.li("t1", 0x80)
    a.sw("t1", "s10", 0)
    a.addi("s10", "s10", 4)
    a.label("padz")                              # zero words until the address is 56 mod 64 (pvbuf is 64-byte aligned by construction below)
    a.andi("t1", "s10", 63)
    a.addi("t1", "t1", -56)
    a.beq("t1", "zero", "padded")
    a.sw("zero", "s10", 0)
    a.addi("s10", "s10", 4)
    a.j("padz")
    a.label("padded")
    a.sw("zero", "s10", 0)                       # bit length, big-endian 64-bit: high word 0
    a.slli("t1", "s11", 3)
    _bswap(a, "t3", "t1", "t2", "t6")
    a.sw("t3", "s10", 4)
    a.addi("s10", "s10", 8)
    a.li("s9", pvbuf)
    a.sub("a1", "s10", "s9")
    a.srli("a1", "a1", 6)
    a.li("a0", pvbuf)
    copy_words(state, h0, 8)
    a.li("a2", state)
    a.call("sha256_blocks")
    a.li("s8", state)
    for k in range(8):
        a.lw("t1", "s8", 4 * k)
        _bswap(a, "a1", "t1", "t2", "t6")
        a.li("a0", k)
        a.li("t0", SYS_COMMIT)
        a.ecall()
    a.halt(0)
    return a.elf()


def dkg_like_expected(stdin_buf: bytes, kind="finalization", sig_iters=2, pt_iters=1, pair_iters=1, curve_precompiles=False):
    """Python model of dkg_like: the public-value bytes the guest must commit for this input"""
    import hashlib

    L, L2 = DKG_L, DKG_L2
    buf = bytearray(stdin_buf + b"\0" * (-len(stdin_buf) % 4))
    LEN = len(stdin_buf)
    at = stdin_buf.find(b"\xa3\x61\x6e", 8) + 1
    assert at >= 1

    def uint(p):
        return (stdin_buf[p + 1], p + 2) if stdin_buf[p] == 0x18 else (stdin_buf[p], p + 1)

    n, p = uint(at + 2)
    assert stdin_buf[p:p + 2] == b"\x61\x6b"
    k, _ = uint(p + 2)
    S = (LEN - 72) // n
    x, y = sum(w << (32 * i) for i, w in enumerate(DKG_A)), sum(w << (32 * i) for i, w in enumerate(DKG_B))
    x2, y2w = sum(w << (32 * i) for i, w in enumerate(DKG_A2)), list(DKG_B2)
    t = t2 = 0
    pv = b""
    if curve_precompiles:
        from tools import bls12_381 as bls

        Y, Cs = bls.G1, [bls.E1.mul(bls.G1, m) for m in range(2, 2 + DKG_NC)]
    if kind == "encshare":
        key = hashlib.sha256(bytes(buf[8:72])).digest()
        kw = list(struct.unpack("<8I", key))
        for blk in range(3):
            ks = chacha20_block_py([0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + kw + [blk] + kw[:3])
            for w in range(16):
                o = 72 + 64 * blk + 4 * w
                v = struct.unpack_from("<I", buf, o)[0] ^ ks[w]
                struct.pack_into("<I", buf, o, v)
                pv += struct.pack("<I", v)
    for i in range(n):
        off = 8 + ((i * S) & ~3)
        d = hashlib.sha256(bytes(buf[off:off + 64])).digest()
        h = struct.unpack(">8I", d)
        y ^= sum(w << (32 * j) for j, w in enumerate(h))
        pv += d
        for _ in range(sig_iters + (0 if curve_precompiles else k * pt_iters)):
            t += x * y
            x = t & ((1 << (32 * L)) - 1)
        if curve_precompiles:
            for j in range(k):
                Y = bls.E1.add(bls.E1.mul(Y, i + 1), Cs[j % DKG_NC])
        y2w[0] = h[0] ^ i
        for j in range(n):
            y2w[1] = j
            y2 = sum(w << (32 * q) for q, w in enumerate(y2w))
            for _ in range(pair_iters):
                t2 += x2 * y2
                x2 = t2 & ((1 << (32 * L2)) - 1)
    assert t < 1 << (32 * (2 * L + 2)) and t2 < 1 << (32 * (2 * L2 + 2))
    pv += t.to_bytes(4 * (2 * L + 2), "little") + t2.to_bytes(4 * (2 * L2 + 2), "little")
    if curve_precompiles:
        pv += bls.g1_compress(Y)
    return pv


# ------------------------------------------------------------------------------------------------ field / curve precompiles
# SP1's syscall numbers as best recalled [EXTERNAL, unverified] (tools/airgen/rv32.py)
SYS_SECP256K1_ADD, SYS_SECP256K1_DOUBLE = 0x0001010A, 0x0000010B
SYS_BLS12381_ADD, SYS_BLS12381_DOUBLE = 0x0001011E, 0x0000011F
SYS_BLS12381_FP_ADD, SYS_BLS12381_FP_SUB, SYS_BLS12381_FP_MUL = 0x00010120, 0x00010121, 0x00010122
SYS_BLS12381_FP2_ADD, SYS_BLS12381_FP2_SUB, SYS_BLS12381_FP2_MUL = 0x00010123, 0x00010124, 0x00010125
SYS_UINT256_MUL = 0x0001011D
BLS_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
BLS_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
SECP_P = (1 << 256) - (1 << 32) - 977
SECP_G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
BLS_G1 = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
          0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)


def words_of(v, n):
    return [(v >> (32 * i)) & M32 for i in range(n)]


def _syscall(a, code, a0, a1):
    a.li("a0", a0)
    a.li("a1", a1)
    a.li("t0", code)
    a.ecall()


def ec_add(p, q, mod):
    """affine addition on y^2 = x^3 + b (a = 0), p != +-q, no point at infinity: plain Python restatement"""
    (x1, y1), (x2, y2) = p, q
    lam = (y2 - y1) * pow(x2 - x1, -1, mod) % mod
    x3 = (lam * lam - x1 - x2) % mod
    return x3, (lam * (x1 - x3) - y1) % mod


def ec_double(p, mod):
    x1, y1 = p
    lam = 3 * x1 * x1 * pow(2 * y1, -1, mod) % mod
    x3 = (lam * lam - 2 * x1) % mod
    return x3, (lam * (x1 - x3) - y1) % mod


def field_ops(bad=None):
    """BLS12-381 Fp and Fp2 add / sub / mul precompiles on reduced, unreduced and aliased operands.  Returns
    (elf, expected result bytes): every result (12 or 24 words) goes to fd 1, their checksum() to the public values.
    bad: "misaligned" / "low" / "high" pointer -> the call traps."""
    p = BLS_P
    top = (1 << 384) - 1
    vals = [0, 1, p - 1, p, p + 5, top, 0x1234567 << 300 | 0xABCDEF, (p * 7) // 11, top - 12345, 2, (p + 1) // 2]
    a = Asm()
    cases = []     # (code, x words, y words or None for y = x (same pointer), expected words)
    f = {0: lambda x, y: (x + y) % p, 1: lambda x, y: (x - y) % p, 2: lambda x, y: x * y % p}
    k = 0
    for op in (0, 1, 2):
        for i in range(4):
            x, y = vals[(k * 3 + 1) % len(vals)], vals[(k * 5 + 2) % len(vals)]
            k += 1
            cases.append((SYS_BLS12381_FP_ADD + op, words_of(x, 12), words_of(y, 12), words_of(f[op](x, y), 12)))
        x = vals[(k + 5) % len(vals)]
        cases.append((SYS_BLS12381_FP_ADD + op, words_of(x, 12), None, words_of(f[op](x, x), 12)))       # x op x through ONE pointer
    for op in (0, 1, 2):
        for i in range(3):
            x0, x1, y0, y1 = (vals[(k * 7 + j) % len(vals)] for j in (1, 4, 6, 9))
            k += 1
            if op == 2:
                r0, r1 = (x0 * y0 - x1 * y1) % p, (x0 * y1 + x1 * y0) % p
            else:
                r0, r1 = f[op](x0, y0), f[op](x1, y1)
            cases.append((SYS_BLS12381_FP2_ADD + op, words_of(x0, 12) + words_of(x1, 12), words_of(y0, 12) + words_of(y1, 12), words_of(r0, 12) + words_of(r1, 12)))
        x0, x1 = vals[(k + 2) % len(vals)], vals[(k + 7) % len(vals)]
        r = ((x0 * x0 - x1 * x1) % p, 2 * x0 * x1 % p) if op == 2 else (f[op](x0, x0), f[op](x1, x1))
        cases.append((SYS_BLS12381_FP2_ADD + op, words_of(x0, 12) + words_of(x1, 12), None, words_of(r[0], 12) + words_of(r[1], 12)))
    exp = []
    nres = sum(len(c[3]) for c in cases)
    out = a.dword("out", [0] * (nres + 4))
    at = out
    for n_, (code, xw, yw, rw) in enumerate(cases):
        # the operand that is replaced lives in the output area (so the result lands where _finish reads it)
        src = a.dword(f"x{n_}", xw)
        a.li("s1", src)
        a.li("s2", at)
        for i in range(len(xw)):
            a.lw("a5", "s1", 4 * i)
            a.sw("a5", "s2", 4 * i)
        yp = a.dword(f"y{n_}", yw) if yw is not None else at
        xp = at
        if bad and n_ == 0:
            xp = {"misaligned": at + 2, "low": 16, "high": 0x38000000 - 8}[bad]
        _syscall(a, code, xp, yp)
        at += 4 * len(xw)
        exp += rw
    _finish(a, out, 4 * nres)
    return a.elf(), b"".join(struct.pack("<I", v) for v in exp)


def curve_ops(bad=None):
    """BLS12-381 G1 and secp256k1 affine ADD / DOUBLE precompiles: 2G, 3G = 2G + G, 4G = 2(2G), 5G = 4G + G, 7G = 5G + 2G
    on both curves.  Returns (elf, expected result bytes): the points (x || y little-endian words) go to fd 1.
    bad: "equal" (ADD of a point to itself), "unreduced" (x + p), "a1" (DOUBLE with a1 != 0) -> the call traps."""
    a = Asm()
    exp = []
    plan = []
    for name, G, mod, W, c_add, c_dbl in (("bls", BLS_G1, BLS_P, 12, SYS_BLS12381_ADD, SYS_BLS12381_DOUBLE), ("secp", SECP_G, SECP_P, 8, SYS_SECP256K1_ADD, SYS_SECP256K1_DOUBLE)):
        g2 = ec_double(G, mod)
        g3 = ec_add(g2, G, mod)
        g4 = ec_double(g2, mod)
        g5 = ec_add(g4, G, mod)
        g7 = ec_add(g5, g2, mod)
        plan.append((name, W, c_add, c_dbl, G, [g2, g3, g4, g5, g7], mod))
    nres = sum(2 * W * 5 for _, W, *_ in plan)
    out = a.dword("out", [0] * (nres + 4))
    at = out
    for name, W, c_add, c_dbl, G, pts, mod in plan:
        pw = lambda pt: words_of(pt[0], W) + words_of(pt[1], W)
        gaddr = a.dword(f"{name}_g", pw(G))
        slots = [at + 8 * W * i for i in range(5)]          # 2G, 3G, 4G, 5G, 7G in the output area

        def copy(dst, src):
            a.li("s1", src)
            a.li("s2", dst)
            for i in range(2 * W):
                a.lw("a5", "s1", 4 * i)
                a.sw("a5", "s2", 4 * i)
        copy(slots[0], gaddr)
        _syscall(a, c_dbl, slots[0], 4 if bad == "a1" and name == "bls" else 0)                 # 2G
        copy(slots[1], slots[0])
        if bad == "unreduced" and name == "bls":
            bogus = a.dword("bogus", words_of(G[0] + mod, W) + words_of(G[1], W))
            _syscall(a, c_add, slots[1], bogus)
        _syscall(a, c_add, slots[1], slots[1] if bad == "equal" and name == "bls" else gaddr)    # 3G = 2G + G
        copy(slots[2], slots[0])
        _syscall(a, c_dbl, slots[2], 0)                      # 4G
        copy(slots[3], slots[2])
        _syscall(a, c_add, slots[3], gaddr)                  # 5G
        copy(slots[4], slots[3])
        _syscall(a, c_add, slots[4], slots[0])               # 7G = 5G + 2G
        for pt in pts:
            exp += pw(pt)
        at += 4 * 2 * W * 5
    _finish(a, out, 4 * nres)
    return a.elf(), b"".join(struct.pack("<I", v) for v in exp)


def u256_ops(bad=None):
    """UINT256_MUL (a0 = x, a1 = y followed by the modulus): x := x * y mod m on the BLS12-381 scalar field, on an even
    modulus, on m = 1, on m = 0 (= 2^256, SP1's convention) and with x = y through aliasing-free copies.  Returns
    (elf, expected result bytes).  bad = "misaligned": the call traps."""
    top = (1 << 256) - 1
    cases = [(3, 5, BLS_R), (BLS_R - 1, BLS_R - 1, BLS_R), (top, top, BLS_R), (top, top, 0), (0x1234 << 200, 77, 1 << 255),
             (top - 5, 2, top), (12345, 67890, 1), (top, top, (1 << 256) - 189), (1 << 128, 1 << 128, 0), (5, 0, 7)]
    a = Asm()
    out = a.dword("out", [0] * (8 * len(cases) + 4))
    exp = []
    for n_, (x, y, m) in enumerate(cases):
        src = a.dword(f"x{n_}", words_of(x, 8))
        ym = a.dword(f"ym{n_}", words_of(y, 8) + words_of(m, 8))
        at = out + 32 * n_
        a.li("s1", src)
        a.li("s2", at)
        for i in range(8):
            a.lw("a5", "s1", 4 * i)
            a.sw("a5", "s2", 4 * i)
        _syscall(a, SYS_UINT256_MUL, at + (2 if bad == "misaligned" and n_ == 0 else 0), ym)
        exp += words_of(x * y % (m or 1 << 256), 8)
    _finish(a, out, 32 * len(cases))
    return a.elf(), b"".join(struct.pack("<I", v) for v in exp)
