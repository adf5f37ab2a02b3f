"""BLS12-381 in plain Python big integers — TEST TOOLING: what the synthetic DKG input
generator (tools/gen_dkg_input.py) and the Python restatement of the reference's finalization check
(tools/dkg_verify.py) need: G1 / G2 arithmetic with the zcash compressed encodings, the RFC 9380 hash-to-G2 suite the
reference uses (BLS12381G2_XMD:SHA-256_SSWU_RO_, DST "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_",
reference crates/dkg/src/crypto/bls_common.rs:11-24), and the pairing for signature self-checks (:26-40).

Pinned by the reference's own known-answer tests (tests/test_bls_tooling.py): the signature KAT of
crates/dkg/src/dkg_math.rs:258-278 / crypto/bls_common.rs:134-159, the Horner and Lagrange KATs of dkg_math.rs:281-431,
and the reference's real finalization vectors.  Nothing is copied from the reference (it delegates curve arithmetic
to the bls12_381 crate); constants that could not be trusted from memory are DERIVED here: the 3-isogeny E2' -> E2 of
the SSWU suite comes from Velu's formulas on the 3-torsion point x0 = -6 + 6i of E2', cofactor clearing uses the
psi endomorphism with computed Frobenius coefficients (RFC 9380 appendix G.4), and the module checks at import that
the generators lie on their curves and have order r.

Slow by design (a pairing takes seconds): never on the product path.
"""
import hashlib

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
X_PARAM = -0xD201000000010000          # the BLS parameter x (negative)


# ------------------------------------------------------------------------------------------------ Fp2 = Fp[i]/(i^2 + 1)
class Fp2:
    __slots__ = ("a", "b")

    def __init__(self, a, b=0):
        self.a, self.b = a % P, b % P

    def __add__(self, o):
        return Fp2(self.a + o.a, self.b + o.b)

    def __sub__(self, o):
        return Fp2(self.a - o.a, self.b - o.b)

    def __neg__(self):
        return Fp2(-self.a, -self.b)

    def __mul__(self, o):
        if isinstance(o, int):
            return Fp2(self.a * o, self.b * o)
        return Fp2(self.a * o.a - self.b * o.b, self.a * o.b + self.b * o.a)

    __rmul__ = __mul__

    def __eq__(self, o):
        return isinstance(o, Fp2) and self.a == o.a and self.b == o.b

    def __hash__(self):
        return hash((self.a, self.b))

    def sq(self):
        return self * self

    def conj(self):
        return Fp2(self.a, -self.b)

    def inv(self):
        n = pow(self.a * self.a + self.b * self.b, P - 2, P)
        return Fp2(self.a * n, -self.b * n)

    def is_zero(self):
        return self.a == 0 and self.b == 0

    def pow(self, e):
        r, x = Fp2(1), self
        while e:
            if e & 1:
                r = r * x
            x = x * x
            e >>= 1
        return r

    def is_square(self):
        return pow(self.a * self.a + self.b * self.b, (P - 1) // 2, P) in (0, 1)     # the norm is a square in Fp

    def sqrt(self):
        """a square root, or None (p = 3 mod 4: complex method)"""
        if self.is_zero():
            return Fp2(0)
        a1 = self.pow((P - 3) // 4)
        alpha = a1 * a1 * self
        x0 = a1 * self
        if alpha == Fp2(-1):
            r = Fp2(0, 1) * x0
        else:
            r = (alpha + Fp2(1)).pow((P - 1) // 2) * x0
        return r if r * r == self else None

    def __repr__(self):
        return "Fp2(%#x, %#x)" % (self.a, self.b)


def fp_sqrt(v):
    r = pow(v, (P + 1) // 4, P)
    return r if r * r % P == v % P else None


# ------------------------------------------------------------------------------------------------ curves (affine, None = infinity)
class Curve:
    """y^2 = x^3 + a x + b over Fp (ints) or Fp2"""

    def __init__(self, a, b, zero, one):
        self.a, self.b, self.zero, self.one = a, b, zero, one

    def inv(self, v):
        return pow(v, P - 2, P) if isinstance(v, int) else v.inv()

    def red(self, v):
        return v % P if isinstance(v, int) else v

    def on_curve(self, pt):
        if pt is None:
            return True
        x, y = pt
        return self.red(y * y - (x * x * x + self.a * x + self.b)) == self.zero

    def neg(self, pt):
        return None if pt is None else (pt[0], self.red(-pt[1]))

    def add(self, p1, p2):
        if p1 is None:
            return p2
        if p2 is None:
            return p1
        x1, y1 = p1
        x2, y2 = p2
        if x1 == x2:
            if self.red(y1 + y2) == self.zero:
                return None
            m = self.red((3 * x1 * x1 + self.a) * self.inv(self.red(2 * y1)))
        else:
            m = self.red((y2 - y1) * self.inv(self.red(x2 - x1)))
        x3 = self.red(m * m - x1 - x2)
        return (x3, self.red(m * (x1 - x3) - y1))

    def mul(self, pt, k):
        if k < 0:
            return self.mul(self.neg(pt), -k)
        acc, q = None, pt
        while k:
            if k & 1:
                acc = self.add(acc, q)
            q = self.add(q, q)
            k >>= 1
        return acc


E1 = Curve(0, 4, 0, 1)
E2 = Curve(Fp2(0), Fp2(4, 4), Fp2(0), Fp2(1))
G1 = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
      0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)
G2 = (Fp2(0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
          0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
      Fp2(0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
          0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))
assert E1.on_curve(G1) and E2.on_curve(G2), "generator constants are wrong"
assert E1.mul(G1, R) is None and E2.mul(G2, R) is None, "generators do not have order r"


# ------------------------------------------------------------------------------------------------ zcash encodings
def _lex_largest_fp(y):
    return y > (P - 1) // 2


def _lex_largest_fp2(y):
    return _lex_largest_fp(y.b) if y.b else _lex_largest_fp(y.a)


def g1_compress(pt) -> bytes:
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if _lex_largest_fp(y) else 0)
    return bytes(b)


def g1_decompress(data: bytes, check_subgroup=True):
    """-> point, or raises ValueError (what G1Affine::from_compressed rejects)"""
    if len(data) != 48 or not data[0] & 0x80:
        raise ValueError("not a compressed G1 encoding")
    if data[0] & 0x40:
        if any(data[1:]) or data[0] & 0x3F:
            raise ValueError("non-canonical infinity")
        return None
    x = int.from_bytes(bytes([data[0] & 0x1F]) + data[1:], "big")
    if x >= P:
        raise ValueError("x not in the field")
    y = fp_sqrt((x * x * x + 4) % P)
    if y is None:
        raise ValueError("not on the curve")
    if _lex_largest_fp(y) != bool(data[0] & 0x20):
        y = P - y
    if check_subgroup and E1.mul((x, y), R) is not None:
        raise ValueError("not in the subgroup")
    return (x, y)


def g2_compress(pt) -> bytes:
    if pt is None:
        return bytes([0xC0]) + bytes(95)
    x, y = pt
    b = bytearray(x.b.to_bytes(48, "big") + x.a.to_bytes(48, "big"))
    b[0] |= 0x80 | (0x20 if _lex_largest_fp2(y) else 0)
    return bytes(b)


def g2_decompress(data: bytes, check_subgroup=True):
    if len(data) != 96 or not data[0] & 0x80:
        raise ValueError("not a compressed G2 encoding")
    if data[0] & 0x40:
        if any(data[1:]) or data[0] & 0x3F:
            raise ValueError("non-canonical infinity")
        return None
    x1 = int.from_bytes(bytes([data[0] & 0x1F]) + data[1:48], "big")
    x0 = int.from_bytes(data[48:], "big")
    if x0 >= P or x1 >= P:
        raise ValueError("x not in the field")
    x = Fp2(x0, x1)
    y = (x * x * x + Fp2(4, 4)).sqrt()
    if y is None:
        raise ValueError("not on the curve")
    if _lex_largest_fp2(y) != bool(data[0] & 0x20):
        y = -y
    if check_subgroup and E2.mul((x, y), R) is not None:
        raise ValueError("not in the subgroup")
    return (x, y)


# ------------------------------------------------------------------------------------------------ hash to G2 (RFC 9380)
DST_POP = b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_"


def expand_message_xmd(msg: bytes, dst: bytes, n: int) -> bytes:
    ell = -(-n // 32)
    assert ell <= 255 and len(dst) <= 255
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(bytes(64) + msg + n.to_bytes(2, "big") + b"\0" + dst_prime).digest()
    bi = hashlib.sha256(b0 + b"\x01" + dst_prime).digest()
    out = bi
    for i in range(2, ell + 1):
        bi = hashlib.sha256(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
        out += bi
    return out[:n]


def hash_to_field_fp2(msg: bytes, dst: bytes, count=2):
    L = 64
    u = expand_message_xmd(msg, dst, count * 2 * L)
    return [Fp2(int.from_bytes(u[L * (2 * i):L * (2 * i + 1)], "big"), int.from_bytes(u[L * (2 * i + 1):L * (2 * i + 2)], "big")) for i in range(count)]


# the isogenous curve E2' : y^2 = x^3 + 240 i x + 1012 (1 + i), Z = -(2 + i)
ISO_A, ISO_B, SSWU_Z = Fp2(0, 240), Fp2(1012, 1012), Fp2(-2, -1)
E2P = Curve(ISO_A, ISO_B, Fp2(0), Fp2(1))


def sgn0(x: Fp2) -> int:
    return (x.a & 1) | ((x.a == 0) & (x.b & 1))


def map_to_curve_sswu(u: Fp2):
    """simplified SWU onto E2' (RFC 9380 section 6.6.2)"""
    z_u2 = SSWU_Z * u * u
    tv1 = z_u2 * z_u2 + z_u2
    if tv1.is_zero():
        x1 = ISO_B * (SSWU_Z * ISO_A).inv()
    else:
        x1 = (-ISO_B) * ISO_A.inv() * (Fp2(1) + tv1.inv())
    gx1 = x1 * x1 * x1 + ISO_A * x1 + ISO_B
    if gx1.is_square():
        x, y = x1, gx1.sqrt()
    else:
        x = z_u2 * x1
        y = (x * x * x + ISO_A * x + ISO_B).sqrt()
    assert y is not None
    if sgn0(u) != sgn0(y):
        y = -y
    return (x, y)


def _derive_isogeny():
    """The 3-isogeny E2' -> E2 of the suite, from Velu's formulas.  Kernel {O, +-Q} with x(Q) = x0: the codomain
    y^2 = x^3 + (A - 5t) x + (B - 7w) must have a-coefficient 0, i.e. t = A / 5 = 2 (3 x0^2 + A), which gives x0^2 = -72 i:
    x0 = -6 + 6i (the root for which the codomain is E2 up to the scaling below; the import-time check and the reference's
    signature KAT settle the choice of root, of lambda^2 and of the sign of y)."""
    x0 = None
    for cand in (Fp2(-6, 6), Fp2(6, -6)):       # the two square roots of -72 i; the kernel abscissa is the 3-torsion one
        psi3 = cand.pow(4) * 3 + ISO_A * cand * cand * 6 + ISO_B * cand * 12 - ISO_A * ISO_A
        if cand * cand == Fp2(0, -72) and psi3.is_zero():
            x0 = cand
    assert x0 is not None, "no 3-torsion abscissa with x0^2 = -72 i"
    t = (x0 * x0 * 3 + ISO_A) * 2
    assert t * 5 == ISO_A
    u = (x0 * x0 * x0 + ISO_A * x0 + ISO_B) * 4
    w = u + x0 * t
    b2 = ISO_B - w * 7                      # Velu codomain: Y^2 = X^3 + b2
    # scale (X, Y) -> (X / l2, Y / l3) with l2^3 = l3^2 = b2 / (4 + 4i)
    ratio = b2 * Fp2(4, 4).inv()
    return x0, t, u, ratio


_ISO_X0, _ISO_T, _ISO_U, _ISO_RATIO = _derive_isogeny()


def _cube_roots(v: Fp2):
    """all cube roots of v in Fp2.  p^2 - 1 = 3^s m with 3 not dividing m: x = v^(1/3 mod m) satisfies x^3 = v e with e in the
    (cyclic, tiny) 3-Sylow subgroup, which is searched exhaustively for the correction."""
    m, s_ = P * P - 1, 0
    while m % 3 == 0:
        m //= 3
        s_ += 1
    k = 2
    while True:                                  # a generator c of the 3-Sylow subgroup
        c = Fp2(k, 1).pow(m)
        if not (c.pow(3 ** (s_ - 1)) == Fp2(1)):
            break
        k += 1
    x = v.pow(pow(3, -1, m))
    e = x * x * x * v.inv()
    roots, d = [], Fp2(1)
    for _ in range(3 ** s_):
        if d * d * d * e == Fp2(1):
            roots.append(x * d)
        d = d * c
    return roots


def _choose_scaling():
    """lambda^2 (l2) and lambda^3 (l3) with l2^3 = l3^2 = ratio; among the 3 x 2 choices the suite's map is the one whose
    leading x-coefficient 1 / l2 equals RFC 9380's k_(1,3) (a value in Fp: 0x171d...5ed1)"""
    cands = []
    for l2 in _cube_roots(_ISO_RATIO):
        l3 = _ISO_RATIO.sqrt()
        if l3 is None:
            continue
        for sgn in (l3, -l3):
            cands.append((l2, sgn))
    return cands


_ISO_CANDS = _choose_scaling()
_iso_choice = [None]


def iso_map(pt, choice=None):
    """E2' -> E2"""
    if pt is None:
        return None
    l2, l3 = choice if choice is not None else _iso_choice[0]
    x, y = pt
    d = x - _ISO_X0
    if d.is_zero():
        return None
    di = d.inv()
    di2 = di * di
    X = x + _ISO_T * di + _ISO_U * di2
    Y = y * (Fp2(1) - _ISO_T * di2 - _ISO_U * 2 * di2 * di)
    return (X * l2.inv(), Y * l3.inv())


# psi = twist^-1 o Frobenius o twist on E2 and its square (RFC 9380 appendix G.3)
_PSI_CX = Fp2(1, 1).pow((P - 1) // 3).inv()
_PSI_CY = Fp2(1, 1).pow((P - 1) // 2).inv()
_PSI2_CX = Fp2(pow(2, (P - 1) // 3, P)).inv()


def psi(pt):
    return None if pt is None else (_PSI_CX * pt[0].conj(), _PSI_CY * pt[1].conj())


def psi2(pt):
    return None if pt is None else (_PSI2_CX * pt[0], -pt[1])


def clear_cofactor_g2(pt):
    """[x^2 - x - 1] P + [x - 1] psi(P) + psi^2(2 P)  (= h_eff * P, RFC 9380 appendix G.4)"""
    c1 = X_PARAM
    t1 = E2.mul(pt, c1)
    t2 = psi(pt)
    t3 = psi2(E2.add(pt, pt))
    t3 = E2.add(t3, E2.neg(t2))
    t2 = E2.add(t1, t2)
    t2 = E2.mul(t2, c1)
    t3 = E2.add(t3, t2)
    t3 = E2.add(t3, E2.neg(t1))
    return E2.add(t3, E2.neg(pt))


def hash_to_g2(msg: bytes, dst: bytes = DST_POP, choice=None):
    u0, u1 = hash_to_field_fp2(msg, dst, 2)
    q0, q1 = iso_map(map_to_curve_sswu(u0), choice), iso_map(map_to_curve_sswu(u1), choice)
    return clear_cofactor_g2(E2.add(q0, q1))


# ------------------------------------------------------------------------------------------------ pairing (Fp12 = Fp[w]/(w^12 - 2 w^6 + 2))
class Fp12:
    """w^6 = 1 + i, so Fp2 = {u + v i} embeds as (u - v) + v w^6"""
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = [v % P for v in c]

    @staticmethod
    def one():
        return Fp12([1] + [0] * 11)

    def __mul__(self, o):
        t = [0] * 23
        for i, a in enumerate(self.c):
            if a:
                for j, b in enumerate(o.c):
                    if b:
                        t[i + j] += a * b
        for k in range(22, 11, -1):          # w^k = 2 w^(k-6) - 2 w^(k-12)
            v = t[k]
            if v:
                t[k - 6] += 2 * v
                t[k - 12] -= 2 * v
        return Fp12(t[:12])

    def __eq__(self, o):
        return self.c == o.c

    def pow(self, e):
        r, x = Fp12.one(), self
        while e:
            if e & 1:
                r = r * x
            x = x * x
            e >>= 1
        return r


def _line_value(lam: Fp2, r2, p1) -> Fp12:
    """The line of slope lam through r2 on E2, evaluated at the G1 point p1, scaled by w^3 (an element of a proper
    subfield: the final exponentiation removes it).  Through the twist (x, y) -> (x / w^2, y / w^3) the value is
    (y_r - lam x_r) + (lam x_p) w^2 - y_p w^3."""
    xr, yr = r2
    a = yr - lam * xr
    b = lam * p1[0]
    c = [0] * 12
    c[0], c[6] = a.a - a.b, a.b
    c[2], c[8] = b.a - b.b, b.b
    c[3] = -p1[1]
    return Fp12(c)


def miller_loop(q2, p1) -> Fp12:
    """q2 in G2 (Fp2 coordinates), p1 in G1; ate loop over |x| (the common inversion cancels in equality checks).
    Point arithmetic stays in E2(Fp2); only the line values live in Fp12."""
    if q2 is None or p1 is None:
        return Fp12.one()
    r2, f = q2, Fp12.one()
    n = -X_PARAM
    for i in range(n.bit_length() - 2, -1, -1):
        lam = (r2[0] * r2[0] * 3) * (r2[1] * 2).inv()
        f = f * f * _line_value(lam, r2, p1)
        r2 = E2.add(r2, r2)
        if n >> i & 1:
            lam = (q2[1] - r2[1]) * (q2[0] - r2[0]).inv()
            f = f * _line_value(lam, r2, p1)
            r2 = E2.add(r2, q2)
    return f


def final_exponentiation(f: Fp12) -> Fp12:
    return f.pow((P ** 12 - 1) // R)


def pairing_product_is_one(pairs) -> bool:
    """prod e(P_i, Q_i) == 1 for pairs (g1 point, g2 point)"""
    f = Fp12.one()
    for p1, q2 in pairs:
        f = f * miller_loop(q2, p1)
    return final_exponentiation(f) == Fp12.one()


def bls_verify(pk, sig, msg: bytes) -> bool:
    """e(pk, H(msg)) == e(g1, sig)  (reference crypto/bls_common.rs:26-40)"""
    return pairing_product_is_one([(pk, hash_to_g2(msg)), (E1.neg(G1), sig)])


def select_isogeny_by_kat(pk, sig, msg: bytes) -> bool:
    """Fix the remaining choice of the derived isogeny (3 cube roots x 2 signs) with one known-good signature: exactly one
    candidate maps onto E2 in a way that makes the reference's signature KAT verify.  Returns True when one was found."""
    u0, _ = hash_to_field_fp2(msg, DST_POP, 2)
    probe = map_to_curve_sswu(u0)
    for cand in _ISO_CANDS:
        img = iso_map(probe, cand)
        if img is None or not E2.on_curve(img):
            continue
        if pairing_product_is_one([(pk, hash_to_g2(msg, DST_POP, cand)), (E1.neg(G1), sig)]):
            _iso_choice[0] = cand
            return True
    return False


# the reference's signature KAT (crates/dkg/src/dkg_math.rs:258-266) fixes the isogeny choice once per process
KAT_MSG = bytes.fromhex("2f901d5cec8722e44afd59e94d0a56bf1506a72a0a60709920aad714d1a2ece0")
KAT_PK = bytes.fromhex("90346f9c5f3c09d96ea02acd0220daa8459f03866ed938c798e3716e42c7e033c9a7ef66a10f83af06d5c00b508c6d0f")
KAT_SIG = bytes.fromhex("a9c08eff13742f78f1e5929888f223b5b5b12b4836b5417c5a135cf24f4e2a4c66a6cdef91be3098b7e7a6a63903b61302e3cf2b8653101da245cf01a8d82b25"
                        "debe7b18a3a2eb1778f8628fd2c59c8687f6e048a31250fbc2804c20043b8443")


def ensure_ready():
    if _iso_choice[0] is None:
        if not select_isogeny_by_kat(g1_decompress(KAT_PK), g2_decompress(KAT_SIG), KAT_MSG):
            raise RuntimeError("no candidate of the derived 3-isogeny makes the reference's signature KAT verify")


# ------------------------------------------------------------------------------------------------ scalars
def scalar_from_be(b: bytes) -> int:
    return int.from_bytes(b, "big") % R


def sign(sk: int, msg: bytes):
    ensure_ready()
    return E2.mul(hash_to_g2(msg), sk % R)
