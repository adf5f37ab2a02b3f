"""A small symbolic AIR DSL: chips are described once (columns, polynomial
constraints, LogUp interactions) and tools/airgen/emit.py turns the description
into (a) C++ templates shared by the gfx950 quotient / permutation kernels and the
host verifier of the product path and (b) plain C for the CPU oracle.

Nothing here is derived from reference source: the reference delegates its AIRs
to the absent sp1-core-machine crate; the chip set is an
original design (DESIGN.md section "AIR").
"""
from __future__ import annotations

P = 2013265921


class Expr:
    """Hash-consed polynomial expression over F_p."""

    _table = {}
    __slots__ = ("op", "args", "deg", "id")

    def __new__(cls, op, args):
        key = (op, args)
        e = Expr._table.get(key)
        if e is None:
            e = object.__new__(cls)
            e.op, e.args = op, args
            e.id = len(Expr._table)
            if op == "const":
                e.deg = 0
            elif op == "var":
                kind = args[0]
                e.deg = 0 if kind in ("pub",) else 1
            elif op in ("add", "sub"):
                e.deg = max(args[0].deg, args[1].deg)
            elif op == "neg":
                e.deg = args[0].deg
            elif op == "mul":
                e.deg = args[0].deg + args[1].deg
            Expr._table[key] = e
        return e

    # ---- construction helpers
    @staticmethod
    def const(v):
        return Expr("const", (int(v) % P,))

    @staticmethod
    def var(kind, idx, rot=0):
        return Expr("var", (kind, idx, rot))

    @staticmethod
    def wrap(x):
        return x if isinstance(x, Expr) else Expr.const(x)

    def is_const(self):
        return self.op == "const"

    def __add__(self, o):
        o = Expr.wrap(o)
        if self.is_const() and o.is_const():
            return Expr.const(self.args[0] + o.args[0])
        if self.is_const() and self.args[0] == 0:
            return o
        if o.is_const() and o.args[0] == 0:
            return self
        return Expr("add", (self, o))

    __radd__ = lambda self, o: Expr.wrap(o) + self

    def __sub__(self, o):
        o = Expr.wrap(o)
        if self.is_const() and o.is_const():
            return Expr.const(self.args[0] - o.args[0])
        if o.is_const() and o.args[0] == 0:
            return self
        if self is o:
            return Expr.const(0)
        return Expr("sub", (self, o))

    __rsub__ = lambda self, o: Expr.wrap(o) - self

    def __neg__(self):
        if self.is_const():
            return Expr.const(-self.args[0])
        return Expr("neg", (self,))

    def __mul__(self, o):
        o = Expr.wrap(o)
        if self.is_const() and o.is_const():
            return Expr.const(self.args[0] * o.args[0])
        for a, b in ((self, o), (o, self)):
            if a.is_const():
                if a.args[0] == 0:
                    return Expr.const(0)
                if a.args[0] == 1:
                    return b
        return Expr("mul", (self, o))

    __rmul__ = lambda self, o: Expr.wrap(o) * self

    def next(self):
        """Same expression on the next row (only defined on column variables)."""
        assert self.op == "var" and self.args[0] in ("main", "prep") and self.args[2] == 0
        return Expr.var(self.args[0], self.args[1], 1)


def esum(xs):
    acc = Expr.const(0)
    for x in xs:
        acc = acc + x
    return acc


def word(bytes4):
    """little-endian byte limbs -> field element sum b_i * 256^i"""
    return esum(b * (1 << (8 * i)) for i, b in enumerate(bytes4))


class Interaction:
    __slots__ = ("bus", "sign", "mult", "vals", "scope")

    def __init__(self, bus, sign, mult, vals, scope):
        self.bus, self.sign, self.mult, self.vals, self.scope = bus, sign, Expr.wrap(mult), [Expr.wrap(v) for v in vals], scope
        assert self.mult.deg <= 1, "interaction multiplicity must be affine"
        for v in self.vals:
            assert v.deg <= 1, "interaction values must be affine"


class PolyRel:
    """A big-integer identity checked limb by limb (see Chip.assert_poly_zero)."""
    __slots__ = ("name", "terms", "w", "w_off", "sel", "first", "K", "coef_expr", "q", "modulus", "w_lo", "w_top")


class Chip:
    def __init__(self, name):
        self.name = name
        self.main_names, self.prep_names = [], []
        self.constraints = []  # (expr, when) when in {"all","first","last","trans"}
        self.interactions = []
        self.n_pub = 0
        self.poly_rels = []    # PolyRel groups: seal() appends their coefficient constraints as the LAST entries of self.constraints
        self._sealed = False

    # ---- columns
    def col(self, name):
        self.main_names.append(name)
        return Expr.var("main", len(self.main_names) - 1)

    def cols(self, name, n):
        return [self.col(f"{name}[{i}]") for i in range(n)]

    def prep(self, name):
        self.prep_names.append(name)
        return Expr.var("prep", len(self.prep_names) - 1)

    def preps(self, name, n):
        return [self.prep(f"{name}[{i}]") for i in range(n)]

    def pub(self, i):
        self.n_pub = max(self.n_pub, i + 1)
        return Expr.var("pub", i)

    def index_of(self, name):
        return self.main_names.index(name)

    # ---- constraints (degree counted with the selector: first/last/trans add 1)
    def assert_zero(self, e, when="all"):
        assert not self._sealed
        e = Expr.wrap(e)
        if e.is_const():
            assert e.args[0] == 0, f"{self.name}: constant non-zero constraint"
            return
        d = e.deg + (0 if when == "all" else 1)
        assert d <= 3, f"{self.name}: constraint degree {d} > 3"
        self.constraints.append((e, when))

    def assert_eq(self, a, b, when="all"):
        self.assert_zero(Expr.wrap(a) - Expr.wrap(b), when)

    def assert_bool(self, x):
        self.assert_zero(x * (x - 1))

    # ---- big-integer identities over byte limbs
    def assert_poly_zero(self, name, terms, q, modulus, sel, cases):
        """sum_terms coef * s * A(t) * B(t) = 0 at t = 256, as an identity of INTEGERS, where every term is
        (coef, s, A, B): coef an integer, s a column (a 0/1 selector) or None, A and B limb vectors (lists of columns,
        or lists of integers = constants; B may be None).  Column limbs are range-checked bytes (the caller's duty).
        With C(t) the polynomial of the left-hand side, the witness W (K - 1 carries) satisfies C(t) = (256 - t) W(t), i.e.
            c_k + W_(k-1) - 256 W_k = 0      for k = 0 .. K - 1   (W_(-1) = W_(K-1) = 0),
        one degree <= 3 constraint per coefficient; W_k = w_k - sel * off_k with w_k a new column that the caller
        range-checks to 16 bits (returned).  |c_k| < 2^25 and |256 W_k| < 2^25 keep every equation far below p, so it
        holds over the integers, and so does the identity at t = 256.
        The identity is "V = 0 (mod modulus)": the term - sel * q(t) * modulus(t) is added here, q a vector of witness
        columns (range-checked bytes, the caller's duty) and modulus an odd integer given as byte limbs.
        `sel` is the chip's "row is real" selector (the offsets vanish on padding rows); every term must carry a
        selector that is 0 on padding rows.  `cases` lists the selector assignments that can occur ({column: 0/1} dicts,
        e.g. one per operation of the chip): the carry offsets off_k are sized from the extreme values of c_k over them.
        The constraints are appended as ordinary ones (the oracle evaluates them as written); the product's generated
        code evaluates a whole group at the folding challenge instead: sum_k alpha^(s+k) (...) =
        alpha^s (C(alpha) + (alpha - 256) W(alpha)), O(limbs) instead of O(limbs^2) multiplications (emit.py)."""
        def vlen(v):
            return len(v)

        terms = list(terms) + [(-1, sel, q, modulus)]
        K = max((vlen(a) + (vlen(b) - 1 if b is not None else 0)) for _, _, a, b in terms)
        # ---- coefficient expressions and their ranges per case
        coef_expr = [Expr.const(0) for _ in range(K)]
        lo = [[0] * K for _ in cases]
        hi = [[0] * K for _ in cases]
        for coef, s, a, b in terms:
            assert s is not None, "every term needs a selector that vanishes on padding rows"
            bb = b if b is not None else [1]
            for i, ai in enumerate(a):
                for j, bj in enumerate(bb):
                    ca, cb = isinstance(ai, int), isinstance(bj, int)
                    if (ca and ai == 0) or (cb and bj == 0):
                        continue
                    e = Expr.wrap(coef) * s * Expr.wrap(ai) * Expr.wrap(bj)
                    coef_expr[i + j] = coef_expr[i + j] + e
                    mx = coef * (ai if ca else 255) * (bj if cb else 255)
                    mn = 0 if not (ca and cb) else mx
                    for ci, case in enumerate(cases):
                        if case.get(s.id, 0):
                            lo[ci][i + j] += min(mn, mx)
                            hi[ci][i + j] += max(mn, mx)
        # ---- carry ranges: W_k = (c_k + W_(k-1)) / 256
        wlo, whi = [0] * K, [0] * K
        for ci in range(len(cases)):
            l = h = 0
            for k in range(K):
                l, h = (lo[ci][k] + l) // 256, -((-(hi[ci][k] + h)) // 256)
                wlo[k], whi[k] = min(wlo[k], l), max(whi[k], h)
        assert not self._sealed
        rel = PolyRel()
        rel.name, rel.terms, rel.sel, rel.K, rel.q, rel.modulus = name, terms, sel, K, q, modulus
        width = max(whi[k] - wlo[k] for k in range(K - 1))
        assert width < 1 << 17, f"{self.name}.{name}: carries span {width} values"
        for k in range(K):
            assert max(abs(lo_[k]) for lo_ in lo) < 1 << 25 and max(abs(hi_[k]) for hi_ in hi) < 1 << 25
        lo16 = [self.col(f"{name}_w[{k}]") for k in range(K - 1)]
        # carries wider than 16 bits (three or more limb products on a side): one more bit per carry
        top = [self.col(f"{name}_wb[{k}]") for k in range(K - 1)] if width >= 1 << 16 else None
        for t in top or []:
            self.assert_zero(t * (t - 1))
        rel.w = [lo16[k] + 65536 * top[k] for k in range(K - 1)] if top else lo16
        rel.w_lo, rel.w_top = lo16, top
        rel.w_off = [-wlo[k] for k in range(K - 1)]
        rel.coef_expr = coef_expr
        rel.first = None
        self.poly_rels.append(rel)
        return lo16

    def seal(self):
        """append the coefficient constraints of the polynomial identities (after every other constraint)"""
        if self._sealed:
            return
        self._sealed = True
        for rel in self.poly_rels:
            rel.first = len(self.constraints)
            for k in range(rel.K):
                W_prev = rel.w[k - 1] - rel.sel * rel.w_off[k - 1] if k else Expr.const(0)
                W_k = rel.w[k] - rel.sel * rel.w_off[k] if k < rel.K - 1 else Expr.const(0)
                e = rel.coef_expr[k] + W_prev - 256 * W_k
                assert e.deg <= 3
                self.constraints.append((e, "all"))

    # ---- LogUp
    def send(self, bus, vals, mult=1, scope="local"):
        self.interactions.append(Interaction(bus, +1, mult, vals, scope))

    def receive(self, bus, vals, mult=1, scope="local"):
        self.interactions.append(Interaction(bus, -1, mult, vals, scope))

    @property
    def main_width(self):
        return len(self.main_names)

    @property
    def prep_width(self):
        return len(self.prep_names)


class Machine:
    def __init__(self, name, chips, buses):
        self.name, self.chips, self.buses = name, chips, buses
        for c in chips:
            c.seal()
            for it in c.interactions:
                assert it.bus in buses, f"{c.name}: unknown bus {it.bus}"
