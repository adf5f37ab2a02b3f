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


class Chip:
    def __init__(self, name):
        self.name = name
        self.main_names, self.prep_names = [], []
        self.constraints = []  # (expr, when) when in {"all","first","last","trans"}
        self.interactions = []
        self.n_pub = 0

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
            for it in c.interactions:
                assert it.bus in buses, f"{c.name}: unknown bus {it.bus}"
