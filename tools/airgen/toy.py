"""Toy machine used to bring up and unit-test the generic STARK engine (LogUp with
a preprocessed table, next-row / first / last constraints, public values, chips
of different heights, odd and even interaction counts)."""
from .dsl import Chip, Machine

BUSES = {"range": 1}


def build():
    rng = Chip("range8")
    v = rng.prep("v")
    m = rng.col("mult")
    rng.receive("range", [v], m)

    fib = Chip("fib")
    a, b, c, carry = fib.col("a"), fib.col("b"), fib.col("c"), fib.col("carry")
    fib.assert_bool(carry)
    fib.assert_eq(a + b, c + 256 * carry)
    fib.assert_eq(a.next(), b, "trans")
    fib.assert_eq(b.next(), c, "trans")
    fib.assert_eq(a, fib.pub(0), "first")
    fib.assert_eq(b, fib.pub(1), "first")
    fib.assert_eq(c, fib.pub(2), "last")
    fib.send("range", [c])

    pairs = Chip("pairs")
    x, y, z, real = pairs.col("x"), pairs.col("y"), pairs.col("z"), pairs.col("is_real")
    pairs.assert_bool(real)
    pairs.assert_zero(real * (x * y - z))          # degree 3
    pairs.send("range", [x], real)
    pairs.send("range", [y], real)
    pairs.send("range", [z], real)
    return Machine("toy", [rng, fib, pairs], BUSES)
