/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the
 * product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it.  PARITY UNPINNED: the reference (metacraft-labs/dvt-circuits)
 * contains no prover arithmetic; its hot path lives in the un-vendored crate
 * sp1-sdk ^4.2.1 (reference Cargo.toml:31) and transitively p3-baby-bear /
 * p3-field.  This file restates the *published* BabyBear field
 * (p = 2^31 - 2^27 + 1) and its degree-4 binomial extension F_p[x]/(x^4 - 11)
 * in canonical (non-Montgomery) form with plain 64-bit `%`, on purpose a
 * different technique from the Montgomery arithmetic of the HIP product path.
 *
 * Reference call sites that reach this arithmetic: src/main.rs:461-466
 * (`client.setup(elf)`, `client.prove(&pk,&stdin).run()`).
 */
#ifndef DVT_ORACLE_FIELD_H
#define DVT_ORACLE_FIELD_H
#include <stdint.h>

#define BB_P 2013265921u /* 0x78000001 = 15 * 2^27 + 1 */
#define BB_TWO_ADICITY 27
#define BB_GENERATOR 31u /* multiplicative generator; also the LDE coset shift */
#define BB_EXT_W 11u     /* x^4 = 11 */

typedef uint32_t bb_t;

static inline bb_t bb_add(bb_t a, bb_t b) { uint32_t s = a + b; return s >= BB_P ? s - BB_P : s; }
static inline bb_t bb_sub(bb_t a, bb_t b) { return a >= b ? a - b : a + BB_P - b; }
static inline bb_t bb_neg(bb_t a) { return a ? BB_P - a : 0; }
static inline bb_t bb_mul(bb_t a, bb_t b) { return (bb_t)(((uint64_t)a * b) % BB_P); }
static inline bb_t bb_pow(bb_t a, uint64_t e) {
    bb_t r = 1;
    while (e) { if (e & 1) r = bb_mul(r, a); a = bb_mul(a, a); e >>= 1; }
    return r;
}
static inline bb_t bb_inv(bb_t a) { return bb_pow(a, BB_P - 2); }
/* primitive 2^k-th root of unity: 31^((p-1)/2^k) */
static inline bb_t bb_two_adic_gen(unsigned k) { return bb_pow(BB_GENERATOR, (uint64_t)(BB_P - 1) >> k); }

/* ---- F_{p^4} = F_p[x]/(x^4 - 11), element = c[0] + c[1] x + c[2] x^2 + c[3] x^3 ---- */
typedef struct { bb_t c[4]; } ef_t;

static inline ef_t ef_zero(void) { ef_t r = {{0, 0, 0, 0}}; return r; }
static inline ef_t ef_one(void) { ef_t r = {{1, 0, 0, 0}}; return r; }
static inline ef_t ef_from_base(bb_t a) { ef_t r = {{a, 0, 0, 0}}; return r; }
static inline int ef_eq(ef_t a, ef_t b) { return a.c[0]==b.c[0] && a.c[1]==b.c[1] && a.c[2]==b.c[2] && a.c[3]==b.c[3]; }
static inline ef_t ef_add(ef_t a, ef_t b) { ef_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_add(a.c[i], b.c[i]); return r; }
static inline ef_t ef_sub(ef_t a, ef_t b) { ef_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_sub(a.c[i], b.c[i]); return r; }
static inline ef_t ef_neg(ef_t a) { ef_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_neg(a.c[i]); return r; }
static inline ef_t ef_mul_base(ef_t a, bb_t b) { ef_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_mul(a.c[i], b); return r; }
static inline ef_t ef_mul(ef_t a, ef_t b) {
    /* schoolbook then reduce x^4 -> 11 */
    uint64_t t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) t[i + j] = (t[i + j] + (uint64_t)a.c[i] * b.c[j]) % BB_P;
    ef_t r;
    for (int i = 0; i < 4; i++) {
        uint64_t v = t[i];
        if (i + 4 < 7) v = (v + t[i + 4] * BB_EXT_W) % BB_P;
        r.c[i] = (bb_t)v;
    }
    return r;
}
static inline ef_t ef_pow(ef_t a, uint64_t e) {
    ef_t r = ef_one();
    while (e) { if (e & 1) r = ef_mul(r, a); a = ef_mul(a, a); e >>= 1; }
    return r;
}
/* inverse via the norm to the quadratic subfield and then to F_p:
 * a^-1 = conj-products / Norm(a).  We use the Frobenius-free route:
 * write a = A + x*B with A = a0 + a2 x^2, B = a1 + a3 x^2 over K = F_p[y]/(y^2-11), y = x^2;
 * a^-1 = (A - x B) / (A^2 - y B^2). */
static inline ef_t ef_inv(ef_t a) {
    /* elements of K as pairs (u0 + u1 y), y^2 = 11 */
    bb_t A0 = a.c[0], A1 = a.c[2], B0 = a.c[1], B1 = a.c[3];
    /* A^2 = (A0^2 + 11 A1^2) + (2 A0 A1) y */
    bb_t A2_0 = bb_add(bb_mul(A0, A0), bb_mul(BB_EXT_W, bb_mul(A1, A1)));
    bb_t A2_1 = bb_mul(2, bb_mul(A0, A1));
    bb_t B2_0 = bb_add(bb_mul(B0, B0), bb_mul(BB_EXT_W, bb_mul(B1, B1)));
    bb_t B2_1 = bb_mul(2, bb_mul(B0, B1));
    /* y * B^2 = 11 B2_1 + B2_0 y */
    bb_t D0 = bb_sub(A2_0, bb_mul(BB_EXT_W, B2_1));
    bb_t D1 = bb_sub(A2_1, B2_0);
    /* D^-1 in K: (D0 - D1 y) / (D0^2 - 11 D1^2) */
    bb_t n = bb_sub(bb_mul(D0, D0), bb_mul(BB_EXT_W, bb_mul(D1, D1)));
    bb_t ni = bb_inv(n);
    bb_t I0 = bb_mul(D0, ni), I1 = bb_neg(bb_mul(D1, ni));
    /* (A - xB) * I : A*I and B*I in K */
    bb_t AI0 = bb_add(bb_mul(A0, I0), bb_mul(BB_EXT_W, bb_mul(A1, I1)));
    bb_t AI1 = bb_add(bb_mul(A0, I1), bb_mul(A1, I0));
    bb_t BI0 = bb_add(bb_mul(B0, I0), bb_mul(BB_EXT_W, bb_mul(B1, I1)));
    bb_t BI1 = bb_add(bb_mul(B0, I1), bb_mul(B1, I0));
    ef_t r = {{AI0, bb_neg(BI0), AI1, bb_neg(BI1)}};
    return r;
}
#endif
