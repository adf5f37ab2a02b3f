// Exact dot products with wave-uniform field coefficients on the FP64 pipe (used by K4, K5, K6, K7).
//
// sum_i w_i * v_i with field coefficients w and per-row field words v.  A Montgomery product costs three quarter-rate
// integer multiplies plus a reduction per term; here w is a centred residue in a double (|w| < 2^30) and v is split
// into 16-bit halves, so that w * v_lo and w * v_hi are exact (< 2^46) and an FMA accumulates them exactly: two
// full-rate operations per term and coefficient, one reduction every 32 terms (32 * 2^46 + 2^30 < 2^52).  The sum is
// lo + 2^16 * hi.  v may be a Montgomery word: the sum is then the Montgomery word of the true dot product (the map
// is linear), which is what all callers store.
#pragma once
#include "poseidon2_f64.cuh"

namespace dvt {

struct DotAcc {
    double lo, hi;
    DVT_HD void add(double w, double vlo, double vhi) { lo = fma(w, vlo, lo); hi = fma(w, vhi, hi); }
    DVT_HD void reduce() { lo = p2f::red(lo); hi = p2f::red(hi); }
    // |lo|, |hi| < 2^52: red(hi) * 2^16 + lo < 2^52.2 is exact
    DVT_HD Fp value() const { return Fp::raw(p2f::fix(p2f::red(fma(p2f::red(hi), 65536.0, lo)))); }
};
// F_p^4 accumulator: sum_i W_i * v_i with W_i in F_p^4 given as 4 centred doubles, v_i a field word
struct DotAcc4 {
    DotAcc c[4];
    DVT_HD DotAcc4() { for (int k = 0; k < 4; k++) c[k] = DotAcc{0.0, 0.0}; }
    DVT_HD void add(const double *w4, Fp v) {
        const double vlo = (double)(v.v & 0xffffu), vhi = (double)(v.v >> 16);
#pragma unroll
        for (int k = 0; k < 4; k++) c[k].add(w4[k], vlo, vhi);
    }
    DVT_HD void reduce() {
#pragma unroll
        for (int k = 0; k < 4; k++) c[k].reduce();
    }
    DVT_HD Fp4 value() const {
        Fp4 r;
#pragma unroll
        for (int k = 0; k < 4; k++) r.c[k] = c[k].value();
        return r;
    }
    // the same sum times 2^-32 as lazily reduced doubles: with Montgomery words v this is the CANONICAL dot product
    DVT_HD void value_canonical(double out[4]) const {
#pragma unroll
        for (int k = 0; k < 4; k++) out[k] = p2f::mm(p2f::red(fma(p2f::red(c[k].hi), 65536.0, c[k].lo)), p2f::MONT_RINV);
    }
};
// F_p^4 = F_p[x]/(x^4 - 11) on lazily reduced doubles (canonical residues, not Montgomery words).  A product reduces every
// one of its 16 coefficient products (p2f::mm, 6 operations, |result| < 0.51 p), so its coefficients are bounded by
// (1 + 3 * 11) * 0.51 p < 2^35.2 whatever the inputs (which must stay below 2^38 so that |a_i b_j| < 2^76); sums and
// differences are plain adds.  Costs 96 full-rate operations against 16 Montgomery products (3 quarter-rate multiplies
// + 4 operations each) and 12 three-instruction modular additions.
struct Fd4 {
    double c[4];
};
DVT_HD Fd4 operator*(const Fd4 &a, const Fd4 &b) {
    // every b_j takes part in four products: its quotient estimate b_j / p is computed once (p2f::mm_pre)
    const double q0 = b.c[0] * p2f::PINV, q1 = b.c[1] * p2f::PINV, q2 = b.c[2] * p2f::PINV, q3 = b.c[3] * p2f::PINV;
    auto m = [&](int i, int j, double qj) { return p2f::mm_pre(a.c[i], b.c[j], qj); };
    Fd4 r;
    r.c[0] = fma(11.0, m(1, 3, q3) + m(2, 2, q2) + m(3, 1, q1), m(0, 0, q0));
    r.c[1] = fma(11.0, m(2, 3, q3) + m(3, 2, q2), m(0, 1, q1) + m(1, 0, q0));
    r.c[2] = fma(11.0, m(3, 3, q3), m(0, 2, q2) + m(1, 1, q1) + m(2, 0, q0));
    r.c[3] = (m(0, 3, q3) + m(1, 2, q2)) + (m(2, 1, q1) + m(3, 0, q0));
    return r;
}
DVT_HD Fd4 operator*(const Fd4 &a, double b) {   // |b| < 2^38
    const double bq = b * p2f::PINV;
    Fd4 r;
    for (int k = 0; k < 4; k++) r.c[k] = p2f::mm_pre(a.c[k], b, bq);
    return r;
}
DVT_HD Fd4 operator+(const Fd4 &a, const Fd4 &b) { Fd4 r; for (int k = 0; k < 4; k++) r.c[k] = a.c[k] + b.c[k]; return r; }
DVT_HD Fd4 operator-(const Fd4 &a, const Fd4 &b) { Fd4 r; for (int k = 0; k < 4; k++) r.c[k] = a.c[k] - b.c[k]; return r; }

// centred canonical residue of a Montgomery word, as a double
DVT_HD double centred_from_mont(uint32_t m) { return p2f::from_mont(m); }

}  // namespace dvt
