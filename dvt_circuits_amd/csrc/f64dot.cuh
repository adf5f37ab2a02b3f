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
};
// centred canonical residue of a Montgomery word, as a double
DVT_HD double centred_from_mont(uint32_t m) { return p2f::mm((double)m, p2f::MONT_RINV); }

}  // namespace dvt
