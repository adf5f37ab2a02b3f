// Host-side witness solver for the big-integer identities of the field / curve precompile chips
// (tools/airgen/dsl.py Chip.assert_poly_zero; tables in gen/<machine>_rels.h).
//
// An identity "V = 0 (mod m)" over byte limbs is proven as  sum_terms coef * s * A(t) * B(t) - s_real * q(t) * m(t) = 0
// at t = 256: given the row's operand and result cells, solve_poly_rel() fills the quotient q (bytes) and the carries
// w_k = W_k + off_k  (W_k = (c_k + W_(k-1)) / 256, c_k the k-th coefficient), splitting a carry wider than 16 bits into
// a 16-bit column and a top bit.  q is found without a long division: V is a multiple of the odd modulus, so
// q = (V mod 256^nq) * m^-1 mod 256^nq.
#pragma once
#include <cstdint>

namespace dvt {

struct PolyVec {
    const int16_t *cols;   // main-column index of every limb, or
    const uint8_t *cst;    // constant limbs
    int len;               // 0: the term has no second factor
};
struct PolyTerm {
    int coef;
    int sel_col;           // 0/1 selector column of the term
    PolyVec a, b;
};
struct PolyRelDesc {
    const char *name;
    int n_terms;
    const PolyTerm *terms;   // every term but the quotient's
    int K;                   // number of coefficients (constraints)
    const int16_t *q;
    int nq;
    const uint8_t *mod;
    int nmod;
    const uint8_t *pinv;     // modulus^-1 mod 256^nq
    const int16_t *w, *wb;   // carry columns (low 16 bits; top bit or nullptr)
    const int32_t *w_off;
    const int16_t *modv;     // non-null: the modulus is a vector of the row's own cells (mod / pinv are null): q by long division
};

constexpr int POLY_MAX_K = 200;

// Row: uint32_t get(int col) const;  void put(int col, uint32_t v);
template <class Row>
bool solve_poly_rel(const PolyRelDesc &d, Row &row) {
    int64_t c[POLY_MAX_K] = {0};
    if (d.K > POLY_MAX_K) return false;
    auto limb = [&](const PolyVec &v, int i) -> int64_t { return v.cols ? (int64_t)row.get(v.cols[i]) : (int64_t)v.cst[i]; };
    for (int t = 0; t < d.n_terms; t++) {
        const PolyTerm &tm = d.terms[t];
        if (!row.get(tm.sel_col)) continue;
        for (int i = 0; i < tm.a.len; i++) {
            const int64_t ai = tm.coef * limb(tm.a, i);
            if (!ai) continue;
            if (tm.b.len == 0) c[i] += ai;
            else for (int j = 0; j < tm.b.len; j++) c[i + j] += ai * limb(tm.b, j);
        }
    }
    uint8_t low[POLY_MAX_K + 1], q[POLY_MAX_K], modb[POLY_MAX_K];
    for (int j = 0; j < d.nmod; j++) modb[j] = d.modv ? (uint8_t)row.get(d.modv[j]) : d.mod[j];
    if (!d.modv) {
        // the low nq digits of V, then q = low * pinv mod 256^nq
        int64_t t = 0;
        for (int k = 0; k < d.nq; k++) {
            if (k < d.K) t += c[k];
            low[k] = (uint8_t)(t & 255);
            t = (t - low[k]) / 256;
        }
        uint64_t carry = 0;
        for (int k = 0; k < d.nq; k++) {
            uint64_t s = carry;
            for (int i = 0; i <= k; i++) s += (uint64_t)low[i] * d.pinv[k - i];
            q[k] = (uint8_t)(s & 255);
            carry = s >> 8;
        }
    } else {
        // a modulus from the row may be even: all K digits of V (non-negative by construction), then schoolbook
        // division in base 256 (quotient digit by repeated subtraction: at most 255 steps of nmod bytes)
        int64_t t = 0;
        for (int k = 0; k < d.K; k++) {
            t += c[k];
            low[k] = (uint8_t)(t & 255);
            t = (t - low[k]) / 256;
        }
        if (t != 0) return false;
        int nm = d.nmod;
        while (nm > 0 && modb[nm - 1] == 0) nm--;
        if (nm == 0) return false;
        uint8_t rem[POLY_MAX_K + 2] = {0};    // running remainder, little-endian, nm + 1 bytes
        for (int k = 0; k < d.nq; k++) q[k] = 0;
        for (int k = d.K - 1; k >= 0; k--) {
            for (int i = nm; i > 0; i--) rem[i] = rem[i - 1];
            rem[0] = low[k];
            int digit = 0;
            for (;;) {
                bool ge = rem[nm] != 0;
                if (!ge) {
                    ge = true;
                    for (int i = nm - 1; i >= 0; i--)
                        if (rem[i] != modb[i]) { ge = rem[i] > modb[i]; break; }
                }
                if (!ge) break;
                int borrow = 0;
                for (int i = 0; i <= nm; i++) {
                    int v = (int)rem[i] - (i < nm ? modb[i] : 0) - borrow;
                    borrow = v < 0;
                    rem[i] = (uint8_t)(v + (borrow << 8));
                }
                digit++;
            }
            if (digit) {
                if (k >= d.nq || digit > 255) return false;
                q[k] = (uint8_t)digit;
            }
        }
        for (int i = 0; i <= nm; i++)
            if (rem[i]) return false;         // V is not a multiple of the modulus
    }
    for (int k = 0; k < d.nq; k++) row.put(d.q[k], q[k]);
    for (int i = 0; i < d.nq; i++)
        if (q[i])
            for (int j = 0; j < d.nmod; j++) c[i + j] -= (int64_t)q[i] * modb[j];
    int64_t W = 0;
    for (int k = 0; k + 1 < d.K; k++) {
        const int64_t s = c[k] + W;
        if (s & 255) return false;          // V is not a multiple of the modulus: the caller's witness is wrong
        W = s / 256;
        const int64_t wv = W + d.w_off[k];
        if (wv < 0 || wv >= (d.wb ? 131072 : 65536)) return false;
        row.put(d.w[k], (uint32_t)(wv & 0xffff));
        if (d.wb) row.put(d.wb[k], (uint32_t)(wv >> 16));
    }
    return c[d.K - 1] + W == 0;
}

}  // namespace dvt
