// Witness solver for the big-integer identities of the field / curve precompile chips
// (tools/airgen/dsl.py Chip.assert_poly_zero; tables in gen/<machine>_rels.h).  Host and device: the product builds the
// precompile rows on the GPU (rv32_bigops.hip, one thread per row), the CPU-only debug / test entry points on the host.
//
// An identity "V = 0 (mod m)" over byte limbs is proven as  sum_terms coef * s * A(t) * B(t) - s_real * q(t) * m(t) = 0
// at t = 256: given the row's operand and result cells, solve_poly_rel() fills the quotient q (bytes) and the carries
// w_k = W_k + off_k  (W_k = (c_k + W_(k-1)) / 256, c_k the k-th coefficient), splitting a carry wider than 16 bits into
// a 16-bit column and a top bit.  q is found without a long division: V is a multiple of the odd modulus, so
// q = (V mod 256^nq) * m^-1 mod 256^nq.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define DVT_PR_HD __host__ __device__
#else
#define DVT_PR_HD
#endif
// the generated tables: host statics in the host pass, device statics in the device pass (same names in both)
#if defined(__HIP_DEVICE_COMPILE__)
#define DVT_RELS_Q static __device__ const
#else
#define DVT_RELS_Q static const
#endif

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

// t[i + j] += a[i] * b[j]: byte limbs in 32-bit lanes (a coefficient of a product of two vectors of at most 96 bytes is
// below 2^23).  Contiguous operands and a branch-free inner loop: the compiler vectorises it; an AVX2 clone is picked at load
// time where the host has it (the build is generic x86-64, the GPU boxes are not).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline void poly_conv_acc(int32_t *__restrict t, const int32_t *__restrict a, int la, const int32_t *__restrict b, int lb) {
    for (int i = 0; i < la; i++) {
        const int32_t ai = a[i];
        for (int j = 0; j < lb; j++) t[i + j] += ai * b[j];
    }
}
#else
#if defined(__x86_64__) && defined(__clang__)
__attribute__((target_clones("avx2", "default")))
#endif
inline void poly_conv_acc(int32_t *__restrict t, const int32_t *__restrict a, int la, const int32_t *__restrict b, int lb) {
    for (int i = 0; i < la; i++) {
        const int32_t ai = a[i];
        int32_t *__restrict ti = t + i;
        for (int j = 0; j < lb; j++) ti[j] += ai * b[j];
    }
}
#endif

// q = u / v, r = u mod v on little-endian 32-bit digits: u has m digits, v has n, 1 <= n <= m, v[n-1] != 0; q gets m - n + 1
// digits, r gets n (D. Knuth, TAOCP vol. 2, 4.3.1 algorithm D, in the form of H. Warren, "Hacker's Delight", divmnu)
DVT_PR_HD inline void poly_divmnu(uint32_t *q, uint32_t *r, const uint32_t *u, const uint32_t *v, int m, int n) {
    if (n == 1) {
        uint64_t k = 0;
        for (int j = m - 1; j >= 0; j--) { const uint64_t t = (k << 32) | u[j]; q[j] = (uint32_t)(t / v[0]); k = t % v[0]; }
        r[0] = (uint32_t)k;
        return;
    }
    const int s = __builtin_clz(v[n - 1]);
    uint32_t vn[POLY_MAX_K / 4 + 2], un[POLY_MAX_K / 4 + 3];
    for (int i = n - 1; i > 0; i--) vn[i] = (v[i] << s) | (s ? v[i - 1] >> (32 - s) : 0);
    vn[0] = v[0] << s;
    un[m] = s ? u[m - 1] >> (32 - s) : 0;
    for (int i = m - 1; i > 0; i--) un[i] = (u[i] << s) | (s ? u[i - 1] >> (32 - s) : 0);
    un[0] = u[0] << s;
    for (int j = m - n; j >= 0; j--) {
        const uint64_t num = ((uint64_t)un[j + n] << 32) | un[j + n - 1];
        uint64_t qhat = num / vn[n - 1], rhat = num % vn[n - 1];
        while (qhat >> 32 || qhat * vn[n - 2] > ((rhat << 32) | un[j + n - 2])) {
            qhat--;
            rhat += vn[n - 1];
            if (rhat >> 32) break;
        }
        int64_t k = 0, t;
        for (int i = 0; i < n; i++) {
            const uint64_t p = qhat * vn[i];
            t = (int64_t)un[i + j] - k - (int64_t)(p & 0xffffffffull);
            un[i + j] = (uint32_t)t;
            k = (int64_t)(p >> 32) - (t >> 32);
        }
        t = (int64_t)un[j + n] - k;
        un[j + n] = (uint32_t)t;
        q[j] = (uint32_t)qhat;
        if (t < 0) {   // one too many: add the divisor back
            q[j]--;
            uint64_t c = 0;
            for (int i = 0; i < n; i++) { c += (uint64_t)un[i + j] + vn[i]; un[i + j] = (uint32_t)c; c >>= 32; }
            un[j + n] += (uint32_t)c;
        }
    }
    for (int i = 0; i < n - 1; i++) r[i] = (un[i] >> s) | (s ? un[i + 1] << (32 - s) : 0);
    r[n - 1] = un[n - 1] >> s;
}

// Row: uint32_t get(int col) const;  void put(int col, uint32_t v);
template <class Row>
DVT_PR_HD bool solve_poly_rel(const PolyRelDesc &d, Row &row) {
    int64_t c[POLY_MAX_K] = {0};
    if (d.K > POLY_MAX_K) return false;
    int32_t av[POLY_MAX_K], bv[POLY_MAX_K], tv[2 * POLY_MAX_K];
    auto fetch = [&](const PolyVec &v, int32_t *o) {
        if (v.cols) for (int i = 0; i < v.len; i++) o[i] = (int32_t)row.get(v.cols[i]);
        else for (int i = 0; i < v.len; i++) o[i] = (int32_t)v.cst[i];
    };
    for (int t = 0; t < d.n_terms; t++) {
        const PolyTerm &tm = d.terms[t];
        if (!row.get(tm.sel_col)) continue;
        fetch(tm.a, av);
        if (tm.b.len == 0) {
            for (int i = 0; i < tm.a.len; i++) c[i] += (int64_t)tm.coef * av[i];
            continue;
        }
        fetch(tm.b, bv);
        const int n = tm.a.len + tm.b.len - 1;
        for (int k = 0; k < n; k++) tv[k] = 0;
        poly_conv_acc(tv, av, tm.a.len, bv, tm.b.len);
        for (int k = 0; k < n; k++) c[k] += (int64_t)tm.coef * tv[k];
    }
    uint8_t low[POLY_MAX_K + 1], q[POLY_MAX_K], modb[POLY_MAX_K];
    for (int j = 0; j < d.nmod; j++) modb[j] = d.modv ? (uint8_t)row.get(d.modv[j]) : d.mod[j];
    if (!d.modv) {
        // the low nq digits of V, then q = low * pinv mod 256^nq (a truncated convolution, then one carry sweep)
        int64_t t = 0;
        for (int k = 0; k < d.nq; k++) {
            if (k < d.K) t += c[k];
            low[k] = (uint8_t)(t & 255);
            t = (t - low[k]) / 256;
        }
        for (int k = 0; k < d.nq; k++) { av[k] = low[k]; bv[k] = d.pinv[k]; }
        for (int k = 0; k < 2 * d.nq; k++) tv[k] = 0;
        poly_conv_acc(tv, av, d.nq, bv, d.nq);
        uint64_t carry = 0;
        for (int k = 0; k < d.nq; k++) {
            const uint64_t s = carry + (uint32_t)tv[k];
            q[k] = (uint8_t)(s & 255);
            carry = s >> 8;
        }
    } else {
        // a modulus from the row may be even: all K digits of V (non-negative by construction), then a long division
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
        // 32-bit digits, Knuth's algorithm D
        uint32_t u[POLY_MAX_K / 4 + 2] = {0}, v[POLY_MAX_K / 4 + 2] = {0}, qw[POLY_MAX_K / 4 + 2] = {0}, rw[POLY_MAX_K / 4 + 2] = {0};
        for (int k = 0; k < d.K; k++) u[k >> 2] |= (uint32_t)low[k] << (8 * (k & 3));
        for (int k = 0; k < nm; k++) v[k >> 2] |= (uint32_t)modb[k] << (8 * (k & 3));
        const int mu = (d.K + 3) / 4, nv = (nm + 3) / 4;
        if (mu >= nv) poly_divmnu(qw, rw, u, v, mu, nv);
        else for (int k = 0; k < mu; k++) rw[k] = u[k];
        for (int k = 0; k < nv; k++)
            if (rw[k]) return false;          // V is not a multiple of the modulus
        for (int k = 0; k < d.nq; k++) q[k] = 0;
        for (int k = 0; k < 4 * (mu >= nv ? mu - nv + 1 : 0); k++) {
            const uint8_t digit = (uint8_t)(qw[k >> 2] >> (8 * (k & 3)));
            if (!digit) continue;
            if (k >= d.nq) return false;
            q[k] = digit;
        }
    }
    for (int k = 0; k < d.nq; k++) row.put(d.q[k], q[k]);
    {
        for (int k = 0; k < d.nq; k++) av[k] = q[k];
        for (int k = 0; k < d.nmod; k++) bv[k] = modb[k];
        const int n = d.nq + d.nmod - 1;
        for (int k = 0; k < n; k++) tv[k] = 0;
        poly_conv_acc(tv, av, d.nq, bv, d.nmod);
        for (int k = 0; k < n && k < POLY_MAX_K; k++) c[k] -= tv[k];
    }
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
