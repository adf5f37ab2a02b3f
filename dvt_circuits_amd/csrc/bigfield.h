// Prime-field arithmetic on 64-bit limbs for the guest machine's field / curve precompiles (BLS12-381 Fp: N = 6,
// secp256k1: N = 4).  Montgomery form (CIOS multiplication), inversion by Fermat's little theorem.  Host only.
#pragma once
#include <cstdint>
#include <cstring>

namespace dvt {

template <int N>
struct MontField {
    using u128 = unsigned __int128;
    uint64_t p[N], r2[N], one[N], pm2[N], n0;

    explicit MontField(const uint64_t (&modulus)[N]) {
        memcpy(p, modulus, sizeof p);
        // n0 = -p^-1 mod 2^64 (Newton iteration on the odd low limb)
        uint64_t x = p[0];
        for (int i = 0; i < 6; i++) x *= 2 - p[0] * x;
        n0 = (uint64_t)0 - x;
        // R mod p and R^2 mod p by doubling
        uint64_t t[N] = {1};
        for (int i = 0; i < 64 * N; i++) dbl(t);
        memcpy(one, t, sizeof t);
        for (int i = 0; i < 64 * N; i++) dbl(t);
        memcpy(r2, t, sizeof t);
        memcpy(pm2, p, sizeof p);
        pm2[0] -= 2;    // (p is odd and > 2: no borrow)
    }
    static int cmp(const uint64_t *a, const uint64_t *b) {
        for (int i = N - 1; i >= 0; i--)
            if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
        return 0;
    }
    bool is_canonical(const uint64_t *a) const { return cmp(a, p) < 0; }
    static bool is_zero(const uint64_t *a) {
        uint64_t o = 0;
        for (int i = 0; i < N; i++) o |= a[i];
        return o == 0;
    }
    static uint64_t add_n(uint64_t *o, const uint64_t *a, const uint64_t *b) {
        u128 c = 0;
        for (int i = 0; i < N; i++) { c += (u128)a[i] + b[i]; o[i] = (uint64_t)c; c >>= 64; }
        return (uint64_t)c;
    }
    static uint64_t sub_n(uint64_t *o, const uint64_t *a, const uint64_t *b) {
        uint64_t br = 0;
        for (int i = 0; i < N; i++) {
            const u128 d = (u128)a[i] - b[i] - br;
            o[i] = (uint64_t)d;
            br = (uint64_t)(d >> 64) & 1;
        }
        return br;
    }
    void dbl(uint64_t *t) const {   // t < p  ->  2 t mod p
        uint64_t c = add_n(t, t, t);
        uint64_t s[N];
        if (c || cmp(t, p) >= 0) { sub_n(s, t, p); memcpy(t, s, sizeof s); }
    }
    void add(uint64_t *o, const uint64_t *a, const uint64_t *b) const {   // canonical operands
        uint64_t c = add_n(o, a, b), s[N];
        if (c || cmp(o, p) >= 0) { sub_n(s, o, p); memcpy(o, s, sizeof s); }
    }
    void sub(uint64_t *o, const uint64_t *a, const uint64_t *b) const {
        uint64_t s[N];
        if (sub_n(o, a, b)) { add_n(s, o, p); memcpy(o, s, sizeof s); }
    }
    // o = a b R^-1 mod p for a b < p R (one operand may be any N-limb number)
    void mul(uint64_t *o, const uint64_t *a, const uint64_t *b) const {
        uint64_t t[N + 2] = {0};
        for (int i = 0; i < N; i++) {
            u128 c = 0;
            for (int j = 0; j < N; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[N]; t[N] = (uint64_t)c; t[N + 1] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * n0;
            c = (u128)m * p[0] + t[0];
            c >>= 64;
            for (int j = 1; j < N; j++) { c += (u128)m * p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[N]; t[N - 1] = (uint64_t)c; t[N] = t[N + 1] + (uint64_t)(c >> 64);
        }
        uint64_t s[N];
        if (t[N] || cmp(t, p) >= 0) { sub_n(s, t, p); memcpy(o, s, sizeof s); }
        else memcpy(o, t, sizeof s);
    }
    void to_mont(uint64_t *o, const uint64_t *a) const { mul(o, a, r2); }
    void from_mont(uint64_t *o, const uint64_t *a) const {
        uint64_t u[N] = {1};
        mul(o, a, u);
    }
    void inv(uint64_t *o, const uint64_t *a) const {   // Montgomery in, Montgomery out: a^(p-2)
        uint64_t r[N], b[N];
        memcpy(r, one, sizeof r);
        memcpy(b, a, sizeof b);
        for (int i = 0; i < 64 * N; i++) {
            if ((pm2[i >> 6] >> (i & 63)) & 1) mul(r, r, b);
            mul(b, b, b);
        }
        memcpy(o, r, sizeof r);
    }
};

}  // namespace dvt
