// Prime-field arithmetic on 64-bit limbs for the guest machine's field / curve precompiles (BLS12-381 Fp: N = 6,
// secp256k1: N = 4).  Montgomery form (CIOS multiplication), inversion by Fermat's little theorem.  Host only.
#pragma once
#include <cstdint>
#include <cstring>
#include <initializer_list>

namespace dvt {

template <int N>
struct MontField {
    using u128 = unsigned __int128;
    uint64_t p[N], r2[N], r3[N], one[N], pm2[N], n0;

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
        mul(r3, r2, r2);   // R^3 (needs n0 and p: set above)
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
    // ---- modular inverse of a canonical value, canonical result (0 for 0): binary GCD on 62-bit approximations
    // (T. Pornin, "Optimized Binary GCD for Modular Inversion", 2020, algorithm 2 with k = 31; variable time — the guest's
    // data is not secret).  Invariants a = y u, b = y v (mod p) with (a, b) -> ((a f0 + b g0) / 2^31, (a f1 + b g1) / 2^31)
    // exactly and the same map on (u, v) modulo p; 31 halving steps per round are decided on the low 31 and top 33 bits of
    // a and b.  len(a) + len(b) drops by at least one per step: ceil((2 * 64 N - 1) / 31) rounds always suffice; at the end
    // b = gcd = 1 and v = y^-1.  About 3 us for 381 bits against 28 us for y^(p-2).
    static int top_bit(const uint64_t *a) {   // bit length
        for (int i = N - 1; i >= 0; i--)
            if (a[i]) return 64 * i + 64 - __builtin_clzll(a[i]);
        return 0;
    }
    // t = (x f + y g) as N + 1 limbs, two's complement (|f|, |g| <= 2^31): unsigned 64 x 64 -> 128 products of the magnitudes,
    // signs applied to the products (a signed 128 x 64 product costs two multiplications each)
    static void lin2(uint64_t *t, const uint64_t *x, int64_t f, const uint64_t *y, int64_t g) {
        const uint64_t af = (uint64_t)(f < 0 ? -f : f), ag = (uint64_t)(g < 0 ? -g : g);
        const bool nf = f < 0, ng = g < 0;
        __int128 carry = 0;
        for (int i = 0; i < N; i++) {
            const __int128 p1 = (__int128)((u128)x[i] * af), p2 = (__int128)((u128)y[i] * ag);   // (< 2^95: positive)
            const __int128 acc = carry + (nf ? -p1 : p1) + (ng ? -p2 : p2);
            t[i] = (uint64_t)acc;
            carry = acc >> 64;
        }
        t[N] = (uint64_t)carry;
    }
    static void shr31(uint64_t *t) {   // arithmetic shift of an (N + 1)-limb two's complement number
        for (int i = 0; i < N; i++) t[i] = (t[i] >> 31) | (t[i + 1] << 33);
        t[N] = (uint64_t)((int64_t)t[N] >> 31);
    }
    static void neg_n1(uint64_t *t) {
        uint64_t c = 1;
        for (int i = 0; i <= N; i++) { t[i] = ~t[i] + c; c = c && t[i] == 0; }
    }
    bool inv_canonical(uint64_t *o, const uint64_t *y) const {
        uint64_t a[N + 1], b[N + 1], u[N + 1], v[N + 1], t0[N + 1], t1[N + 1];
        memcpy(a, y, sizeof(uint64_t) * N); a[N] = 0;
        memcpy(b, p, sizeof(uint64_t) * N); b[N] = 0;
        memset(u, 0, sizeof u); u[0] = 1;
        memset(v, 0, sizeof v);
        // -p^-1 mod 2^31 (from n0 = -p^-1 mod 2^64)
        const uint64_t minv = n0 & 0x7fffffffull;
        constexpr int ROUNDS = (2 * 64 * N - 1 + 30) / 31;
        for (int round = 0; round < ROUNDS; round++) {
            const int la = top_bit(a), lb = top_bit(b), n = la > lb ? la : lb;
            uint64_t xa, xb;
            if (n <= 64) { xa = a[0]; xb = b[0]; }
            else {
                // low 31 bits and the 33 bits below position n
                auto top33 = [&](const uint64_t *z) -> uint64_t {
                    const int sh = n - 33, w = sh >> 6, r = sh & 63;
                    uint64_t hi = z[w] >> r;
                    if (r && w + 1 <= N) hi |= z[w + 1] << (64 - r);
                    return hi & 0x1ffffffffull;
                };
                xa = (a[0] & 0x7fffffffull) | (top33(a) << 31);
                xb = (b[0] & 0x7fffffffull) | (top33(b) << 31);
            }
            int64_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
            for (int j = 0; j < 31; j++) {
                // branch-free (the conditions are data-dependent coin flips: a mispredicted branch costs more than the masks):
                // odd -> (swap when xa < xb), subtract; then halve
                const uint64_t odd = (uint64_t)0 - (xa & 1), sw = odd & ((uint64_t)0 - (uint64_t)(xa < xb));
                const uint64_t tx = (xa ^ xb) & sw;
                xa ^= tx; xb ^= tx;
                const int64_t tf = (f0 ^ f1) & (int64_t)sw, tg = (g0 ^ g1) & (int64_t)sw;
                f0 ^= tf; f1 ^= tf; g0 ^= tg; g1 ^= tg;
                xa -= xb & odd; f0 -= f1 & (int64_t)odd; g0 -= g1 & (int64_t)odd;
                xa >>= 1;
                f1 <<= 1; g1 <<= 1;
            }
            lin2(t0, a, f0, b, g0);
            lin2(t1, a, f1, b, g1);
            shr31(t0); shr31(t1);
            if ((int64_t)t0[N] < 0) { neg_n1(t0); f0 = -f0; g0 = -g0; }
            if ((int64_t)t1[N] < 0) { neg_n1(t1); f1 = -f1; g1 = -g1; }
            memcpy(a, t0, sizeof a); memcpy(b, t1, sizeof b);
            // (u, v) <- ((u f0 + v g0) / 2^31, (u f1 + v g1) / 2^31) mod p: add the multiple of p that clears the low 31 bits
            lin2(t0, u, f0, v, g0);
            lin2(t1, u, f1, v, g1);
            for (uint64_t *t : {t0, t1}) {
                const uint64_t z = (t[0] * minv) & 0x7fffffffull;
                u128 c = 0;
                for (int i = 0; i < N; i++) { c += (u128)p[i] * z + t[i]; t[i] = (uint64_t)c; c >>= 64; }
                t[N] += (uint64_t)c;
                shr31(t);
                // now in (-2 p, 2 p): into [0, p)
                while ((int64_t)t[N] < 0) { u128 k = 0; for (int i = 0; i < N; i++) { k += (u128)t[i] + p[i]; t[i] = (uint64_t)k; k >>= 64; } t[N] += (uint64_t)k; }
                while (t[N] || cmp(t, p) >= 0) { uint64_t br = sub_n(t, t, p); t[N] -= br; }
            }
            memcpy(u, t0, sizeof u); memcpy(v, t1, sizeof v);
        }
        bool one = b[0] == 1;
        for (int i = 1; i <= N; i++) one = one && b[i] == 0;
        if (!one) { memset(o, 0, sizeof(uint64_t) * N); return false; }
        memcpy(o, v, sizeof(uint64_t) * N);
        return true;
    }
    void inv(uint64_t *o, const uint64_t *a) const {   // Montgomery in, Montgomery out
        uint64_t c[N];
        inv_canonical(c, a);      // a holds x R: c = x^-1 R^-1
        mul(o, c, r3);            // x^-1 R^-1 R^3 R^-1 = x^-1 R
    }
    void inv_fermat(uint64_t *o, const uint64_t *a) const {   // Montgomery in, Montgomery out: a^(p-2) (the cross-check of inv)
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
