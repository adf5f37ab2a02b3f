// Cross-check of MontField::inv (binary GCD on 62-bit approximations, csrc/bigfield.h) against y^(p-2) on 20 000 values per
// field (small, near-p, one-limb and random ones), and their timings.  Built and run by tests/test_bigfield.py.
#include "bigfield.h"
#include <cstdio>
#include <chrono>
using namespace dvt;
static uint64_t st = 88172645463325252ull;
static uint64_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
template <int N> int run(const uint64_t (&P)[N], const char *name) {
    MontField<N> F(P);
    int bad = 0;
    for (int it = 0; it < 20000; it++) {
        uint64_t a[N], am[N], i1[N], i2[N];
        for (int k = 0; k < N; k++) a[k] = rnd();
        if (it < 64) { for (int k = 0; k < N; k++) a[k] = 0; a[0] = it + 1; }
        else if (it < 128) { memcpy(a, P, sizeof a); a[0] -= (it - 63); }
        else if (it % 7 == 0) { for (int k = 1; k < N; k++) a[k] = 0; }
        a[N - 1] &= (P[N - 1] >> 1);   // below p
        F.to_mont(am, a);
        F.inv(i1, am);
        F.inv_fermat(i2, am);
        if (memcmp(i1, i2, sizeof i1)) { if (bad++ < 5) printf("%s mismatch at %d\n", name, it); }
    }
    uint64_t a[N], am[N], o[N];
    for (int k = 0; k < N; k++) a[k] = rnd();
    a[N - 1] &= (P[N - 1] >> 1);
    F.to_mont(am, a);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20000; i++) { F.inv(o, am); for (int k = 0; k < N; k++) am[k] = o[k]; am[0] ^= (uint64_t)i + 2; if (F.cmp(am, F.p) >= 0) am[N - 1] >>= 1; }   // (a different operand every time: the branches of the algorithm are data-dependent)
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2000; i++) { F.inv_fermat(o, am); am[0] ^= o[0] & 1; }
    auto t2 = std::chrono::steady_clock::now();
    printf("%s: %d mismatches; bingcd %.2f us, fermat %.2f us\n", name, bad, std::chrono::duration<double, std::micro>(t1 - t0).count() / 20000, std::chrono::duration<double, std::micro>(t2 - t1).count() / 2000);
    uint64_t z[N] = {0}, zo[N];
    printf("  inverse of 0 reported as %s\n", F.inv_canonical(zo, z) ? "invertible (BUG)" : "not invertible");
    return bad;
}
int main() {
    static const uint64_t BLS_P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    static const uint64_t SECP_P[4] = {0xfffffffefffffc2full, 0xffffffffffffffffull, 0xffffffffffffffffull, 0xffffffffffffffffull};
    return run<6>(BLS_P, "bls") + run<4>(SECP_P, "secp");
}
