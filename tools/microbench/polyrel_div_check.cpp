// Long division of the witness solver (csrc/polyrel.h poly_divmnu: Knuth's algorithm D on 32-bit digits) on crafted and random
// operands: prints "u v q r" (hex, most significant digit first) per case for tests/test_bigfield.py, which checks
// u = q v + r and r < v with Python integers.  The crafted cases reach the branches random operands practically never take
// (q-hat one or two too large, the add-back step, a one-digit divisor, equal lengths, a zero dividend).
#include "polyrel.h"
#include <cstdio>
#include <vector>
using namespace dvt;
static uint64_t st = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); }
static void show(const uint32_t *a, int n) { for (int i = n - 1; i >= 0; i--) printf("%08x", a[i]); }
static void run(const std::vector<uint32_t> &u, const std::vector<uint32_t> &v) {
    const int m = (int)u.size(), n = (int)v.size();
    uint32_t q[80] = {0}, r[80] = {0};
    poly_divmnu(q, r, u.data(), v.data(), m, n);
    show(u.data(), m); printf(" "); show(v.data(), n); printf(" "); show(q, m - n + 1); printf(" "); show(r, n); printf("\n");
}
int main() {
    // Hacker's Delight's divmnu test vectors (little-endian digits), among them the add-back cases
    run({3, 0, 0x80000000u}, {1, 0, 0x20000000u});
    run({0, 0, 0x8000, 0x7fff}, {1, 0, 0x8000});
    run({0, 0xfffe, 0x8000}, {0xffff, 0x8000});
    run({0x00000003, 0x00000000, 0x00008000}, {0x00000001, 0x00000000, 0x00002000});
    run({0, 0, 0x80000000u, 0x7fffffffu}, {1, 0, 0x80000000u});
    run({0, 0xfffffffeu, 0x80000000u}, {0xffffffffu, 0x80000000u});
    run({0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {0xffffffffu});
    run({0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, {1});
    run({0, 0, 0, 0}, {5, 7});
    run({1, 2, 3}, {1, 2, 3});
    run({0, 2, 3}, {1, 2, 3});
    run({0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu});
    run({0x89abcdefu, 0x01234567u, 0, 0, 0, 0, 0, 0, 0x89abcdefu, 0x01234567u, 0xffffffffu, 0xffffffffu, 0, 0, 0, 1}, {0xffffffffu, 0, 0, 0, 0, 0, 0, 0x80000000u});
    // random: the shapes of UINT256_MUL (a 512-bit product by a 1..8-digit modulus) and arbitrary ones, with runs of 0 / f digits
    for (int it = 0; it < 20000; it++) {
        const int n = 1 + rnd() % 8, m = (it & 1) ? 16 : n + rnd() % 12;
        std::vector<uint32_t> u(m), v(n);
        for (auto &x : u) { uint32_t k = rnd() % 8; x = k == 0 ? 0 : k == 1 ? 0xffffffffu : k == 2 ? 0x80000000u : rnd(); }
        for (auto &x : v) { uint32_t k = rnd() % 8; x = k == 0 ? 0 : k == 1 ? 0xffffffffu : k == 2 ? 0x80000000u : rnd(); }
        if (v[n - 1] == 0) v[n - 1] = 1 + rnd() % 3;
        run(u, v);
    }
    return 0;
}
