// BabyBear (p = 2^31 - 2^27 + 1) and F_p[x]/(x^4 - 11) for gfx950 kernels and
// the host-side transcript/verifier code of the product path.
//
// Representation: every field element that lives in HBM or in a register is in
// MONTGOMERY form (x*2^32 mod p).  The map is a field isomorphism, so NTTs,
// Poseidon2 and constraint evaluation run unchanged on Montgomery values;
// conversion to canonical form happens only where bytes leave the prover
// (transcript observations, proof serialisation).
//
// Replaces (behind reference src/main.rs:461-466) the arithmetic of the absent
// crates p3-baby-bear / p3-field that sp1-sdk ^4.2.1 pulls in (Cargo.toml:31).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
// forced inlining only in the device pass: on the host it would fold a chip's whole generated verifier (thousands of F_p^4
// operations, every operator inlined) into one function whose -O3 compile time grows quadratically (18 minutes for the
// rv32 machine); the host compiler's own inlining heuristics are enough there
#if defined(__HIP_DEVICE_COMPILE__)
#define DVT_HD __host__ __device__ __forceinline__
#else
#define DVT_HD __host__ __device__ inline
#endif
#else
#define DVT_HD inline
#endif

namespace dvt {

constexpr uint32_t P = 0x78000001u;
constexpr uint32_t MONTY_MU = 0x88000001u;   // p^-1 mod 2^32
constexpr uint32_t MONTY_R = 0x0ffffffeu;    // 2^32 mod p  (= Montgomery form of 1)
constexpr uint32_t MONTY_R2 = 0x45dddde3u;   // 2^64 mod p
static_assert((uint32_t)(P * MONTY_MU) == 1u, "mu");
static_assert(((uint64_t)1 << 32) % P == MONTY_R, "R");
static_assert((((uint64_t)MONTY_R * MONTY_R) % P) == MONTY_R2, "R2");

DVT_HD uint32_t mulhi_u32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

struct Fp {
    uint32_t v;  // Montgomery form, always in [0, p)
    DVT_HD static Fp raw(uint32_t m) { Fp r; r.v = m; return r; }
    DVT_HD static Fp zero() { return raw(0); }
    DVT_HD static Fp one() { return raw(MONTY_R); }
    DVT_HD static Fp two() { return raw(2 * MONTY_R % P); }
    DVT_HD static uint32_t reduce64(uint64_t x) {  // x < p * 2^32  ->  x / 2^32 mod p
        uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        uint32_t t = lo * MONTY_MU;
        uint32_t u = mulhi_u32(t, P);
        // hi - u lies in (-p, p); add p when it is negative.  Written as an unsigned min so that hipcc
        // emits v_sub / v_add / v_min instead of a 64-bit compare + select: if hi >= u then r < p and
        // r + p does not wrap (so min = r); otherwise r = 2^32 - d wraps high and r + p = p - d is smaller.
        uint32_t r = hi - u, r2 = r + P;
        return r2 < r ? r2 : r;
    }
    DVT_HD static Fp from_canonical(uint32_t c) { return raw(reduce64((uint64_t)c * MONTY_R2)); }
    DVT_HD static Fp from_u64(uint64_t c) { return from_canonical((uint32_t)(c % P)); }
    DVT_HD uint32_t canonical() const { return reduce64((uint64_t)v); }
    DVT_HD bool operator==(Fp o) const { return v == o.v; }
    DVT_HD bool operator!=(Fp o) const { return v != o.v; }
    DVT_HD bool is_zero() const { return v == 0; }
};

DVT_HD Fp operator+(Fp a, Fp b) {
    uint32_t s = a.v + b.v, t = s - P;
    return Fp::raw(t < s ? t : s);
}
DVT_HD Fp operator-(Fp a, Fp b) {
    uint32_t s = a.v - b.v, t = s + P;
    return Fp::raw(t < s ? t : s);
}
DVT_HD Fp operator-(Fp a) { return Fp::raw(a.v ? P - a.v : 0); }
DVT_HD Fp operator*(Fp a, Fp b) { return Fp::raw(Fp::reduce64((uint64_t)a.v * b.v)); }
DVT_HD Fp &operator+=(Fp &a, Fp b) { a = a + b; return a; }
DVT_HD Fp &operator-=(Fp &a, Fp b) { a = a - b; return a; }
DVT_HD Fp &operator*=(Fp &a, Fp b) { a = a * b; return a; }
DVT_HD Fp dbl(Fp a) { return a + a; }

DVT_HD Fp pow(Fp a, uint64_t e) {
    Fp r = Fp::one();
    while (e) { if (e & 1) r = r * a; a = a * a; e >>= 1; }
    return r;
}
// a^(p-2), p - 2 = 0x77FFFFFF = 111 0 then 27 ones: a^7, one squaring, then nine times (three squarings, times a^7):
// 30 squarings + 11 products instead of the 58 of plain square-and-multiply
DVT_HD Fp inv(Fp a) {
    const Fp a2 = a * a, a3 = a2 * a, a7 = a3 * a3 * a;
    Fp r = a7 * a7;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r = r * r; r = r * r; r = r * r;
        r = r * a7;
    }
    return r;
}
// primitive 2^k-th root of unity 31^((p-1)/2^k); host-side helper (slow)
inline Fp two_adic_generator(unsigned k) { return pow(Fp::from_canonical(31), (uint64_t)(P - 1) >> k); }
constexpr uint32_t COSET_SHIFT = 31;

// the representative of a in (-p/2, p/2], exact in a double (operand format of the FP64 dot-product kernels)
inline double centred_canonical(Fp a) {
    const uint32_t c = a.canonical();
    return c > P / 2 ? (double)c - (double)P : (double)c;
}

// ---- F_{p^4} = F_p[x]/(x^4 - 11) ----
struct Fp4 {
    Fp c[4];
    DVT_HD static Fp4 zero() { Fp4 r; r.c[0] = r.c[1] = r.c[2] = r.c[3] = Fp::zero(); return r; }
    DVT_HD static Fp4 one() { Fp4 r = zero(); r.c[0] = Fp::one(); return r; }
    DVT_HD static Fp4 from_base(Fp a) { Fp4 r = zero(); r.c[0] = a; return r; }
    DVT_HD bool operator==(const Fp4 &o) const { return c[0] == o.c[0] && c[1] == o.c[1] && c[2] == o.c[2] && c[3] == o.c[3]; }
    DVT_HD bool operator!=(const Fp4 &o) const { return !(*this == o); }
};
DVT_HD Fp mul11(Fp a) {  // 11a = 8a + 2a + a
    Fp a2 = dbl(a), a4 = dbl(a2), a8 = dbl(a4);
    return a8 + a2 + a;
}
DVT_HD Fp4 operator+(const Fp4 &a, const Fp4 &b) { Fp4 r; for (int i = 0; i < 4; i++) r.c[i] = a.c[i] + b.c[i]; return r; }
DVT_HD Fp4 operator-(const Fp4 &a, const Fp4 &b) { Fp4 r; for (int i = 0; i < 4; i++) r.c[i] = a.c[i] - b.c[i]; return r; }
DVT_HD Fp4 operator-(const Fp4 &a) { Fp4 r; for (int i = 0; i < 4; i++) r.c[i] = -a.c[i]; return r; }
DVT_HD Fp4 operator+(const Fp4 &a, Fp b) { Fp4 r = a; r.c[0] = r.c[0] + b; return r; }
DVT_HD Fp4 operator-(const Fp4 &a, Fp b) { Fp4 r = a; r.c[0] = r.c[0] - b; return r; }
DVT_HD Fp4 operator*(const Fp4 &a, Fp b) { Fp4 r; for (int i = 0; i < 4; i++) r.c[i] = a.c[i] * b; return r; }
DVT_HD Fp4 operator*(const Fp4 &a, const Fp4 &b) {
    Fp4 r;
    r.c[0] = a.c[0] * b.c[0] + mul11(a.c[1] * b.c[3] + a.c[2] * b.c[2] + a.c[3] * b.c[1]);
    r.c[1] = a.c[0] * b.c[1] + a.c[1] * b.c[0] + mul11(a.c[2] * b.c[3] + a.c[3] * b.c[2]);
    r.c[2] = a.c[0] * b.c[2] + a.c[1] * b.c[1] + a.c[2] * b.c[0] + mul11(a.c[3] * b.c[3]);
    r.c[3] = a.c[0] * b.c[3] + a.c[1] * b.c[2] + a.c[2] * b.c[1] + a.c[3] * b.c[0];
    return r;
}
DVT_HD Fp4 &operator+=(Fp4 &a, const Fp4 &b) { a = a + b; return a; }
DVT_HD Fp4 &operator-=(Fp4 &a, const Fp4 &b) { a = a - b; return a; }
DVT_HD Fp4 &operator*=(Fp4 &a, const Fp4 &b) { a = a * b; return a; }
DVT_HD Fp4 pow(Fp4 a, uint64_t e) {
    Fp4 r = Fp4::one();
    while (e) { if (e & 1) r = r * a; a = a * a; e >>= 1; }
    return r;
}
// a = A + xB over K = F_p[y]/(y^2 - 11), y = x^2:  a^-1 = (A - xB) / (A^2 - y B^2)
DVT_HD Fp4 inv(const Fp4 &a) {
    Fp A0 = a.c[0], A1 = a.c[2], B0 = a.c[1], B1 = a.c[3];
    Fp A2_0 = A0 * A0 + mul11(A1 * A1), A2_1 = dbl(A0 * A1);
    Fp B2_0 = B0 * B0 + mul11(B1 * B1), B2_1 = dbl(B0 * B1);
    Fp D0 = A2_0 - mul11(B2_1), D1 = A2_1 - B2_0;
    Fp ni = inv(D0 * D0 - mul11(D1 * D1));
    Fp I0 = D0 * ni, I1 = -(D1 * ni);
    Fp4 r;
    r.c[0] = A0 * I0 + mul11(A1 * I1);
    r.c[2] = A0 * I1 + A1 * I0;
    r.c[1] = -(B0 * I0 + mul11(B1 * I1));
    r.c[3] = -(B0 * I1 + B1 * I0);
    return r;
}

}  // namespace dvt
