// Poseidon2 (the permutation of poseidon2.cuh, bit for bit) for the gfx950 hashing kernels, computed on the
// FP64 pipe instead of the integer multiplier.
//
// Why: v_mul_lo/hi_u32 issue at quarter rate on gfx950 and a Montgomery product needs three of them, which
// bounds the integer permutation at about 4.5 Gperm/s per GPU (measured, tools/microbench).  v_fma_f64 issues at
// full rate, a modular product costs six FP64 operations and additions need no reduction at all, which measures
// 6.5 Gperm/s for the same function.  (No MFMA: nothing here is a contraction.)
//
// How: a field element is an INTEGER carried in a double, congruent mod p to the value, of bounded magnitude
// ("lazy").  Integers below 2^53 are exact in a double, an FMA rounds once, so every step below is exact integer
// arithmetic as long as the stated bounds hold; they are stated at each function and the whole permutation is
// compared with the integer one over 2^24 states by tools/microbench/p2_f64_bench.hip and by the GPU parity
// tests (Merkle roots and proof bytes against the CPU oracle).
//   mm(a, b):  h = RN(a b), l = a b - h (exact, one FMA), q = round(h / p), r = (h - q p) + l.
//              Needs |a b| < 2^82 (q < 2^51 for the rounding trick).  |r| <= p/2 + |h| 2^-52 + |l| < 0.51 p
//              for |a b| < 2^76.
// HBM keeps Montgomery words (bb.cuh); from_mont / to_mont convert at the edges (one product each).
#pragma once
#include <math.h>
#include "bb.cuh"
#include "poseidon2_rc.inc"

namespace dvt {
namespace p2f {

// (host-callable as well: IEEE doubles and fma() behave the same there, which is how the CPU test suite checks
// this file against the integer permutation without a GPU)
#if defined(__HIP_DEVICE_COMPILE__)
#define DVT_F64_TABLE static __constant__ const
#else
#define DVT_F64_TABLE static const
#endif
#define DVT_DEV DVT_HD
DVT_F64_TABLE double RC_EXT[128] = DVT_P2_RC_EXT_F64_INIT;
DVT_F64_TABLE double RC_INT[13] = DVT_P2_RC_INT_F64_INIT;
DVT_F64_TABLE double DIAG[16] = DVT_P2_DIAG_F64_INIT;

constexpr double PD = 2013265921.0;
constexpr double PINV = 1.0 / 2013265921.0;
constexpr double MONT_R = 268435454.0;        // 2^32 mod p
constexpr double MONT_RINV = 943718400.0;     // 2^-32 mod p

// round(a * b) for |a b| < 2^51 in two full-rate operations: the fused multiply-add a * b + 1.5 * 2^52 is rounded once, at
// unit precision (its result lies in [2^52, 2^53)), i.e. to the nearest integer; subtracting the constant is exact.
// (v_mul_f64 + v_rndne_f64 is the same instruction count and measures the same within 1 %.)  Every quotient estimate
// below is such a rounded product; an estimate off by one only moves the representative by p, inside the stated bounds.
// Instruction budget of one permutation (gfx950 ISA of the bare kernel): 4 854 FP64 operations + 117 moves; at four
// issue cycles per wave64 operation that is 19.9 k cycles per wave, measured 20.5 k: the permutation is issue-bound.
constexpr double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
DVT_DEV double rnd_prod(double a, double b) { return fma(a, b, MAGIC) - MAGIC; }
DVT_DEV double mm(double a, double b) {
    double h = a * b;
    double l = fma(a, b, -h);
    double q = rnd_prod(h, PINV);
    return fma(-q, PD, h) + l;
}
// |a| < 2^51  ->  the representative in [-p/2, p/2] (+- a rounding slack far below 1 for |a| < 2^48)
DVT_DEV double red(double a) { return fma(-rnd_prod(a, PINV), PD, a); }

DVT_DEV double from_canonical(uint32_t c) { return (double)c; }
// r in (-p, p) -> canonical word
DVT_DEV uint32_t fix(double r) { return (uint32_t)(r < 0.0 ? r + PD : r); }
DVT_DEV uint32_t to_canonical(double x) { return fix(red(x)); }       // |x| < 2^48

// a * b mod p with bp = b / p (rounded) supplied: q = round(a * bp), then a b - q p = (a b - q (p - 1)) - q, where
// q (p - 1) = 15 q * 2^27 is exact in a double and the FMA result (|r + q| < 2^42) is exact: 5 operations, no error
// term.  Same bounds as mm.  Worth it where one bp serves several products (the S-box: 23 operations instead of 24).
constexpr double P_MINUS_1 = 2013265920.0;
DVT_DEV double mm_pre(double a, double b, double bp) {
    const double q = rnd_prod(a, bp);
    return fma(a, b, -(q * P_MINUS_1)) - q;
}
DVT_DEV double from_mont(uint32_t m) { return mm_pre((double)m, MONT_RINV, MONT_RINV / PD); }
DVT_DEV uint32_t to_mont(double x) { return fix(mm_pre(x, MONT_R, MONT_R / PD)); }      // |x| < 2^48
DVT_DEV double sbox(double x) {  // |x| < 2^38
    const double xp = x * PINV;
    const double x2 = mm_pre(x, x, xp), x3 = mm_pre(x2, x, xp);
    const double x2p = x2 * PINV, x4 = mm_pre(x2, x2, x2p);
    return mm_pre(x3, x4, x4 * PINV);
}

// max |s| = B  ->  <= 35 B
DVT_DEV void external_layer(double s[16]) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        // circ(2,3,1,1) in 9 operations: y0 = 2x0+3x1+x2+x3, y1 = x0+2x1+3x2+x3, y2 = x0+x1+2x2+3x3, y3 = 3x0+x1+x2+2x3
        const double x0 = s[4 * c], x1 = s[4 * c + 1], x2 = s[4 * c + 2], x3 = s[4 * c + 3];
        const double t01 = x0 + x1, t23 = x2 + x3, t = t01 + t23;
        const double u = t + x1, v = t + x3;          // x0+2x1+x2+x3,  x0+x1+x2+2x3
        s[4 * c + 0] = u + t01;
        s[4 * c + 1] = fma(2.0, x2, u);
        s[4 * c + 2] = v + t23;
        s[4 * c + 3] = fma(2.0, x0, v);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double sum = (s[k] + s[4 + k]) + (s[8 + k] + s[12 + k]);
#pragma unroll
        for (int c = 0; c < 4; c++) s[4 * c + k] += sum;
    }
}

// y_i = sum + d_i x_i, d = [-2, 1, 2, 3, 4, -3, -4, 5, -5, 6, -6, 7, 8, -8, 9, -1] (tools/gen_poseidon2_rc.py): one fused
// multiply-add per entry, exact.  max |s| = B (< 2^48) -> <= 25 B.
template <bool REDUCE>
DVT_DEV void internal_layer(double s[16]) {
    if (REDUCE) {
#pragma unroll
        for (int i = 1; i < 16; i++) s[i] = red(s[i]);
    }
    double sum = s[0];
#pragma unroll
    for (int i = 1; i < 16; i++) sum += s[i];
    s[0] = fma(-2.0, s[0], sum);
    s[1] = sum + s[1];
    s[2] = fma(2.0, s[2], sum);
    s[3] = fma(3.0, s[3], sum);
    s[4] = fma(4.0, s[4], sum);
    s[5] = fma(-3.0, s[5], sum);
    s[6] = fma(-4.0, s[6], sum);
    s[7] = fma(5.0, s[7], sum);
    s[8] = fma(-5.0, s[8], sum);
    s[9] = fma(6.0, s[9], sum);
    s[10] = fma(-6.0, s[10], sum);
    s[11] = fma(7.0, s[11], sum);
    s[12] = fma(8.0, s[12], sum);
    s[13] = fma(-8.0, s[13], sum);
    s[14] = fma(9.0, s[14], sum);
    s[15] = sum - s[15];
}

// in: |s_i| < 2^32 (canonical words, from_mont results or a previous output); out: |s_i| < 0.51 p.
// Magnitudes: after an S-box layer every entry is below 0.51 p (2^30.03), the external layer multiplies the bound
// by 35 (2^35.2, the S-box input bound with a round constant added); in the internal rounds s[0] is reduced before
// each S-box and the other entries every third round: 2^35.2 -> x25 -> 2^39.8 -> 2^44.5 -> reduce, and from a
// reduced state 2^30 -> 2^34.7 -> 2^39.3 -> 2^44 (< 2^48 as the sums need).
DVT_DEV void permute(double s[16]) {
    external_layer(s);
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox(s[i] + RC_EXT[16 * r + i]);
        external_layer(s);
    }
#pragma unroll
    for (int r = 0; r < 13; r++) {
        s[0] = sbox(red(s[0]) + RC_INT[r]);
        if (r == 2 || r == 5 || r == 8 || r == 11) internal_layer<true>(s); else internal_layer<false>(s);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = red(s[i]);
#pragma unroll
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox(s[i] + RC_EXT[16 * r + i]);
        external_layer(s);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = red(s[i]);
}

#if defined(__HIPCC__)
// ---- the same permutation, one state per 16 LANES (lane e of a DPP row holds state element e) ---------------------
// A thread that owns a whole state runs ~5 600 FP64 operations back to back: ~9 us however few states there are, which
// is what the small Merkle levels and every tree top cost per level.  Spread over a DPP row the S-boxes of a round run
// in parallel and the linear layers become quad permutes / row rotations (v_mov_b32_dpp, no LDS): ~1 000 dependent
// operations per permutation.  Used where there are too few states to fill the machine anyway (merkle.hip).
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_QUAD_NEXT = 0x39;   // quad_perm [1,2,3,0]: lane k reads lane k+1 of its quad
constexpr int DPP_QUAD_XOR1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;   // quad_perm [2,3,0,1]
constexpr int DPP_ROW_ROR = 0x120;    // + n: rotate the 16-lane row by n

struct CoopConsts {
    double rc[8];   // external round constants of this lane's element
    double diag;    // internal diagonal entry of this lane's element (centred residue)
    bool lane0;
};
__device__ __forceinline__ CoopConsts coop_consts(uint32_t e) {
    CoopConsts k;
#pragma unroll
    for (int r = 0; r < 8; r++) k.rc[r] = RC_EXT[16 * r + e];
    k.diag = DIAG[e];
    k.lane0 = e == 0;
    return k;
}
// max |s| = B -> <= 35 B (as external_layer)
__device__ __forceinline__ double coop_external_layer(double x) {
    const double x1 = dpp_mov<DPP_QUAD_NEXT>(x);
    double t = x + dpp_mov<DPP_QUAD_XOR1>(x);
    t = t + dpp_mov<DPP_QUAD_XOR2>(t);
    const double y = fma(2.0, x1, t + x);
    double a = y + dpp_mov<DPP_ROW_ROR + 8>(y);
    a = a + dpp_mov<DPP_ROW_ROR + 4>(a);
    return y + a;
}
// in: |s| < 2^32; out: |s| < 0.51 p (every lane of the row must be active)
__device__ __forceinline__ double coop_permute(double s, const CoopConsts &k) {
    s = coop_external_layer(s);
#pragma unroll
    for (int r = 0; r < 4; r++) s = coop_external_layer(sbox(s + k.rc[r]));
    s = red(s);
#pragma unroll
    for (int r = 0; r < 13; r++) {
        const double x = sbox(s + RC_INT[r]);
        s = k.lane0 ? x : s;
        double t = s + dpp_mov<DPP_ROW_ROR + 8>(s);
        t = t + dpp_mov<DPP_ROW_ROR + 4>(t);
        t = t + dpp_mov<DPP_ROW_ROR + 2>(t);
        t = t + dpp_mov<DPP_ROW_ROR + 1>(t);
        // every diagonal entry is a small integer and s is reduced: the fused multiply-add is exact
        s = red(fma(s, k.diag, t));                      // every entry stays reduced: |sum| <= 16 * 0.51 p
    }
#pragma unroll
    for (int r = 4; r < 8; r++) s = coop_external_layer(sbox(s + k.rc[r]));
    return red(s);
}
#endif

}  // namespace p2f
}  // namespace dvt
