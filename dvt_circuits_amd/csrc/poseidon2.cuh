// Poseidon2 over BabyBear (width 16, x^7, 8 external + 13 internal rounds),
// sponge (rate 8 -> 8-element digest) and 2-to-1 truncated-permutation
// compression, for gfx950 kernels and the host-side transcript/verifier.
// All values are Montgomery-form Fp (bb.cuh); the permutation commutes with the
// Montgomery isomorphism, so digests are "Montgomery digests" until serialised.
//
// Stands in for p3-poseidon2 / sp1-primitives behind reference
// src/main.rs:461-466 (sources absent; parameters per SURVEY.md Appendix C,
// constants per tools/gen_poseidon2_rc.py).
#pragma once
#include "bb.cuh"
#include "poseidon2_rc.inc"

namespace dvt {

#if defined(__HIP_DEVICE_COMPILE__)
#define DVT_TABLE static __constant__ const
#else
#define DVT_TABLE static const
#endif
DVT_TABLE uint32_t P2_RC_EXT[128] = DVT_P2_RC_EXT_INIT;
DVT_TABLE uint32_t P2_RC_INT[13] = DVT_P2_RC_INT_INIT;
DVT_TABLE uint32_t P2_DIAG[16] = DVT_P2_DIAG_INIT;  // (documentation / cross-check only: the layer below is table-free)

constexpr int P2_WIDTH = 16;
constexpr int P2_RATE = 8;
constexpr int P2_DIGEST = 8;

DVT_HD Fp p2_sbox(Fp x) {
    Fp x2 = x * x, x3 = x2 * x, x4 = x2 * x2;
    return x3 * x4;
}

DVT_HD void p2_external_layer(Fp s[16]) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        Fp x0 = s[4 * c], x1 = s[4 * c + 1], x2 = s[4 * c + 2], x3 = s[4 * c + 3];
        // circ(2,3,1,1): y_i = sum + x_i + 2 x_{i+1}
        Fp t = x0 + x1 + x2 + x3;
        s[4 * c + 0] = t + x0 + dbl(x1);
        s[4 * c + 1] = t + x1 + dbl(x2);
        s[4 * c + 2] = t + x2 + dbl(x3);
        s[4 * c + 3] = t + x3 + dbl(x0);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        Fp sum = s[k] + s[4 + k] + s[8 + k] + s[12 + k];
#pragma unroll
        for (int c = 0; c < 4; c++) s[4 * c + k] += sum;
    }
}

// K x for a small literal K (the internal diagonal): doublings and additions, no Montgomery product
template <int K>
DVT_HD Fp p2_mul_small(Fp x) {
    if constexpr (K < 0) return -p2_mul_small<-K>(x);
    else if constexpr (K == 1) return x;
    else if constexpr (K % 2 == 0) return dbl(p2_mul_small<K / 2>(x));
    else return dbl(p2_mul_small<K / 2>(x)) + x;
}
// y_i = sum(x) + d_i x_i with d = [-2, 1, 2, 3, 4, -3, -4, 5, -5, 6, -6, 7, 8, -8, 9, -1] (tools/gen_poseidon2_rc.py)
DVT_HD void p2_internal_layer(Fp s[16]) {
    Fp sum = s[0];
#pragma unroll
    for (int i = 1; i < 16; i++) sum += s[i];
    s[0] = sum + p2_mul_small<-2>(s[0]);
    s[1] = sum + s[1];
    s[2] = sum + p2_mul_small<2>(s[2]);
    s[3] = sum + p2_mul_small<3>(s[3]);
    s[4] = sum + p2_mul_small<4>(s[4]);
    s[5] = sum + p2_mul_small<-3>(s[5]);
    s[6] = sum + p2_mul_small<-4>(s[6]);
    s[7] = sum + p2_mul_small<5>(s[7]);
    s[8] = sum + p2_mul_small<-5>(s[8]);
    s[9] = sum + p2_mul_small<6>(s[9]);
    s[10] = sum + p2_mul_small<-6>(s[10]);
    s[11] = sum + p2_mul_small<7>(s[11]);
    s[12] = sum + p2_mul_small<8>(s[12]);
    s[13] = sum + p2_mul_small<-8>(s[13]);
    s[14] = sum + p2_mul_small<9>(s[14]);
    s[15] = sum + p2_mul_small<-1>(s[15]);
}

DVT_HD void p2_permute(Fp s[16]) {
    p2_external_layer(s);
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = p2_sbox(s[i] + Fp::raw(P2_RC_EXT[16 * r + i]));
        p2_external_layer(s);
    }
#pragma unroll
    for (int r = 0; r < 13; r++) {
        s[0] = p2_sbox(s[0] + Fp::raw(P2_RC_INT[r]));
        p2_internal_layer(s);
    }
#pragma unroll
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = p2_sbox(s[i] + Fp::raw(P2_RC_EXT[16 * r + i]));
        p2_external_layer(s);
    }
}

struct Digest {
    Fp d[8];
    DVT_HD bool operator==(const Digest &o) const {
        bool e = true;
        for (int i = 0; i < 8; i++) e = e && (d[i] == o.d[i]);
        return e;
    }
};

DVT_HD Digest p2_compress(const Digest &l, const Digest &r) {
    Fp s[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { s[i] = l.d[i]; s[8 + i] = r.d[i]; }
    p2_permute(s);
    Digest o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.d[i] = s[i];
    return o;
}

// Incremental overwrite sponge: absorb() elements one at a time, finish() pads nothing.
struct Sponge {
    Fp s[16];
    int pos;
    DVT_HD Sponge() : pos(0) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = Fp::zero();
    }
    DVT_HD void absorb(Fp x) {
        s[pos++] = x;
        if (pos == P2_RATE) { p2_permute(s); pos = 0; }
    }
    DVT_HD Digest finish() {
        if (pos) { p2_permute(s); pos = 0; }
        Digest o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.d[i] = s[i];
        return o;
    }
};

}  // namespace dvt
