// Duplex-sponge Fiat-Shamir transcript over Poseidon2 (width 16, rate 8), host
// side (prover orchestration and verifier).  Semantics follow the public
// description of p3-challenger's DuplexChallenger (source absent, SURVEY.md
// Appendix C): observing clears pending outputs; sampling pops from the back of
// the squeezed rate; a proof-of-work witness w is valid when, after observe(w),
// the next `bits` sampled bits are zero.
#pragma once
#include <vector>

#include "poseidon2.cuh"

namespace dvt {

struct Challenger {
    Fp state[16];
    std::vector<Fp> input, output;
    Challenger() {
        for (auto &s : state) s = Fp::zero();
    }
    void duplex() {
        for (size_t i = 0; i < input.size(); i++) state[i] = input[i];
        input.clear();
        p2_permute(state);
        output.assign(state, state + P2_RATE);
    }
    void observe(Fp x) {
        output.clear();
        input.push_back(x);
        if (input.size() == (size_t)P2_RATE) duplex();
    }
    void observe(const Digest &d) {
        for (int i = 0; i < 8; i++) observe(d.d[i]);
    }
    void observe(const Fp4 &x) {
        for (int i = 0; i < 4; i++) observe(x.c[i]);
    }
    void observe_u32(uint32_t v) { observe(Fp::from_canonical(v % P)); }
    // The values a chip's matrix opens to at one point enter the transcript as ONE sponge digest of the vector (nothing for an
    // empty vector), not element by element: a shard with precompile chips opens ~5 000 columns at two points, and absorbing
    // ~20 k elements is ~2 500 SEQUENTIAL permutations the GPU would wait for; the digests of different vectors are independent
    // chains (the prover hashes them on a few host threads, engine.hip).
    static Digest hash_values(const std::vector<Fp4> &v) {
        Sponge sp;
        for (auto &x : v)
            for (int i = 0; i < 4; i++) sp.absorb(x.c[i]);
        return sp.finish();
    }
    void observe_values(const std::vector<Fp4> &v) {
        if (!v.empty()) observe(hash_values(v));
    }
    Fp sample() {
        if (!input.empty() || output.empty()) duplex();
        Fp r = output.back();
        output.pop_back();
        return r;
    }
    Fp4 sample_ext() {
        Fp4 r;
        for (int i = 0; i < 4; i++) r.c[i] = sample();
        return r;
    }
    uint32_t sample_bits(unsigned bits) { return sample().canonical() & ((1u << bits) - 1); }
    bool check_witness(unsigned bits, Fp w) {
        observe(w);
        return sample_bits(bits) == 0;
    }
};

}  // namespace dvt
