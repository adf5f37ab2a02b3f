// Shard proof container and its wire format.
//
// The reference stores `SP1ProofWithPublicValues` with bincode (src/main.rs:472-474);
// that layout lives in the absent sp1-sdk crate, so this
// library defines its own: a flat little-endian stream of u32 words, every field
// element in canonical form, every vector preceded by its length.  DESIGN.md
// "Proof format" lists the fields in order.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "poseidon2.cuh"

namespace dvt {

struct ChipOpening {
    uint32_t chip_id = 0, log_n = 0;
    Fp4 cumsum;
    std::vector<Fp4> prep_l, prep_n, main_l, main_n, perm_l, perm_n, quot;  // quot: 2 chunks x 4 columns
};
struct TreeOpening {
    std::vector<std::vector<Fp>> rows;  // one opened row per matrix of the tree, in tree order
    std::vector<Digest> path;
};
struct FriLayerOpening {
    Fp4 sibling;
    std::vector<Digest> path;
};
struct QueryProof {
    TreeOpening trees[4];  // preprocessed, main, permutation, quotient
    std::vector<FriLayerOpening> layers;
};
struct ShardProof {
    Digest main_root, perm_root, quot_root;
    std::vector<Fp> public_values;
    std::vector<ChipOpening> chips;
    std::vector<Digest> fri_roots;
    Fp4 final_poly;
    Fp pow_witness;
    std::vector<QueryProof> queries;
    // the prover fills the query section directly in wire format (canonical words, exactly what write_shard_proof
    // emits for `queries`, including the leading count) instead of building `queries`: one pass over the downloaded
    // gather buffer, no per-query vectors.  Empty = serialise `queries`.
    std::vector<uint32_t> query_words;
};

struct WordWriter {
    std::vector<uint32_t> w;
    void u32(uint32_t v) { w.push_back(v); }
    void fp(Fp x) { w.push_back(x.canonical()); }
    void ef(const Fp4 &x) { for (int i = 0; i < 4; i++) fp(x.c[i]); }
    void dg(const Digest &d) { for (int i = 0; i < 8; i++) fp(d.d[i]); }
    void fps(const std::vector<Fp> &v) { u32((uint32_t)v.size()); for (auto x : v) fp(x); }
    void efs(const std::vector<Fp4> &v) { u32((uint32_t)v.size()); for (auto &x : v) ef(x); }
    void dgs(const std::vector<Digest> &v) { u32((uint32_t)v.size()); for (auto &x : v) dg(x); }
};
struct WordReader {
    const uint32_t *p, *end;
    WordReader(const uint32_t *b, size_t n) : p(b), end(b + n) {}
    uint32_t u32() { if (p >= end) throw std::runtime_error("proof truncated"); return *p++; }
    uint32_t len(uint32_t max) { uint32_t n = u32(); if (n > max) throw std::runtime_error("proof: vector too long"); return n; }
    Fp fp() { uint32_t v = u32(); if (v >= P) throw std::runtime_error("proof: non-canonical field element"); return Fp::from_canonical(v); }
    Fp4 ef() { Fp4 r; for (int i = 0; i < 4; i++) r.c[i] = fp(); return r; }
    Digest dg() { Digest d; for (int i = 0; i < 8; i++) d.d[i] = fp(); return d; }
    std::vector<Fp> fps(uint32_t max = 1 << 20) { uint32_t n = len(max); std::vector<Fp> v(n); for (auto &x : v) x = fp(); return v; }
    std::vector<Fp4> efs(uint32_t max = 1 << 20) { uint32_t n = len(max); std::vector<Fp4> v(n); for (auto &x : v) x = ef(); return v; }
    std::vector<Digest> dgs(uint32_t max = 64) { uint32_t n = len(max); std::vector<Digest> v(n); for (auto &x : v) x = dg(); return v; }
};

constexpr uint32_t SHARD_PROOF_MAGIC = 0x31505644u;  // "DVP1"

inline void write_shard_proof(WordWriter &w, const ShardProof &p) {
    w.u32(SHARD_PROOF_MAGIC);
    w.dg(p.main_root); w.dg(p.perm_root); w.dg(p.quot_root);
    w.fps(p.public_values);
    w.u32((uint32_t)p.chips.size());
    for (auto &c : p.chips) {
        w.u32(c.chip_id); w.u32(c.log_n); w.ef(c.cumsum);
        w.efs(c.prep_l); w.efs(c.prep_n); w.efs(c.main_l); w.efs(c.main_n);
        w.efs(c.perm_l); w.efs(c.perm_n); w.efs(c.quot);
    }
    w.dgs(p.fri_roots); w.ef(p.final_poly); w.fp(p.pow_witness);
    if (!p.query_words.empty()) {
        w.w.insert(w.w.end(), p.query_words.begin(), p.query_words.end());
        return;
    }
    w.u32((uint32_t)p.queries.size());
    for (auto &q : p.queries) {
        for (int t = 0; t < 4; t++) {
            w.u32((uint32_t)q.trees[t].rows.size());
            for (auto &r : q.trees[t].rows) w.fps(r);
            w.dgs(q.trees[t].path);
        }
        w.u32((uint32_t)q.layers.size());
        for (auto &l : q.layers) { w.ef(l.sibling); w.dgs(l.path); }
    }
}

inline ShardProof read_shard_proof(WordReader &r) {
    ShardProof p;
    if (r.u32() != SHARD_PROOF_MAGIC) throw std::runtime_error("proof: bad magic");
    p.main_root = r.dg(); p.perm_root = r.dg(); p.quot_root = r.dg();
    p.public_values = r.fps(1 << 12);
    uint32_t nc = r.len(64);
    p.chips.resize(nc);
    for (auto &c : p.chips) {
        c.chip_id = r.u32(); c.log_n = r.u32(); c.cumsum = r.ef();
        // (untrusted input: bound both before anything indexes with them; the machine's own chip count is checked by the verifier)
        if (c.chip_id >= 64 || c.log_n > 22) throw std::runtime_error("proof: chip id or height out of range");
        c.prep_l = r.efs(1 << 12); c.prep_n = r.efs(1 << 12); c.main_l = r.efs(1 << 12); c.main_n = r.efs(1 << 12);
        c.perm_l = r.efs(1 << 12); c.perm_n = r.efs(1 << 12); c.quot = r.efs(8);
    }
    p.fri_roots = r.dgs(); p.final_poly = r.ef(); p.pow_witness = r.fp();
    uint32_t nq = r.len(1024);
    p.queries.resize(nq);
    for (auto &q : p.queries) {
        for (int t = 0; t < 4; t++) {
            uint32_t nm = r.len(256);
            q.trees[t].rows.resize(nm);
            for (auto &row : q.trees[t].rows) row = r.fps(1 << 12);
            q.trees[t].path = r.dgs();
        }
        uint32_t nl = r.len(64);
        q.layers.resize(nl);
        for (auto &l : q.layers) { l.sibling = r.ef(); l.path = r.dgs(); }
    }
    return p;
}

}  // namespace dvt
