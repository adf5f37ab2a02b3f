// Host-side shard verifier (no device work).  Mirrors Engine::prove_shard's
// transcript step for step; see DESIGN.md "Protocol".  The reference's own
// `verify` sub-command is not a STARK verifier; this
// plays the role of stock `client.verify(&proof, &vk)`.
#include <cstring>

#include "engine.h"

namespace dvt {
namespace {

Digest hash_rows(const std::vector<const std::vector<Fp> *> &rows) {
    Sponge sp;
    for (auto *r : rows)
        for (Fp x : *r) sp.absorb(x);
    return sp.finish();
}

struct TreeShape {
    std::vector<std::pair<uint32_t, uint32_t>> mats;  // (width, log_h) in tree order
    uint32_t log_h = 0;
};

bool verify_tree_opening(const TreeShape &shape, const TreeOpening &op, uint32_t idx, const Digest &root) {
    if (op.rows.size() != shape.mats.size() || op.path.size() != shape.log_h) return false;
    for (size_t i = 0; i < shape.mats.size(); i++)
        if (op.rows[i].size() != shape.mats[i].first) return false;
    auto rows_at = [&](uint32_t lh) {
        std::vector<const std::vector<Fp> *> r;
        for (size_t i = 0; i < shape.mats.size(); i++)
            if (shape.mats[i].second == lh) r.push_back(&op.rows[i]);
        return r;
    };
    Digest cur = hash_rows(rows_at(shape.log_h));
    uint32_t j = idx & ((1u << shape.log_h) - 1);
    for (uint32_t s = shape.log_h; s >= 1; s--) {
        uint32_t half = 1u << (s - 1);
        const Digest &sib = op.path[shape.log_h - s];
        cur = j < half ? p2_compress(cur, sib) : p2_compress(sib, cur);
        j &= half - 1;
        auto inj = rows_at(s - 1);
        if (!inj.empty()) cur = p2_compress(cur, hash_rows(inj));
    }
    return cur == root;
}

bool verify_path(Digest cur, const std::vector<Digest> &path, uint32_t log_h, uint32_t leaf, const Digest &root) {
    if (path.size() != log_h) return false;
    uint32_t j = leaf & ((1u << log_h) - 1);
    for (uint32_t s = log_h; s >= 1; s--) {
        uint32_t half = 1u << (s - 1);
        cur = j < half ? p2_compress(cur, path[log_h - s]) : p2_compress(path[log_h - s], cur);
        j &= half - 1;
    }
    return cur == root;
}

Fp4 ext_from_flat(const Fp4 *p) {  // sum_k x^k * p[k]
    Fp4 acc = p[0];
    for (int k = 1; k < 4; k++) {
        Fp4 b = Fp4::zero();
        b.c[k] = Fp::one();
        acc += b * p[k];
    }
    return acc;
}

}  // namespace

std::string verify_shard(const VerifyingKey &vk, const ShardProof &pf, const StarkConfig &cfg, const PermChallenges *global,
                         Fp4 *cumsum_total) {
    const MachineDesc *m = vk.machine;
    if (!m) return "no machine";
    if (pf.chips.empty()) return "no chips";
    std::vector<ChipRef> refs;
    uint32_t max_log_n = 0;
    int max_pub = 0;
    for (auto &o : pf.chips) {
        if (o.chip_id >= (uint32_t)m->n_chips) return "chip id out of range";
        if (!refs.empty() && (int)o.chip_id <= refs.back().chip_id) return "chips not sorted";
        if (o.log_n > 22) return "log_n too large";
        const ChipDesc &d = m->chips[o.chip_id];
        if (o.prep_l.size() != (size_t)d.prep_w || o.prep_n.size() != (size_t)d.prep_w || o.main_l.size() != (size_t)d.main_w ||
            o.main_n.size() != (size_t)d.main_w || o.perm_l.size() != (size_t)4 * d.perm_ext_w ||
            o.perm_n.size() != (size_t)4 * d.perm_ext_w || o.quot.size() != 8)
            return std::string("opened value shape mismatch for chip ") + d.name;
        refs.push_back({(int)o.chip_id, o.log_n});
        max_log_n = std::max(max_log_n, o.log_n);
        max_pub = std::max(max_pub, d.n_pub);
    }
    if ((int)pf.public_values.size() < max_pub) return "too few public values";
    for (auto &pc : vk.prep_chips) {
        bool ok = false;
        for (auto &r : refs) ok |= r.chip_id == pc.chip_id && r.log_n == pc.log_n;
        if (!ok) return "preprocessed chip missing or of wrong height";
    }
    for (auto &r : refs)
        if (m->chips[r.chip_id].prep_w) {
            bool ok = false;
            for (auto &pc : vk.prep_chips) ok |= pc.chip_id == r.chip_id;
            if (!ok) return "chip with preprocessed columns is not in the verifying key";
        }
    const uint32_t hmax = max_log_n + 1;
    if (pf.fri_roots.size() != hmax - 1) return "wrong number of FRI layers";
    if (pf.queries.size() != cfg.num_queries) return "wrong number of queries";

    // ---- transcript
    Challenger ch;
    transcript_begin(ch, vk, refs);
    ch.observe(pf.main_root);
    ch.observe_u32((uint32_t)pf.public_values.size());
    for (auto x : pf.public_values) ch.observe(x);
    Fp4 perm_alpha, beta;
    if (global) {
        perm_alpha = global->alpha;
        beta = global->beta;
        ch.observe(perm_alpha);
        ch.observe(beta);
    } else {
        perm_alpha = ch.sample_ext();
        beta = ch.sample_ext();
    }
    ch.observe(pf.perm_root);
    Fp4 total = Fp4::zero();
    for (auto &o : pf.chips) {
        const ChipDesc &d = m->chips[o.chip_id];
        if (!d.perm_ext_w && o.cumsum != Fp4::zero()) return "cumulative sum on a chip without interactions";
        ch.observe(o.cumsum);
        total += o.cumsum;
    }
    if (cumsum_total) *cumsum_total = total;
    else if (total != Fp4::zero()) return "LogUp cumulative sums do not cancel";
    Fp4 alpha = ch.sample_ext();
    ch.observe(pf.quot_root);
    Fp4 zeta = ch.sample_ext();
    for (auto &o : pf.chips)
        for (auto *v : {&o.prep_l, &o.prep_n, &o.main_l, &o.main_n, &o.perm_l, &o.perm_n, &o.quot}) ch.observe_values(*v);

    // ---- constraints at zeta
    int max_arity = 1, max_folded = 1;
    for (int i = 0; i < m->n_chips; i++) {
        max_arity = std::max(max_arity, m->chips[i].max_arity);
        max_folded = std::max(max_folded, m->chips[i].n_folded);
    }
    std::vector<Fp4> beta_pows(max_arity), alpha_pows(max_folded);
    { Fp4 x = beta; for (auto &b : beta_pows) { b = x; x = x * beta; } }
    { Fp4 x = Fp4::one(); for (auto &a : alpha_pows) { a = x; x = x * alpha; } }
    const Fp g = Fp::from_canonical(COSET_SHIFT);
    for (auto &o : pf.chips) {
        const ChipDesc &d = m->chips[o.chip_id];
        const size_t n = (size_t)1 << o.log_n;
        VerifierAccess ax{o.main_l.data(), o.main_n.data(), o.prep_l.data(), o.prep_n.data(), o.perm_l.data(), o.perm_n.data(),
                          pf.public_values.data()};
        VerifierPoint pt;
        pt.alpha_pows = alpha_pows.data();
        pt.beta_pows = beta_pows.data();
        pt.perm_alpha = perm_alpha;
        pt.cum_over_n = o.cumsum * inv(Fp::from_canonical((uint32_t)n));
        Fp w_inv = inv(two_adic_generator(o.log_n));
        Fp4 zh = pow(zeta, n) - Fp::one();
        Fp4 d1 = zeta - Fp::one(), d2 = zeta - w_inv;
        if (d1 == Fp4::zero() || d2 == Fp4::zero() || zh == Fp4::zero()) return "zeta lies in the trace domain";
        pt.sel_first = zh * inv(d1);
        pt.sel_last = zh * inv(d2);
        pt.sel_trans = d2;
        Fp4 folded = d.verify_eval(ax, pt);
        Fp4 r0 = ext_from_flat(o.quot.data()), r1 = ext_from_flat(o.quot.data() + 4);
        Fp4 u = pow(zeta * inv(g), n);
        Fp4 zd0 = u - Fp::one(), zd1 = -u - Fp::one();
        Fp4 q = (r0 * zd1 + r1 * zd0) * inv(-Fp::two());
        if (folded != zh * q) return std::string("constraint check failed at zeta for chip ") + d.name;
    }

    // ---- FRI
    Fp4 alpha_fri = ch.sample_ext();
    std::vector<Fp4> fold_betas;
    for (auto &r : pf.fri_roots) {
        ch.observe(r);
        fold_betas.push_back(ch.sample_ext());
    }
    ch.observe(pf.final_poly);
    if (!ch.check_witness(cfg.pow_bits, pf.pow_witness)) return "proof-of-work witness rejected";

    TreeShape shapes[4];
    for (auto &r : refs) {
        const ChipDesc &d = m->chips[r.chip_id];
        if (d.prep_w) shapes[0].mats.push_back({(uint32_t)d.prep_w, r.log_n + 1});
        shapes[1].mats.push_back({(uint32_t)d.main_w, r.log_n + 1});
        if (d.perm_ext_w) shapes[2].mats.push_back({(uint32_t)(4 * d.perm_ext_w), r.log_n + 1});
        shapes[3].mats.push_back({8u, r.log_n + 1});
    }
    for (auto &sh : shapes)
        for (auto &mt : sh.mats) sh.log_h = std::max(sh.log_h, mt.second);
    const Digest *roots[4] = {&vk.prep_root, &pf.main_root, &pf.perm_root, &pf.quot_root};

    std::vector<std::vector<ColRef>> cols_by_h(hmax + 1);
    std::vector<uint32_t> n_two_by_h(hmax + 1, 0);
    size_t max_cols = 1;
    for (uint32_t h = 1; h <= hmax; h++) {
        cols_by_h[h] = fri_columns(m, refs, h, &n_two_by_h[h]);
        max_cols = std::max(max_cols, cols_by_h[h].size());
    }
    std::vector<Fp4> apow(max_cols + 1);
    { Fp4 x = Fp4::one(); for (auto &a : apow) { a = x; x = x * alpha_fri; } }
    const Fp inv2 = inv(Fp::two());

    for (uint32_t qi = 0; qi < cfg.num_queries; qi++) {
        const QueryProof &q = pf.queries[qi];
        const uint32_t idx = ch.sample_bits(hmax);
        for (int t = 0; t < 4; t++) {
            if (shapes[t].mats.empty()) {
                if (!q.trees[t].rows.empty() || !q.trees[t].path.empty()) return "unexpected opening for an empty tree";
                continue;
            }
            if (!verify_tree_opening(shapes[t], q.trees[t], idx, *roots[t])) return "Merkle opening rejected (input tree)";
        }
        auto reduced = [&](uint32_t h) {
            const auto &cols = cols_by_h[h];
            Fp x = g * pow(two_adic_generator(h), idx & ((1u << h) - 1));
            Fp4 s_all = Fp4::zero(), s_two = Fp4::zero();
            for (size_t c = 0; c < cols.size(); c++) {
                const ColRef &r = cols[c];
                const ChipOpening &o = pf.chips[r.chip_pos];
                Fp px = q.trees[r.tree].rows[r.mat][r.col];
                const std::vector<Fp4> &loc = r.tree == 0 ? o.prep_l : r.tree == 1 ? o.main_l : r.tree == 2 ? o.perm_l : o.quot;
                s_all += apow[c] * (Fp4::from_base(px) - loc[r.col]);
                if (r.tree < 3) {
                    const std::vector<Fp4> &nx = r.tree == 0 ? o.prep_n : r.tree == 1 ? o.main_n : o.perm_n;
                    s_two += apow[c] * (Fp4::from_base(px) - nx[r.col]);
                }
            }
            Fp4 zeta_next = zeta * two_adic_generator(h - 1);
            Fp4 r = s_all * inv(Fp4::from_base(x) - zeta);
            if (n_two_by_h[h]) r += apow[cols.size()] * (s_two * inv(Fp4::from_base(x) - zeta_next));
            return r;
        };
        if (q.layers.size() != hmax - 1) return "wrong number of FRI layer openings";
        Fp4 e = reduced(hmax);
        for (uint32_t k = 0; k + 1 < hmax; k++) {
            const uint32_t lm = hmax - k, half = 1u << (lm - 1);
            const uint32_t j = idx & ((1u << lm) - 1), jl = j & (half - 1);
            const FriLayerOpening &lo = q.layers[k];
            Fp4 a = j < half ? e : lo.sibling, b = j < half ? lo.sibling : e;
            Sponge sp;
            for (int c = 0; c < 4; c++) sp.absorb(a.c[c]);
            for (int c = 0; c < 4; c++) sp.absorb(b.c[c]);
            if (!verify_path(sp.finish(), lo.path, lm - 1, jl, pf.fri_roots[k])) return "Merkle opening rejected (FRI layer)";
            Fp xinv = inv(pow(two_adic_generator(lm), jl));
            e = (a + b) * inv2 + fold_betas[k] * ((a - b) * (inv2 * xinv));
            if (!cols_by_h[lm - 1].empty()) e += reduced(lm - 1);
        }
        if (e != pf.final_poly) return "FRI final value mismatch";
    }
    return "";
}

}  // namespace dvt
