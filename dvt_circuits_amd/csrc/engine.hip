// Shard prover orchestration (host side of K1-K9) — see engine.h.
#include "engine.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>

namespace dvt {

// ------------------------------------------------------------------ shared layout helpers
std::vector<ColRef> fri_columns(const MachineDesc *m, const std::vector<ChipRef> &chips, uint32_t h, uint32_t *n_two) {
    std::vector<ColRef> out;
    for (int tree = 0; tree < 3; tree++) {
        int mat = 0;
        for (size_t pos = 0; pos < chips.size(); pos++) {
            const ChipDesc &d = m->chips[chips[pos].chip_id];
            int w = tree == 0 ? d.prep_w : tree == 1 ? d.main_w : 4 * d.perm_ext_w;
            if (w == 0) continue;
            if (chips[pos].log_n + 1 == h)
                for (int c = 0; c < w; c++) out.push_back({tree, mat, c, (int)pos});
            mat++;
        }
    }
    *n_two = (uint32_t)out.size();
    for (size_t pos = 0; pos < chips.size(); pos++)
        if (chips[pos].log_n + 1 == h)
            for (int c = 0; c < 8; c++) out.push_back({3, (int)pos, c, (int)pos});
    return out;
}

void transcript_begin(Challenger &ch, const VerifyingKey &vk, const std::vector<ChipRef> &chips) {
    ch.observe(vk.prep_root);
    ch.observe_u32((uint32_t)chips.size());
    for (auto &c : chips) {
        ch.observe_u32((uint32_t)c.chip_id);
        ch.observe_u32(c.log_n);
    }
}

// ------------------------------------------------------------------ arena / ring
hipError_t Arena::reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    hipError_t e = hipMalloc(&base, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
}
void Arena::release() {
    if (base) (void)hipFree(base);
    base = nullptr;
    cap = off = 0;
}

bool Engine::fail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return false;
}
#define HIPCHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) return fail("%s: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

hipError_t Engine::init(int dev) {
    device = dev;
    hipError_t e = hipSetDevice(dev);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = ntt_tables_create(&tabs);
    ring_bytes = 16u << 20;
    if (e == hipSuccess) e = hipMalloc(&d_ring, ring_bytes);
    if (e == hipSuccess) e = hipHostMalloc(&h_ring, ring_bytes);
    if (e == hipSuccess) e = hipHostMalloc(&h_down, DOWN_BYTES);
    return e;
}
void Engine::shutdown() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    arena.release();
    pool.trim();
    ntt_tables_destroy(&tabs);
    for (auto &t : sel_tables) { if (t) (void)hipFree(t); t = nullptr; }
    if (inject_buf) (void)hipFree(inject_buf);
    inject_buf = nullptr; inject_words = 0;
    if (d_ring) (void)hipFree(d_ring);
    if (h_ring) (void)hipHostFree(h_ring);
    if (h_down) (void)hipHostFree(h_down);
    h_down = nullptr;
    if (stream) (void)hipStreamDestroy(stream);
    d_ring = h_ring = nullptr;
    stream = nullptr;
}
const void *Engine::upload(const void *host, size_t bytes) {
    size_t need = (bytes + 255) & ~(size_t)255;
    if (need > ring_bytes) { err = "upload larger than ring"; return nullptr; }
    if (ring_pos + need > ring_bytes) {
        if (hipStreamSynchronize(stream) != hipSuccess) { err = "ring sync failed"; return nullptr; }
        ring_pos = 0;
    }
    memcpy(h_ring + ring_pos, host, bytes);
    if (hipMemcpyAsync(d_ring + ring_pos, h_ring + ring_pos, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) {
        err = "ring upload failed";
        return nullptr;
    }
    const void *r = d_ring + ring_pos;
    ring_pos += need;
    return r;
}
bool Engine::download(void *host, const void *dev, size_t bytes) {
    if (bytes <= DOWN_BYTES && h_down) {  // through pinned memory: a pageable destination makes the runtime stage and block
        HIPCHK(hipMemcpyAsync(h_down, dev, bytes, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        memcpy(host, h_down, bytes);
        return true;
    }
    HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    return true;
}

const uint32_t *Engine::selector_table(const QuotientArgs &qa) {
    if (qa.log_n >= 32) return nullptr;
    if (!sel_tables[qa.log_n]) {
        // (kept for the life of the prover: 24 bytes per trace row of that height; a failed allocation only means the
        //  kernels keep computing the selectors themselves)
        const size_t m = (size_t)2 << qa.log_n;
        uint32_t *t = nullptr;
        if (hipMalloc(&t, 3 * m * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        QuotientArgs a = qa;
        a.sel = nullptr;
        selector_table_kernel<0><<<(unsigned)((m + 255) / 256), 256, 0, stream>>>(a, t);
        if (hipGetLastError() != hipSuccess) { (void)hipFree(t); return nullptr; }
        sel_tables[qa.log_n] = t;
    }
    return sel_tables[qa.log_n];
}

bool Engine::commit_tree(const std::vector<DevMat> &mats, uint32_t *d_digests) {
    uint32_t mx = 0;
    for (auto &m : mats) mx = std::max(mx, m.log_h);
    if (mx >= MERKLE_MAX_SEGMENTS) return fail("commit: tree too tall");
    // every row sponge of the tree in one launch: the leaves and the row digests of the shorter matrices (joined to the
    // tree at the level of their height)
    MerkleLeafSegments sg;
    std::vector<const uint32_t *> inject(mx + 1, nullptr);
    size_t want = 0;
    for (uint32_t lh = 0; lh < mx; lh++)
        for (auto &m : mats)
            if (m.log_h == lh) { want += (size_t)8 << lh; break; }
    if (want > inject_words) {
        HIPCHK(hipStreamSynchronize(stream));   // (an earlier tree's level kernels may still read the old buffer)
        if (inject_buf) HIPCHK(hipFree(inject_buf));
        inject_buf = nullptr; inject_words = 0;
        HIPCHK(hipMalloc(&inject_buf, want * 4));
        inject_words = want;
    }
    size_t at = 0;
    for (uint32_t lh = mx + 1; lh-- > 0;) {
        std::vector<uint64_t> ptrs;
        for (auto &m : mats)
            if (m.log_h == lh)
                for (uint32_t c = 0; c < m.width; c++) ptrs.push_back((uint64_t)(uintptr_t)(m.ptr + ((size_t)c << lh)));
        if (ptrs.empty()) continue;
        auto d_cols = reinterpret_cast<const uint32_t *const *>(upload(ptrs.data(), ptrs.size() * 8));
        if (!d_cols) return false;
        uint32_t *out = d_digests;
        if (lh != mx) {
            out = inject_buf + at;
            at += (size_t)8 << lh;
            inject[lh] = out;
        }
        sg.add(d_cols, (uint32_t)ptrs.size(), lh, out);
    }
    HIPCHK(launch_merkle_leaves(stream, sg));
    uint32_t *prev = d_digests;
    uint32_t lh = mx;
    while (lh > MERKLE_TOP_LOG) {
        lh--;
        uint32_t *cur = prev + ((size_t)16 << lh);
        HIPCHK(launch_merkle_level(stream, prev, inject[lh], lh, cur));
        prev = cur;
    }
    MerkleTopInject inj;
    for (uint32_t l = 0; l < lh; l++) inj.digests[l] = inject[l];
    HIPCHK(launch_merkle_top(stream, prev, lh, inj));
    return true;
}

bool Engine::commit_tree_levels(uint32_t *d_digests, uint32_t log_h) {
    uint32_t *prev = d_digests;
    uint32_t lh = log_h;
    while (lh > MERKLE_TOP_LOG) {
        lh--;
        uint32_t *cur = prev + ((size_t)16 << lh);
        HIPCHK(launch_merkle_level(stream, prev, nullptr, lh, cur));
        prev = cur;
    }
    HIPCHK(launch_merkle_top(stream, prev, lh, MerkleTopInject()));
    return true;
}

static size_t tree_words(uint32_t log_h) { return (((size_t)2 << log_h) - 1) * 8; }
static const uint32_t *tree_root(const uint32_t *d_digests, uint32_t log_h) { return d_digests + tree_words(log_h) - 8; }

static Digest digest_from_words(const uint32_t w[8]) {
    Digest d;
    for (int i = 0; i < 8; i++) d.d[i] = Fp::raw(w[i]);
    return d;
}

// ------------------------------------------------------------------ setup
bool Engine::setup(const MachineDesc *m, const std::vector<ChipRef> &prep_chips,
                   const std::vector<std::vector<uint32_t>> &host_prep, ProvingKey *pk) {
    HIPCHK(hipSetDevice(device));
    pk->vk.machine = m;
    pk->vk.prep_chips = prep_chips;
    for (int i = 0; i < 8; i++) pk->vk.prep_root.d[i] = Fp::zero();
    if (prep_chips.empty()) return true;
    if (prep_chips.size() != host_prep.size()) return fail("setup: %zu preprocessed chips but %zu traces", prep_chips.size(), host_prep.size());
    std::vector<DevMat> mats;
    uint32_t mx = 0;
    size_t scratch_words = 0;
    for (size_t i = 0; i < prep_chips.size(); i++) {
        const ChipDesc &d = m->chips[prep_chips[i].chip_id];
        size_t n = (size_t)1 << prep_chips[i].log_n;
        if (d.prep_w == 0 || host_prep[i].size() != n * d.prep_w) return fail("setup: bad preprocessed trace for chip %s", d.name);
        scratch_words = std::max(scratch_words, n * d.prep_w);
    }
    uint32_t *d_scratch = nullptr;
    HIPCHK(hipMalloc(&d_scratch, scratch_words * 4));
    for (size_t i = 0; i < prep_chips.size(); i++) {
        const ChipDesc &d = m->chips[prep_chips[i].chip_id];
        size_t n = (size_t)1 << prep_chips[i].log_n, words = n * d.prep_w;
        ProvingKey::Prep pc{prep_chips[i].chip_id, prep_chips[i].log_n, nullptr, nullptr};
        HIPCHK(hipMalloc(&pc.d_trace, words * 4));
        HIPCHK(hipMalloc(&pc.d_lde, words * 8));
        HIPCHK(hipMemcpyAsync(pc.d_trace, host_prep[i].data(), words * 4, hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));  // host_prep[i] is pageable
        HIPCHK(launch_to_internal(stream, pc.d_trace, words));
        HIPCHK(launch_coset_lde(stream, tabs, pc.d_trace, d_scratch, pc.d_lde, d.prep_w, pc.log_n, 0));
        pk->prep.push_back(pc);
        mats.push_back({pc.d_lde, (uint32_t)d.prep_w, pc.log_n + 1});
        mx = std::max(mx, pc.log_n + 1);
    }
    pk->prep_log_h = mx;
    HIPCHK(hipMalloc(&pk->d_prep_digests, tree_words(mx) * 4));
    if (!commit_tree(mats, pk->d_prep_digests)) return false;
    uint32_t root[8];
    if (!download(root, tree_root(pk->d_prep_digests, mx), 32)) return false;
    pk->vk.prep_root = digest_from_words(root);
    HIPCHK(hipFree(d_scratch));
    return true;
}

void Engine::free_key(ProvingKey *pk) {
    (void)hipSetDevice(device);
    for (auto &p : pk->prep) {
        if (p.d_trace) (void)hipFree(p.d_trace);
        if (p.d_lde) (void)hipFree(p.d_lde);
    }
    pk->prep.clear();
    if (pk->d_prep_digests) (void)hipFree(pk->d_prep_digests);
    pk->d_prep_digests = nullptr;
}

// ------------------------------------------------------------------ prove
namespace {
struct ChipState {
    const ChipDesc *d;
    int id;
    uint32_t log_n;
    size_t n;
    const uint32_t *main = nullptr, *prep = nullptr, *prep_lde = nullptr;
    uint32_t *main_lde = nullptr, *perm = nullptr, *perm_lde = nullptr, *quot = nullptr, *quot_lde = nullptr;
    Fp4 cumsum = Fp4::zero();
};
std::vector<Fp4> ext_powers(Fp4 base, size_t n, bool from_one) {
    std::vector<Fp4> v(n);
    Fp4 x = from_one ? Fp4::one() : base;
    for (size_t i = 0; i < n; i++) { v[i] = x; x = x * base; }
    return v;
}
struct EventTimer {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    bool on;
    EventTimer(hipStream_t s, bool enable) : st(s), on(enable) {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
    }
    float stop() {
        if (!on) return 0;
        float ms = 0;
        (void)hipEventRecord(b, st);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
        (void)hipEventRecord(a, st);
        return ms;
    }
    ~EventTimer() { if (on) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } }
};
}  // namespace

bool Engine::commit_main_root(const ProvingKey &pk, const std::vector<ChipTrace> &traces, Digest *root, MainCache *keep) {
    times = StageTimes();
    EventTimer t_all(stream, profile);
    HIPCHK(hipSetDevice(device));
    const MachineDesc *m = pk.vk.machine;
    arena.reset();
    static const bool time_stages = getenv("DVT_TIME_PREPARE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (time_stages) fprintf(stderr, "[commit_main_root] %s at %.2f ms (pool misses so far %zu)\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), pool.misses);
    };
    size_t need = 0, max_mat_words = 0;
    uint32_t max_log_n = 0;
    for (auto &t : traces) {
        if (t.chip_id < 0 || t.chip_id >= m->n_chips || t.log_n > 22 || !t.d_main) return fail("commit: bad chip trace");
        size_t words = (size_t)m->chips[t.chip_id].main_w << t.log_n;
        need += words * 8 + 4096;
        max_mat_words = std::max(max_mat_words, words);
        max_log_n = std::max(max_log_n, t.log_n);
    }
    need += max_mat_words * 4 + (((size_t)4 << max_log_n)) * 32 + (1u << 20);
    if (arena.cap < need && arena.reserve(need + need / 8) != hipSuccess) { (void)hipGetLastError(); pool.trim(); HIPCHK(arena.reserve(need + need / 8)); }
    uint32_t *d_scratch = arena.alloc<uint32_t>(max_mat_words);
    if (!d_scratch) return fail("commit: device arena exhausted");
    lap("arena ready");
    std::vector<DevMat> mats;
    bool reuse = false;
    if (keep) {
        std::vector<size_t> want;
        for (auto &t : traces) want.push_back(((size_t)m->chips[t.chip_id].main_w << t.log_n) * 2);
        reuse = keep->fits(want, max_log_n + 1);
        if (!reuse) { keep->release(); keep->lde_words = want; keep->pool = &pool; }
        keep->valid = false;
    }
    size_t ti = 0;
    lap("cache decided");
    for (auto &t : traces) {
        const ChipDesc &d = m->chips[t.chip_id];
        uint32_t *lde = nullptr;
        if (keep && reuse) {
            lde = keep->lde[ti];
        } else if (keep) {
            HIPCHK(pool.alloc(&lde, ((size_t)d.main_w << t.log_n) * 8));
            keep->lde.push_back(lde);
        } else {
            lde = arena.alloc<uint32_t>(((size_t)d.main_w << t.log_n) * 2);
        }
        if (!lde) return fail("commit: device arena exhausted");
        if (time_stages && ti == 0) lap("first buffer");
        {
            EventTimer t_lde(stream, profile);
            HIPCHK(launch_coset_lde(stream, tabs, const_cast<uint32_t *>(t.d_main), d_scratch, lde, d.main_w, t.log_n, 0));
            if (profile) {
                times.lde_ms += t_lde.stop();
                times.lde_alg_bytes += 12.0 * d.main_w * (double)((size_t)1 << t.log_n);
                times.lde_calls++;
            }
        }
        mats.push_back({lde, (uint32_t)d.main_w, t.log_n + 1});
        ti++;
        if (time_stages) lap(d.name);
    }
    lap("LDEs launched");
    const uint32_t hmax = max_log_n + 1;
    uint32_t *tree = nullptr;
    if (keep && reuse) {
        tree = keep->tree;
    } else if (keep) {
        HIPCHK(pool.alloc(&tree, (((size_t)2 << hmax) - 1) * 32));
        keep->tree = tree;
        keep->log_h = hmax;
    } else {
        tree = arena.alloc<uint32_t>((((size_t)2 << hmax) - 1) * 8);
    }
    if (!tree) return fail("commit: device arena exhausted");
    {
        EventTimer t_mk(stream, profile);
        if (!commit_tree(mats, tree)) return false;
        if (profile) {
            times.merkle_ms += t_mk.stop();
            for (uint32_t lh = 0; lh <= hmax; lh++) {
                uint32_t w = 0;
                for (auto &mm : mats) if (mm.log_h == lh) w += mm.width;
                double rows = (double)((size_t)1 << lh);
                times.merkle_perms += lh == hmax ? rows * ((w + 7) / 8) : rows * (1 + (w ? 1 + (w + 7) / 8 : 0));
            }
        }
    }
    lap("tree launched");
    uint32_t rootw[8];
    if (!download(rootw, tree + ((((size_t)2 << hmax) - 1) * 8) - 8, 32)) return false;
    lap("root downloaded");
    for (int i = 0; i < 8; i++) root->d[i] = Fp::raw(rootw[i]);
    if (keep) { keep->root = *root; keep->valid = true; }
    if (profile) times.commit_main = t_all.stop();
    return true;
}

bool Engine::prove_shard(const ProvingKey &pk, const std::vector<ChipTrace> &traces, const std::vector<Fp> &pubs,
                         const StarkConfig &cfg, ShardProof *out, const PermChallenges *global, const MainCache *cached) {
    HIPCHK(hipSetDevice(device));
    const MachineDesc *m = pk.vk.machine;
    arena.reset();
    const StageTimes phase1 = times;
    times = StageTimes();
    if (cached && cached->valid) {  // carry the phase-1 measurements of this shard into its totals
        times.lde_ms = phase1.lde_ms; times.lde_alg_bytes = phase1.lde_alg_bytes; times.lde_calls = phase1.lde_calls;
        times.merkle_ms = phase1.merkle_ms; times.merkle_perms = phase1.merkle_perms;
    }
    for (auto &t : traces) {
        const ChipDesc &d = m->chips[t.chip_id];
        const double n = (double)((size_t)1 << t.log_n);
        times.cells_m += d.main_w * n; times.cells_p += 4.0 * d.perm_ext_w * n; times.cells_q += 8.0 * n; times.cells_pre += d.prep_w * n;
    }
    EventTimer tm(stream, profile), tm_total(stream, profile);
    // per-family timing (profile mode only: the extra event synchronisations serialise the stream)
    auto lde = [&](uint32_t *in, uint32_t *scratch, uint32_t *outp, uint32_t width, uint32_t log_n, uint32_t mode) -> hipError_t {
        if (!profile) return launch_coset_lde(stream, tabs, in, scratch, outp, width, log_n, mode);
        EventTimer t(stream, true);
        hipError_t e = launch_coset_lde(stream, tabs, in, scratch, outp, width, log_n, mode);
        times.lde_ms += t.stop();
        times.lde_alg_bytes += 12.0 * width * (double)((size_t)1 << log_n);
        times.lde_calls++;
        return e;
    };
    auto commit = [&](const std::vector<DevMat> &ms, uint32_t *d_digests) -> bool {
        if (!profile) return commit_tree(ms, d_digests);
        EventTimer t(stream, true);
        bool ok = commit_tree(ms, d_digests);
        times.merkle_ms += t.stop();
        uint32_t mxh = 0;
        for (auto &mm : ms) mxh = std::max(mxh, mm.log_h);
        for (uint32_t lh = 0; lh <= mxh; lh++) {
            uint32_t w = 0;
            for (auto &mm : ms) if (mm.log_h == lh) w += mm.width;
            double rows = (double)((size_t)1 << lh);
            if (lh == mxh) times.merkle_perms += rows * ((w + 7) / 8);
            else times.merkle_perms += rows * (1 + (w ? 1 + (w + 7) / 8 : 0));
        }
        return ok;
    };

    // ---- validate inputs and lay out per-chip state
    std::vector<ChipState> cs;
    std::vector<ChipRef> refs;
    size_t max_mat_words = 0;
    uint32_t max_log_n = 0;
    size_t need = 0, max_open_cols = 8;
    for (auto &t : traces) {
        if (t.chip_id < 0 || t.chip_id >= m->n_chips) return fail("prove: chip id %d out of range", t.chip_id);
        if (!cs.empty() && t.chip_id <= cs.back().id) return fail("prove: chip traces must be sorted by chip id");
        if (t.log_n > 22) return fail("prove: log_n %u > 22", t.log_n);
        ChipState s;
        s.d = &m->chips[t.chip_id];
        s.id = t.chip_id;
        s.log_n = t.log_n;
        s.n = (size_t)1 << t.log_n;
        s.main = t.d_main;
        if (s.d->main_w == 0 || !s.main) return fail("prove: chip %s has no main trace", s.d->name);
        cs.push_back(s);
        refs.push_back({t.chip_id, t.log_n});
        max_log_n = std::max(max_log_n, t.log_n);
        size_t w = std::max<size_t>({(size_t)s.d->main_w, (size_t)4 * s.d->perm_ext_w, 4});
        max_mat_words = std::max(max_mat_words, w * s.n);
        need += (size_t)(s.d->main_w * 2 + 4 * s.d->perm_ext_w * 3 + 4 + 8 * 3) * s.n * 4 + 16 * 256;
        max_open_cols = std::max<size_t>(max_open_cols, (size_t)s.d->main_w + 4 * s.d->perm_ext_w + s.d->prep_w);
    }
    if (cs.empty()) return fail("prove: no chips");
    for (auto &pc : pk.prep) {
        bool found = false;
        for (auto &s : cs)
            if (s.id == pc.chip_id) {
                if (s.log_n != pc.log_n) return fail("prove: chip %s height differs from its preprocessed trace", s.d->name);
                s.prep = pc.d_trace;
                s.prep_lde = pc.d_lde;
                found = true;
            }
        if (!found) return fail("prove: preprocessed chip %d missing from the shard", pc.chip_id);
    }
    for (auto &s : cs)
        if (s.d->prep_w && !s.prep) return fail("prove: chip %s needs a preprocessed trace (setup)", s.d->name);
    const uint32_t hmax = max_log_n + 1;
    // arena: traces/LDEs + trees + FRI vectors + opening scratch (generous bound)
    need += max_mat_words * 4;
    need += 3 * tree_words(hmax) * 4;
    need += ((size_t)1 << hmax) * 16 * 4 + 2 * tree_words(hmax) * 4;
    need += ((size_t)1 << max_log_n) * 16 * 3 + (64u << 20);
    need += ((size_t)PARTS_MAX * 32) << std::min(max_log_n, PARTS_PARALLEL_LOG);
    need += (size_t)OPEN_MAX_ROW_BLOCKS * max_open_cols * 2 * sizeof(Fp4);   // K6 partial sums
    if (arena.cap < need && arena.reserve(need + need / 8) != hipSuccess) { (void)hipGetLastError(); pool.trim(); HIPCHK(arena.reserve(need + need / 8)); }
#define ALLOC(var, T, count)                                                        \
    do {                                                                            \
        var = arena.alloc<T>(count);                                                \
        if (!var) return fail("prove: device arena exhausted (%s)", #var);          \
    } while (0)

    uint32_t *d_scratch;
    ALLOC(d_scratch, uint32_t, max_mat_words);
    const uint32_t *d_pub = upload_vec(std::vector<uint32_t>([&] {
        std::vector<uint32_t> w(std::max<size_t>(pubs.size(), 1), 0);
        for (size_t i = 0; i < pubs.size(); i++) w[i] = pubs[i].v;
        return w;
    }()));
    if (!d_pub) return false;

    ShardProof &pf = *out;
    pf = ShardProof();
    pf.public_values = pubs;
    Challenger ch;
    transcript_begin(ch, pk.vk, refs);

    // ---- 1. main trace: LDE + commit  (K1, K2, K3)
    std::vector<DevMat> mats;
    uint32_t *d_main_tree;
    uint32_t rootw[8];
    if (cached && cached->valid) {  // phase 1 already did K1-K3 of the main traces and kept the results
        if (cached->lde.size() != cs.size() || cached->log_h != hmax) return fail("prove: main-trace cache does not match the shard");
        for (size_t k = 0; k < cs.size(); k++) cs[k].main_lde = cached->lde[k];
        d_main_tree = cached->tree;
        pf.main_root = cached->root;
    } else {
        for (auto &s : cs) {
            ALLOC(s.main_lde, uint32_t, (size_t)s.d->main_w * 2 * s.n);
            HIPCHK(lde(const_cast<uint32_t *>(s.main), d_scratch, s.main_lde, s.d->main_w, s.log_n, 0));
            mats.push_back({s.main_lde, (uint32_t)s.d->main_w, s.log_n + 1});
        }
        ALLOC(d_main_tree, uint32_t, tree_words(hmax));
        if (!commit(mats, d_main_tree)) return false;
        if (!download(rootw, tree_root(d_main_tree, hmax), 32)) return false;
        pf.main_root = digest_from_words(rootw);
    }
    ch.observe(pf.main_root);
    ch.observe_u32((uint32_t)pubs.size());
    for (auto x : pubs) ch.observe(x);
    times.commit_main = tm.stop() + (cached && cached->valid ? phase1.commit_main : 0.f);

    // ---- 2. permutation trace (K4) + LDE + commit
    Fp4 perm_alpha, beta;
    if (global) {  // common to all shards of the execution; bound into this shard's transcript
        perm_alpha = global->alpha;
        beta = global->beta;
        ch.observe(perm_alpha);
        ch.observe(beta);
    } else {
        perm_alpha = ch.sample_ext();
        beta = ch.sample_ext();
    }
    int max_arity = 1, max_folded = 1;
    for (int i = 0; i < m->n_chips; i++) {
        max_arity = std::max(max_arity, m->chips[i].max_arity);
        max_folded = std::max(max_folded, m->chips[i].n_folded);
    }
    std::vector<Fp4> beta_pows = ext_powers(beta, max_arity, false);
    const Fp4 *d_beta = upload_vec(beta_pows);
    if (!d_beta) return false;
    // the same powers as centred doubles: operands of the exact FP64 dot products of K4 / K5 (f64dot.cuh)
    auto upload_centred = [&](const std::vector<Fp4> &v) -> const double * {
        std::vector<double> d(4 * v.size());
        for (size_t i = 0; i < v.size(); i++)
            for (int k = 0; k < 4; k++) d[4 * i + k] = centred_canonical(v[i].c[k]);
        return reinterpret_cast<const double *>(upload(d.data(), d.size() * sizeof(double)));
    };
    const double *d_beta_f64 = upload_centred(beta_pows);
    if (!d_beta_f64) return false;
    mats.clear();
    uint32_t perm_hmax = 0;
    size_t n_cumsum = 0;
    // scratch of the part-parallel K4 / K5 launches of short tables (stark.cuh): [PARTS_MAX][8][rows]
    uint32_t *d_parts;
    ALLOC(d_parts, uint32_t, (size_t)PARTS_MAX * 8 << std::min(max_log_n, PARTS_PARALLEL_LOG));
    for (auto &s : cs) {
        if (!s.d->perm_ext_w) continue;
        const size_t bw = 4 * (size_t)s.d->perm_ext_w;
        ALLOC(s.perm, uint32_t, bw * s.n);
        ALLOC(s.perm_lde, uint32_t, bw * 2 * s.n);
        uint32_t *totals, *scan_scratch;
        ALLOC(totals, uint32_t, 4 * s.n);
        PermArgs pa{s.main, s.prep, d_pub, s.perm, totals, d_beta, d_beta_f64, perm_alpha, s.log_n};
        if (s.log_n <= PARTS_PARALLEL_LOG) pa.partial = d_parts;   // (tall tables: measured, no gain - 68.14 against 68.09 M cycles/s)
        HIPCHK(s.d->launch_perm(stream, pa));
        ALLOC(scan_scratch, uint32_t, prefix_sum_scratch_words(4, s.n));
        HIPCHK(launch_prefix_sum_columns(stream, totals, 4, s.n, scan_scratch));
        // the cumulative sum is only needed for the transcript, after the permutation tree: the kernel writes its four
        // words into the tail of the pinned staging buffer, read after the next synchronisation
        uint32_t *cw = reinterpret_cast<uint32_t *>(h_down + DOWN_BYTES - 4096) + 4 * n_cumsum++;
        HIPCHK(launch_phi_from_prefix_sums(stream, totals, s.perm + (bw - 4) * s.n, s.log_n, cw));
        HIPCHK(lde(s.perm, d_scratch, s.perm_lde, (uint32_t)bw, s.log_n, 0));
        mats.push_back({s.perm_lde, (uint32_t)bw, s.log_n + 1});
        perm_hmax = std::max(perm_hmax, s.log_n + 1);
    }
    uint32_t *d_perm_tree = nullptr;
    for (int i = 0; i < 8; i++) pf.perm_root.d[i] = Fp::zero();
    if (!mats.empty()) {
        ALLOC(d_perm_tree, uint32_t, tree_words(perm_hmax));
        if (!commit(mats, d_perm_tree)) return false;
        if (!download(rootw, tree_root(d_perm_tree, perm_hmax), 32)) return false;
        pf.perm_root = digest_from_words(rootw);
    }
    HIPCHK(hipStreamSynchronize(stream));
    {
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(h_down + DOWN_BYTES - 4096);
        size_t at = 0;
        for (auto &s : cs) {
            if (!s.d->perm_ext_w) continue;
            for (int k = 0; k < 4; k++) s.cumsum.c[k] = Fp::raw(cw[4 * at + k]);
            at++;
        }
    }
    ch.observe(pf.perm_root);
    for (auto &s : cs) ch.observe(s.cumsum);
    times.perm = tm.stop();

    // ---- 3. quotient (K5) + chunk LDE + commit
    Fp4 alpha = ch.sample_ext();
    std::vector<Fp4> alpha_pows = ext_powers(alpha, max_folded, true);
    const Fp4 *d_alpha = upload_vec(alpha_pows);
    const double *d_alpha_f64 = upload_centred(alpha_pows);
    if (!d_alpha || !d_alpha_f64) return false;
    mats.clear();
    const Fp g = Fp::from_canonical(COSET_SHIFT);
    for (auto &s : cs) {
        ALLOC(s.quot, uint32_t, 8 * s.n);
        ALLOC(s.quot_lde, uint32_t, 16 * s.n);
        QuotientArgs qa;
        qa.main_lde = s.main_lde; qa.prep_lde = s.prep_lde; qa.perm_lde = s.perm_lde; qa.pub = d_pub; qa.out = s.quot;
        qa.alpha_pows = d_alpha; qa.beta_pows = d_beta; qa.alpha_d = d_alpha_f64; qa.beta_d = d_beta_f64;
        qa.perm_alpha = perm_alpha; qa.cum_over_n = s.cumsum * inv(Fp::from_canonical((uint32_t)s.n));
        Fp gn = pow(g, s.n);
        qa.z_even = gn - Fp::one();
        qa.z_odd = -gn - Fp::one();
        qa.zinv_even = inv(qa.z_even);
        qa.zinv_odd = inv(qa.z_odd);
        qa.w_inv = inv(two_adic_generator(s.log_n));
        qa.log_n = s.log_n;
        qa.tabs = tabs;
        qa.sel = selector_table(qa);
        if (s.log_n <= PARTS_PARALLEL_LOG) qa.partial = d_parts;
        HIPCHK(s.d->launch_quotient(stream, qa));
        HIPCHK(lde(s.quot, d_scratch, s.quot_lde, 4, s.log_n, 1));
        HIPCHK(lde(s.quot + 4 * s.n, d_scratch, s.quot_lde + 8 * s.n, 4, s.log_n, 2));
        mats.push_back({s.quot_lde, 8, s.log_n + 1});
    }
    uint32_t *d_quot_tree;
    ALLOC(d_quot_tree, uint32_t, tree_words(hmax));
    if (!commit(mats, d_quot_tree)) return false;
    if (!download(rootw, tree_root(d_quot_tree, hmax), 32)) return false;
    pf.quot_root = digest_from_words(rootw);
    ch.observe(pf.quot_root);
    times.quotient = tm.stop();

    // ---- 4. openings at zeta and zeta*omega (K6)
    Fp4 zeta = ch.sample_ext();
    {
        Fp4 *d_w;
        ALLOC(d_w, Fp4, (size_t)1 << max_log_n);
        size_t max_cols = 8;
        for (auto &s : cs) max_cols = std::max<size_t>(max_cols, (size_t)s.d->main_w + 4 * s.d->perm_ext_w + s.d->prep_w);
        // every matrix writes its opened values into one result buffer; ONE download after the last launch
        // (a host synchronisation per matrix costs more than the small chips' kernels)
        size_t total_vals = 0;
        for (auto &s : cs) total_vals += 2 * ((size_t)s.d->prep_w + s.d->main_w + 4 * s.d->perm_ext_w + 8);
        Fp4 *d_partial, *d_res;
        ALLOC(d_partial, Fp4, (size_t)OPEN_MAX_ROW_BLOCKS * max_cols * 2);
        ALLOC(d_res, Fp4, total_vals);
        struct Pending { size_t off; uint32_t width; Fp4 scale; std::vector<Fp4> *local, *next; };
        std::vector<Pending> pending;
        size_t res_off = 0;
        // the column pointers of every matrix in ONE upload (a small host-to-device copy between two kernels costs a pipeline
        // bubble of tens of microseconds, and there are ~40 matrices in a shard with precompile chips)
        std::vector<uint64_t> all_ptrs;
        {
            auto add = [&](const uint32_t *base, uint32_t width, size_t n) {
                for (uint32_t c = 0; c < width; c++) all_ptrs.push_back((uint64_t)(uintptr_t)(base + (size_t)c * n));
            };
            for (auto &s : cs) {
                if (s.d->prep_w) add(s.prep, s.d->prep_w, s.n);
                add(s.main, s.d->main_w, s.n);
                if (s.d->perm_ext_w) add(s.perm, 4 * s.d->perm_ext_w, s.n);
                for (int c = 0; c < 2; c++) add(s.quot + (size_t)4 * c * s.n, 4, s.n);
            }
        }
        auto d_all_cols = reinterpret_cast<const uint32_t *const *>(upload(all_ptrs.data(), all_ptrs.size() * 8));
        if (!d_all_cols) return false;
        size_t cols_at = 0;
        // matrices that share their weights (the prep / main / permutation traces of a chip) are opened by ONE launch over the
        // concatenation of their columns: note() registers a matrix, open_noted() launches what has been noted
        size_t group_cols = 0, group_res = 0;
        uint32_t group_w = 0;
        auto note = [&](const uint32_t *base, uint32_t width, Fp4 scale, std::vector<Fp4> *local, std::vector<Fp4> *next) -> bool {
            if (cols_at + width > all_ptrs.size() || all_ptrs[cols_at] != (uint64_t)(uintptr_t)base) return fail("prove: internal error, opening order");
            if (!group_w) { group_cols = cols_at; group_res = res_off; }
            pending.push_back({res_off, width, scale, local, next});
            cols_at += width;
            res_off += (size_t)width * 2;
            group_w += width;
            return true;
        };
        auto open_noted = [&](const ChipState &s) -> bool {
            HIPCHK(launch_open_columns(stream, d_all_cols + group_cols, group_w, s.log_n, d_w, d_partial, d_res + group_res));
            group_w = 0;
            return true;
        };
        pf.chips.resize(cs.size());
        std::vector<std::vector<Fp4>> quot_vals(2 * cs.size());
        for (size_t k = 0; k < cs.size(); k++) {
            ChipState &s = cs[k];
            ChipOpening &o = pf.chips[k];
            o.chip_id = (uint32_t)s.id;
            o.log_n = s.log_n;
            o.cumsum = s.cumsum;
            Fp ninv = inv(Fp::from_canonical((uint32_t)(s.n % P)));
            // trace matrices: point zeta over H
            HIPCHK(launch_open_weights(stream, tabs, zeta, s.log_n, d_w));
            Fp4 scale = (pow(zeta, s.n) - Fp::one()) * ninv;
            if (s.d->prep_w && !note(s.prep, s.d->prep_w, scale, &o.prep_l, &o.prep_n)) return false;
            if (!note(s.main, s.d->main_w, scale, &o.main_l, &o.main_n)) return false;
            if (s.d->perm_ext_w && !note(s.perm, 4 * s.d->perm_ext_w, scale, &o.perm_l, &o.perm_n)) return false;
            if (!open_noted(s)) return false;
            // quotient chunks: values on s_c * H, opened at zeta  <=>  f(y) = r(s_c y) at y = zeta / s_c
            o.quot.resize(8);
            for (int c = 0; c < 2; c++) {
                Fp sc = c == 0 ? g : g * two_adic_generator(s.log_n + 1);
                Fp4 y = zeta * inv(sc);
                HIPCHK(launch_open_weights(stream, tabs, y, s.log_n, d_w));
                Fp4 qs = (pow(y, s.n) - Fp::one()) * ninv;
                if (!note(s.quot + (size_t)4 * c * s.n, 4, qs, &quot_vals[2 * k + c], nullptr) || !open_noted(s)) return false;
            }
        }
        std::vector<Fp4> host_res(res_off);
        if (!download(host_res.data(), d_res, res_off * sizeof(Fp4))) return false;
        for (auto &pd : pending) {
            pd.local->resize(pd.width);
            if (pd.next) pd.next->resize(pd.width);
            for (uint32_t c = 0; c < pd.width; c++) {
                (*pd.local)[c] = host_res[pd.off + 2 * c] * pd.scale;
                if (pd.next) (*pd.next)[c] = host_res[pd.off + 2 * c + 1] * pd.scale;
            }
        }
        for (size_t k = 0; k < cs.size(); k++)
            for (int c = 0; c < 2; c++)
                for (int j = 0; j < 4; j++) pf.chips[k].quot[4 * c + j] = quot_vals[2 * k + c][j];
        {
            // one sponge digest per opened vector (challenger.h observe_values): independent chains, hashed by a few host threads
            // when there is enough of them (a shard with precompile chips: ~2 500 permutations, 4.4 ms on one thread)
            std::vector<const std::vector<Fp4> *> vecs;
            size_t total = 0;
            for (auto &o : pf.chips)
                for (auto *v : {&o.prep_l, &o.prep_n, &o.main_l, &o.main_n, &o.perm_l, &o.perm_n, &o.quot})
                    if (!v->empty()) { vecs.push_back(v); total += v->size(); }
            std::vector<Digest> digs(vecs.size());
            const unsigned n_threads = total < 2048 ? 1u : std::min<unsigned>(8u, (unsigned)vecs.size());
            if (n_threads <= 1) {
                for (size_t i = 0; i < vecs.size(); i++) digs[i] = Challenger::hash_values(*vecs[i]);
            } else {
                std::vector<size_t> order(vecs.size());
                for (size_t i = 0; i < order.size(); i++) order[i] = i;
                std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return vecs[a]->size() > vecs[b]->size(); });   // longest chains first
                std::atomic<size_t> next{0};
                auto work = [&] {
                    for (size_t k; (k = next.fetch_add(1)) < order.size();) digs[order[k]] = Challenger::hash_values(*vecs[order[k]]);
                };
                std::vector<std::thread> pool;
                for (unsigned t = 1; t < n_threads; t++) pool.emplace_back(work);
                work();
                for (auto &t : pool) t.join();
            }
            for (auto &d : digs) ch.observe(d);
        }
    }
    times.open = tm.stop();

    // ---- 5. FRI input (K7)
    Fp4 alpha_fri = ch.sample_ext();
    std::vector<Fp4 *> ro(hmax + 1, nullptr);
    // matrices per tree in tree order, for pointer tables
    struct TreeMat { const uint32_t *lde; uint32_t width, log_h; };
    std::vector<TreeMat> tree_mats[4];
    for (auto &s : cs) {
        if (s.d->prep_w) tree_mats[0].push_back({s.prep_lde, (uint32_t)s.d->prep_w, s.log_n + 1});
        tree_mats[1].push_back({s.main_lde, (uint32_t)s.d->main_w, s.log_n + 1});
        if (s.d->perm_ext_w) tree_mats[2].push_back({s.perm_lde, (uint32_t)(4 * s.d->perm_ext_w), s.log_n + 1});
        tree_mats[3].push_back({s.quot_lde, 8, s.log_n + 1});
    }
    {
        size_t max_cols_h = 1;
        std::vector<std::vector<ColRef>> cols_by_h(hmax + 1);
        std::vector<uint32_t> n_two_by_h(hmax + 1, 0);
        for (uint32_t h = 1; h <= hmax; h++) {
            cols_by_h[h] = fri_columns(m, refs, h, &n_two_by_h[h]);
            max_cols_h = std::max(max_cols_h, cols_by_h[h].size());
        }
        std::vector<Fp4> apow = ext_powers(alpha_fri, max_cols_h + 1, true);
        std::vector<double> apow_d(4 * apow.size());
        for (size_t c = 0; c < apow.size(); c++)
            for (int k = 0; k < 4; k++) apow_d[4 * c + k] = centred_canonical(apow[c].c[k]);
        const double *d_apow = reinterpret_cast<const double *>(upload(apow_d.data(), apow_d.size() * sizeof(double)));
        if (!d_apow) return false;
        // pointer tables of every height in one upload (see the openings above), then the launches
        std::vector<uint64_t> all_ptrs;
        std::vector<size_t> ptr_at(hmax + 1, 0);
        std::vector<Fp4> sz_all_h(hmax + 1, Fp4::zero()), sz_two_h(hmax + 1, Fp4::zero());
        for (uint32_t h = 1; h <= hmax; h++) {
            auto &cols = cols_by_h[h];
            ptr_at[h] = all_ptrs.size();
            Fp4 sz_all = Fp4::zero(), sz_two = Fp4::zero();
            for (size_t c = 0; c < cols.size(); c++) {
                const ColRef &r = cols[c];
                const TreeMat &tmx = tree_mats[r.tree][r.mat];
                all_ptrs.push_back((uint64_t)(uintptr_t)(tmx.lde + ((size_t)r.col << h)));
                const ChipOpening &o = pf.chips[r.chip_pos];
                const std::vector<Fp4> &loc = r.tree == 0 ? o.prep_l : r.tree == 1 ? o.main_l : r.tree == 2 ? o.perm_l : o.quot;
                sz_all += apow[c] * loc[r.col];
                if (r.tree < 3) {
                    const std::vector<Fp4> &nx = r.tree == 0 ? o.prep_n : r.tree == 1 ? o.main_n : o.perm_n;
                    sz_two += apow[c] * nx[r.col];
                }
            }
            sz_all_h[h] = sz_all; sz_two_h[h] = sz_two;
        }
        auto d_all_cols = reinterpret_cast<const uint32_t *const *>(upload(all_ptrs.data(), all_ptrs.size() * 8));
        if (!d_all_cols) return false;
        for (uint32_t h = 1; h <= hmax; h++) {
            auto &cols = cols_by_h[h];
            if (cols.empty()) continue;
            ALLOC(ro[h], Fp4, (size_t)1 << h);
            Fp4 zeta_next = zeta * two_adic_generator(h - 1);
            HIPCHK(launch_reduced_opening(stream, tabs, d_all_cols + ptr_at[h], n_two_by_h[h], (uint32_t)cols.size(), h, d_apow, sz_all_h[h], sz_two_h[h],
                                          zeta, zeta_next, apow[cols.size()], ro[h]));
        }
    }

    // ---- 6. FRI commit phase (K8)
    struct Layer { Fp4 *v; uint32_t *tree; uint32_t log_m; };
    std::vector<Layer> layers;
    Fp4 *cur = ro[hmax];
    // The transcript steps of the commit phase run on the device (fri_challenge_kernel): observe(root), sample beta.
    // The host transcript is in the state those steps assume (alpha_fri was just sampled: no pending input); it gets the
    // final sponge state and all the roots back with ONE download after the last round instead of one per round.
    if (!ch.input.empty()) return fail("prove: internal error, transcript has pending input before FRI");
    const uint32_t n_rounds = hmax > 1 ? hmax - 1 : 0;
    uint32_t *d_fs;   // [16] sponge state, then n_rounds x [8] roots
    Fp4 *d_betas;
    ALLOC(d_fs, uint32_t, 16 + 8 * (size_t)n_rounds + 8);
    ALLOC(d_betas, Fp4, (size_t)n_rounds + 1);
    {
        uint32_t st16[16];
        for (int k = 0; k < 16; k++) st16[k] = ch.state[k].v;
        const void *src = upload(st16, sizeof st16);
        if (!src) return false;
        HIPCHK(hipMemcpyAsync(d_fs, src, sizeof st16, hipMemcpyDeviceToDevice, stream));
    }
    uint32_t round = 0;
    for (uint32_t lm = hmax; lm > 1; lm--, round++) {
        Layer L{cur, nullptr, lm};
        ALLOC(L.tree, uint32_t, tree_words(lm - 1));
        HIPCHK(launch_fri_leaves(stream, cur, lm, L.tree));
        if (!commit_tree_levels(L.tree, lm - 1)) return false;
        HIPCHK(launch_fri_challenge(stream, tree_root(L.tree, lm - 1), d_fs, d_betas + round, d_fs + 16 + 8 * (size_t)round));
        Fp4 *nxt;
        ALLOC(nxt, Fp4, (size_t)1 << (lm - 1));
        HIPCHK(launch_fri_fold(stream, tabs, cur, nxt, ro[lm - 1], Fp4::zero(), lm, d_betas + round));
        layers.push_back(L);
        cur = nxt;
    }
    if (n_rounds) {
        std::vector<uint32_t> fs(16 + 8 * (size_t)n_rounds);
        if (!download(fs.data(), d_fs, fs.size() * 4)) return false;
        for (uint32_t r = 0; r < n_rounds; r++) pf.fri_roots.push_back(digest_from_words(&fs[16 + 8 * (size_t)r]));
        for (int k = 0; k < 16; k++) ch.state[k] = Fp::raw(fs[k]);
        ch.input.clear();
        ch.output.assign(ch.state, ch.state + 4);   // the squeezed rate minus the four elements beta popped from its back
    }
    {
        uint32_t fw[8];
        if (!download(fw, cur, 32)) return false;  // two values, both equal to the constant final polynomial
        for (int k = 0; k < 4; k++) pf.final_poly.c[k] = Fp::raw(fw[k]);
        for (int k = 0; k < 4; k++)
            if (fw[k] != fw[4 + k]) return fail("prove: FRI final polynomial is not constant (trace does not satisfy the AIR?)");
        ch.observe(pf.final_poly);
    }

    // ---- 7. proof of work + queries (K9)
    {
        uint32_t st16[16];
        for (int k = 0; k < 16; k++) st16[k] = ch.state[k].v;
        for (size_t k = 0; k < ch.input.size(); k++) st16[k] = ch.input[k].v;
        uint32_t pos = (uint32_t)ch.input.size();
        uint32_t *d_found;
        ALLOC(d_found, uint32_t, 1);
        uint32_t found = 0xffffffffu;
        // batches of 4 x the expected number of tries (the smallest witness is wanted, so batches go in order): the first
        // one succeeds with probability 1 - e^-4; a fixed 2^22-candidate batch cost 0.6 ms at 16 bits for nothing
        const uint32_t batch = cfg.pow_bits + 2 >= 22 ? 1u << 22 : (cfg.pow_bits + 2 < 12 ? 1u << 12 : 1u << (cfg.pow_bits + 2));
        for (uint32_t base = 0; found == 0xffffffffu && base < P - batch; base += batch) {
            HIPCHK(hipMemsetAsync(d_found, 0xff, 4, stream));
            HIPCHK(launch_pow_grind(stream, st16, pos, cfg.pow_bits, base, batch, d_found));
            if (!download(&found, d_found, 4)) return false;
        }
        if (found == 0xffffffffu) return fail("prove: no proof-of-work witness found");
        pf.pow_witness = Fp::from_canonical(found);
        if (!ch.check_witness(cfg.pow_bits, pf.pow_witness)) return fail("prove: internal error, grind witness rejected by transcript");
    }
    const uint32_t nq = cfg.num_queries;
    std::vector<uint32_t> idx(nq);
    for (auto &i : idx) i = ch.sample_bits(hmax);
    const uint32_t *d_idx = upload_vec(idx);
    if (!d_idx) return false;
    const uint32_t *trees_dev[4] = {pk.d_prep_digests, d_main_tree, d_perm_tree, d_quot_tree};
    const uint32_t trees_h[4] = {pk.prep_log_h, hmax, perm_hmax, hmax};
    // every gather writes into one device buffer; ONE download afterwards (50 small synchronous copies cost ~2 ms)
    struct Slot { size_t rows_at = 0, paths_at = 0; uint32_t ncols = 0; };
    Slot tree_slot[4];
    std::vector<Slot> layer_slot(layers.size());
    size_t q_words = 0;
    uint32_t tree_ncols[4] = {0, 0, 0, 0};
    for (int t = 0; t < 4; t++) {
        for (auto &tmx : tree_mats[t]) tree_ncols[t] += tmx.width;
        if (tree_mats[t].empty()) continue;
        tree_slot[t].ncols = tree_ncols[t];
        tree_slot[t].rows_at = q_words; q_words += (size_t)nq * tree_ncols[t];
        q_words = (q_words + 3) & ~(size_t)3;
        tree_slot[t].paths_at = q_words; q_words += (size_t)nq * trees_h[t] * 8 + 8;
    }
    for (size_t l = 0; l < layers.size(); l++) {
        q_words = (q_words + 3) & ~(size_t)3;  // 16-byte aligned: gather_siblings stores uint4
        layer_slot[l].rows_at = q_words; q_words += (size_t)nq * 4;
        layer_slot[l].paths_at = q_words; q_words += (size_t)nq * (layers[l].log_m - 1) * 8 + 8;
    }
    uint32_t *d_q;
    ALLOC(d_q, uint32_t, q_words);
    for (int t = 0; t < 4; t++) {
        if (tree_mats[t].empty()) continue;
        std::vector<uint64_t> ptrs;
        std::vector<uint32_t> lhs;
        for (auto &tmx : tree_mats[t])
            for (uint32_t c = 0; c < tmx.width; c++) {
                ptrs.push_back((uint64_t)(uintptr_t)(tmx.lde + ((size_t)c << tmx.log_h)));
                lhs.push_back(tmx.log_h);
            }
        auto d_cols = reinterpret_cast<const uint32_t *const *>(upload(ptrs.data(), ptrs.size() * 8));
        const uint32_t *d_lh = upload_vec(lhs);
        if (!d_cols || !d_lh) return false;
        HIPCHK(launch_gather_rows(stream, d_cols, d_lh, tree_ncols[t], d_idx, nq, d_q + tree_slot[t].rows_at));
        HIPCHK(launch_gather_paths(stream, trees_dev[t], trees_h[t], d_idx, nq, d_q + tree_slot[t].paths_at));
    }
    for (size_t l = 0; l < layers.size(); l++) {
        auto &L = layers[l];
        HIPCHK(launch_gather_siblings(stream, L.v, L.log_m, d_idx, nq, reinterpret_cast<Fp4 *>(d_q + layer_slot[l].rows_at)));
        HIPCHK(launch_gather_paths(stream, L.tree, L.log_m - 1, d_idx, nq, d_q + layer_slot[l].paths_at));
    }
    // canonical words on the device, one download, then the query section of the proof straight in wire format
    HIPCHK(launch_from_internal(stream, d_q, q_words));
    std::vector<uint32_t> hq(q_words);
    if (!download(hq.data(), d_q, q_words * 4)) return false;
    {
        std::vector<uint32_t> &w = pf.query_words;
        w.reserve(q_words + (size_t)nq * 64 + 16);
        w.push_back(nq);
        auto put = [&](const uint32_t *src, size_t n) { w.insert(w.end(), src, src + n); };
        for (uint32_t q = 0; q < nq; q++) {
            for (int t = 0; t < 4; t++) {
                if (tree_mats[t].empty()) { w.push_back(0); w.push_back(0); continue; }   // no rows, empty path
                w.push_back((uint32_t)tree_mats[t].size());
                const uint32_t *row = hq.data() + tree_slot[t].rows_at + (size_t)q * tree_ncols[t];
                for (auto &tmx : tree_mats[t]) {
                    w.push_back(tmx.width);
                    put(row, tmx.width);
                    row += tmx.width;
                }
                w.push_back(trees_h[t]);
                put(hq.data() + tree_slot[t].paths_at + (size_t)q * trees_h[t] * 8, (size_t)trees_h[t] * 8);
            }
            w.push_back((uint32_t)layers.size());
            for (size_t li = 0; li < layers.size(); li++) {
                const uint32_t th = layers[li].log_m - 1;
                put(hq.data() + layer_slot[li].rows_at + (size_t)q * 4, 4);
                w.push_back(th);
                put(hq.data() + layer_slot[li].paths_at + (size_t)q * th * 8, (size_t)th * 8);
            }
        }
    }
    times.fri = tm.stop();
    times.total = tm_total.stop() + (cached && cached->valid ? phase1.commit_main : 0.f);
    return true;
#undef ALLOC
}

}  // namespace dvt
