// Shard prover / verifier for a generic multi-chip machine.  The prover keeps
// every matrix in HBM (column-major, Montgomery form) for the whole shard and runs
// K1-K9 on one HIP stream; the host only drives the transcript.
//
// Stands behind reference src/main.rs:462-466 (`setup`, `prove(..).run()`); the
// protocol itself is an original restatement of the public multi-table STARK
// structure, see DESIGN.md "Protocol".
#pragma once
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "challenger.h"
#include "machine.h"
#include "proof.h"

namespace dvt {

struct StarkConfig {
    uint32_t num_queries = 100;
    uint32_t pow_bits = 16;
};

struct ChipRef {
    int chip_id;
    uint32_t log_n;
};

struct VerifyingKey {
    const MachineDesc *machine = nullptr;
    Digest prep_root;                 // all-zero when the machine has no preprocessed chip
    std::vector<ChipRef> prep_chips;  // chips with preprocessed columns (always part of every shard)
    std::vector<uint32_t> extra;      // machine-specific words bound into the key (rv32: entry pc)
};

// FRI batching order of the LDE columns of log-height h: every column opened at two
// points (preprocessed, main, permutation; tree-major, then chip, then column) first,
// then the quotient columns (opened at zeta only).
struct ColRef {
    int tree, mat, col, chip_pos;
};
std::vector<ColRef> fri_columns(const MachineDesc *m, const std::vector<ChipRef> &chips, uint32_t h, uint32_t *n_two);
void transcript_begin(Challenger &ch, const VerifyingKey &vk, const std::vector<ChipRef> &chips);

// LogUp challenges shared by all shards of one execution (derived from every shard's main commitment);
// nullptr = the shard samples its own (single, self-contained proof).
struct PermChallenges {
    Fp4 alpha, beta;
};
// returns "" on success, otherwise the reason for rejection.  With cumsum_total != nullptr the sum of the
// chips' cumulative sums is returned instead of being required to vanish (the caller balances it across shards).
std::string verify_shard(const VerifyingKey &vk, const ShardProof &proof, const StarkConfig &cfg,
                         const PermChallenges *global = nullptr, Fp4 *cumsum_total = nullptr);

#if defined(__HIPCC__)
struct Arena {
    char *base = nullptr;
    size_t cap = 0, off = 0;
    hipError_t reserve(size_t bytes);
    void release();
    void reset() { off = 0; }
    template <class T> T *alloc(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        if (off + bytes > cap) return nullptr;
        T *p = reinterpret_cast<T *>(base + off);
        off += bytes;
        return p;
    }
};

// Size-keyed cache of freed device buffers.  The per-shard buffers of a prove (cycle records, cpu trace, main-trace LDEs,
// Merkle tree: ~3 GB per 2^21-cycle shard) have the same sizes shard after shard and call after call; hipMalloc / hipFree
// of them cost ~80 ms per shard (measured: 3.85 s instead of 1.3 s per 32-shard prove), a cached buffer costs nothing.
// Not thread-safe: callers hold the prover's mutex.
struct DevPool {
    std::multimap<size_t, void *> cached;
    std::unordered_map<void *, size_t> live;
    size_t cached_bytes = 0, misses = 0;   // (misses: requests that went to hipMalloc)
    hipError_t alloc_bytes(void **out, size_t bytes) {
        bytes = (bytes + 4095) & ~(size_t)4095;
        auto it = cached.lower_bound(bytes);
        if (it != cached.end() && it->first <= bytes + bytes / 8) {
            *out = it->second;
            live[*out] = it->first;
            cached_bytes -= it->first;
            cached.erase(it);
            return hipSuccess;
        }
        misses++;
        hipError_t e = hipMalloc(out, bytes);
        if (e != hipSuccess) {   // give the cache back to the driver and try once more
            (void)hipGetLastError();
            trim();
            e = hipMalloc(out, bytes);
        }
        if (e == hipSuccess) live[*out] = bytes;
        return e;
    }
    template <class T> hipError_t alloc(T **out, size_t bytes) { return alloc_bytes(reinterpret_cast<void **>(out), bytes); }
    void free(void *p) {
        if (!p) return;
        auto it = live.find(p);
        if (it == live.end()) { (void)hipFree(p); return; }
        cached.emplace(it->second, p);
        cached_bytes += it->second;
        live.erase(it);
    }
    void trim() {
        for (auto &kv : cached) (void)hipFree(kv.second);
        cached.clear();
        cached_bytes = 0;
    }
};

struct ChipTrace {
    int chip_id;
    uint32_t log_n;
    const uint32_t *d_main;  // device [main_w][2^log_n], Montgomery form
};

struct ProvingKey {
    VerifyingKey vk;
    struct Prep { int chip_id; uint32_t log_n; uint32_t *d_trace, *d_lde; };
    std::vector<Prep> prep;
    uint32_t *d_prep_digests = nullptr;
    uint32_t prep_log_h = 0;  // log2 of the tallest preprocessed LDE
};

// Phase-1 results of one shard kept in HBM for phase 2 (main-trace LDEs + their Merkle tree), so that the
// common-challenge protocol does not pay K1-K3 of the main trace twice.  Owned by the caller.
struct MainCache {
    bool valid = false;
    DevPool *pool = nullptr;     // where lde / tree came from
    std::vector<uint32_t *> lde;  // one per chip trace, in trace order
    std::vector<size_t> lde_words;
    uint32_t *tree = nullptr;
    uint32_t log_h = 0;
    Digest root;
    // buffers are reused by the next commit of a shard of the same shape (benchmarks prove the same job repeatedly)
    bool fits(const std::vector<size_t> &words, uint32_t h) const { return tree && log_h == h && lde_words == words; }
    void release() {
        for (auto p : lde) if (p) { if (pool) pool->free(p); else (void)hipFree(p); }
        lde.clear();
        lde_words.clear();
        if (tree) { if (pool) pool->free(tree); else (void)hipFree(tree); }
        tree = nullptr;
        valid = false;
    }
};

struct StageTimes {  // milliseconds, HIP events on the prover stream (profile mode)
    float commit_main = 0, perm = 0, quotient = 0, open = 0, fri = 0, total = 0;
    // kernel families, summed over the launches of one prove_shard call
    float lde_ms = 0;          // K1: ntt_strided<inverse> + lde_block + ntt_strided<forward>
    double lde_alg_bytes = 0;  // sum over LDE calls of 12 * width * N  (read N words, write 2N words per column)
    int lde_calls = 0;
    float merkle_ms = 0;       // K2 + K3 of the three trace commitments (leaf hashing + levels)
    double merkle_perms = 0;   // Poseidon2 permutations executed by them
    // committed cells of the shard in BabyBear elements: main, permutation and quotient
    // (both flattened over F_p^4), preprocessed
    double cells_m = 0, cells_p = 0, cells_q = 0, cells_pre = 0;
};

class Engine {
  public:
    int device = 0;
    hipStream_t stream = nullptr;
    NttTables tabs;
    std::string err;
    StageTimes times;
    bool profile = false;

    hipError_t init(int dev);
    void shutdown();
    // small host->device tables (column pointer lists, powers, indices); stream-ordered
    const void *upload(const void *host, size_t bytes);
    template <class T> const T *upload_vec(const std::vector<T> &v) { return static_cast<const T *>(upload(v.data(), v.size() * sizeof(T))); }
    bool download(void *host, const void *dev, size_t bytes);  // synchronises the stream

    struct DevMat { const uint32_t *ptr; uint32_t width, log_h; };
    bool commit_tree(const std::vector<DevMat> &mats, uint32_t *d_digests);
    // upper levels of a tree whose 2^log_h leaf digests are already in place (no injection)
    bool commit_tree_levels(uint32_t *d_digests, uint32_t log_h);

    // host_prep[i]: canonical column-major trace of the i-th chip that has preprocessed columns
    bool setup(const MachineDesc *m, const std::vector<ChipRef> &prep_chips, const std::vector<std::vector<uint32_t>> &host_prep,
               ProvingKey *pk);
    void free_key(ProvingKey *pk);
    bool prove_shard(const ProvingKey &pk, const std::vector<ChipTrace> &traces, const std::vector<Fp> &pubs,
                     const StarkConfig &cfg, ShardProof *out, const PermChallenges *global = nullptr,
                     const MainCache *cached = nullptr);
    // phase 1 of a multi-shard proof: K1-K3 of the main traces only -> main_root; with keep != nullptr the
    // LDEs and the tree stay in HBM (hipMalloc'ed into *keep) for prove_shard(..., cached = keep)
    bool commit_main_root(const ProvingKey &pk, const std::vector<ChipTrace> &traces, Digest *root, MainCache *keep = nullptr);

    Arena arena;
    DevPool pool;
    // K5 selector tables, one per trace height this prover has seen ([3][2N] words each, stark.cuh QuotientArgs::sel)
    uint32_t *sel_tables[32] = {};
    // row digests of the shorter matrices of the tree being committed (commit_tree): reused tree after tree on the one stream
    uint32_t *inject_buf = nullptr;
    size_t inject_words = 0;
    const uint32_t *selector_table(const QuotientArgs &qa);

  private:
    char *d_ring = nullptr, *h_ring = nullptr;
    char *h_down = nullptr;  // pinned staging for downloads (roots, opened values, query data): no pageable-memory path
    static constexpr size_t DOWN_BYTES = 16u << 20;   // (every download of a proof goes through this pinned buffer: see the note on pageable memory in capi.hip)
    size_t ring_bytes = 0, ring_pos = 0;
    bool fail(const char *fmt, ...);
};
#endif

}  // namespace dvt
