// C-ABI layer (include/dvt_prover.h) over the gfx950 engine.  There is no CPU
// fallback anywhere in this file: without a HIP device every entry point that
// computes returns DVT_ERR_DEVICE.  (dvt_machine_verify is host-only by nature.)
#include "../../include/dvt_prover.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#include "engine.h"
#include "poseidon2_f64.cuh"
#include "rv32.h"
#include "sha256.h"

using namespace dvt;

struct dvt_prover {
    Engine eng;
    StarkConfig cfg;
    uint32_t log_shard = 21;          // cycles per shard = 2^log_shard (SP1's default shard size)
    uint64_t max_cycles = 1ull << 36;
    bool keep_phase1 = true;          // keep K0 output, main LDEs and tree of phase 1 in HBM for phase 2 ("keep_phase1": 0 recomputes)
    uint32_t exec_threads = 0;        // trace-mode executor threads of the prove pipeline ("exec_threads", 0 = from the host's core count)
    hipStream_t copy_stream = nullptr;            // record uploads overlap the previous shard's kernels
    std::vector<rv32::CycleRec *> pinned;         // pinned staging buffers of 2^log_shard records each, reused across calls
    // Pinned staging of everything else a shard uploads (auxiliary traces, precompile calls).  Handing the runtime PAGEABLE
    // memory makes it pin the pages on the fly; when the vectors are freed afterwards the driver quiesces every queue of the
    // process to drop that mapping - measured as a 20-30 ms stall of the GPU right before phase 1 of a single-shard proof.
    uint8_t *aux_pinned = nullptr;
    size_t aux_pinned_bytes = 0;
    std::string err;
    std::mutex mu;
};
struct dvt_pk {
    ProvingKey key;
    bool is_rv32 = false;
    rv32::Program prog;
    rv32::HostPrep prep;
    rv32::Instr *d_instrs = nullptr;   // device copy of prog.instrs (K0)
    uint32_t *d_prog_row = nullptr;    // instruction index -> program-table row
};
static thread_local std::string g_create_err;

static int fail(dvt_prover *p, int code, const char *fmt, ...) {
    char buf[600];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (p) p->err = buf; else g_create_err = buf;
    return code;
}
#define HIP_TRY(p, expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail(p, DVT_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static int cfg_int(const char *json, const char *key, int dflt) {
    if (!json) return dflt;
    std::string pat = std::string("\"") + key + "\"";
    const char *s = strstr(json, pat.c_str());
    if (!s) return dflt;
    s = strchr(s + pat.size(), ':');
    if (!s) return dflt;
    return atoi(s + 1);
}

namespace dvt {
const MachineDesc *machine_by_name(const char *name) {
    if (!name) return nullptr;
    if (!strcmp(name, "toy")) return machine_toy();
    if (!strcmp(name, "rv32")) return machine_rv32();
    return nullptr;
}
}  // namespace dvt

static uint8_t *copy_out(const std::vector<uint32_t> &w, size_t *len) {
    uint8_t *b = (uint8_t *)malloc(w.size() * 4 + 1);
    if (!b) return nullptr;
    memcpy(b, w.data(), w.size() * 4);
    *len = w.size() * 4;
    return b;
}

static std::vector<uint32_t> vk_words(const VerifyingKey &vk) {
    WordWriter w;
    w.u32(0x314b5644u);  // "DVK1"
    char name[16] = {0};
    strncpy(name, vk.machine->name, 15);
    for (int i = 0; i < 4; i++) { uint32_t v; memcpy(&v, name + 4 * i, 4); w.u32(v); }
    w.dg(vk.prep_root);
    w.u32((uint32_t)vk.prep_chips.size());
    for (auto &c : vk.prep_chips) { w.u32((uint32_t)c.chip_id); w.u32(c.log_n); }
    w.u32((uint32_t)vk.extra.size());
    for (auto x : vk.extra) w.u32(x);
    return w.w;
}
static bool vk_parse(const uint8_t *b, size_t len, VerifyingKey *vk) {
    if (len % 4 || len < 4 * 15) return false;
    std::vector<uint32_t> wv(len / 4);
    memcpy(wv.data(), b, len);
    try {
        WordReader r(wv.data(), wv.size());
        if (r.u32() != 0x314b5644u) return false;
        char name[17] = {0};
        for (int i = 0; i < 4; i++) { uint32_t v = r.u32(); memcpy(name + 4 * i, &v, 4); }
        vk->machine = machine_by_name(name);
        if (!vk->machine) return false;
        for (size_t i = strlen(name); i < 16; i++)
            if (name[i]) return false;   // one encoding per key: the padding after the machine name is zero
        vk->prep_root = r.dg();
        uint32_t n = r.len(64);
        for (uint32_t i = 0; i < n; i++) {
            ChipRef c;
            c.chip_id = (int)r.u32();
            c.log_n = r.u32();
            if (c.chip_id < 0 || c.chip_id >= vk->machine->n_chips || c.log_n > 22) return false;
            vk->prep_chips.push_back(c);
        }
        uint32_t ne = r.len(16);
        for (uint32_t i = 0; i < ne; i++) vk->extra.push_back(r.u32());
        if (r.p != r.end) return false;
    } catch (const std::exception &) { return false; }
    return true;
}

extern "C" {

uint32_t dvt_abi_version(void) { return 3; }

int dvt_prover_create(const char *cfg_json, dvt_prover **out) {
    if (!out) return fail(nullptr, DVT_ERR_INPUT, "out == NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, DVT_ERR_DEVICE, "no HIP device (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    int dev = cfg_int(cfg_json, "device", 0);
    if (dev < 0 || dev >= ndev) return fail(nullptr, DVT_ERR_INPUT, "device %d out of range (%d present)", dev, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipSetDevice(dev));
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, DVT_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
    dvt_prover *p = new dvt_prover();
    p->cfg.num_queries = (uint32_t)cfg_int(cfg_json, "fri_queries", 100);
    p->cfg.pow_bits = (uint32_t)cfg_int(cfg_json, "pow_bits", 16);
    p->eng.profile = cfg_int(cfg_json, "profile", 0) != 0;
    p->log_shard = (uint32_t)cfg_int(cfg_json, "log_shard_size", 21);
    p->keep_phase1 = cfg_int(cfg_json, "keep_phase1", 1) != 0;
    p->exec_threads = (uint32_t)std::max(0, cfg_int(cfg_json, "exec_threads", 0));
    if (p->log_shard < 4 || p->log_shard > 22) { delete p; return fail(nullptr, DVT_ERR_INPUT, "log_shard_size must be 4..22"); }
    if (p->cfg.num_queries == 0 || p->cfg.num_queries > 1024 || p->cfg.pow_bits > 30) {
        delete p;
        return fail(nullptr, DVT_ERR_INPUT, "fri_queries must be 1..1024 and pow_bits <= 30");
    }
    e = p->eng.init(dev);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        fail(nullptr, DVT_ERR_DEVICE, "handle setup: %s", hipGetErrorString(e));
        dvt_prover_destroy(p);
        return DVT_ERR_DEVICE;
    }
    *out = p;
    return DVT_OK;
}

void dvt_prover_destroy(dvt_prover *p) {
    if (!p) return;
    (void)hipSetDevice(p->eng.device);
    if (p->copy_stream) { (void)hipStreamSynchronize(p->copy_stream); (void)hipStreamDestroy(p->copy_stream); }
    for (auto b : p->pinned) (void)hipHostFree(b);
    if (p->aux_pinned) (void)hipHostFree(p->aux_pinned);
    p->eng.shutdown();
    delete p;
}

const char *dvt_last_error(const dvt_prover *p) { return p ? p->err.c_str() : g_create_err.c_str(); }
void dvt_free(void *ptr) { free(ptr); }
void *dvt_stream(dvt_prover *p) { return p ? (void *)p->eng.stream : nullptr; }

int dvt_sync(dvt_prover *p) {
    if (!p) return DVT_ERR_INPUT;
    HIP_TRY(p, hipSetDevice(p->eng.device));
    HIP_TRY(p, hipStreamSynchronize(p->eng.stream));
    return DVT_OK;
}

int dvt_dev_to_internal(dvt_prover *p, uint32_t *d, size_t n) {
    if (!p || (!d && n)) return fail(p, DVT_ERR_INPUT, "null argument");
    HIP_TRY(p, hipSetDevice(p->eng.device));
    HIP_TRY(p, launch_to_internal(p->eng.stream, d, n));
    return DVT_OK;
}
int dvt_dev_from_internal(dvt_prover *p, uint32_t *d, size_t n) {
    if (!p || (!d && n)) return fail(p, DVT_ERR_INPUT, "null argument");
    HIP_TRY(p, hipSetDevice(p->eng.device));
    HIP_TRY(p, launch_from_internal(p->eng.stream, d, n));
    return DVT_OK;
}

int dvt_stage_coset_lde(dvt_prover *p, uint32_t *d_in, uint32_t *d_scratch, uint32_t *d_out, uint32_t width, uint32_t log_n,
                        uint32_t shift_mode) {
    if (!p) return DVT_ERR_INPUT;
    if (width && (!d_in || !d_out)) return fail(p, DVT_ERR_INPUT, "null matrix");
    if (log_n > 22) return fail(p, DVT_ERR_INPUT, "log_n %u > 22", log_n);
    if (shift_mode > 2) return fail(p, DVT_ERR_INPUT, "shift_mode %u", shift_mode);
    HIP_TRY(p, hipSetDevice(p->eng.device));
    HIP_TRY(p, launch_coset_lde(p->eng.stream, p->eng.tabs, d_in, d_scratch, d_out, width, log_n, shift_mode));
    return DVT_OK;
}

size_t dvt_merkle_digest_words(const dvt_dev_matrix *mats, size_t n) {
    uint32_t mx = 0;
    for (size_t i = 0; i < n; i++) mx = std::max(mx, mats[i].log_height);
    return (((size_t)2 << mx) - 1) * 8;
}

int dvt_stage_merkle_commit(dvt_prover *p, const dvt_dev_matrix *mats, size_t n, uint32_t *d_digests) {
    if (!p) return DVT_ERR_INPUT;
    if (!mats || !n || !d_digests) return fail(p, DVT_ERR_INPUT, "null argument");
    std::lock_guard<std::mutex> lk(p->mu);
    HIP_TRY(p, hipSetDevice(p->eng.device));
    std::vector<Engine::DevMat> dm;
    for (size_t i = 0; i < n; i++) {
        if (mats[i].log_height > 30) return fail(p, DVT_ERR_INPUT, "log_height too large");
        if (mats[i].width && !mats[i].d_data) return fail(p, DVT_ERR_INPUT, "null matrix data");
        dm.push_back({mats[i].d_data, mats[i].width, mats[i].log_height});
    }
    if (!p->eng.commit_tree(dm, d_digests)) return fail(p, DVT_ERR_DEVICE, "%s", p->eng.err.c_str());
    return DVT_OK;
}

int dvt_stage_poseidon2_permute(dvt_prover *p, uint32_t *d_states, size_t n) {
    if (!p || (!d_states && n)) return fail(p, DVT_ERR_INPUT, "null argument");
    HIP_TRY(p, hipSetDevice(p->eng.device));
    HIP_TRY(p, launch_poseidon2_permute(p->eng.stream, d_states, n));
    return DVT_OK;
}

int dvt_stage_fri_fold(dvt_prover *p, const uint32_t *d_v, uint32_t *d_out, const uint32_t *d_ro, const uint32_t beta[4],
                       uint32_t log_m) {
    if (!p || !d_v || !d_out || !beta) return fail(p, DVT_ERR_INPUT, "null argument");
    if (log_m < 1 || log_m > 23) return fail(p, DVT_ERR_INPUT, "log_m out of range");
    Fp4 b;
    for (int k = 0; k < 4; k++) {
        if (beta[k] >= P) return fail(p, DVT_ERR_INPUT, "beta not canonical");
        b.c[k] = Fp::from_canonical(beta[k]);
    }
    HIP_TRY(p, hipSetDevice(p->eng.device));
    HIP_TRY(p, launch_fri_fold(p->eng.stream, p->eng.tabs, reinterpret_cast<const Fp4 *>(d_v), reinterpret_cast<Fp4 *>(d_out),
                               reinterpret_cast<const Fp4 *>(d_ro), b, log_m));
    return DVT_OK;
}

// ------------------------------------------------------------------ machine level
int dvt_machine_setup(dvt_prover *p, const char *machine, const dvt_host_trace *prep, size_t nprep, dvt_pk **pk_out,
                      uint8_t **vk, size_t *vk_len) {
    if (!p || !pk_out) return fail(p, DVT_ERR_INPUT, "null argument");
    std::lock_guard<std::mutex> lk(p->mu);
    const MachineDesc *m = machine_by_name(machine);
    if (!m) return fail(p, DVT_ERR_INPUT, "unknown machine '%s'", machine ? machine : "(null)");
    std::vector<ChipRef> refs;
    std::vector<std::vector<uint32_t>> host;
    for (size_t i = 0; i < nprep; i++) {
        if ((int)prep[i].chip_id >= m->n_chips || prep[i].log_n > 22 || !prep[i].data) return fail(p, DVT_ERR_INPUT, "bad preprocessed trace %zu", i);
        if (i && prep[i].chip_id <= prep[i - 1].chip_id) return fail(p, DVT_ERR_INPUT, "preprocessed traces must be sorted by chip id");
        size_t words = (size_t)m->chips[prep[i].chip_id].prep_w << prep[i].log_n;
        for (size_t k = 0; k < words; k++)
            if (prep[i].data[k] >= P) return fail(p, DVT_ERR_INPUT, "preprocessed trace %zu holds a non-canonical value", i);
        refs.push_back({(int)prep[i].chip_id, prep[i].log_n});
        host.emplace_back(prep[i].data, prep[i].data + words);
    }
    for (int c = 0; c < m->n_chips; c++)
        if (m->chips[c].prep_w) {
            bool have = false;
            for (auto &r : refs) have |= r.chip_id == c;
            if (!have) return fail(p, DVT_ERR_INPUT, "chip %s needs a preprocessed trace", m->chips[c].name);
        }
    dvt_pk *pk = new dvt_pk();
    if (!p->eng.setup(m, refs, host, &pk->key)) {
        p->eng.free_key(&pk->key);
        delete pk;
        return fail(p, DVT_ERR_DEVICE, "%s", p->eng.err.c_str());
    }
    if (vk && vk_len) {
        *vk = copy_out(vk_words(pk->key.vk), vk_len);
        if (!*vk) { p->eng.free_key(&pk->key); delete pk; return fail(p, DVT_ERR_DEVICE, "out of host memory"); }
    }
    *pk_out = pk;
    return DVT_OK;
}

void dvt_pk_free(dvt_prover *p, dvt_pk *pk) {
    if (!p || !pk) return;
    std::lock_guard<std::mutex> lk(p->mu);
    p->eng.free_key(&pk->key);
    if (pk->d_instrs) (void)hipFree(pk->d_instrs);
    if (pk->d_prog_row) (void)hipFree(pk->d_prog_row);
    delete pk;
}

int dvt_machine_prove(dvt_prover *p, const dvt_pk *pk, const dvt_host_trace *main, size_t nmain, const uint32_t *pubs, size_t npub,
                      uint8_t **proof, size_t *proof_len) {
    if (!p || !pk || !main || !nmain || !proof || !proof_len || (npub && !pubs)) return fail(p, DVT_ERR_INPUT, "null argument");
    std::lock_guard<std::mutex> lk(p->mu);
    const MachineDesc *m = pk->key.vk.machine;
    HIP_TRY(p, hipSetDevice(p->eng.device));
    std::vector<uint32_t *> dev(nmain, nullptr);
    std::vector<ChipTrace> traces;
    int rc = DVT_OK;
    auto cleanup = [&] { for (auto d : dev) if (d) (void)hipFree(d); };
    for (size_t i = 0; i < nmain && rc == DVT_OK; i++) {
        if ((int)main[i].chip_id >= m->n_chips || main[i].log_n > 22 || !main[i].data) { rc = fail(p, DVT_ERR_INPUT, "bad main trace %zu", i); break; }
        size_t words = (size_t)m->chips[main[i].chip_id].main_w << main[i].log_n;
        for (size_t k = 0; k < words; k++)
            if (main[i].data[k] >= P) { rc = fail(p, DVT_ERR_INPUT, "main trace %zu holds a non-canonical value", i); break; }
        if (rc) break;
        if (hipMalloc(&dev[i], words * 4) != hipSuccess || hipMemcpy(dev[i], main[i].data, words * 4, hipMemcpyHostToDevice) != hipSuccess ||
            launch_to_internal(p->eng.stream, dev[i], words) != hipSuccess) {
            rc = fail(p, DVT_ERR_DEVICE, "uploading main trace %zu failed", i);
            break;
        }
        traces.push_back({(int)main[i].chip_id, main[i].log_n, dev[i]});
    }
    if (rc) { cleanup(); return rc; }
    std::vector<Fp> pv(npub);
    for (size_t i = 0; i < npub; i++) {
        if (pubs[i] >= P) { cleanup(); return fail(p, DVT_ERR_INPUT, "public value %zu not canonical", i); }
        pv[i] = Fp::from_canonical(pubs[i]);
    }
    ShardProof sp;
    bool ok = p->eng.prove_shard(pk->key, traces, pv, p->cfg, &sp);
    (void)hipStreamSynchronize(p->eng.stream);
    cleanup();
    if (!ok) return fail(p, DVT_ERR_DEVICE, "%s", p->eng.err.c_str());
    WordWriter w;
    write_shard_proof(w, sp);
    *proof = copy_out(w.w, proof_len);
    if (!*proof) return fail(p, DVT_ERR_DEVICE, "out of host memory");
    return DVT_OK;
}

int dvt_machine_verify(const uint8_t *vk, size_t vk_len, const uint8_t *proof, size_t proof_len, uint32_t fri_queries,
                       uint32_t pow_bits, char **reason) {
    if (reason) *reason = nullptr;
    auto reject = [&](int code, const std::string &why) {
        if (reason) *reason = strdup(why.c_str());
        return code;
    };
    if (!vk || !proof) return reject(DVT_ERR_INPUT, "null argument");
    VerifyingKey key;
    if (!vk_parse(vk, vk_len, &key)) return reject(DVT_ERR_INPUT, "malformed verifying key");
    if (proof_len % 4) return reject(DVT_ERR_INPUT, "proof length is not a multiple of 4");
    std::vector<uint32_t> words(proof_len / 4);
    memcpy(words.data(), proof, proof_len);
    ShardProof sp;
    try {
        WordReader r(words.data(), words.size());
        sp = read_shard_proof(r);
        if (r.p != r.end) return reject(DVT_ERR_REJECTED, "trailing bytes after proof");
    } catch (const std::exception &e) { return reject(DVT_ERR_REJECTED, e.what()); }
    StarkConfig cfg;
    cfg.num_queries = fri_queries;
    cfg.pow_bits = pow_bits;
    std::string why = verify_shard(key, sp, cfg);
    if (!why.empty()) return reject(DVT_ERR_REJECTED, why);
    return DVT_OK;
}

int dvt_last_kernel_stats(dvt_prover *p, double out[9]) {
    if (!p || !out) return DVT_ERR_INPUT;
    const StageTimes &t = p->eng.times;
    out[0] = t.lde_ms; out[1] = t.lde_alg_bytes; out[2] = t.lde_calls; out[3] = t.merkle_ms; out[4] = t.merkle_perms;
    out[5] = t.cells_m; out[6] = t.cells_p; out[7] = t.cells_q; out[8] = t.cells_pre;
    return DVT_OK;
}

int dvt_last_stage_ms(dvt_prover *p, float out[6]) {
    if (!p || !out) return DVT_ERR_INPUT;
    const StageTimes &t = p->eng.times;
    out[0] = t.commit_main; out[1] = t.perm; out[2] = t.quotient; out[3] = t.open; out[4] = t.fri; out[5] = t.total;
    return DVT_OK;
}

}  // extern "C"

// ====================================================================== rv32 boundary
namespace {
constexpr uint32_t CORE_PROOF_MAGIC = 0x33435644u;  // "DVC3"
constexpr uint32_t N_PUB = rv32::N_PUBLIC;           // start_pc, next_pc, exit_code, shard, is_last
constexpr uint32_t HEADER_WORDS = 8 + N_PUB;         // per-shard commitment header: main root + public values (canonical)
constexpr uint32_t PV_BUS = 5;                       // tools/airgen/rv32.py BUSES["sys"]

std::vector<std::vector<uint8_t>> collect_stdin(const dvt_buf *bufs, size_t n) {
    std::vector<std::vector<uint8_t>> v(n);
    for (size_t i = 0; i < n; i++)
        if (bufs[i].len) v[i].assign(bufs[i].data, bufs[i].data + bufs[i].len);
    return v;
}
uint8_t *dup_bytes(const std::vector<uint8_t> &v, size_t *len) {
    uint8_t *b = (uint8_t *)malloc(v.size() + 1);
    if (b && !v.empty()) memcpy(b, v.data(), v.size());
    if (len) *len = v.size();
    return b;
}
// LogUp challenges common to all shards: transcript over the key and every shard's header
PermChallenges global_challenges(const VerifyingKey &vk, const uint32_t *headers, size_t n) {
    Challenger g;
    g.observe(vk.prep_root);
    g.observe_u32((uint32_t)n);
    for (size_t i = 0; i < n; i++) {
        const uint32_t *h = headers + i * HEADER_WORDS;
        for (uint32_t k = 0; k < 8; k++) g.observe(Fp::from_canonical(h[k]));
        g.observe_u32(N_PUB);
        for (uint32_t k = 0; k < N_PUB; k++) g.observe(Fp::from_canonical(h[8 + k] % P));
    }
    PermChallenges c;
    c.alpha = g.sample_ext();
    c.beta = g.sample_ext();
    return c;
}
// SP1's committed-value digest: word k = little-endian u32 of bytes 4k..4k+3 of SHA-256(public-value bytes)
// (an empty stream commits 42c4b0e3 141cfc98 ... = sha256("") read as LE words)
void pv_digest_words(const std::vector<uint8_t> &pv, uint32_t out[8]) {
    uint8_t dg[32];
    sha256(pv.data(), pv.size(), dg);
    for (int k = 0; k < 8; k++) out[k] = dg[4 * k] | (dg[4 * k + 1] << 8) | (dg[4 * k + 2] << 16) | ((uint32_t)dg[4 * k + 3] << 24);
}
}  // namespace

struct ShardJob {
    uint32_t index = 0;       // shard number (1-based) = position in the execution + 1
    size_t n_recs = 0;
    uint32_t next_pc = 0;
    rv32::CycleRec *d_recs = nullptr;
    uint32_t log_n[rv32::N_CHIPS] = {};
    bool present[rv32::N_CHIPS] = {};
    uint32_t *d_aux[rv32::N_CHIPS] = {};  // main traces except cpu
    std::vector<Fp> pubs;
    MainCache cache;  // phase-1 LDEs + tree of the main traces, consumed by phase 2
    // K0 output of this shard kept from phase 1 to phase 2 (with the cache, while HBM allows); otherwise the
    // job's working buffers are used and phase 2 runs K0 again
    uint32_t *d_cpu = nullptr, *d_byte = nullptr, *d_prog = nullptr;
    bool traces_valid = false;
    uint32_t header[HEADER_WORDS] = {};
    bool header_valid = false;   // phase 1 ran (inside the prepare pipeline, or by commit_shard) and no phase 2 has consumed it
};
// one prepared execution: cut into shards by the executor; the shards this job owns (first, first + stride, ...) are
// resident in HBM, ready for K0..K9
struct dvt_job {
    int exit_code = -1;
    uint64_t cycles = 0;
    std::vector<uint8_t> public_values;
    size_t n_total = 0, first = 0, stride = 1;   // shards of the execution / which of them this job holds
    std::vector<ShardJob> shards;
    uint32_t *d_cpu = nullptr, *d_byte = nullptr, *d_prog = nullptr;  // working buffers (largest shard)
    uint32_t work_log_cpu = 0;
    size_t byte_words = 0, prog_words = 0;
    double t_exec_wait = 0;   // seconds the GPU thread spent waiting for the executor inside prepare
    ShardJob *at(size_t pos) { return pos >= first && (pos - first) % stride == 0 && (pos - first) / stride < shards.size() ? &shards[(pos - first) / stride] : nullptr; }
};

static void job_release(dvt_prover *p, dvt_job *j) {
    if (!j) return;
    DevPool &pool = p->eng.pool;
    for (auto &s : j->shards) {
        pool.free(s.d_recs);
        for (auto &d : s.d_aux) pool.free(d);
        for (uint32_t *d : {s.d_cpu, s.d_byte, s.d_prog}) pool.free(d);
        s.cache.release();
    }
    pool.free(j->d_cpu);
    pool.free(j->d_byte);
    pool.free(j->d_prog);
    delete j;
}

// K0 of a shard (into the shard's own buffers when it has them, else the job's working buffers); fills the chip
// trace list of that shard.  `reuse`: phase 2 takes the traces phase 1 left behind instead of generating them again.
static int shard_traces(dvt_prover *p, const dvt_pk *pk, dvt_job *j, ShardJob &s, std::vector<ChipTrace> *traces, bool reuse) {
    hipStream_t st = p->eng.stream;
    const MachineDesc *m = machine_rv32();
    if (!s.d_cpu && (!j->d_cpu || j->work_log_cpu < s.log_n[RV32_CHIP_CPU])) {   // working buffers, sized for the largest shard seen
        HIP_TRY(p, hipStreamSynchronize(st));
        for (uint32_t **d : {&j->d_cpu, &j->d_byte, &j->d_prog}) { p->eng.pool.free(*d); *d = nullptr; }
        j->work_log_cpu = s.log_n[RV32_CHIP_CPU];
        HIP_TRY(p, p->eng.pool.alloc(&j->d_cpu, ((size_t)RV32_CPU_MAIN_W << j->work_log_cpu) * 4));
        HIP_TRY(p, p->eng.pool.alloc(&j->d_byte, j->byte_words * 4));
        HIP_TRY(p, p->eng.pool.alloc(&j->d_prog, j->prog_words * 4));
    }
    uint32_t *cpu = s.d_cpu ? s.d_cpu : j->d_cpu, *byte = s.d_cpu ? s.d_byte : j->d_byte, *prog = s.d_cpu ? s.d_prog : j->d_prog;
    if (!(reuse && s.d_cpu && s.traces_valid)) {
        bool ok = hipMemcpyAsync(byte, s.d_aux[RV32_CHIP_BYTE], j->byte_words * 4, hipMemcpyDeviceToDevice, st) == hipSuccess &&
                  hipMemcpyAsync(prog, s.d_aux[RV32_CHIP_PROGRAM], j->prog_words * 4, hipMemcpyDeviceToDevice, st) == hipSuccess &&
                  rv32::launch_k0_cpu_rows(st, s.d_recs, s.n_recs, s.index, s.next_pc, pk->d_instrs, pk->d_prog_row, cpu, s.log_n[RV32_CHIP_CPU], byte, prog) == hipSuccess &&
                  launch_to_internal(st, byte, j->byte_words) == hipSuccess && launch_to_internal(st, prog, j->prog_words) == hipSuccess;
        if (!ok) return fail(p, DVT_ERR_DEVICE, "trace generation (K0) failed: %s", hipGetErrorString(hipGetLastError()));
        s.traces_valid = s.d_cpu != nullptr;
    }
    traces->clear();
    for (int c = 0; c < m->n_chips; c++) {
        if (!s.present[c]) continue;
        const uint32_t *ptr = c == RV32_CHIP_CPU ? cpu : c == RV32_CHIP_BYTE ? byte : c == RV32_CHIP_PROGRAM ? prog : s.d_aux[c];
        traces->push_back({c, s.log_n[c], ptr});
    }
    return DVT_OK;
}

// phase 1 of a shard: K0 + K1..K3 of the main traces -> header
static int shard_commit(dvt_prover *p, const dvt_pk *pk, dvt_job *j, ShardJob &s) {
    std::vector<ChipTrace> traces;
    const bool time_stages = getenv("DVT_TIME_PREPARE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (time_stages) fprintf(stderr, "[commit] %s at %.2f ms (pool misses so far %zu)\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), p->eng.pool.misses);
    };
    // keep the phase-1 results in HBM while they fit (about 3 GB per 2^21-cycle shard); otherwise phase 2 recomputes
    size_t free_b = 0, total_b = 0;
    if (p->keep_phase1 && !s.cache.tree) { (void)hipMemGetInfo(&free_b, &total_b); free_b += p->eng.pool.cached_bytes; }   // (only the first commit of a shard asks)
    lap("memory asked");
    MainCache *keep = p->keep_phase1 && (s.cache.tree || free_b > ((size_t)24 << 30)) ? &s.cache : nullptr;
    if (keep && !s.d_cpu) {
        DevPool &pool = p->eng.pool;
        bool ok = pool.alloc(&s.d_cpu, ((size_t)RV32_CPU_MAIN_W << s.log_n[RV32_CHIP_CPU]) * 4) == hipSuccess && pool.alloc(&s.d_byte, j->byte_words * 4) == hipSuccess &&
                  pool.alloc(&s.d_prog, j->prog_words * 4) == hipSuccess;
        if (!ok) {  // not fatal: fall back to the shared working buffers
            (void)hipGetLastError();
            for (uint32_t **d : {&s.d_cpu, &s.d_byte, &s.d_prog}) { pool.free(*d); *d = nullptr; }
        }
    }
    lap("trace buffers");
    int rc = shard_traces(p, pk, j, s, &traces, false);
    if (rc) return rc;
    lap("K0 launched");
    Digest root;
    if (!p->eng.commit_main_root(pk->key, traces, &root, keep)) return fail(p, DVT_ERR_DEVICE, "%s", p->eng.err.c_str());
    lap("main root");
    for (int k = 0; k < 8; k++) s.header[k] = root.d[k].canonical();
    for (uint32_t k = 0; k < N_PUB; k++) s.header[8 + k] = s.pubs[k].canonical();
    s.header_valid = true;
    return DVT_OK;
}

// phase 2 of a shard: K0..K9 with the common challenges -> shard proof words
static int shard_prove(dvt_prover *p, const dvt_pk *pk, dvt_job *j, ShardJob &s, const PermChallenges &gc, std::vector<uint32_t> *words) {
    std::vector<ChipTrace> traces;
    int rc = shard_traces(p, pk, j, s, &traces, s.cache.valid);
    if (rc) return rc;
    ShardProof sp;
    bool ok = p->eng.prove_shard(pk->key, traces, s.pubs, p->cfg, &sp, &gc, &s.cache);
    (void)hipStreamSynchronize(p->eng.stream);
    s.cache.valid = false;  // the buffers stay for the next commit of this shard (released with the job)
    s.traces_valid = false;
    s.header_valid = false;
    if (!ok) return fail(p, DVT_ERR_DEVICE, "%s", p->eng.err.c_str());
    WordWriter w;
    w.w.reserve((size_t)1 << 20);  // a shard proof is about 2.4 MB at 100 queries
    write_shard_proof(w, sp);
    *words = std::move(w.w);
    return DVT_OK;
}

// ------------------------------------------------------------------ the prepare pipeline
// One sequential FAST pass of the guest finds the shard boundaries and snapshots the machine there; trace-mode
// executor threads re-run the owned shards from the snapshots into pinned buffers and build the small auxiliary
// traces; the calling thread uploads shard i+1 on the copy stream while the GPU runs phase 1 (K0 + K1..K3 of the main
// traces) of shard i.  (reference src/main.rs:461-466: prove() executes AND proves in one call.)
namespace {
struct ReadyShard {
    rv32::CycleRec *buf = nullptr;
    rv32::ShardMeta meta{};
    rv32::HostTraces aux;
    rv32::BigOpBatches big;   // the precompile calls of the shard: their chips' rows are built on the GPU
    std::string err;
    bool unsupported = false;
};
struct Pipeline {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<size_t, rv32::Snapshot>> snaps;
    bool snaps_closed = false, fast_done = false;
    std::atomic<bool> abort{false};   // (also read outside the mutex by a worker that is about to build auxiliary traces)
    std::vector<rv32::CycleRec *> free_bufs;
    std::map<size_t, ReadyShard> ready;
    // results of the fast pass
    size_t n_total = 0;
    std::vector<rv32::MemInitRow> mem_rows;
    int exit_code = -1;
    bool halted = false, unsupported = false;
    uint64_t cycles = 0;
    std::string error, unsupported_what;
    std::vector<uint8_t> public_values;
    uint32_t committed[8] = {}, committed_mask = 0;
};
}  // namespace

static int job_prepare(dvt_prover *p, const dvt_pk *pk, const dvt_buf *stdin_bufs, size_t nbuf, size_t first, size_t stride, dvt_job **out,
                       dvt_report *report) {
    HIP_TRY(p, hipSetDevice(p->eng.device));
    if (stride == 0 || first >= stride) return fail(p, DVT_ERR_INPUT, "bad shard partition %zu / %zu", first, stride);
    const std::vector<std::vector<uint8_t>> inputs = collect_stdin(stdin_bufs, nbuf);
    const rv32::Program &prog = pk->prog;
    const uint32_t log_shard = p->log_shard;
    const uint64_t max_cycles = p->max_cycles;
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_workers = p->exec_threads ? p->exec_threads : std::max(1u, std::min(6u, hw > 3 ? hw - 2 : 1u));
    // pinned staging: one buffer per worker + two in flight on the GPU side
    const size_t want_bufs = n_workers + 2;
    while (p->pinned.size() < want_bufs) {
        rv32::CycleRec *b = nullptr;
        HIP_TRY(p, hipHostMalloc(&b, sizeof(rv32::CycleRec) << log_shard));
        p->pinned.push_back(b);
    }
    Pipeline pl;
    pl.free_bufs = p->pinned;
    const bool time_stages = getenv("DVT_TIME_PREPARE") != nullptr;   // (stderr: where the host side of a prepare goes)
    const auto t_begin = std::chrono::steady_clock::now();
    auto since_begin = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };

    rv32::CurveLog curve_log;
    std::thread fast([&] {
        rv32::Vm vm(prog, &inputs, log_shard);
        vm.curve_log = &curve_log;
        struct Close { rv32::CurveLog &l; ~Close() { l.closed.store(true, std::memory_order_release); } } close_log{curve_log};
        size_t pos = 0;
        for (;; pos++) {
            if (pos % stride == first) {
                rv32::Snapshot snap = vm.snapshot();
                std::unique_lock<std::mutex> lk(pl.mu);
                pl.cv.wait(lk, [&] { return pl.snaps.size() < 2 * (size_t)n_workers + 2 || pl.abort; });
                if (pl.abort) break;
                pl.snaps.emplace_back(pos, std::move(snap));
                pl.cv.notify_all();
            }
            vm.run_shard(false, nullptr, max_cycles);
            if (!vm.error.empty() || vm.halted || !vm.next_shard()) break;
        }
        std::vector<rv32::MemInitRow> rows;
        if (vm.halted) rows = vm.mem_rows();
        std::lock_guard<std::mutex> lk(pl.mu);
        pl.n_total = pos + 1;
        pl.mem_rows = std::move(rows);
        pl.exit_code = vm.exit_code; pl.halted = vm.halted; pl.cycles = vm.cycles; pl.error = vm.error;
        pl.unsupported = vm.unsupported; pl.unsupported_what = vm.unsupported_what;
        pl.public_values = std::move(vm.public_values);
        for (int k = 0; k < 8; k++) pl.committed[k] = vm.committed[k];
        pl.committed_mask = vm.committed_mask;
        pl.fast_done = pl.snaps_closed = true;
        pl.cv.notify_all();
    });
    std::vector<std::thread> workers;
    for (unsigned w = 0; w < n_workers; w++)
        workers.emplace_back([&] {
            for (;;) {
                rv32::CycleRec *buf = nullptr;
                size_t pos = 0;
                rv32::Snapshot snap;
                {
                    // a buffer first, then the OLDEST snapshot: buffers are handed out in shard order, so the shard the GPU
                    // thread waits for always has one
                    std::unique_lock<std::mutex> lk(pl.mu);
                    pl.cv.wait(lk, [&] { return pl.abort || ((!pl.snaps.empty() || pl.snaps_closed) && (!pl.free_bufs.empty() || pl.snaps.empty())); });
                    if (pl.abort || pl.snaps.empty()) return;
                    buf = pl.free_bufs.back();
                    pl.free_bufs.pop_back();
                    pos = pl.snaps.front().first;
                    snap = std::move(pl.snaps.front().second);
                    pl.snaps.pop_front();
                    pl.cv.notify_all();
                }
                ReadyShard r;
                r.buf = buf;
                const auto tw0 = std::chrono::steady_clock::now();
                {
                    rv32::Vm vm(prog, &inputs, log_shard, snap);
                    vm.curve_log = &curve_log;
                    snap = rv32::Snapshot();
                    rv32::ShardOut so;
                    so.recs = buf;
                    vm.run_shard(true, &so, max_cycles);
                    const auto tw1 = std::chrono::steady_clock::now();
                    r.meta = rv32::ShardMeta{so.index, so.start_pc, so.next_pc, so.n_recs};
                    if (!vm.error.empty()) { r.err = vm.error; r.unsupported = vm.unsupported; }
                    else {
                        const std::vector<rv32::MemInitRow> *rows = nullptr;
                        int ec = 0;
                        if (vm.halted) {   // the last shard carries the mem_init table: final memory state of the fast pass
                            std::unique_lock<std::mutex> lk(pl.mu);
                            pl.cv.wait(lk, [&] { return pl.fast_done || pl.abort; });
                            rows = &pl.mem_rows;
                            ec = pl.exit_code;
                        }
                        std::string e;
                        const auto tw2 = std::chrono::steady_clock::now();
                        if (!pl.abort && !rv32::build_aux_host(r.meta, so.alu, so.sha_ext, so.sha_cmp, so.big, rows, ec, pk->prep, &r.aux, &e, &r.big)) r.err = e;
                        if (time_stages) {
                            auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                            fprintf(stderr, "[prepare] shard %u: traced execution %.2f ms, wait for the fast pass %.2f ms, auxiliary traces %.2f ms\n", so.index, ms(tw0, tw1), ms(tw1, tw2),
                                    ms(tw2, std::chrono::steady_clock::now()));
                        }
                    }
                }
                std::lock_guard<std::mutex> lk(pl.mu);
                pl.ready[pos] = std::move(r);
                pl.cv.notify_all();
            }
        });

    dvt_job *j = new dvt_job();
    j->first = first; j->stride = stride;
    j->byte_words = (size_t)rv32::N_BYTE_OPS * 65536;
    j->prog_words = (size_t)1 << pk->prep.log_n[RV32_CHIP_PROGRAM];
    const MachineDesc *m = machine_rv32();
    int rc = DVT_OK;
    hipEvent_t ev = nullptr;
    (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    auto give_back = [&](rv32::CycleRec *b) {
        std::lock_guard<std::mutex> lk(pl.mu);
        pl.free_bufs.push_back(b);
        pl.cv.notify_all();
    };
    // upload of one ready shard: records on the copy stream (pinned source, overlaps the compute stream), the small
    // auxiliary traces on the compute stream
    auto upload = [&](ReadyShard &r) -> int {
        j->shards.emplace_back();
        ShardJob &s = j->shards.back();
        s.index = r.meta.index; s.n_recs = r.meta.n_recs; s.next_pc = r.meta.next_pc;
        for (int c = 0; c < m->n_chips; c++) { s.log_n[c] = r.aux.log_n[c]; s.present[c] = r.aux.present[c]; }
        HIP_TRY(p, p->eng.pool.alloc(&s.d_recs, s.n_recs * sizeof(rv32::CycleRec)));
        HIP_TRY(p, hipMemcpyAsync(s.d_recs, r.buf, s.n_recs * sizeof(rv32::CycleRec), hipMemcpyHostToDevice, p->copy_stream));
        HIP_TRY(p, hipEventRecord(ev, p->copy_stream));
        uint32_t *d_calls[rv32::N_CHIPS] = {};   // per precompile chip: [error word, padding to 16 bytes, the calls]
        // what each chip uploads: its calls / events when K0 of the chip runs on the GPU, else the rows the executor thread built
        auto events_of = [&](int c, const void **src, size_t *bytes) -> bool {
            if (!r.big.ev[c].empty()) { *src = r.big.ev[c].data(); *bytes = r.big.ev[c].size() * sizeof(rv32::BigOpEvent); return true; }
            if (c == RV32_CHIP_SHIFT && !r.big.shifts.empty()) { *src = r.big.shifts.data(); *bytes = r.big.shifts.size() * sizeof(rv32::AluEvent); return true; }
            if (c == RV32_CHIP_MEM_INIT && r.big.mem_rows && !r.big.mem_rows->empty()) { *src = r.big.mem_rows->data(); *bytes = r.big.mem_rows->size() * sizeof(rv32::MemInitRow); return true; }
            return false;
        };
        size_t stage_bytes = 0;
        for (int c = 0; c < m->n_chips; c++) {
            if (c == RV32_CHIP_CPU || !s.present[c]) continue;
            const void *src = nullptr;
            size_t bytes = 0;
            if (!events_of(c, &src, &bytes)) bytes = r.aux.main[c].size() * 4;
            stage_bytes += (bytes + 255) & ~(size_t)255;
        }
        if (stage_bytes > p->aux_pinned_bytes) {
            HIP_TRY(p, hipStreamSynchronize(p->eng.stream));
            if (p->aux_pinned) HIP_TRY(p, hipHostFree(p->aux_pinned));
            p->aux_pinned = nullptr; p->aux_pinned_bytes = 0;
            HIP_TRY(p, hipHostMalloc(&p->aux_pinned, stage_bytes + stage_bytes / 4));
            p->aux_pinned_bytes = stage_bytes + stage_bytes / 4;
        }
        size_t stage_at = 0;
        auto staged = [&](const void *src, size_t bytes) -> const void * {
            uint8_t *dst = p->aux_pinned + stage_at;
            memcpy(dst, src, bytes);
            stage_at += (bytes + 255) & ~(size_t)255;
            return dst;
        };
        size_t n_events[rv32::N_CHIPS] = {};
        for (int c = 0; c < m->n_chips; c++) {
            if (c == RV32_CHIP_CPU || !s.present[c]) continue;
            const void *src = nullptr;
            size_t bytes = 0;
            if (events_of(c, &src, &bytes)) {   // K0 of this chip on the GPU (after the byte counts are in): [error word, padding to 16 bytes, the events]
                const size_t words = (size_t)m->chips[c].main_w << s.log_n[c];
                HIP_TRY(p, p->eng.pool.alloc(&s.d_aux[c], words * 4));
                HIP_TRY(p, hipMemsetAsync(s.d_aux[c], 0, words * 4, p->eng.stream));
                HIP_TRY(p, p->eng.pool.alloc(&d_calls[c], 16 + bytes));
                HIP_TRY(p, hipMemsetAsync(d_calls[c], 0, 16, p->eng.stream));
                HIP_TRY(p, hipMemcpyAsync(d_calls[c] + 4, staged(src, bytes), bytes, hipMemcpyHostToDevice, p->eng.stream));
                n_events[c] = c == RV32_CHIP_SHIFT ? r.big.shifts.size() : c == RV32_CHIP_MEM_INIT ? r.big.mem_rows->size() : r.big.ev[c].size();
                continue;
            }
            size_t words = r.aux.main[c].size();
            HIP_TRY(p, p->eng.pool.alloc(&s.d_aux[c], words * 4));
            HIP_TRY(p, hipMemcpyAsync(s.d_aux[c], staged(r.aux.main[c].data(), words * 4), words * 4, hipMemcpyHostToDevice, p->eng.stream));
            // byte / program multiplicities stay plain integers until K0 has added the cpu rows' lookups
            if (c != RV32_CHIP_BYTE && c != RV32_CHIP_PROGRAM) HIP_TRY(p, launch_to_internal(p->eng.stream, s.d_aux[c], words));
        }
        for (int c = 0; c < m->n_chips; c++) {
            if (!d_calls[c]) continue;
            const size_t words = (size_t)m->chips[c].main_w << s.log_n[c];
            if (c == RV32_CHIP_SHIFT) {   // (these two write Montgomery words themselves)
                HIP_TRY(p, rv32::launch_k0_shift_rows(p->eng.stream, reinterpret_cast<const rv32::AluEvent *>(d_calls[c] + 4), n_events[c], s.d_aux[c], s.log_n[c], s.d_aux[RV32_CHIP_BYTE]));
            } else if (c == RV32_CHIP_MEM_INIT) {
                HIP_TRY(p, rv32::launch_k0_mem_init_rows(p->eng.stream, reinterpret_cast<const rv32::MemInitRow *>(d_calls[c] + 4), n_events[c], s.d_aux[c], s.log_n[c], s.d_aux[RV32_CHIP_BYTE]));
            } else {
                HIP_TRY(p, rv32::launch_k0_bigop_rows(p->eng.stream, c, reinterpret_cast<const rv32::BigOpEvent *>(d_calls[c] + 4), (uint32_t)n_events[c], s.index,
                                                      s.d_aux[c], s.log_n[c], s.d_aux[RV32_CHIP_BYTE], d_calls[c]));
                HIP_TRY(p, launch_to_internal(p->eng.stream, s.d_aux[c], words));
            }
        }
        HIP_TRY(p, hipStreamSynchronize(p->eng.stream));   // (the staging buffer is reused by the next shard; the error words below)
        for (int c = 0; c < m->n_chips; c++) {
            if (!d_calls[c]) continue;
            uint32_t row_err = 0;
            const hipError_t e = hipMemcpy(&row_err, d_calls[c], 4, hipMemcpyDeviceToHost);
            p->eng.pool.free(d_calls[c]);
            HIP_TRY(p, e);
            if (row_err) return fail(p, DVT_ERR_DEVICE, "K0 of chip %s: %s", m->chips[c].name, rv32::bigop_row_error_text(row_err));
        }
        for (auto x : r.aux.pubs) s.pubs.push_back(Fp::from_canonical(x));
        return DVT_OK;
    };
    {
        bool have_prev = false;
        rv32::CycleRec *prev_buf = nullptr;
        for (size_t pos = first; rc == DVT_OK; pos += stride) {
            ReadyShard r;
            bool got = false;
            {
                std::unique_lock<std::mutex> lk(pl.mu);
                const auto t0 = std::chrono::steady_clock::now();
                pl.cv.wait(lk, [&] { return pl.ready.count(pos) || (pl.fast_done && pos >= pl.n_total) || (pl.fast_done && !pl.error.empty()); });
                j->t_exec_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                auto it = pl.ready.find(pos);
                if (it != pl.ready.end()) { r = std::move(it->second); pl.ready.erase(it); got = true; }
            }
            if (got && !r.err.empty()) {
                rc = r.unsupported ? fail(p, DVT_ERR_UNSUPPORTED, "no chip for an instruction of the guest (%s)", r.err.c_str())
                                   : fail(p, DVT_ERR_GUEST, "guest trapped: %s", r.err.c_str());
                give_back(r.buf);
                break;
            }
            if (time_stages && got) fprintf(stderr, "[prepare] %.2f ms: shard at position %zu ready\n", since_begin(), pos);
            if (got) rc = upload(r);
            if (time_stages && got) fprintf(stderr, "[prepare] %.2f ms: uploaded\n", since_begin());
            // phase 1 of the previous shard runs while the copy engine brings this one in
            if (rc == DVT_OK && have_prev) {
                ShardJob &ps = j->shards[j->shards.size() - (got ? 2 : 1)];
                rc = shard_commit(p, pk, j, ps);
                give_back(prev_buf);
            }
            if (!got) { have_prev = false; break; }
            if (rc == DVT_OK) {
                // (no early return in this scope: the executor threads are joined and the job released below on every path)
                // the event is re-recorded per shard: wait for this copy before the buffer can be reused / the event re-armed.
                // (the host waits, so the compute stream needs no dependency on the copy stream)
                if (hipEventSynchronize(ev) != hipSuccess) rc = fail(p, DVT_ERR_DEVICE, "record upload failed");
            }
            have_prev = true;
            prev_buf = r.buf;
        }
        if (rc == DVT_OK && have_prev) {
            rc = shard_commit(p, pk, j, j->shards.back());
            give_back(prev_buf);
        }
        if (time_stages) fprintf(stderr, "[prepare] %.2f ms: phase 1 of the last shard done\n", since_begin());
    }
    {
        std::lock_guard<std::mutex> lk(pl.mu);
        pl.abort = true;   // (everything is done on the success path; on errors this stops the threads)
        pl.cv.notify_all();
    }
    fast.join();
    for (auto &w : workers) w.join();
    (void)hipEventDestroy(ev);
    if (time_stages) fprintf(stderr, "[prepare] %.2f ms: threads joined\n", since_begin());
    if (report) {
        report->cycles = pl.cycles;
        report->exit_code = pl.halted ? pl.exit_code : -1;
        report->halted = pl.halted;
        report->unprovable = pl.unsupported;
    }
    if (rc == DVT_OK) {
        if (!pl.error.empty()) rc = fail(p, DVT_ERR_GUEST, "guest trapped: %s", pl.error.c_str());
        else if (pl.unsupported) rc = fail(p, DVT_ERR_UNSUPPORTED, "no chip for %s", pl.unsupported_what.c_str());
        else if (pl.exit_code != 0) rc = fail(p, DVT_ERR_GUEST, "guest halted with exit code %d", pl.exit_code);
        else {
            uint32_t want[8];
            pv_digest_words(pl.public_values, want);
            bool ok = pl.committed_mask == 0xff;
            for (int k = 0; ok && k < 8; k++) ok = pl.committed[k] == want[k];
            if (!ok) rc = fail(p, DVT_ERR_GUEST, "guest did not COMMIT the SHA-256 digest of the %zu public-value bytes it wrote to fd 3", pl.public_values.size());
        }
    } else if (pl.unsupported && rc == DVT_ERR_GUEST) {
        rc = fail(p, DVT_ERR_UNSUPPORTED, "no chip for %s", pl.unsupported_what.c_str());
    }
    if (rc) { job_release(p, j); return rc; }
    j->exit_code = pl.exit_code;
    j->cycles = pl.cycles;
    j->public_values = std::move(pl.public_values);
    j->n_total = pl.n_total;
    *out = j;
    return DVT_OK;
}

static std::vector<uint32_t> assemble_container(const dvt_job *j, const std::vector<std::vector<uint32_t>> &shards) {
    WordWriter w;
    w.u32(CORE_PROOF_MAGIC);
    w.u32((uint32_t)shards.size());
    w.u32((uint32_t)j->exit_code);
    const auto &pv = j->public_values;
    w.u32((uint32_t)pv.size());
    for (size_t i = 0; i < pv.size(); i += 4) {
        uint32_t v = 0;
        for (size_t k = 0; k < 4 && i + k < pv.size(); k++) v |= (uint32_t)pv[i + k] << (8 * k);
        w.u32(v);
    }
    for (auto &s : shards) {
        w.u32((uint32_t)s.size());
        w.w.insert(w.w.end(), s.begin(), s.end());
    }
    return w.w;
}

// both phases on one GPU (no lock); the job must hold every shard of the execution
static int job_prove(dvt_prover *p, const dvt_pk *pk, dvt_job *j, uint8_t **proof, size_t *proof_len) {
    HIP_TRY(p, hipSetDevice(p->eng.device));
    const size_t n = j->shards.size();
    if (n != j->n_total) return fail(p, DVT_ERR_INPUT, "this job holds %zu of the execution's %zu shards: prove them shard by shard", n, j->n_total);
    std::vector<uint32_t> headers(n * HEADER_WORDS);
    for (size_t i = 0; i < n; i++) {
        if (!j->shards[i].header_valid) {
            int rc = shard_commit(p, pk, j, j->shards[i]);
            if (rc) return rc;
        }
        memcpy(headers.data() + i * HEADER_WORDS, j->shards[i].header, sizeof(uint32_t) * HEADER_WORDS);
    }
    PermChallenges gc = global_challenges(pk->key.vk, headers.data(), n);
    std::vector<std::vector<uint32_t>> shards(n);
    for (size_t i = 0; i < n; i++) {
        int rc = shard_prove(p, pk, j, j->shards[i], gc, &shards[i]);
        if (rc) return rc;
    }
    (void)hipStreamSynchronize(p->eng.stream);
    if (!proof) return DVT_OK;  // timing runs may discard the bytes
    *proof = copy_out(assemble_container(j, shards), proof_len);
    if (!*proof) return fail(p, DVT_ERR_DEVICE, "out of host memory");
    return DVT_OK;
}

extern "C" {

int dvt_setup(dvt_prover *p, const uint8_t *elf, size_t elf_len, dvt_pk **pk_out, uint8_t **vk, size_t *vk_len) {
    if (!p || !elf || !pk_out) return fail(p, DVT_ERR_INPUT, "null argument");
    std::lock_guard<std::mutex> lk(p->mu);
    dvt_pk *pk = new dvt_pk();
    std::string err;
    if (!rv32::load_elf(elf, elf_len, &pk->prog, &err)) { delete pk; return fail(p, DVT_ERR_INPUT, "ELF: %s", err.c_str()); }
    rv32::build_prep(pk->prog, &pk->prep);
    pk->is_rv32 = true;
    std::vector<ChipRef> refs;
    std::vector<std::vector<uint32_t>> host;
    for (int c : {RV32_CHIP_PROGRAM, RV32_CHIP_BYTE, RV32_CHIP_MEM_IMAGE}) {
        refs.push_back({c, pk->prep.log_n[c]});
        host.push_back(pk->prep.prep[c]);
    }
    if (!p->eng.setup(machine_rv32(), refs, host, &pk->key)) {
        p->eng.free_key(&pk->key);
        delete pk;
        return fail(p, DVT_ERR_DEVICE, "%s", p->eng.err.c_str());
    }
    pk->key.vk.extra = {pk->prog.entry};
    {
        std::vector<uint32_t> rowmap = rv32::program_row_map(pk->prog);
        size_t ib = pk->prog.instrs.size() * sizeof(rv32::Instr);
        if (hipMalloc(&pk->d_instrs, ib) != hipSuccess || hipMalloc(&pk->d_prog_row, rowmap.size() * 4) != hipSuccess ||
            hipMemcpy(pk->d_instrs, pk->prog.instrs.data(), ib, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(pk->d_prog_row, rowmap.data(), rowmap.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            p->eng.free_key(&pk->key);
            if (pk->d_instrs) (void)hipFree(pk->d_instrs);
            if (pk->d_prog_row) (void)hipFree(pk->d_prog_row);
            delete pk;
            return fail(p, DVT_ERR_DEVICE, "uploading the program table failed");
        }
    }
    if (vk && vk_len) {
        *vk = copy_out(vk_words(pk->key.vk), vk_len);
        if (!*vk) { p->eng.free_key(&pk->key); delete pk; return fail(p, DVT_ERR_DEVICE, "out of host memory"); }
    }
    *pk_out = pk;
    return DVT_OK;
}

int dvt_execute_io(const uint8_t *elf, size_t elf_len, const dvt_buf *stdin_bufs, size_t nbuf, uint64_t max_cycles,
                   uint8_t **public_values, size_t *pv_len, uint8_t **stdout_bytes, size_t *stdout_len, dvt_report *report, char **err_text) {
    if (err_text) *err_text = nullptr;
    if (public_values) *public_values = nullptr;
    if (stdout_bytes) *stdout_bytes = nullptr;
    if (!elf || (nbuf && !stdin_bufs)) return DVT_ERR_INPUT;
    rv32::Program prog;
    std::string err;
    if (!rv32::load_elf(elf, elf_len, &prog, &err)) {
        if (err_text) *err_text = strdup(("ELF: " + err).c_str());
        return DVT_ERR_INPUT;
    }
    rv32::ExecResult res;
    rv32::execute(prog, collect_stdin(stdin_bufs, nbuf), false, max_cycles ? max_cycles : ~0ull, 21, &res);
    if (report) {
        report->cycles = res.cycles;
        report->exit_code = res.halted ? res.exit_code : -1;
        report->halted = res.halted;
        report->unprovable = res.unsupported;
    }
    if (public_values) *public_values = dup_bytes(res.public_values, pv_len);
    if (stdout_bytes) *stdout_bytes = dup_bytes(res.stdout_bytes, stdout_len);
    if (!res.error.empty()) {
        if (err_text) *err_text = strdup(res.error.c_str());
        return DVT_ERR_GUEST;
    }
    if (res.exit_code != 0) {
        if (err_text) *err_text = strdup("guest halted with a non-zero exit code");
        return DVT_ERR_GUEST;
    }
    return DVT_OK;
}
int dvt_execute(const uint8_t *elf, size_t elf_len, const dvt_buf *stdin_bufs, size_t nbuf, uint64_t max_cycles,
                uint8_t **public_values, size_t *pv_len, dvt_report *report, char **err_text) {
    return dvt_execute_io(elf, elf_len, stdin_bufs, nbuf, max_cycles, public_values, pv_len, nullptr, nullptr, report, err_text);
}

int dvt_rv32_prepare(dvt_prover *p, const dvt_pk *pk, const dvt_buf *stdin_bufs, size_t nbuf, dvt_job **job, dvt_report *report) {
    return dvt_rv32_prepare_part(p, pk, stdin_bufs, nbuf, 0, 1, job, report);
}
int dvt_rv32_prepare_part(dvt_prover *p, const dvt_pk *pk, const dvt_buf *stdin_bufs, size_t nbuf, size_t first, size_t stride, dvt_job **job,
                          dvt_report *report) {
    if (!p || !pk || !job || (nbuf && !stdin_bufs)) return fail(p, DVT_ERR_INPUT, "null argument");
    if (!pk->is_rv32) return fail(p, DVT_ERR_INPUT, "proving key was not made by dvt_setup");
    std::lock_guard<std::mutex> lk(p->mu);
    return job_prepare(p, pk, stdin_bufs, nbuf, first, stride, job, report);
}
int dvt_rv32_prove_job(dvt_prover *p, const dvt_pk *pk, dvt_job *job, uint8_t **proof, size_t *proof_len) {
    if (!p || !pk || !job || (proof && !proof_len)) return fail(p, DVT_ERR_INPUT, "null argument");
    std::lock_guard<std::mutex> lk(p->mu);
    return job_prove(p, pk, job, proof, proof_len);
}
void dvt_job_free(dvt_prover *p, dvt_job *job) {
    if (!p || !job) return;
    std::lock_guard<std::mutex> lk(p->mu);
    (void)hipSetDevice(p->eng.device);
    job_release(p, job);
}
size_t dvt_rv32_job_shards(const dvt_job *job) { return job ? job->n_total : 0; }
double dvt_rv32_job_exec_wait_seconds(const dvt_job *job) { return job ? job->t_exec_wait : 0.0; }

int dvt_rv32_commit_shard(dvt_prover *p, const dvt_pk *pk, dvt_job *job, size_t shard, uint32_t *header) {
    if (!p || !pk || !job || !header) return fail(p, DVT_ERR_INPUT, "bad argument");
    ShardJob *s = job->at(shard);
    if (!s) return fail(p, DVT_ERR_INPUT, "shard %zu is not held by this job", shard);
    std::lock_guard<std::mutex> lk(p->mu);
    HIP_TRY(p, hipSetDevice(p->eng.device));
    if (!s->header_valid) {   // (the prepare pipeline already ran phase 1; a second proof of the same job runs it again)
        int rc = shard_commit(p, pk, job, *s);
        if (rc) return rc;
    }
    memcpy(header, s->header, sizeof(uint32_t) * HEADER_WORDS);
    return DVT_OK;
}
uint32_t dvt_rv32_header_words(void) { return HEADER_WORDS; }
int dvt_rv32_challenges(const uint8_t *vk, size_t vk_len, const uint32_t *headers, size_t n, uint32_t out[8]) {
    VerifyingKey key;
    if (!vk || !headers || !out || !n || !vk_parse(vk, vk_len, &key)) return DVT_ERR_INPUT;
    for (size_t i = 0; i < n * HEADER_WORDS; i++)
        if (headers[i] >= P && (i % HEADER_WORDS) < 8) return DVT_ERR_INPUT;
    PermChallenges c = global_challenges(key, headers, n);
    for (int k = 0; k < 4; k++) { out[k] = c.alpha.c[k].canonical(); out[4 + k] = c.beta.c[k].canonical(); }
    return DVT_OK;
}
int dvt_rv32_prove_shard(dvt_prover *p, const dvt_pk *pk, dvt_job *job, size_t shard, const uint32_t challenges[8], uint8_t **proof,
                         size_t *proof_len) {
    if (!p || !pk || !job || !challenges || (proof && !proof_len)) return fail(p, DVT_ERR_INPUT, "bad argument");
    ShardJob *s = job->at(shard);
    if (!s) return fail(p, DVT_ERR_INPUT, "shard %zu is not held by this job", shard);
    std::lock_guard<std::mutex> lk(p->mu);
    HIP_TRY(p, hipSetDevice(p->eng.device));
    PermChallenges gc;
    for (int k = 0; k < 4; k++) {
        if (challenges[k] >= P || challenges[4 + k] >= P) return fail(p, DVT_ERR_INPUT, "challenge not canonical");
        gc.alpha.c[k] = Fp::from_canonical(challenges[k]);
        gc.beta.c[k] = Fp::from_canonical(challenges[4 + k]);
    }
    std::vector<uint32_t> words;
    int rc = shard_prove(p, pk, job, *s, gc, &words);
    if (rc || !proof) return rc;
    *proof = copy_out(words, proof_len);
    return *proof ? DVT_OK : fail(p, DVT_ERR_DEVICE, "out of host memory");
}
int dvt_rv32_assemble(const dvt_job *job, const uint8_t *const *shard_proofs, const size_t *lens, size_t n, uint8_t **proof, size_t *proof_len) {
    if (!job || !shard_proofs || !lens || !proof || !proof_len || n != job->n_total) return DVT_ERR_INPUT;
    std::vector<std::vector<uint32_t>> shards(n);
    for (size_t i = 0; i < n; i++) {
        if (lens[i] % 4 || !shard_proofs[i]) return DVT_ERR_INPUT;
        shards[i].resize(lens[i] / 4);
        memcpy(shards[i].data(), shard_proofs[i], lens[i]);
    }
    *proof = copy_out(assemble_container(job, shards), proof_len);
    return *proof ? DVT_OK : DVT_ERR_DEVICE;
}

int dvt_prove_core(dvt_prover *p, const dvt_pk *pk, const dvt_buf *stdin_bufs, size_t nbuf, uint8_t **proof, size_t *proof_len,
                   dvt_report *report) {
    if (!p || !pk || !proof || !proof_len || (nbuf && !stdin_bufs)) return fail(p, DVT_ERR_INPUT, "null argument");
    if (!pk->is_rv32) return fail(p, DVT_ERR_INPUT, "proving key was not made by dvt_setup");
    std::lock_guard<std::mutex> lk(p->mu);
    dvt_job *j = nullptr;
    int rc = job_prepare(p, pk, stdin_bufs, nbuf, 0, 1, &j, report);
    if (rc) return rc;
    rc = job_prove(p, pk, j, proof, proof_len);
    job_release(p, j);
    return rc;
}

int dvt_verify(const uint8_t *vk, size_t vk_len, const uint8_t *proof, size_t proof_len, uint32_t fri_queries, uint32_t pow_bits,
               int32_t *exit_code, uint8_t **public_values, size_t *pv_len, char **reason) {
    if (reason) *reason = nullptr;
    if (public_values) *public_values = nullptr;
    auto reject = [&](int code, const std::string &why) {
        if (reason) *reason = strdup(why.c_str());
        return code;
    };
    if (!vk || !proof) return reject(DVT_ERR_INPUT, "null argument");
    VerifyingKey key;
    if (!vk_parse(vk, vk_len, &key) || key.machine != machine_rv32() || key.extra.size() != 1) return reject(DVT_ERR_INPUT, "malformed verifying key");
    if (proof_len % 4) return reject(DVT_ERR_REJECTED, "proof length is not a multiple of 4");
    std::vector<uint32_t> words(proof_len / 4);
    memcpy(words.data(), proof, proof_len);
    // the caller states the FRI parameters it accepts; refuse settings that verify nothing
    // (blow-up 2: one bit of security per query, plus the proof-of-work bits)
    if (fri_queries == 0 || fri_queries > 1024 || pow_bits > 30) return reject(DVT_ERR_INPUT, "fri_queries must be 1..1024 and pow_bits <= 30");
    StarkConfig cfg;
    cfg.num_queries = fri_queries;
    cfg.pow_bits = pow_bits;
    try {
        WordReader r(words.data(), words.size());
        if (r.u32() != CORE_PROOF_MAGIC) return reject(DVT_ERR_REJECTED, "bad container magic");
        const uint32_t nshards = r.len(1 << 16);
        if (nshards == 0) return reject(DVT_ERR_REJECTED, "no shards");
        const uint32_t ec = r.u32(), pvl = r.len(1 << 24);
        // the AIR pins exit codes below 2^24 (the code itself, not a residue): a container word ec + p would pass the
        // comparison mod p below and be reported to the caller as is
        if (ec >> 24) return reject(DVT_ERR_REJECTED, "exit code out of range");
        std::vector<uint8_t> pv(pvl);
        for (uint32_t i = 0; i < pvl; i += 4) {
            uint32_t v = r.u32();
            for (uint32_t k = 0; k < 4; k++) {
                if (i + k < pvl) pv[i + k] = (uint8_t)(v >> (8 * k));
                else if ((v >> (8 * k)) & 0xff) return reject(DVT_ERR_REJECTED, "non-zero padding after the public values");   // (one encoding per proof)
            }
        }
        std::vector<ShardProof> sps(nshards);
        for (uint32_t i = 0; i < nshards; i++) {
            uint32_t nw = r.len(1u << 30);
            if ((size_t)(r.end - r.p) < nw) return reject(DVT_ERR_REJECTED, "container truncated");
            WordReader sr(r.p, nw);
            sps[i] = read_shard_proof(sr);
            if (sr.p != sr.end) return reject(DVT_ERR_REJECTED, "trailing words after a shard proof");
            r.p += nw;
        }
        if (r.p != r.end) return reject(DVT_ERR_REJECTED, "trailing bytes after proof");
        // shard chaining through the public values
        std::vector<uint32_t> headers(nshards * HEADER_WORDS);
        for (uint32_t i = 0; i < nshards; i++) {
            const ShardProof &sp = sps[i];
            if (sp.public_values.size() != N_PUB) return reject(DVT_ERR_REJECTED, "wrong number of public values");
            uint32_t pubv[N_PUB];
            for (uint32_t k = 0; k < N_PUB; k++) pubv[k] = sp.public_values[k].canonical();
            const bool last = i + 1 == nshards;
            if (pubv[3] != i + 1) return reject(DVT_ERR_REJECTED, "shard index out of sequence");
            if (pubv[4] != (last ? 1u : 0u)) return reject(DVT_ERR_REJECTED, "is_last flag does not match the shard's position");
            if (i == 0 && pubv[0] != key.extra[0] % P) return reject(DVT_ERR_REJECTED, "first shard does not start at the entry point");
            if (i > 0 && pubv[0] != sps[i - 1].public_values[1].canonical()) return reject(DVT_ERR_REJECTED, "shards do not chain (pc)");
            // only a HALT row has next_pc = HALT_PC (tools/airgen/rv32.py): control flow that merely reaches address 0 does not count
            if (last && pubv[1] != rv32::HALT_PC) return reject(DVT_ERR_REJECTED, "execution did not halt");
            if (!last && pubv[1] == rv32::HALT_PC) return reject(DVT_ERR_REJECTED, "halt before the last shard");
            if (last && pubv[2] != ec % P) return reject(DVT_ERR_REJECTED, "exit code mismatch");
            // chip set: program, byte, cpu, mem_image always; mem_init in the last shard only; shift / muldiv when the shard uses them
            bool have[rv32::N_CHIPS] = {};
            for (auto &c : sp.chips) {
                if (c.chip_id >= (uint32_t)rv32::N_CHIPS) return reject(DVT_ERR_REJECTED, "chip id out of range");
                have[c.chip_id] = true;
            }
            for (int c : {RV32_CHIP_PROGRAM, RV32_CHIP_BYTE, RV32_CHIP_CPU, RV32_CHIP_MEM_IMAGE})
                if (!have[c]) return reject(DVT_ERR_REJECTED, "a mandatory chip is missing from a shard");
            if (have[RV32_CHIP_MEM_INIT] != last) return reject(DVT_ERR_REJECTED, "mem_init must be part of exactly the last shard");
            for (int k = 0; k < 8; k++) headers[i * HEADER_WORDS + k] = sp.main_root.d[k].canonical();
            for (uint32_t k = 0; k < N_PUB; k++) headers[i * HEADER_WORDS + 8 + k] = pubv[k];
        }
        PermChallenges gc = global_challenges(key, headers.data(), nshards);
        Fp4 total = Fp4::zero();
        for (uint32_t i = 0; i < nshards; i++) {
            Fp4 t;
            std::string why = verify_shard(key, sps[i], cfg, &gc, &t);
            if (!why.empty()) return reject(DVT_ERR_REJECTED, "shard " + std::to_string(i + 1) + ": " + why);
            total += t;
        }
        // The receiving side of the COMMIT rows' sys-bus tuples is supplied here, from the claimed bytes: an SP1 guest commits
        // the eight words of SHA-256(public-value bytes) with COMMIT(k, word k); the cpu chip sends
        // (t0 bytes = 0x10 0 0 0, a0 bytes = k 0 0 0, a1 bytes = the bytes of digest word k, 0, 0), tuple k contributes
        // 1 / (alpha + bus + beta 0x10 + beta^5 k + beta^9 b0 + ... + beta^12 b3).
        {
            uint8_t dg[32];
            sha256(pv.data(), pv.size(), dg);
            Fp4 bp[13];
            bp[1] = gc.beta;
            for (int k = 2; k < 13; k++) bp[k] = bp[k - 1] * gc.beta;
            Fp4 expect = Fp4::zero();
            for (uint32_t k = 0; k < 8; k++) {
                Fp4 d = gc.alpha + Fp::from_canonical(PV_BUS) + bp[1] * Fp::from_canonical(rv32::SYS_COMMIT) + bp[5] * Fp::from_canonical(k);
                for (int b = 0; b < 4; b++) d += bp[9 + b] * Fp::from_canonical(dg[4 * k + b]);
                expect += inv(d);
            }
            if (total != expect) return reject(DVT_ERR_REJECTED, "LogUp cumulative sums do not cancel across the shards (memory bus or public-values digest)");
        }
        if (exit_code) *exit_code = (int32_t)ec;
        if (public_values) *public_values = dup_bytes(pv, pv_len);
    } catch (const std::exception &e) { return reject(DVT_ERR_REJECTED, e.what()); }
    return DVT_OK;
}

static std::vector<uint32_t> trace_blob(const rv32::HostTraces &T, const rv32::HostPrep *prep) {
    const MachineDesc *m = machine_rv32();
    std::vector<uint32_t> w;
    uint32_t present = 0;
    for (int c = 0; c < m->n_chips; c++) present += T.present[c];
    w.push_back(present);
    for (int c = 0; c < m->n_chips; c++)
        if (T.present[c]) { w.push_back(c); w.push_back(T.log_n[c]); w.push_back(m->chips[c].main_w); w.push_back(prep ? m->chips[c].prep_w : 0); }
    w.push_back((uint32_t)T.pubs.size());
    w.insert(w.end(), T.pubs.begin(), T.pubs.end());
    for (int c = 0; c < m->n_chips; c++) {
        if (!T.present[c]) continue;
        w.insert(w.end(), T.main[c].begin(), T.main[c].end());
        if (prep) w.insert(w.end(), prep->prep[c].begin(), prep->prep[c].end());
    }
    return w;
}

int dvt_rv32_debug_traces(const uint8_t *elf, size_t elf_len, const dvt_buf *stdin_bufs, size_t nbuf, uint32_t log_shard, uint32_t shard,
                          uint32_t *n_shards, uint32_t **blob, size_t *blob_words, char **err_text) {
    if (err_text) *err_text = nullptr;
    if (!elf || !blob || !blob_words) return DVT_ERR_INPUT;
    auto bad = [&](int code, const std::string &m) { if (err_text) *err_text = strdup(m.c_str()); return code; };
    rv32::Program prog;
    std::string err;
    if (!rv32::load_elf(elf, elf_len, &prog, &err)) return bad(DVT_ERR_INPUT, "ELF: " + err);
    rv32::HostPrep prep;
    rv32::build_prep(prog, &prep);
    rv32::ExecResult res;
    rv32::execute(prog, collect_stdin(stdin_bufs, nbuf), true, 1ull << 32, log_shard ? log_shard : 21, &res);
    if (!res.error.empty()) return bad(DVT_ERR_GUEST, res.error);
    if (n_shards) *n_shards = (uint32_t)res.shards.size();
    rv32::HostTraces T;
    if (!rv32::build_traces_host(prog, res, shard, prep, &T, &err)) return bad(DVT_ERR_UNSUPPORTED, err);
    std::vector<uint32_t> w = trace_blob(T, &prep);
    *blob = (uint32_t *)malloc(w.size() * 4);
    if (!*blob) return bad(DVT_ERR_DEVICE, "out of host memory");
    memcpy(*blob, w.data(), w.size() * 4);
    *blob_words = w.size();
    return DVT_OK;
}

// test hook: run K0 on shard `shard` of a prepared job and return the device-generated main traces (canonical),
// same blob layout as dvt_rv32_debug_traces but without preprocessed columns (prep_width = 0)
int dvt_rv32_debug_device_traces(dvt_prover *p, const dvt_pk *pk, dvt_job *j, size_t shard, uint32_t **blob, size_t *blob_words) {
    if (!p || !pk || !j || !blob || !blob_words || !j->at(shard)) return fail(p, DVT_ERR_INPUT, "bad argument");
    std::lock_guard<std::mutex> lk(p->mu);
    HIP_TRY(p, hipSetDevice(p->eng.device));
    std::vector<ChipTrace> traces;
    ShardJob &sj = *j->at(shard);
    int rc = shard_traces(p, pk, j, sj, &traces, false);
    if (rc) return rc;
    HIP_TRY(p, hipStreamSynchronize(p->eng.stream));
    const MachineDesc *m = machine_rv32();
    rv32::HostTraces T;
    for (int c = 0; c < rv32::N_CHIPS; c++) T.present[c] = false;
    for (auto &t : traces) {
        size_t words = (size_t)m->chips[t.chip_id].main_w << t.log_n;
        T.present[t.chip_id] = true;
        T.log_n[t.chip_id] = t.log_n;
        T.main[t.chip_id].resize(words);
        HIP_TRY(p, hipMemcpy(T.main[t.chip_id].data(), t.d_main, words * 4, hipMemcpyDeviceToHost));
        for (auto &x : T.main[t.chip_id]) x = Fp::raw(x).canonical();
    }
    for (auto x : sj.pubs) T.pubs.push_back(x.canonical());
    std::vector<uint32_t> w = trace_blob(T, nullptr);
    *blob = (uint32_t *)malloc(w.size() * 4);
    if (!*blob) return fail(p, DVT_ERR_DEVICE, "out of host memory");
    memcpy(*blob, w.data(), w.size() * 4);
    *blob_words = w.size();
    return DVT_OK;
}

// measurement hook (host only): guest cycles per second of the executor alone, fast mode (trace = 0: what the
// sequential pass of the prove pipeline runs) or trace mode (one 48-byte record per cycle into a reused buffer)
double dvt_debug_exec_rate(const uint8_t *elf, size_t elf_len, const dvt_buf *stdin_bufs, size_t nbuf, uint32_t log_shard, int trace) {
    rv32::Program prog;
    std::string err;
    if (!elf || !rv32::load_elf(elf, elf_len, &prog, &err)) return 0.0;
    const std::vector<std::vector<uint8_t>> inputs = collect_stdin(stdin_bufs, nbuf);
    std::vector<rv32::CycleRec> buf;
    if (trace) buf.resize((size_t)1 << log_shard);
    rv32::ShardOut out;
    out.recs = buf.data();
    const auto t0 = std::chrono::steady_clock::now();
    rv32::Vm vm(prog, &inputs, log_shard);
    for (;;) {
        vm.run_shard(trace != 0, &out, ~0ull);
        if (vm.halted || !vm.error.empty() || !vm.next_shard()) break;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return vm.halted ? (double)vm.cycles / dt : 0.0;
}

// test hook (host only): the FP64 formulation of Poseidon2 that the hashing kernels run, evaluated on the host
// (IEEE doubles + fma, the same arithmetic) against the integer permutation on n pseudo-random and edge-case states,
// through the same Montgomery conversions the kernels use.  Returns the number of differing words.
uint64_t dvt_debug_p2_f64_selfcheck(uint32_t n, uint32_t seed) {
    uint64_t bad = 0, x = 0x9e3779b97f4a7c15ull ^ seed;
    for (uint32_t t = 0; t < n; t++) {
        Fp a[16];
        double b[16];
        for (int i = 0; i < 16; i++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            uint32_t v = (uint32_t)(x % P);
            if (t < 4) v = t == 0 ? 0 : t == 1 ? P - 1 : t == 2 ? (P - 1) / 2 + (i & 1) : (i ? P - i : 1);
            a[i] = Fp::from_canonical(v);
            b[i] = p2f::from_mont(a[i].v);
        }
        for (int rep = 0; rep < 3; rep++) {  // chained: the second and third calls start from lazy (signed) outputs
            p2_permute(a);
            p2f::permute(b);
            for (int i = 0; i < 16; i++) bad += (a[i].v != p2f::to_mont(b[i])) + (a[i].canonical() != p2f::to_canonical(b[i]));
        }
    }
    return bad;
}

}  // extern "C"
