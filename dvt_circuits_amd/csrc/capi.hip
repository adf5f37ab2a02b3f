// C-ABI layer (include/dvt_prover.h) over the gfx950 engine.  There is no CPU
// fallback anywhere in this file: without a HIP device every entry point that
// computes returns DVT_ERR_DEVICE.  (dvt_machine_verify is host-only by nature.)
#include "../../include/dvt_prover.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "engine.h"
#include "poseidon2_f64.cuh"
#include "rv32.h"

using namespace dvt;

struct dvt_prover {
    Engine eng;
    StarkConfig cfg;
    uint32_t log_shard = 21;          // cycles per shard = 2^log_shard (SP1's default shard size)
    uint64_t max_cycles = 1ull << 36;
    bool keep_phase1 = true;          // keep K0 output, main LDEs and tree of phase 1 in HBM for phase 2 ("keep_phase1": 0 recomputes)
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

uint32_t dvt_abi_version(void) { return 2; }

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
    if (p->log_shard < 4 || p->log_shard > 22) { delete p; return fail(nullptr, DVT_ERR_INPUT, "log_shard_size must be 4..22"); }
    if (p->cfg.num_queries == 0 || p->cfg.num_queries > 1024 || p->cfg.pow_bits > 30) {
        delete p;
        return fail(nullptr, DVT_ERR_INPUT, "fri_queries must be 1..1024 and pow_bits <= 30");
    }
    e = p->eng.init(dev);
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
constexpr uint32_t CORE_PROOF_MAGIC = 0x32435644u;  // "DVC2"
constexpr uint32_t N_PUB = rv32::N_PUBLIC;           // start_pc, next_pc, exit_code, shard, is_last, pv_start, pv_end
constexpr uint32_t HEADER_WORDS = 8 + N_PUB;         // per-shard commitment header: main root + public values (canonical)

std::vector<std::vector<uint8_t>> collect_stdin(const dvt_buf *bufs, size_t n) {
    std::vector<std::vector<uint8_t>> v(n);
    for (size_t i = 0; i < n; i++)
        if (bufs[i].len) v[i].assign(bufs[i].data, bufs[i].data + bufs[i].len);
    return v;
}
void fill_report(dvt_report *rep, const rv32::ExecResult &r) {
    if (!rep) return;
    rep->cycles = r.cycles;
    rep->exit_code = r.halted ? r.exit_code : -1;
    rep->halted = r.halted;
    rep->unprovable = r.unsupported;
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
}  // namespace

struct ShardJob {
    uint32_t index = 0;
    size_t n_recs = 0;
    uint32_t pv_end = 0;
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
};
// one prepared execution: executor output cut into shards, resident in HBM, ready for K0..K9
struct dvt_job {
    rv32::ExecResult res;  // per-cycle records are released after the upload
    std::vector<ShardJob> shards;
    uint32_t *d_cpu = nullptr, *d_byte = nullptr, *d_prog = nullptr;  // working buffers (largest shard)
    size_t byte_words = 0, prog_words = 0;
};

static void job_release(dvt_job *j) {
    if (!j) return;
    for (auto &s : j->shards) {
        if (s.d_recs) (void)hipFree(s.d_recs);
        for (auto &d : s.d_aux) if (d) (void)hipFree(d);
        for (uint32_t *d : {s.d_cpu, s.d_byte, s.d_prog}) if (d) (void)hipFree(d);
        s.cache.release();
    }
    if (j->d_cpu) (void)hipFree(j->d_cpu);
    if (j->d_byte) (void)hipFree(j->d_byte);
    if (j->d_prog) (void)hipFree(j->d_prog);
    delete j;
}

// execute + auxiliary traces + upload (no lock: callers hold p->mu)
static int job_prepare(dvt_prover *p, const dvt_pk *pk, const dvt_buf *stdin_bufs, size_t nbuf, dvt_job **out, dvt_report *report) {
    HIP_TRY(p, hipSetDevice(p->eng.device));
    dvt_job *j = new dvt_job();
    rv32::execute(pk->prog, collect_stdin(stdin_bufs, nbuf), true, p->max_cycles, p->log_shard, &j->res);
    fill_report(report, j->res);
    int rc = DVT_OK;
    if (!j->res.error.empty()) rc = fail(p, DVT_ERR_GUEST, "guest trapped: %s", j->res.error.c_str());
    else if (j->res.exit_code != 0) rc = fail(p, DVT_ERR_GUEST, "guest halted with exit code %d", j->res.exit_code);
    else if (j->res.unsupported) rc = fail(p, DVT_ERR_UNSUPPORTED, "no chip for %s", j->res.unsupported_what.c_str());
    if (rc) { job_release(j); return rc; }
    const MachineDesc *m = machine_rv32();
    uint32_t max_log_cpu = 0;
    bool ok = true;
    for (size_t si = 0; ok && si < j->res.shards.size(); si++) {
        rv32::HostTraces T;
        std::string err;
        if (!rv32::build_aux_host(pk->prog, j->res, si, pk->prep, &T, &err)) { job_release(j); return fail(p, DVT_ERR_UNSUPPORTED, "%s", err.c_str()); }
        ShardJob s;
        auto &recs = j->res.shards[si].recs;
        s.index = j->res.shards[si].index;
        s.n_recs = recs.size();
        s.pv_end = j->res.shards[si].pv_end;
        for (int c = 0; c < m->n_chips; c++) { s.log_n[c] = T.log_n[c]; s.present[c] = T.present[c]; }
        max_log_cpu = std::max(max_log_cpu, s.log_n[RV32_CHIP_CPU]);
        ok = hipMalloc(&s.d_recs, s.n_recs * sizeof(rv32::CycleRec)) == hipSuccess &&
             hipMemcpy(s.d_recs, recs.data(), s.n_recs * sizeof(rv32::CycleRec), hipMemcpyHostToDevice) == hipSuccess;
        for (int c = 0; ok && c < m->n_chips; c++) {
            if (c == RV32_CHIP_CPU || !s.present[c]) continue;
            size_t words = T.main[c].size();
            ok = hipMalloc(&s.d_aux[c], words * 4) == hipSuccess && hipMemcpy(s.d_aux[c], T.main[c].data(), words * 4, hipMemcpyHostToDevice) == hipSuccess;
            // byte / program multiplicities stay plain integers until K0 has added the cpu rows' lookups
            if (ok && c != RV32_CHIP_BYTE && c != RV32_CHIP_PROGRAM) ok = launch_to_internal(p->eng.stream, s.d_aux[c], words) == hipSuccess;
        }
        for (auto x : T.pubs) s.pubs.push_back(Fp::from_canonical(x));
        std::vector<rv32::CycleRec>().swap(recs);
        j->shards.push_back(std::move(s));
    }
    j->byte_words = (size_t)rv32::N_BYTE_OPS * 65536;
    j->prog_words = (size_t)1 << pk->prep.log_n[RV32_CHIP_PROGRAM];
    if (ok) ok = hipMalloc(&j->d_cpu, ((size_t)RV32_CPU_MAIN_W << max_log_cpu) * 4) == hipSuccess &&
                 hipMalloc(&j->d_byte, j->byte_words * 4) == hipSuccess && hipMalloc(&j->d_prog, j->prog_words * 4) == hipSuccess;
    if (ok) ok = hipStreamSynchronize(p->eng.stream) == hipSuccess;
    if (!ok) { job_release(j); return fail(p, DVT_ERR_DEVICE, "uploading the shards to the device failed"); }
    *out = j;
    return DVT_OK;
}

// K0 of shard i (into the shard's own buffers when it has them, else the job's working buffers); fills the chip
// trace list of that shard.  `reuse`: phase 2 takes the traces phase 1 left behind instead of generating them again.
static int shard_traces(dvt_prover *p, const dvt_pk *pk, dvt_job *j, size_t i, std::vector<ChipTrace> *traces, bool reuse) {
    hipStream_t st = p->eng.stream;
    ShardJob &s = j->shards[i];
    const MachineDesc *m = machine_rv32();
    uint32_t *cpu = s.d_cpu ? s.d_cpu : j->d_cpu, *byte = s.d_cpu ? s.d_byte : j->d_byte, *prog = s.d_cpu ? s.d_prog : j->d_prog;
    if (!(reuse && s.d_cpu && s.traces_valid)) {
        bool ok = hipMemcpyAsync(byte, s.d_aux[RV32_CHIP_BYTE], j->byte_words * 4, hipMemcpyDeviceToDevice, st) == hipSuccess &&
                  hipMemcpyAsync(prog, s.d_aux[RV32_CHIP_PROGRAM], j->prog_words * 4, hipMemcpyDeviceToDevice, st) == hipSuccess &&
                  rv32::launch_k0_cpu_rows(st, s.d_recs, s.n_recs, s.index, s.pv_end, pk->d_instrs, pk->d_prog_row, cpu, s.log_n[RV32_CHIP_CPU], byte, prog) == hipSuccess &&
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

static int shard_commit(dvt_prover *p, const dvt_pk *pk, dvt_job *j, size_t i, uint32_t header[HEADER_WORDS]) {
    std::vector<ChipTrace> traces;
    ShardJob &s = j->shards[i];
    // keep the phase-1 results in HBM while they fit (about 3 GB per 2^21-cycle shard); otherwise phase 2 recomputes
    size_t free_b = 0, total_b = 0;
    if (p->keep_phase1 && !s.cache.tree) (void)hipMemGetInfo(&free_b, &total_b);   // (only the first commit of a shard asks)
    MainCache *keep = p->keep_phase1 && (s.cache.tree || free_b > ((size_t)24 << 30)) ? &s.cache : nullptr;
    if (keep && !s.d_cpu) {
        bool ok = hipMalloc(&s.d_cpu, ((size_t)RV32_CPU_MAIN_W << s.log_n[RV32_CHIP_CPU]) * 4) == hipSuccess && hipMalloc(&s.d_byte, j->byte_words * 4) == hipSuccess &&
                  hipMalloc(&s.d_prog, j->prog_words * 4) == hipSuccess;
        if (!ok) {  // not fatal: fall back to the shared working buffers
            (void)hipGetLastError();
            for (uint32_t **d : {&s.d_cpu, &s.d_byte, &s.d_prog}) { if (*d) (void)hipFree(*d); *d = nullptr; }
        }
    }
    int rc = shard_traces(p, pk, j, i, &traces, false);
    if (rc) return rc;
    Digest root;
    if (!p->eng.commit_main_root(pk->key, traces, &root, keep)) return fail(p, DVT_ERR_DEVICE, "%s", p->eng.err.c_str());
    for (int k = 0; k < 8; k++) header[k] = root.d[k].canonical();
    for (uint32_t k = 0; k < N_PUB; k++) header[8 + k] = j->shards[i].pubs[k].canonical();
    return DVT_OK;
}

static int shard_prove(dvt_prover *p, const dvt_pk *pk, dvt_job *j, size_t i, const PermChallenges &gc, std::vector<uint32_t> *words) {
    std::vector<ChipTrace> traces;
    int rc = shard_traces(p, pk, j, i, &traces, j->shards[i].cache.valid);
    if (rc) return rc;
    ShardProof sp;
    bool ok = p->eng.prove_shard(pk->key, traces, j->shards[i].pubs, p->cfg, &sp, &gc, &j->shards[i].cache);
    (void)hipStreamSynchronize(p->eng.stream);
    j->shards[i].cache.valid = false;  // the buffers stay for the next commit of this shard (released with the job)
    j->shards[i].traces_valid = false;
    if (!ok) return fail(p, DVT_ERR_DEVICE, "%s", p->eng.err.c_str());
    WordWriter w;
    w.w.reserve((size_t)1 << 20);  // a shard proof is about 2.4 MB at 100 queries
    write_shard_proof(w, sp);
    *words = std::move(w.w);
    return DVT_OK;
}

static std::vector<uint32_t> assemble_container(const dvt_job *j, const std::vector<std::vector<uint32_t>> &shards) {
    WordWriter w;
    w.u32(CORE_PROOF_MAGIC);
    w.u32((uint32_t)shards.size());
    w.u32((uint32_t)j->res.exit_code);
    const auto &pv = j->res.public_values;
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

// both phases on one GPU (no lock)
static int job_prove(dvt_prover *p, const dvt_pk *pk, dvt_job *j, uint8_t **proof, size_t *proof_len) {
    HIP_TRY(p, hipSetDevice(p->eng.device));
    const size_t n = j->shards.size();
    std::vector<uint32_t> headers(n * HEADER_WORDS);
    for (size_t i = 0; i < n; i++) {
        int rc = shard_commit(p, pk, j, i, headers.data() + i * HEADER_WORDS);
        if (rc) return rc;
    }
    PermChallenges gc = global_challenges(pk->key.vk, headers.data(), n);
    std::vector<std::vector<uint32_t>> shards(n);
    for (size_t i = 0; i < n; i++) {
        int rc = shard_prove(p, pk, j, i, gc, &shards[i]);
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

int dvt_execute(const uint8_t *elf, size_t elf_len, const dvt_buf *stdin_bufs, size_t nbuf, uint64_t max_cycles,
                uint8_t **public_values, size_t *pv_len, dvt_report *report, char **err_text) {
    if (err_text) *err_text = nullptr;
    if (public_values) *public_values = nullptr;
    if (!elf || (nbuf && !stdin_bufs)) return DVT_ERR_INPUT;
    rv32::Program prog;
    std::string err;
    if (!rv32::load_elf(elf, elf_len, &prog, &err)) {
        if (err_text) *err_text = strdup(("ELF: " + err).c_str());
        return DVT_ERR_INPUT;
    }
    rv32::ExecResult res;
    rv32::execute(prog, collect_stdin(stdin_bufs, nbuf), false, max_cycles ? max_cycles : ~0ull, 21, &res);
    fill_report(report, res);
    if (public_values) *public_values = dup_bytes(res.public_values, pv_len);
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

int dvt_rv32_prepare(dvt_prover *p, const dvt_pk *pk, const dvt_buf *stdin_bufs, size_t nbuf, dvt_job **job, dvt_report *report) {
    if (!p || !pk || !job || (nbuf && !stdin_bufs)) return fail(p, DVT_ERR_INPUT, "null argument");
    if (!pk->is_rv32) return fail(p, DVT_ERR_INPUT, "proving key was not made by dvt_setup");
    std::lock_guard<std::mutex> lk(p->mu);
    return job_prepare(p, pk, stdin_bufs, nbuf, job, report);
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
    job_release(job);
}
size_t dvt_rv32_job_shards(const dvt_job *job) { return job ? job->shards.size() : 0; }

int dvt_rv32_commit_shard(dvt_prover *p, const dvt_pk *pk, dvt_job *job, size_t shard, uint32_t header[15]) {
    if (!p || !pk || !job || !header || shard >= job->shards.size()) return fail(p, DVT_ERR_INPUT, "bad argument");
    std::lock_guard<std::mutex> lk(p->mu);
    HIP_TRY(p, hipSetDevice(p->eng.device));
    return shard_commit(p, pk, job, shard, header);
}
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
    if (!p || !pk || !job || !challenges || shard >= job->shards.size() || (proof && !proof_len)) return fail(p, DVT_ERR_INPUT, "bad argument");
    std::lock_guard<std::mutex> lk(p->mu);
    HIP_TRY(p, hipSetDevice(p->eng.device));
    PermChallenges gc;
    for (int k = 0; k < 4; k++) {
        if (challenges[k] >= P || challenges[4 + k] >= P) return fail(p, DVT_ERR_INPUT, "challenge not canonical");
        gc.alpha.c[k] = Fp::from_canonical(challenges[k]);
        gc.beta.c[k] = Fp::from_canonical(challenges[4 + k]);
    }
    std::vector<uint32_t> words;
    int rc = shard_prove(p, pk, job, shard, gc, &words);
    if (rc || !proof) return rc;
    *proof = copy_out(words, proof_len);
    return *proof ? DVT_OK : fail(p, DVT_ERR_DEVICE, "out of host memory");
}
int dvt_rv32_assemble(const dvt_job *job, const uint8_t *const *shard_proofs, const size_t *lens, size_t n, uint8_t **proof, size_t *proof_len) {
    if (!job || !shard_proofs || !lens || !proof || !proof_len || n != job->shards.size()) return DVT_ERR_INPUT;
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
    int rc = job_prepare(p, pk, stdin_bufs, nbuf, &j, report);
    if (rc) return rc;
    rc = job_prove(p, pk, j, proof, proof_len);
    job_release(j);
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
    StarkConfig cfg;
    cfg.num_queries = fri_queries;
    cfg.pow_bits = pow_bits;
    try {
        WordReader r(words.data(), words.size());
        if (r.u32() != CORE_PROOF_MAGIC) return reject(DVT_ERR_REJECTED, "bad container magic");
        const uint32_t nshards = r.len(1 << 16);
        if (nshards == 0) return reject(DVT_ERR_REJECTED, "no shards");
        const uint32_t ec = r.u32(), pvl = r.len(1 << 24);
        std::vector<uint8_t> pv(pvl);
        for (uint32_t i = 0; i < pvl; i += 4) {
            uint32_t v = r.u32();
            for (uint32_t k = 0; k < 4 && i + k < pvl; k++) pv[i + k] = (uint8_t)(v >> (8 * k));
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
            if (last && pubv[1] != 0) return reject(DVT_ERR_REJECTED, "execution did not halt");
            if (!last && pubv[1] == 0) return reject(DVT_ERR_REJECTED, "halt before the last shard");
            if (last && pubv[2] != ec % P) return reject(DVT_ERR_REJECTED, "exit code mismatch");
            if (pubv[5] != (i ? sps[i - 1].public_values[6].canonical() : 0u)) return reject(DVT_ERR_REJECTED, "public-value counters do not chain");
            if (pubv[6] < pubv[5]) return reject(DVT_ERR_REJECTED, "public-value counter decreases");
            if (last && (pvl % 4 || pubv[6] != pvl / 4)) return reject(DVT_ERR_REJECTED, "number of committed public-value words does not match");
            // chip set: program, byte, cpu, mem_image always; mem_init in the last shard only; shift / muldiv when the shard uses them
            bool have[rv32::N_CHIPS] = {};
            for (auto &c : sp.chips) have[c.chip_id] = true;
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
        // the receiving side of the public-values bus is supplied here, from the claimed bytes:
        // word k contributes 1 / (alpha + bus + beta*k + beta^2 b0 + ... + beta^5 b3)
        {
            Fp4 bp[5];
            bp[0] = gc.beta;
            for (int k = 1; k < 5; k++) bp[k] = bp[k - 1] * gc.beta;
            Fp4 expect = Fp4::zero();
            for (uint32_t k = 0; k < pvl / 4; k++) {
                Fp4 d = gc.alpha + Fp::from_canonical(5) + bp[0] * Fp::from_canonical(k);
                for (int b = 0; b < 4; b++) d += bp[1 + b] * Fp::from_canonical(pv[4 * k + b]);
                expect += inv(d);
            }
            if (total != expect) return reject(DVT_ERR_REJECTED, "LogUp cumulative sums do not cancel across the shards (memory bus or public values)");
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
    if (!p || !pk || !j || !blob || !blob_words || shard >= j->shards.size()) return fail(p, DVT_ERR_INPUT, "bad argument");
    std::lock_guard<std::mutex> lk(p->mu);
    HIP_TRY(p, hipSetDevice(p->eng.device));
    std::vector<ChipTrace> traces;
    int rc = shard_traces(p, pk, j, shard, &traces, false);
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
    for (auto x : j->shards[shard].pubs) T.pubs.push_back(x.canonical());
    std::vector<uint32_t> w = trace_blob(T, nullptr);
    *blob = (uint32_t *)malloc(w.size() * 4);
    if (!*blob) return fail(p, DVT_ERR_DEVICE, "out of host memory");
    memcpy(*blob, w.data(), w.size() * 4);
    *blob_words = w.size();
    return DVT_OK;
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
