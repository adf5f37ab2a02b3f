/*
 * dvt_prover.h — C ABI of the MI355X-native STARK prover that replaces the SP1
 * prover behind metacraft-labs/dvt-circuits' prove()/execute() boundary.
 *
 * The reference reaches its prover through six call sites in src/main.rs
 * (ProverClient::from_env :438,:461,:481; setup :462,:482; prove(..).run()
 * :463-466; execute(..).run() :439-442,:498-501; SP1Stdin::write :434-437,
 * :458-460; proof.save :472-474).  Each entry point below names the call site
 * it stands in for.  INTEGRATION.md shows the Rust `extern "C"` stub a
 * maintainer would add.
 *
 * Conventions
 *   - return 0 = ok; DVT_ERR_GUEST = guest halted non-zero / panicked (what the
 *     reference's 92 test vectors observe as process exit code 1, script/run.sh:82-89);
 *     DVT_ERR_INPUT = malformed argument / ELF / proof; DVT_ERR_DEVICE = HIP failure
 *     or no gfx950 device (there is NO CPU fallback); DVT_ERR_UNSUPPORTED = the
 *     program uses an instruction the prover has no chip for yet;
 *     DVT_ERR_REJECTED = a proof failed verification.
 *   - buffers returned through `uint8_t **` are library-allocated, release with dvt_free().
 *   - a dvt_prover is re-entrant per handle: one handle per caller thread (the
 *     reference's HTTP node calls prove() from concurrent tokio workers,
 *     src/service/node.rs:72-81); calls on one handle are serialised internally.
 *   - "device field array": uint32_t words in HBM holding BabyBear elements in
 *     the library's internal (Montgomery) representation, COLUMN-MAJOR
 *     ([width][height], element (r,c) at c*height + r), natural row order.
 *   - all device work of a handle runs on the handle's own HIP stream (dvt_stream).
 */
#ifndef DVT_PROVER_H
#define DVT_PROVER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DVT_OK 0
#define DVT_ERR_GUEST 1
#define DVT_ERR_INPUT 2
#define DVT_ERR_DEVICE 3
#define DVT_ERR_UNSUPPORTED 4
#define DVT_ERR_REJECTED 5

typedef struct dvt_prover dvt_prover;
typedef struct dvt_pk dvt_pk;

/* ---------------------------------------------------------------- lifecycle */
/* ProverClient::from_env() (src/main.rs:438,461,481).  cfg_json may be NULL or
 * a JSON object: {"device":0,"fri_queries":100,"pow_bits":16,"profile":0,"log_shard_size":21,
 * "keep_phase1":1,"exec_threads":0}.  exec_threads = trace-mode executor threads of the prove pipeline
 * (0 = from the host's core count).  keep_phase1 = 0 makes phase 2 of a shard recompute K0 and the main-trace commitment
 * instead of keeping them in HBM (about 3 GB per 2^21-cycle shard) between the two phases. */
int dvt_prover_create(const char *cfg_json, dvt_prover **out);
void dvt_prover_destroy(dvt_prover *p);
/* last error text of this handle (or of the failed create when p == NULL) */
const char *dvt_last_error(const dvt_prover *p);
void dvt_free(void *ptr);
/* ABI version of this header */
uint32_t dvt_abi_version(void);
/* the handle's hipStream_t (for event timing by the caller) */
void *dvt_stream(dvt_prover *p);
int dvt_sync(dvt_prover *p);

/* ------------------------------------------------- stage-level entry points
 * One call = one kernel family of SURVEY.md section 8(a) on caller-owned device
 * memory; used by the parity tests and by bench.py's roofline measurement.
 * Asynchronous on dvt_stream(p): call dvt_sync() before reading results. */
/* canonical <-> internal representation, in place, n words */
int dvt_dev_to_internal(dvt_prover *p, uint32_t *d_words, size_t n);
int dvt_dev_from_internal(dvt_prover *p, uint32_t *d_words, size_t n);

/* K1: coset low-degree extension, blow-up 2.  d_in [width][2^log_n] holds
 * evaluations over the subgroup H; d_out [width][2^(log_n+1)] receives the
 * evaluations on shift*H', |H'| = 2|H|.  shift_mode: 0 = the generator 31 (trace
 * commitments), 1 = 1, 2 = w_{2N}^-1 (the two quotient chunks).  d_scratch
 * ([width][2^log_n]) holds the intermediate of the first pass when log_n > 12;
 * pass NULL to run that pass in place, which clobbers d_in.  log_n <= 22. */
int dvt_stage_coset_lde(dvt_prover *p, uint32_t *d_in, uint32_t *d_scratch, uint32_t *d_out,
                        uint32_t width, uint32_t log_n, uint32_t shift_mode);

/* K2+K3: mixed-height Poseidon2 Merkle commitment (natural-order pairing). */
typedef struct {
    const uint32_t *d_data; /* device field array [width][2^log_height] */
    uint32_t width;
    uint32_t log_height;
} dvt_dev_matrix;
/* words the digest buffer must hold: (2*H - 1) * 8, H = tallest height */
size_t dvt_merkle_digest_words(const dvt_dev_matrix *mats, size_t n);
/* d_digests: layer 0 (H digests of 8 words) first, then H/2, ..., the root last */
int dvt_stage_merkle_commit(dvt_prover *p, const dvt_dev_matrix *mats, size_t n, uint32_t *d_digests);
/* raw permutation on n states of 16 words each (device array [n][16]); test hook */
int dvt_stage_poseidon2_permute(dvt_prover *p, uint32_t *d_states, size_t n);

/* K8: one FRI fold of d_v (2^log_m extension elements, 4 words each, natural
 * order) into d_out (2^(log_m-1)); d_ro (may be NULL) is added element-wise;
 * beta = 4 canonical words. */
int dvt_stage_fri_fold(dvt_prover *p, const uint32_t *d_v, uint32_t *d_out, const uint32_t *d_ro,
                       const uint32_t beta[4], uint32_t log_m);

/* ------------------------------------------------- machine-level entry points
 * A "machine" is a fixed list of chips (AIRs) compiled into the library:
 * "toy" (engine unit tests) and "rv32" (the RISC-V core machine).  Traces are
 * host arrays, canonical form, column-major. */
typedef struct {
    uint32_t chip_id;
    uint32_t log_n;
    const uint32_t *data; /* [width][2^log_n] */
} dvt_host_trace;
/* preprocessed commitment = the prover half of client.setup(elf) (src/main.rs:462) */
int dvt_machine_setup(dvt_prover *p, const char *machine, const dvt_host_trace *prep, size_t nprep,
                      dvt_pk **pk, uint8_t **vk, size_t *vk_len);
void dvt_pk_free(dvt_prover *p, dvt_pk *pk);
/* prove one shard from explicit main traces (sorted by chip id) */
int dvt_machine_prove(dvt_prover *p, const dvt_pk *pk, const dvt_host_trace *main, size_t nmain,
                      const uint32_t *public_values, size_t npub, uint8_t **proof, size_t *proof_len);
/* host-only; DVT_ERR_REJECTED + reason in *reason (release with dvt_free) when the proof is bad */
int dvt_machine_verify(const uint8_t *vk, size_t vk_len, const uint8_t *proof, size_t proof_len,
                       uint32_t fri_queries, uint32_t pow_bits, char **reason);
/* per-stage milliseconds of the last dvt_machine_prove on this handle (needs "profile":1):
 * out[0..5] = commit_main, permutation, quotient, openings, fri, total */
int dvt_last_stage_ms(dvt_prover *p, float out[6]);
/* kernel-family totals of the last prove on this handle (needs "profile":1), measured with HIP
 * events on the prover stream: out[0] = K1 LDE milliseconds, out[1] = K1 algorithmic bytes
 * (12 B per input element: read N, write 2N words per column), out[2] = K1 calls,
 * out[3] = K2+K3 (trace commitments) milliseconds, out[4] = Poseidon2 permutations they ran,
 * out[5..8] = committed cells of the shard in field elements: main, permutation, quotient (both
 * flattened over F_p^4), preprocessed -- the M, P, Q, Pre of SURVEY.md section 8d */
int dvt_last_kernel_stats(dvt_prover *p, double out[9]);

/* ------------------------------------------------- the reference's boundary
 * These five calls are what the reference's FFI for this path binds
 * (INTEGRATION.md).  ELF = RV32IM guest; stdin = list of byte buffers exactly
 * as SP1Stdin::write / write_vec push them (src/main.rs:434-437,458-460). */
typedef struct {
    const uint8_t *data;
    size_t len;
} dvt_buf;
typedef struct {
    uint64_t cycles;
    int32_t exit_code;    /* a0 at HALT; -1 if the guest never halted */
    uint32_t halted;
    uint32_t unprovable;  /* retired an instruction without a chip (prove would return DVT_ERR_UNSUPPORTED) */
} dvt_report;
/* client.setup(elf) (src/main.rs:462,482): decode the ELF, commit the preprocessed
 * tables (program, byte, memory image) on the GPU. */
int dvt_setup(dvt_prover *p, const uint8_t *elf, size_t elf_len, dvt_pk **pk, uint8_t **vk, size_t *vk_len);
/* client.execute(elf,&stdin).run() (src/main.rs:439-442,498-501): host-only emulation.
 * Returns DVT_ERR_GUEST when the guest halts with a non-zero exit code or traps
 * (the reference maps both to process exit code 1).  *public_values = the bytes the
 * guest wrote to fd 3 with the WRITE syscall (sp1_zkvm::io::commit, reference
 * crates/finalization_prove/src/main.rs:26-32), i.e. what SP1PublicValues holds;
 * *err_text (optional) = trap reason; both via dvt_free.
 * Guest syscall ABI (SP1's): t0 = id, a0..a2 = arguments: 0x00 HALT(code),
 * 0x02 WRITE(fd, ptr, len), 0x10 COMMIT(index, word) - the eight words of SHA-256(public-value bytes),
 * which the proof binds -, 0x1A COMMIT_DEFERRED_PROOFS (no-op), 0xF0 HINT_LEN, 0xF1 HINT_READ(ptr, len).
 * Precompiles (SP1 syscall codes with byte 1 = 1, proven by their own chips): 0x00_30_01_05 SHA_EXTEND(w) and
 * 0x00_01_01_06 SHA_COMPRESS(w, state), the two calls of SP1's patched `sha2` crate.  The BLS12-381 / secp256k1
 * accelerators are not implemented: such a call traps ("unknown syscall"), no proof is produced. */
int dvt_execute(const uint8_t *elf, size_t elf_len, const dvt_buf *stdin_bufs, size_t nbuf, uint64_t max_cycles,
                uint8_t **public_values, size_t *pv_len, dvt_report *report, char **err_text);
/* the same, also handing back what the guest wrote to the other file descriptors (SP1 forwards fd 1 / 2 to the
 * host's stdout / stderr while executing) */
int dvt_execute_io(const uint8_t *elf, size_t elf_len, const dvt_buf *stdin_bufs, size_t nbuf, uint64_t max_cycles,
                   uint8_t **public_values, size_t *pv_len, uint8_t **stdout_bytes, size_t *stdout_len,
                   dvt_report *report, char **err_text);
/* client.prove(&pk,&stdin).run() (src/main.rs:463-466), SP1 "core" mode: execute,
 * generate traces, prove on the GPU.  The returned bytes are what proof.save(path)
 * (src/main.rs:472-474) would write. */
int dvt_prove_core(dvt_prover *p, const dvt_pk *pk, const dvt_buf *stdin_bufs, size_t nbuf, uint8_t **proof,
                   size_t *proof_len, dvt_report *report);
/* dvt_prove_core in pieces, so that callers (and bench.py) can spread the shards of one execution over
 * several GPUs.  An execution is cut into shards of 2^log_shard_size cycles (cfg "log_shard_size",
 * default 21).  All shards are proven with COMMON LogUp challenges derived from every shard's header
 * (dvt_rv32_header_words() = 13 words: main-trace Merkle root + 5 public values), which is the one
 * exchange step of the path (an all-gather of 52 bytes per shard):
 *   prepare        the executor pipeline: one sequential fast pass of the guest cuts the execution into
 *                  shards; the shards this job owns (prepare: all; prepare_part: first, first + stride, ...)
 *                  are re-executed in trace mode on host threads, uploaded (compact 48-byte per-cycle
 *                  records) and taken through phase 1 on the GPU as they arrive
 *   commit_shard   header of shard i (global position in the execution); phase 1 = K0 + K1..K3 of the
 *                  main traces runs here only if the pipeline's result has been consumed by an earlier proof
 *   challenges     host-only: the common challenges from ALL headers (in shard order)
 *   prove_shard    phase 2 of shard i: K0..K9 with those challenges -> shard proof bytes
 *   assemble       container (what proof.save would write) from the shard proofs, in order
 * dvt_rv32_prove_job runs everything on the handle's GPU; proof may be NULL to discard the bytes. */
typedef struct dvt_job dvt_job;
int dvt_rv32_prepare(dvt_prover *p, const dvt_pk *pk, const dvt_buf *stdin_bufs, size_t nbuf, dvt_job **job,
                     dvt_report *report);
int dvt_rv32_prepare_part(dvt_prover *p, const dvt_pk *pk, const dvt_buf *stdin_bufs, size_t nbuf, size_t first,
                          size_t stride, dvt_job **job, dvt_report *report);
/* words per shard header (13) */
uint32_t dvt_rv32_header_words(void);
/* seconds the GPU-side thread of the last prepare spent waiting for the host executor (0 = fully hidden) */
double dvt_rv32_job_exec_wait_seconds(const dvt_job *job);
int dvt_rv32_prove_job(dvt_prover *p, const dvt_pk *pk, dvt_job *job, uint8_t **proof, size_t *proof_len);
void dvt_job_free(dvt_prover *p, dvt_job *job);
/* shards of the whole execution (a prepare_part job holds only its share of them) */
size_t dvt_rv32_job_shards(const dvt_job *job);
int dvt_rv32_commit_shard(dvt_prover *p, const dvt_pk *pk, dvt_job *job, size_t shard, uint32_t *header);
int dvt_rv32_challenges(const uint8_t *vk, size_t vk_len, const uint32_t *headers, size_t n_shards, uint32_t out[8]);
int dvt_rv32_prove_shard(dvt_prover *p, const dvt_pk *pk, dvt_job *job, size_t shard, const uint32_t challenges[8],
                         uint8_t **proof, size_t *proof_len);
int dvt_rv32_assemble(const dvt_job *job, const uint8_t *const *shard_proofs, const size_t *lens, size_t n_shards,
                      uint8_t **proof, size_t *proof_len);
/* test hook: run K0 on one shard of a prepared job and return the device-generated main traces
 * (canonical); blob layout as dvt_rv32_debug_traces with prep_width = 0. */
int dvt_rv32_debug_device_traces(dvt_prover *p, const dvt_pk *pk, dvt_job *job, size_t shard, uint32_t **blob,
                                 size_t *blob_words);
/* stock `client.verify(&proof,&vk)` semantics (NOT the reference's re-execution
 * `verify` sub-command).  Host-only.  *public_values = the guest's fd-3 byte
 * stream; the proof binds it through the eight COMMITted words of its SHA-256 digest, which the
 * verifier recomputes.  fri_queries / pow_bits are the parameters the CALLER accepts
 * (1..1024 queries, at most 30 bits; anything else is DVT_ERR_INPUT). */
int dvt_verify(const uint8_t *vk, size_t vk_len, const uint8_t *proof, size_t proof_len, uint32_t fri_queries,
               uint32_t pow_bits, int32_t *exit_code, uint8_t **public_values, size_t *pv_len, char **reason);
/* measurement hook, host only: guest cycles per second of the executor alone; trace = 0: fast mode (the sequential
 * pass of the prove pipeline), 1: trace mode (48-byte record per cycle).  0.0 when the guest does not halt. */
double dvt_debug_exec_rate(const uint8_t *elf, size_t elf_len, const dvt_buf *stdin_bufs, size_t nbuf, uint32_t log_shard,
                           int trace);
/* test hook, host only: FP64 formulation of Poseidon2 (csrc/poseidon2_f64.cuh, what the hashing kernels run)
 * against the integer permutation on n states; returns the number of differing words (0 = identical) */
uint64_t dvt_debug_p2_f64_selfcheck(uint32_t n, uint32_t seed);

/* Host side of the reference's prove()/execute() above the prover call: the typed JSON input of
 * `--type` (bad-share | finalization | bad-partial-key | bad-encrypted-share; crates/dkg/src/types.rs:26-203)
 * -> serde_cbor::to_vec(data) (src/main.rs:435,459) -> the one SP1Stdin buffer (`stdin.write(&bin)`,
 * :437,:460: u64-LE length + CBOR bytes).  auth_commitment != 0 adds the fields of the reference's
 * cargo feature of that name.  Host-only; *out via dvt_free. */
int dvt_stdin_from_json(const char *type, const char *json, size_t json_len, int auth_commitment, uint8_t **out,
                        size_t *out_len, char **err_text);
/* `--json-schema-file` of the reference's CLI (src/main.rs:509-541: JSONSchema::compile + validate): checks
 * `json` against the draft-07 `schema` (the keyword subset of the schema files under the reference's spec/json/: type, $ref
 * into #/definitions, required, properties, items, minLength, maxLength, pattern, minimum, maximum).
 * DVT_OK, or DVT_ERR_INPUT with one violation per line in *err_text (dvt_free).  Host-only. */
int dvt_json_schema_validate(const char *schema, size_t schema_len, const char *json, size_t json_len, char **err_text);

/* test hook, host-only: the traces (canonical, column-major) the prover would commit for shard
 * `shard` (0-based position) of this run cut at 2^log_shard cycles (0 = default 21).  Layout of *blob
 * (u32 words): n_chips present, then per chip {chip_id, log_n, main_width, prep_width}, then n_pub,
 * pubs..., then per chip the main words followed by the preprocessed words. */
int dvt_rv32_debug_traces(const uint8_t *elf, size_t elf_len, const dvt_buf *stdin_bufs, size_t nbuf, uint32_t log_shard,
                          uint32_t shard, uint32_t *n_shards, uint32_t **blob, size_t *blob_words, char **err_text);

#ifdef __cplusplus
}
#endif
#endif
