// RV32IM guest execution and trace generation for the rv32 machine
// (tools/airgen/rv32.py).  Stands behind reference src/main.rs:439-442
// (`client.execute(elf,&stdin).run()`) and the executor half of :463-466.
//
// VM contract (this library's own; SP1's lives in the absent sp1-core-executor):
//   - ELF32 little-endian RISC-V executable; PT_LOAD segments form the initial
//     memory image, executable segments are decoded into the program table;
//     registers start at 0; pc starts at e_entry; all addresses < 2^30.
//   - ecall: t0 = syscall id, a0..a2 = arguments; SP1's guest ABI as observed in the reference's
//     bundled guest: 0x00 HALT(a0 = exit code), 0x02 WRITE(fd,ptr,len) —
//     fd 3 appends to the public-value byte stream (sp1_zkvm::io::commit, reference
//     crates/finalization_prove/src/main.rs:26-32), other fds are guest output; 0x10 COMMIT(a0 =
//     index, a1 = digest word): the guest commits the eight words of SHA-256(public-value bytes)
//     before HALT, the proof binds them and the verifier recomputes them from the claimed bytes;
//     0x1A COMMIT_DEFERRED_PROOFS (no-op: core proofs have no recursion), 0xF0 HINT_LEN (-> t0),
//     0xF1 HINT_READ(ptr,len): the next stdin buffer becomes the initial value of untouched memory.
//     stdin is a list of byte buffers (SP1Stdin::write, src/main.rs:434-437).
//   - FENCE / EBREAK / CSR instructions execute but have no chip: a program that retires one
//     cannot be proven (prove returns DVT_ERR_UNSUPPORTED).  Precompile syscalls are not implemented.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "bb.cuh"
#include "gen/rv32_cols.h"

namespace dvt {
namespace rv32 {

// flag bit positions, in the FLAGS order of tools/airgen/rv32.py
enum Flag : uint32_t {
    F_RD_EN, F_RS1_EN, F_RS2_EN, F_IMM_C, F_ADD, F_SUB, F_AND, F_OR, F_XOR, F_SLT, F_SLTU, F_MUL, F_MULHU,
    F_LUI, F_JAL, F_JALR, F_BEQ, F_BNE, F_BLT, F_BGE, F_BLTU, F_BGEU, F_LW, F_SW, F_ECALL,
    F_LB, F_LBU, F_LH, F_LHU, F_SB, F_SH, F_ALU, N_FLAGS
};
constexpr uint32_t B_AND = 1, B_OR = 2, B_XOR = 3, B_LTU = 4, B_MSB = 5, B_RANGE = 6, B_U16 = 7, B_ADDR = 8;
constexpr int N_BYTE_OPS = 8;  // multiplicity columns of the byte chip, in this order: and or xor ltu msb range u16 addr
constexpr uint32_t ADDR_LIMIT = 0x38000000u;  // tools/airgen/rv32.py ADDR_TOP_BYTE: address + address gap stays below p
constexpr uint32_t SYS_COMMIT = 0x10;
constexpr uint32_t SYS_SHA_EXTEND = 0x00300105u;  // SP1 syscall code (byte 1 = 1: the call has a precompile table)
constexpr uint32_t SYS_SHA_COMPRESS = 0x00010106u;
constexpr uint32_t REG_A1 = 11;
constexpr uint32_t REG_BASE = ADDR_LIMIT + (1u << 23);   // the registers are words REG_BASE + 0..31 of the memory argument: above every guest address (and every multi-word precompile access that starts below ADDR_LIMIT)
constexpr uint32_t HALT_PC = 1u << 30;        // next_pc of a HALT row (tools/airgen/rv32.py): no other row can produce it
constexpr uint32_t BAD_PC = 1;                // program-table target of a JAL / branch whose static target lies outside the text
constexpr uint32_t MAX_SHARDS = 65535;        // shard numbers travel as 16-bit halves of the cycle records
constexpr uint32_t ALU_SLL = 1, ALU_SRL = 2, ALU_SRA = 3;  // alu-bus opcodes (chips outside the cpu chip): shift chip
constexpr uint32_t ALU_MULH = 4, ALU_MULHSU = 5, ALU_DIV = 6, ALU_DIVU = 7, ALU_REM = 8, ALU_REMU = 9;  // muldiv chip
constexpr int N_CHIPS = 14;  // program, byte, cpu, mem_image, mem_init, shift, muldiv, sha_extend, sha_compress, fp_op, fp2_op, bls_g1, secp_k1, u256_mul
constexpr uint32_t SYS_UINT256_MUL = 0x0001011Du;   // x := x * y mod m, the modulus after y in memory (0 = 2^256)
// field / curve precompiles (SP1's syscall numbers as best recalled [EXTERNAL, unverified]; tools/airgen/rv32.py)
constexpr uint32_t SYS_SECP256K1_ADD = 0x0001010Au, SYS_SECP256K1_DOUBLE = 0x0000010Bu;
constexpr uint32_t SYS_BLS12381_ADD = 0x0001011Eu, SYS_BLS12381_DOUBLE = 0x0000011Fu;
constexpr uint32_t SYS_BLS12381_FP_ADD = 0x00010120u, SYS_BLS12381_FP_SUB = 0x00010121u, SYS_BLS12381_FP_MUL = 0x00010122u;
constexpr uint32_t SYS_BLS12381_FP2_ADD = 0x00010123u, SYS_BLS12381_FP2_SUB = 0x00010124u, SYS_BLS12381_FP2_MUL = 0x00010125u;
constexpr uint32_t N_PUBLIC = 5;  // start_pc, next_pc, exit_code, shard, is_last

// dense dispatch code of the interpreter (one case per instruction form: which ports it drives is static per case)
enum Kind : uint8_t {
    K_UNSUP, K_ADD, K_SUB, K_AND, K_OR, K_XOR, K_SLT, K_SLTU, K_MUL, K_MULHU,       // register forms: rs1, rs2 -> rd
    K_ADDI, K_ANDI, K_ORI, K_XORI, K_SLTI, K_SLTIU,                                 // immediate forms: rs1, imm -> rd
    K_LUI, K_JAL, K_JALR, K_BEQ, K_BNE, K_BLT, K_BGE, K_BLTU, K_BGEU,
    K_LW, K_LB, K_LBU, K_LH, K_LHU, K_SW, K_SB, K_SH, K_ALU_R, K_ALU_I, K_ECALL,
    K_OOB,   // the sentinel after the last instruction of the text: control reached an address outside it
    N_KINDS
};

struct Instr {
    uint32_t pc, rd, rs1, rs2, imm, off, tgt, flags;
    uint32_t alu_op;  // for F_ALU instructions: which chip / operation receives (a, b, c)
    uint32_t raw;
    uint8_t supported;  // has a chip (otherwise executes only)
    uint8_t kind;       // Kind
    uint32_t tgt_idx;   // branches / JAL: index of the target in Program::instrs (the sentinel's when it lies outside the text)
    uint32_t tgt_raw;   // branches / JAL: the target as decoded (tgt is BAD_PC when it lies outside the text)
};

struct Program {
    uint32_t entry = 0, text_base = 0;
    std::vector<Instr> instrs;                            // index (pc - text_base) / 4, plus one K_OOB sentinel at the end
    std::vector<std::pair<uint32_t, uint32_t>> image;     // (byte address, word), sorted, word aligned
};

// The decode-time columns of an instruction as the AIR names them (tools/airgen/rv32.py FLAGS / VALUE_FLAGS): families
// that differ only in a byte-table opcode or in signedness share a selector and carry that value beside it.
struct ColFlags {
    uint32_t rd_en, rs1_en, rs2_en, imm_c, is_add, is_sub, is_bit, is_set, is_mul, is_mulhu, is_lui, is_jal, is_jalr, is_beq, is_bne, is_brlt, is_brge,
        is_lw, is_sw, is_ecall, is_lb, is_lbu, is_lh, is_lhu, is_sb, is_sh, is_alu, bit_op, cmp_signed, imm;
};
DVT_HD ColFlags column_flags(const Instr &in) {
    const uint32_t fl = in.flags;
    auto F = [&](uint32_t bit) -> uint32_t { return (fl >> bit) & 1u; };
    ColFlags c;
    c.rd_en = F(F_RD_EN); c.rs1_en = F(F_RS1_EN); c.rs2_en = F(F_RS2_EN); c.imm_c = F(F_IMM_C);
    c.is_add = F(F_ADD); c.is_sub = F(F_SUB);
    c.is_bit = F(F_AND) | F(F_OR) | F(F_XOR);
    c.bit_op = F(F_AND) * B_AND + F(F_OR) * B_OR + F(F_XOR) * B_XOR;
    c.is_set = F(F_SLT) | F(F_SLTU);
    c.cmp_signed = F(F_SLT) | F(F_BLT) | F(F_BGE);
    c.is_mul = F(F_MUL); c.is_mulhu = F(F_MULHU); c.is_lui = F(F_LUI); c.is_jal = F(F_JAL); c.is_jalr = F(F_JALR);
    c.is_beq = F(F_BEQ); c.is_bne = F(F_BNE);
    c.is_brlt = F(F_BLT) | F(F_BLTU);
    c.is_brge = F(F_BGE) | F(F_BGEU);
    c.is_lw = F(F_LW); c.is_sw = F(F_SW); c.is_ecall = F(F_ECALL);
    c.is_lb = F(F_LB); c.is_lbu = F(F_LBU); c.is_lh = F(F_LH); c.is_lhu = F(F_LHU); c.is_sb = F(F_SB); c.is_sh = F(F_SH);
    c.is_alu = F(F_ALU);
    c.imm = in.imm | in.off;   // one immediate field: operand / LUI constant (imm) or address offset (off), never both
    return c;
}
constexpr uint32_t LINK_TOP_BYTE = 0x78;   // tools/airgen/rv32.py: the top byte of a link value pc + 4 is below it
constexpr uint32_t SYS_HINT_LEN = 0xF0;

// One retired instruction, compact (what the host uploads for K0): 12 words = three 16-byte loads.
// Timestamps are (shard, clk) pairs: *_ts = clk of the previous access of that port, its shard is a
// 16-bit half of sh_ab / sh_cm.  Everything else of the row (next pc, memory word after a store, limbs,
// carries ...) is recomputed by fill_cpu_row on the device.
struct CycleRec {
    uint32_t idx;      // instruction index in Program::instrs
    uint32_t a, b, c;  // rd value written / rs1 value / rs2-or-immediate value
    uint32_t pa_prev, pa_ts, pb_ts, pc_ts;
    uint32_t m_prev, m_ts;   // memory port (loads, stores, the a1 read of COMMIT): word before the access, previous clk
    uint32_t sh_ab;    // pa_sh | pb_sh << 16
    uint32_t sh_cm;    // pc_sh | m_sh << 16
};
static_assert(sizeof(CycleRec) == 48, "CycleRec is uploaded as is");

struct MemInitRow {
    uint32_t addr, v, f, fts, fsh, is_img;
};

struct AluEvent {
    uint32_t op, a, b, c;
};
// one SHA_EXTEND call: w[0..15] as read, w[16..63] as written, the words those replaced, and the (shard, clk) each of the
// 64 words carried before the call
struct ShaExtEvent {
    uint32_t clk, ptr;
    uint32_t w[64], old[64], prev_ts[64];
    uint16_t prev_sh[64];
};

// One shard = up to 2^log_shard consecutive cycles; shards are numbered from 1.
// one SHA_COMPRESS call: the 64 schedule words and the 8 state words as read, with the (shard, clk) each carried before
struct ShaCmpEvent {
    uint32_t clk, w_ptr, h_ptr;
    uint32_t w[64], hs[8], w_ts[64], h_ts[8];
    uint16_t w_sh[64], h_sh[8];
};
// one field / curve precompile call: a0 = pointer to the operand that is replaced by the result (read and written at
// clk + 3), a1 = pointer to the second operand (read at clk + 2; absent for DOUBLE).  Little-endian words.
constexpr int BIGOP_MAX_WORDS = 24;
struct BigOpEvent {
    uint32_t code, clk, a_ptr, b_ptr;
    uint32_t a[BIGOP_MAX_WORDS], b[BIGOP_MAX_WORDS], r[BIGOP_MAX_WORDS];
    uint32_t lam[BIGOP_MAX_WORDS / 2];                  // curve operations: the slope (canonical)
    uint32_t a_ts[BIGOP_MAX_WORDS], b_ts[BIGOP_MAX_WORDS];
    uint16_t a_sh[BIGOP_MAX_WORDS], b_sh[BIGOP_MAX_WORDS];
};
// static shape of a call: the chip that proves it, words of the two operands (words_b = 0: a1 must be 0)
struct BigOpInfo {
    int chip, words_a, words_b;
};
bool bigop_info(uint32_t code, BigOpInfo *out);
// r (and lam for curve operations) from a and b; nullptr or the reason the call traps (non-canonical coordinates, ...)
const char *bigop_compute(uint32_t code, const uint32_t *a, const uint32_t *b, uint32_t *r, uint32_t *lam);
struct ShardRec {
    std::vector<AluEvent> alu;   // instructions of this shard proven by chips outside the cpu chip
    std::vector<ShaExtEvent> sha_ext;   // precompile calls of this shard
    std::vector<ShaCmpEvent> sha_cmp;
    std::vector<BigOpEvent> big;
    uint32_t index = 0, start_pc = 0, next_pc = 0;
    std::vector<CycleRec> recs;
};

struct ExecResult {
    int exit_code = -1;
    bool halted = false;
    uint64_t cycles = 0;
    bool unsupported = false;   // retired an instruction the prover has no chip for
    std::string unsupported_what;
    std::vector<uint8_t> public_values;  // bytes the guest wrote to fd 3
    std::vector<uint8_t> stdout_bytes;   // bytes written to any other fd
    uint32_t committed[8] = {};          // digest words passed to COMMIT (index -> word)
    uint32_t committed_mask = 0;
    std::vector<ShardRec> shards;        // filled only when tracing
    std::vector<MemInitRow> mem_rows;    // sorted by address, filled only when tracing (part of the LAST shard)
    std::string error;                   // non-empty: the guest trapped (bad access, bad pc, ...)
};

bool load_elf(const uint8_t *elf, size_t n, Program *out, std::string *err);

// ---- the guest machine --------------------------------------------------------
// A resumable RV32IM interpreter.  Every register / memory word carries the (shard, clk) of its last access,
// which is what the memory argument of the AIR consumes.  Two modes share one loop:
//   fast   no per-cycle records: finds the shard boundaries and the architectural state there (snapshots)
//   trace  one CycleRec per retired instruction + the events of the shift / muldiv chips
// so that a long execution is cut into shards by ONE sequential fast pass while the shards are re-executed
// from the snapshots in trace mode on other threads (capi.hip), overlapped with the GPU.
// Guest memory is paged copy-on-write: a snapshot shares every page the next shard does not touch.
constexpr uint32_t PAGE_WORD_BITS = 12;
// flags: 1 = accessed by a load / store, 2 = part of the program image; tsh = clk | shard << 32 of the last access
struct Cell {
    uint32_t val, flags;
    uint64_t tsh;
    uint32_t ts() const { return (uint32_t)tsh; }
    uint32_t sh() const { return (uint32_t)(tsh >> 32); }
};
struct Page { Cell c[1u << PAGE_WORD_BITS]; };
constexpr uint32_t N_PAGES = 1u << (30 - 2 - PAGE_WORD_BITS);

// Results of the curve precompile calls of one execution in call order.  The prove pipeline executes every shard twice (the
// sequential fast pass finds the shard boundaries, a traced re-execution produces the records); an affine G1 / secp256k1
// operation is a field inversion (about 3 us), 80 % of the reference example's execution.  The fast pass appends, the traced
// passes take what is there and compute themselves when the writer is behind or gone.
struct CurveLog {
    static constexpr size_t CHUNK = 4096, MAX_CHUNKS = 1024;   // 4 M calls (600 MB) at most: later calls are recomputed
    struct Entry { uint32_t r[BIGOP_MAX_WORDS], lam[BIGOP_MAX_WORDS / 2]; };
    std::unique_ptr<Entry[]> chunks[MAX_CHUNKS];
    std::atomic<size_t> published{0};
    std::atomic<bool> closed{false};    // the writer has finished, trapped or filled the log
    void append(size_t idx, const uint32_t *r, const uint32_t *lam) {   // writer only; idx == published
        if (idx != published.load(std::memory_order_relaxed) || idx >= CHUNK * MAX_CHUNKS) { closed.store(true, std::memory_order_release); return; }
        if (idx % CHUNK == 0) chunks[idx / CHUNK].reset(new Entry[CHUNK]);
        Entry &e = chunks[idx / CHUNK][idx % CHUNK];
        memcpy(e.r, r, sizeof e.r); memcpy(e.lam, lam, sizeof e.lam);
        published.store(idx + 1, std::memory_order_release);
    }
    bool fetch(size_t idx, uint32_t *r, uint32_t *lam) const {
        // (the writer is never blocked while a reader of an EARLIER call waits: it only pauses at shard boundaries, after
        // everything of the shards before has been published)
        while (published.load(std::memory_order_acquire) <= idx) {
            if (closed.load(std::memory_order_acquire) && published.load(std::memory_order_acquire) <= idx) return false;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        const Entry &e = chunks[idx / CHUNK][idx % CHUNK];
        memcpy(r, e.r, sizeof e.r); memcpy(lam, e.lam, sizeof e.lam);
        return true;
    }
};

struct Snapshot {
    Cell regs[32];
    uint32_t pc = 0, shard = 1;
    uint64_t cycles = 0;
    size_t next_input = 0;
    size_t curve_index = 0;   // curve precompile calls retired so far (position in a CurveLog)
    std::vector<std::pair<uint32_t, std::shared_ptr<Page>>> pages;
};

// where trace mode writes: recs must hold 2^log_shard entries (pinned host memory in the prover)
struct ShardOut {
    CycleRec *recs = nullptr;
    size_t n_recs = 0;
    std::vector<AluEvent> alu;
    std::vector<ShaExtEvent> sha_ext;
    std::vector<ShaCmpEvent> sha_cmp;
    std::vector<BigOpEvent> big;
    uint32_t index = 0, start_pc = 0, next_pc = 0;
};

class Vm {
  public:
    Vm(const Program &prog, const std::vector<std::vector<uint8_t>> *stdin_bufs, uint32_t log_shard);
    Vm(const Program &prog, const std::vector<std::vector<uint8_t>> *stdin_bufs, uint32_t log_shard, const Snapshot &at);
    Vm(const Vm &) = delete;
    Snapshot snapshot();   // state at the start of the current shard (call between shards)
    // Runs the current shard to its end: 2^log_shard cycles, HALT, a trap (error non-empty) or the cycle budget.
    // Trace mode fills *out.  Afterwards: halted / !error.empty() -> done; otherwise next_shard() and call again.
    void run_shard(bool trace, ShardOut *out, uint64_t max_total_cycles);
    bool next_shard();     // false (and error set) when the execution would exceed MAX_SHARDS
    // mem_init rows of the whole execution (image words + every other word a load / store touched), sorted by address
    std::vector<MemInitRow> mem_rows() const;

    const Program &prog;
    uint32_t pc, shard = 1, in_shard = 0, log_shard;
    uint64_t cycles = 0;
    bool halted = false, unsupported = false, collect_output = true;
    int exit_code = -1;
    std::string error, unsupported_what;
    std::vector<uint8_t> public_values, stdout_bytes;
    uint32_t committed[8] = {}, committed_mask = 0;
    // optional (the prove pipeline): the fast pass (run_shard(false, ...)) appends the results of its curve precompile calls,
    // traced passes read them
    CurveLog *curve_log = nullptr;
    size_t curve_index = 0;

  private:
    Cell regs[32];
    const std::vector<std::vector<uint8_t>> *stdin_bufs;
    size_t next_input = 0;
    std::vector<std::shared_ptr<Page>> holders;   // [N_PAGES]
    std::vector<Page *> raw;
    std::vector<uint8_t> own;                      // page may be written in place (not shared with a snapshot)
    std::vector<uint32_t> alloc_pages, owned_pages;
    std::vector<std::pair<uint32_t, uint32_t>> first_touch;   // (address, value at the first access) of non-image words
    void make_own(uint32_t pg);
    Cell &at(uint32_t addr) {   // addr: byte address below ADDR_LIMIT
        const uint32_t pg = addr >> (PAGE_WORD_BITS + 2);
        if (!own[pg]) make_own(pg);
        return raw[pg]->c[(addr >> 2) & ((1u << PAGE_WORD_BITS) - 1)];
    }
    const Cell *peek(uint32_t addr) const {
        const Page *p = raw[addr >> (PAGE_WORD_BITS + 2)];
        return p ? &p->c[(addr >> 2) & ((1u << PAGE_WORD_BITS) - 1)] : nullptr;
    }
    template <bool TRACE> void run(ShardOut *out, uint64_t budget);
    void trap(const std::string &m);
};

// Single-threaded convenience over Vm (dvt_execute, the debug trace hooks, tests): stdin = list of buffers;
// trace = keep per-cycle records, cut into shards of 2^log_shard cycles.  max_cycles bounds the run.
void execute(const Program &prog, const std::vector<std::vector<uint8_t>> &stdin_bufs, bool trace, uint64_t max_cycles,
             uint32_t log_shard, ExecResult *res);

// ---- rows of the shift and mem_init chips (K0; Sink: put(col, canonical value); byte(op_index, table_row)) ----------------
// one SLL / SRL / SRA instruction (tools/airgen/rv32.py build_shift)
template <class Sink>
DVT_HD void fill_shift_row(const AluEvent &e, Sink &s) {
    auto Bt = [](uint32_t w, int i) -> uint32_t { return (w >> (8 * i)) & 0xffu; };
    const uint32_t sh = e.c & 31, q = sh >> 3, rb = sh & 7, m = 1u << rb, mi = 1u << (8 - rb);
    const bool left = e.op == ALU_SLL;
    const uint32_t sgn = e.op == ALU_SRA ? e.b >> 31 : 0;
    s.put(RV32_SHIFT_is_real, 1);
    s.put(left ? RV32_SHIFT_is_sll : e.op == ALU_SRL ? RV32_SHIFT_is_srl : RV32_SHIFT_is_sra, 1);
    s.put(RV32_SHIFT_sh, sh);
    if (q) s.put(RV32_SHIFT_q_0 + q - 1, 1);
    s.put(RV32_SHIFT_r_0 + rb, 1);
    s.put(RV32_SHIFT_sgn, sgn);
    uint32_t lo[4], hi[4];
    for (int i = 0; i < 4; i++) {
        s.put(RV32_SHIFT_a_0 + i, Bt(e.a, i)); s.put(RV32_SHIFT_b_0 + i, Bt(e.b, i)); s.put(RV32_SHIFT_c_0 + i, Bt(e.c, i));
        if (left) { uint32_t pr = Bt(e.b, i) * m; lo[i] = pr & 0xff; hi[i] = pr >> 8; }
        else { hi[i] = Bt(e.b, i) >> rb; lo[i] = Bt(e.b, i) & (m - 1); }
        s.put(RV32_SHIFT_lo_0 + i, lo[i]); s.put(RV32_SHIFT_hi_0 + i, hi[i]);
    }
    for (int i = 0; i < 4; i++) {
        uint32_t t = left ? lo[i] + (i ? hi[i - 1] : 0) : hi[i] + (i < 3 ? lo[i + 1] * mi : sgn * (256 - mi));
        s.put(RV32_SHIFT_t_0 + i, t);
        if (!left) s.byte(B_LTU - 1, (lo[i] << 8) | m);
    }
    s.byte(B_AND - 1, (Bt(e.c, 0) << 8) | 31);
    s.byte(B_RANGE - 1, (lo[0] << 8) | lo[1]); s.byte(B_RANGE - 1, (lo[2] << 8) | lo[3]);
    s.byte(B_RANGE - 1, (hi[0] << 8) | hi[1]); s.byte(B_RANGE - 1, (hi[2] << 8) | hi[3]);
    if (e.op == ALU_SRA) s.byte(B_MSB - 1, Bt(e.b, 3) << 8);
}
// row r of the mem_init table (tools/airgen/rv32.py build_mem_init): the word at rows[r].addr, its initial and final values, the
// gap to the previous address
template <class Sink>
DVT_HD void fill_mem_init_row(const MemInitRow *rows, size_t r, Sink &s) {
    const MemInitRow m = rows[r];
    s.put(RV32_MEM_INIT_fts, m.fts); s.put(RV32_MEM_INIT_fsh, m.fsh);
    s.put(RV32_MEM_INIT_is_img, m.is_img); s.put(RV32_MEM_INIT_is_real, 1);
    const uint32_t d = r ? m.addr - rows[r - 1].addr - 1 : 0;
    for (int i = 0; i < 4; i++) {
        s.put(RV32_MEM_INIT_ab_0 + i, (m.addr >> (8 * i)) & 0xff);
        s.put(RV32_MEM_INIT_v_0 + i, (m.v >> (8 * i)) & 0xff);
        s.put(RV32_MEM_INIT_f_0 + i, (m.f >> (8 * i)) & 0xff);
        s.put(RV32_MEM_INIT_d_0 + i, (d >> (8 * i)) & 0xff);
    }
    const uint32_t ws[2] = {m.addr, d};
    for (int k = 0; k < 2; k++) {
        const uint32_t w = ws[k];
        s.byte(B_RANGE - 1, ((w & 0xff) << 8) | ((w >> 8) & 0xff));
        s.byte(B_RANGE - 1, (((w >> 16) & 0xff) << 8) | (w >> 24));
    }
    s.byte(B_LTU - 1, ((m.addr >> 24) << 8) | ((ADDR_LIMIT >> 24) + 1));   // (+ 1: the registers sit at REG_BASE)
    s.byte(B_LTU - 1, ((d >> 24) << 8) | ((ADDR_LIMIT >> 24) + 1));
    if (!m.is_img) {
        s.byte(B_RANGE - 1, ((m.v & 0xff) << 8) | ((m.v >> 8) & 0xff));
        s.byte(B_RANGE - 1, (((m.v >> 16) & 0xff) << 8) | (m.v >> 24));
    }
}

// ---- trace generation (K0) ---------------------------------------------------
// Sink interface used by fill_cpu_row:  put(col, canonical value);  byte(op_index, table_row);  prog(idx)
// next_pc: pc of the next retired instruction (the shard's next_pc public value after its last row).
template <class Sink>
DVT_HD void fill_cpu_row(const CycleRec &r, const Instr &in, uint32_t row, uint32_t shard, uint32_t next_pc, Sink &s) {
    const uint32_t fl = in.flags;
    auto F = [&](uint32_t bit) -> uint32_t { return (fl >> bit) & 1u; };
    auto B = [](uint32_t w, int i) -> uint32_t { return (w >> (8 * i)) & 0xffu; };
    const uint32_t clk = 4 * (row + 1);
    const uint32_t pa_sh = r.sh_ab & 0xffffu, pb_sh = r.sh_ab >> 16, pc_sh = r.sh_cm & 0xffffu, m_sh = r.sh_cm >> 16;
    s.put(RV32_CPU_clk, clk);
    s.put(RV32_CPU_pc, in.pc);
    s.put(RV32_CPU_next_pc, next_pc);
    s.put(RV32_CPU_rd, in.rd);
    s.put(RV32_CPU_rs1, in.rs1);
    s.put(RV32_CPU_rs2, in.rs2);
    const ColFlags cf = column_flags(in);
    for (int i = 0; i < 4; i++) {
        s.put(RV32_CPU_imm_0 + i, B(cf.imm, i));
        s.put(RV32_CPU_a_0 + i, B(r.a, i));
        s.put(RV32_CPU_b_0 + i, B(r.b, i));
        s.put(RV32_CPU_c_0 + i, B(r.c, i));
        s.put(RV32_CPU_pa_prev_0 + i, B(r.pa_prev, i));
    }
    s.put(RV32_CPU_aux, in.tgt + in.alu_op);   // target of the control-flow families / alu-bus opcode of F_ALU rows: never both
    // decode-time columns (rs1_en, rs2_en and is_real are linear in the family flags: no columns)
    s.put(RV32_CPU_bit_op, cf.bit_op); s.put(RV32_CPU_cmp_signed, cf.cmp_signed);
    s.put(RV32_CPU_rd_en, cf.rd_en); s.put(RV32_CPU_imm_c, cf.imm_c);
    s.put(RV32_CPU_is_add, cf.is_add); s.put(RV32_CPU_is_sub, cf.is_sub); s.put(RV32_CPU_is_bit, cf.is_bit); s.put(RV32_CPU_is_set, cf.is_set);
    s.put(RV32_CPU_is_mul, cf.is_mul); s.put(RV32_CPU_is_mulhu, cf.is_mulhu); s.put(RV32_CPU_is_lui, cf.is_lui); s.put(RV32_CPU_is_jal, cf.is_jal);
    s.put(RV32_CPU_is_jalr, cf.is_jalr); s.put(RV32_CPU_is_beq, cf.is_beq); s.put(RV32_CPU_is_bne, cf.is_bne); s.put(RV32_CPU_is_brlt, cf.is_brlt);
    s.put(RV32_CPU_is_brge, cf.is_brge); s.put(RV32_CPU_is_lw, cf.is_lw); s.put(RV32_CPU_is_sw, cf.is_sw); s.put(RV32_CPU_is_ecall, cf.is_ecall);
    s.put(RV32_CPU_is_lb, cf.is_lb); s.put(RV32_CPU_is_lbu, cf.is_lbu); s.put(RV32_CPU_is_lh, cf.is_lh); s.put(RV32_CPU_is_lhu, cf.is_lhu);
    s.put(RV32_CPU_is_sb, cf.is_sb); s.put(RV32_CPU_is_sh, cf.is_sh); s.put(RV32_CPU_is_alu, cf.is_alu);
    s.prog(r.idx);
    // register ports
    uint32_t pb_hi = 0, pc_hi = 0, pa_hi = 0, m_hi = 0;
    // (shard', clk') < (shard, clk): same shard -> clk difference, earlier shard -> shard difference
    auto gap = [&](uint32_t prev_sh, uint32_t prev_ts, uint32_t ts) -> uint32_t { return prev_sh == shard ? ts - prev_ts - 1 : shard - prev_sh - 1; };
    if (F(F_RS2_EN)) {
        uint32_t d = gap(pc_sh, r.pc_ts, clk);
        s.put(RV32_CPU_pc_ts, r.pc_ts); s.put(RV32_CPU_pc_sh, pc_sh); s.put(RV32_CPU_pc_same, pc_sh == shard);
        s.put(RV32_CPU_pc_lo, d & 0xffff); pc_hi = d >> 16;
        s.byte(B_U16 - 1, d & 0xffff);
    }
    if (F(F_RS1_EN)) {
        uint32_t d = gap(pb_sh, r.pb_ts, clk + 1);
        s.put(RV32_CPU_pb_ts, r.pb_ts); s.put(RV32_CPU_pb_sh, pb_sh); s.put(RV32_CPU_pb_same, pb_sh == shard);
        s.put(RV32_CPU_pb_lo, d & 0xffff); pb_hi = d >> 16;
        s.byte(B_U16 - 1, d & 0xffff);
    }
    if (F(F_RD_EN)) {
        uint32_t d = gap(pa_sh, r.pa_ts, clk + 3);
        s.put(RV32_CPU_pa_ts, r.pa_ts); s.put(RV32_CPU_pa_sh, pa_sh); s.put(RV32_CPU_pa_same, pa_sh == shard);
        s.put(RV32_CPU_pa_lo, d & 0xffff); pa_hi = d >> 16;
        s.byte(B_U16 - 1, d & 0xffff);
    }
    s.put(RV32_CPU_pb_hi, pb_hi); s.put(RV32_CPU_pc_hi, pc_hi); s.put(RV32_CPU_pa_hi, pa_hi);
    const int U = RV32_CPU_u_0;
    const uint32_t a = r.a, b = r.b, c = r.c;
    // the memory port (loads / stores, the a1 read of COMMIT): u[9..12] word after, u[13..16] word before, u[17] previous
    // clk, u[8] / u[18] limbs of the timestamp gap, u[19] previous shard, u[20] same-shard flag
    auto mem_port = [&](uint32_t m_val) {
        for (int i = 0; i < 4; i++) { s.put(U + 9 + i, B(m_val, i)); s.put(U + 13 + i, B(r.m_prev, i)); }
        const uint32_t d = gap(m_sh, r.m_ts, clk + 2);
        m_hi = d >> 16;
        s.put(U + 17, r.m_ts); s.put(U + 8, d & 0xffff);
        s.put(U + 18, m_hi); s.put(U + 19, m_sh); s.put(U + 20, m_sh == shard);
        s.byte(B_U16 - 1, d & 0xffff);
    };
    if (F(F_ADD) | F(F_SUB)) {
        // carries of (b + c) for ADD, of (a + c) for SUB
        uint32_t x = F(F_ADD) ? b : a, cin = 0;
        for (int i = 0; i < 4; i++) {
            uint32_t t = B(x, i) + B(c, i) + cin;
            cin = t >> 8;
            s.put(U + i, cin);
        }
    } else if (F(F_AND) | F(F_OR) | F(F_XOR)) {
        // the four byte lookups go through shared slots (tools/airgen/rv32.py): operands copied to u[11+i], u[4+i], u[19+i]
        int op = F(F_AND) ? 0 : F(F_OR) ? 1 : 2;
        for (int i = 0; i < 4; i++) {
            s.put(U + 11 + i, B(a, i)); s.put(U + 4 + i, B(b, i)); s.put(U + 19 + i, B(c, i));
            s.byte(op, (B(b, i) << 8) | B(c, i));
        }
    } else if (F(F_SLT) | F(F_SLTU) | F(F_BEQ) | F(F_BNE) | F(F_BLT) | F(F_BGE) | F(F_BLTU) | F(F_BGEU)) {
        const uint32_t sg = F(F_SLT) | F(F_BLT) | F(F_BGE);
        uint32_t bb[4], cc[4];
        for (int i = 0; i < 4; i++) { bb[i] = B(b, i); cc[i] = B(c, i); }
        uint32_t msb_b = sg ? bb[3] >> 7 : 0, msb_c = sg ? cc[3] >> 7 : 0;
        if (sg) {
            bb[3] ^= 0x80; cc[3] ^= 0x80;
            s.byte(B_MSB - 1, B(b, 3) << 8); s.byte(B_MSB - 1, B(c, 3) << 8);
        }
        int k = -1;
        for (int i = 3; i >= 0; i--) if (bb[i] != cc[i]) { k = i; break; }
        uint32_t bc = k >= 0 ? bb[k] : 0, ccv = k >= 0 ? cc[k] : 0;
        // u[0..3] differing-byte flags, u[4] 1/(b_cmp - c_cmp), u[10] b_cmp, u[20] c_cmp, u[19] lt; signed: u[9] = c_3 with
        // its top bit in u[21], u[24] = b_3 with its top bit in u[25] (shared lookup slots, tools/airgen/rv32.py)
        if (k >= 0) s.put(U + k, 1);
        s.put(U + 10, bc); s.put(U + 20, ccv);
        if (k >= 0) s.put(U + 4, inv(Fp::from_canonical(bc) - Fp::from_canonical(ccv)).canonical());
        s.put(U + 19, bc < ccv);
        if (sg) {
            s.put(U + 24, B(b, 3)); s.put(U + 25, msb_b);
            s.put(U + 9, B(c, 3)); s.put(U + 21, msb_c);
        }
        s.byte(B_LTU - 1, (bc << 8) | ccv);
    } else if (F(F_MUL) | F(F_MULHU)) {
        // (hipcc 7.2 miscompiles this carry chain for gfx950 once the k-loop is fully unrolled: carry 1 comes out as
        //  (t0 + S1) >> 8, the accumulator unshifted.  s.fence() pins it; tools/microbench/k0_fill_repro.hip is the
        //  standalone reproducer (this very template with a store-only sink) and tests/test_gpu_k0_parity.py the guard.)
        uint32_t pbyte[8], pcarry[8], acc = 0;
        for (int k = 0; k < 8; k++) {
            uint32_t t = acc;
            for (int i = 0; i < 4; i++) { int j = k - i; if (j >= 0 && j < 4) t += B(b, i) * B(c, j); }
            s.fence(t);  // keep the compiler from re-associating the carry chain across iterations
            pbyte[k] = t & 0xff;
            pcarry[k] = t >> 8;
            acc = t >> 8;
        }
        // u[0..3] = the half of the product that is not the result (the result half is `a`, range-checked below with
        // the other arithmetic families); u[4..10] = carries out of bytes 0..6 (the one out of byte 7 is 0)
        const int xo = F(F_MUL) ? 4 : 0;
        for (int i = 0; i < 4; i++) s.put(U + i, pbyte[xo + i]);
        for (int k = 0; k < 7; k++) s.put(U + 4 + k, pcarry[k]);
        for (int k = 0; k < 7; k++) s.byte(B_U16 - 1, pcarry[k]);
        s.byte(B_RANGE - 1, (pbyte[xo + 1] << 8) | pbyte[xo + 2]);   // (the two lookups of the address adder's sum bytes)
        s.byte(B_RANGE - 1, (pbyte[xo] << 8) | pbyte[xo + 3]);
    } else if (F(F_LW) | F(F_SW) | F(F_JALR) | F(F_LB) | F(F_LBU) | F(F_LH) | F(F_LHU) | F(F_SB) | F(F_SH)) {
        uint32_t sum = b + in.off, cin = 0;
        for (int i = 0; i < 4; i++) {
            uint32_t t = B(b, i) + B(in.off, i) + cin;
            cin = t >> 8;
            s.put(U + i, t & 0xff);
            s.put(U + 4 + i, cin);
        }
        // [RANGE, 0, s1, s2] and [ADDR, s0 & 3, s0, s3]: byte ranges, top byte below 0x38, byte offset (u[21..23] one-hot)
        s.byte(B_RANGE - 1, (B(sum, 1) << 8) | B(sum, 2));
        s.byte(B_ADDR - 1, (B(sum, 0) << 8) | B(sum, 3));
        if (sum & 3) s.put(U + 20 + (sum & 3), 1);
        if (F(F_JALR)) {
            s.put(U + 8, sum & 1);
            // link value a = pc + 4: its top byte is below 0x78 (comparator slot: u[19] = 1 = "u[10] < u[20]")
            s.put(U + 19, 1); s.put(U + 10, B(a, 3)); s.put(U + 20, LINK_TOP_BYTE);
            s.byte(B_LTU - 1, (B(a, 3) << 8) | LINK_TOP_BYTE);
        } else {
            // the access moves the whole aligned word; stores patch it
            const uint32_t o = sum & 3, sh8 = 8 * o;
            uint32_t m_val = r.m_prev;
            if (F(F_SW)) m_val = c;
            else if (F(F_SB)) m_val = (r.m_prev & ~(0xffu << sh8)) | ((c & 0xffu) << sh8);
            else if (F(F_SH)) m_val = (r.m_prev & ~(0xffffu << sh8)) | ((c & 0xffffu) << sh8);
            mem_port(m_val);
            if (F(F_LB) | F(F_LH)) {
                const uint32_t sbyte = F(F_LB) ? B(a, 0) : B(a, 1);
                s.put(U + 24, sbyte);
                s.put(U + 25, sbyte >> 7);
                s.byte(B_MSB - 1, sbyte << 8);
            }
        }
    } else if (F(F_JAL)) {
        s.put(U + 19, 1); s.put(U + 10, B(a, 3)); s.put(U + 20, LINK_TOP_BYTE);   // as for JALR: the link value's top byte
        s.byte(B_LTU - 1, (B(a, 3) << 8) | LINK_TOP_BYTE);
    } else if (F(F_ECALL)) {
        // u[4] is_halt, u[5] 1/id, u[6] is_commit, u[7] 1/(id - COMMIT), u[24] is HINT_LEN (the one call that returns a value
        // in t0), u[25] 1/(id - HINT_LEN); u[1] is_pre (byte 1 of the code = 1: the call has a precompile chip), u[2] 1/(byte 1 - 1).
        // COMMIT and precompile rows ("sys rows") read a1 (x11) through the memory port and send on the sys bus; u[3] / u[21] =
        // clk / shard on precompile rows; u[0] balances the port's address expression u0 + 256 u1 + .. - (u21 + 2 u22 + 3 u23) = 11
        // the id as the AIR compares it: b0 + 256 b1 + 65536 b2 + 2^22 b3 (< p, and equal to a one-byte id only for (id, 0, 0, 0))
        const uint32_t idc = B(b, 0) + 256u * B(b, 1) + 65536u * B(b, 2) + (B(b, 3) << 22);
        s.put(U + 24, idc == SYS_HINT_LEN);
        if (idc != SYS_HINT_LEN) s.put(U + 25, inv(Fp::from_canonical(idc) - Fp::from_canonical(SYS_HINT_LEN)).canonical());
        s.put(U + 4, idc == 0);
        if (idc) s.put(U + 5, inv(Fp::from_canonical(idc)).canonical());
        const bool is_commit = idc == SYS_COMMIT, is_pre = B(b, 1) == 1;
        s.put(U + 6, is_commit);
        if (!is_commit) s.put(U + 7, inv(Fp::from_canonical(idc) - Fp::from_canonical(SYS_COMMIT)).canonical());
        const uint32_t pre_inv = is_pre ? 0u : inv(Fp::from_canonical(B(b, 1)) - Fp::one()).canonical();
        s.put(U + 1, is_pre); s.put(U + 2, pre_inv);
        s.put(RV32_CPU_sys_m, is_commit || is_pre);
        if (is_commit || is_pre) {
            const uint32_t u_clk = is_pre ? clk : 0u, u_sh = is_pre ? shard : 0u;
            s.put(U + 3, u_clk); s.put(U + 21, u_sh);
            // the port's address expression u0 + 256 u1 + 65536 u2 + 2^24 u3 - (u21 + 2 u22 + 3 u23) must be REG_BASE + 11
            const Fp u0 = Fp::from_canonical(REG_BASE + REG_A1) - Fp::from_canonical(256u * is_pre) - Fp::from_canonical(65536) * Fp::from_canonical(pre_inv) -
                          Fp::from_canonical(1u << 24) * Fp::from_canonical(u_clk) + Fp::from_canonical(u_sh);
            s.put(U + 0, u0.canonical());
            mem_port(r.m_prev);
        }
    }
    s.byte(B_RANGE - 1, (pb_hi << 8) | pc_hi);
    s.byte(B_RANGE - 1, (pa_hi << 8) | m_hi);
    if (F(F_ADD) | F(F_SUB) | F(F_MUL) | F(F_MULHU) | F(F_ECALL) | F(F_JAL) | F(F_JALR)) {
        s.byte(B_RANGE - 1, (B(a, 0) << 8) | B(a, 1));
        s.byte(B_RANGE - 1, (B(a, 2) << 8) | B(a, 3));
    }
}

// Host-side trace bundle of one shard (canonical, column-major), for the debug C-ABI and tests.
// present[c] = chip c is part of this shard (mem_init only in the last one).
struct HostTraces {
    uint32_t log_n[N_CHIPS];
    bool present[N_CHIPS];
    std::vector<uint32_t> main[N_CHIPS];
    std::vector<uint32_t> pubs;   // start_pc, next_pc, exit_code, shard, is_last
};
// preprocessed traces (program, byte, mem_image) for setup
struct HostPrep {
    uint32_t log_n[N_CHIPS];
    std::vector<uint32_t> prep[N_CHIPS];
};
void build_prep(const Program &prog, HostPrep *out);
// Everything of a shard except the cpu trace and the lookups the cpu rows make: shift / muldiv rows, in the last
// shard (mem_rows != nullptr) the mem_init rows, the byte-table multiplicities those cause, zeroed program / mem_image
// columns, public values.  The device path (K0) adds the cpu chip and its lookup counts on top.
struct ShardMeta {
    uint32_t index, start_pc, next_pc;
    size_t n_recs;
};
// the precompile calls of a shard grouped by the chip that proves them, in call order: what the GPU builds the rows of those
// chips from (launch_k0_bigop_rows)
struct BigOpBatches {
    std::vector<BigOpEvent> ev[N_CHIPS];
    // likewise the shift instructions of the shard and (last shard) the rows of the mem_init table: K0 of those chips runs
    // on the GPU too (launch_k0_shift_rows / launch_k0_mem_init_rows)
    std::vector<AluEvent> shifts;
    const std::vector<MemInitRow> *mem_rows = nullptr;
};
// device_rows != nullptr (the product path): the field / curve precompile chips get their shape only (present, log_n) and
// their calls are handed back in *device_rows; nullptr (CPU-only debug / test entry points): their rows are built here
bool build_aux_host(const ShardMeta &meta, const std::vector<AluEvent> &alu, const std::vector<ShaExtEvent> &sha_ext,
                    const std::vector<ShaCmpEvent> &sha_cmp, const std::vector<BigOpEvent> &big, const std::vector<MemInitRow> *mem_rows, int exit_code,
                    const HostPrep &prep, HostTraces *out, std::string *err, BigOpBatches *device_rows = nullptr);
// the rows of the fp_op / fp2_op / bls_g1 / secp_k1 / u256_mul chips for the calls of one shard (rv32_bigops.hip);
// byte_mult[op][65536] receives the byte-table lookups those rows make
bool build_bigop_traces(const std::vector<BigOpEvent> &big, uint32_t shard, HostTraces *out, uint32_t *byte_mult, std::string *err,
                        BigOpBatches *device_rows = nullptr);
// the whole shard on the host, cpu chip included (debug C-ABI, tests)
bool build_traces_host(const Program &prog, const ExecResult &res, size_t shard_pos, const HostPrep &prep, HostTraces *out, std::string *err);
// instruction index -> row of the program table (provable instructions only; others map to row 0 and never occur)
std::vector<uint32_t> program_row_map(const Program &prog);

#if defined(__HIPCC__)
hipError_t launch_k0_cpu_rows(hipStream_t st, const CycleRec *d_recs, size_t n_recs, uint32_t shard, uint32_t shard_next_pc, const Instr *d_instrs,
                              const uint32_t *d_prog_row, uint32_t *d_cpu, uint32_t log_n, uint32_t *d_byte_mult, uint32_t *d_prog_mult);
// K0 of a precompile chip: row i of the zeroed column-major trace d_main [W][2^log_n] (canonical words) from call d_ev[i], the
// rows' byte-table lookups added to d_byte_mult (plain counts); *d_err receives the largest row error code (0 = none)
hipError_t launch_k0_bigop_rows(hipStream_t st, int chip, const BigOpEvent *d_ev, uint32_t n_ev, uint32_t shard, uint32_t *d_main, uint32_t log_n,
                                uint32_t *d_byte_mult, uint32_t *d_err);
const char *bigop_row_error_text(uint32_t code);
// K0 of the shift / mem_init chips: row i of the zeroed column-major trace (Montgomery words written directly) from event i,
// byte-table lookups added to d_byte_mult (plain counts) through the same workgroup LDS cache as K0 of the cpu chip
hipError_t launch_k0_shift_rows(hipStream_t st, const AluEvent *d_ev, size_t n_ev, uint32_t *d_main, uint32_t log_n, uint32_t *d_byte_mult);
hipError_t launch_k0_mem_init_rows(hipStream_t st, const MemInitRow *d_rows, size_t n_rows, uint32_t *d_main, uint32_t log_n, uint32_t *d_byte_mult);
#endif

}  // namespace rv32
}  // namespace dvt
