// RV32IM ELF loader, decoder, interpreter and host-side trace assembly (see rv32.h).
#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <unordered_map>

#include "rv32.h"

namespace dvt {
namespace rv32 {
static const uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};


// ------------------------------------------------------------------ decode
static inline uint32_t bits(uint32_t x, int hi, int lo) { return (x >> lo) & ((1u << (hi - lo + 1)) - 1); }
static inline uint32_t sext(uint32_t v, int b) { return (v & (1u << (b - 1))) ? v | ~((1u << b) - 1) : v; }
#define FL(x) (1u << (x))

static Instr decode(uint32_t w, uint32_t pc) {
    Instr in{};
    in.pc = pc;
    in.raw = w;
    in.supported = 1;
    const uint32_t op = w & 0x7f, rd = bits(w, 11, 7), f3 = bits(w, 14, 12), rs1 = bits(w, 19, 15), rs2 = bits(w, 24, 20), f7 = bits(w, 31, 25);
    const uint32_t imm_i = sext(bits(w, 31, 20), 12);
    const uint32_t imm_s = sext((bits(w, 31, 25) << 5) | bits(w, 11, 7), 12);
    const uint32_t imm_b = sext((bits(w, 31, 31) << 12) | (bits(w, 7, 7) << 11) | (bits(w, 30, 25) << 5) | (bits(w, 11, 8) << 1), 13);
    const uint32_t imm_u = w & 0xfffff000u;
    const uint32_t imm_j = sext((bits(w, 31, 31) << 20) | (bits(w, 19, 12) << 12) | (bits(w, 20, 20) << 11) | (bits(w, 30, 21) << 1), 21);
    auto set_rd = [&] { if (rd) { in.rd = rd; in.flags |= FL(F_RD_EN); } };
    auto set_rs1 = [&] { in.rs1 = rs1; in.flags |= FL(F_RS1_EN); };
    auto set_rs2 = [&] { in.rs2 = rs2; in.flags |= FL(F_RS2_EN); };
    switch (op) {
    case 0x37: set_rd(); in.imm = imm_u; in.flags |= FL(F_LUI); break;
    case 0x17: set_rd(); in.imm = pc + imm_u; in.flags |= FL(F_LUI); break;  // AUIPC folded at decode time
    case 0x6f: set_rd(); in.tgt = pc + imm_j; in.flags |= FL(F_JAL); break;    // (the link value pc + 4 is constrained, not tabulated)
    case 0x67:
        if (f3 != 0) { in.supported = 0; break; }
        set_rd(); set_rs1(); in.off = imm_i; in.flags |= FL(F_JALR);
        break;
    case 0x63: {
        static const int map[8] = {F_BEQ, F_BNE, -1, -1, F_BLT, F_BGE, F_BLTU, F_BGEU};
        if (map[f3] < 0) { in.supported = 0; break; }
        set_rs1(); set_rs2(); in.tgt = pc + imm_b; in.flags |= FL(map[f3]);
        break;
    }
    case 0x03: {
        set_rd(); set_rs1(); in.off = imm_i;
        static const int map[8] = {F_LB, F_LH, F_LW, -1, F_LBU, F_LHU, -1, -1};
        if (map[f3] < 0) in.supported = 0; else in.flags |= FL(map[f3]);
        break;
    }
    case 0x23: {
        set_rs1(); set_rs2(); in.off = imm_s;
        static const int map[8] = {F_SB, F_SH, F_SW, -1, -1, -1, -1, -1};
        if (map[f3] < 0) in.supported = 0; else in.flags |= FL(map[f3]);
        break;
    }
    case 0x13: {
        set_rd(); set_rs1(); in.imm = imm_i; in.flags |= FL(F_IMM_C);
        static const int map[8] = {F_ADD, -1, F_SLT, F_SLTU, F_XOR, -1, F_OR, F_AND};
        if (map[f3] >= 0) in.flags |= FL(map[f3]);
        else {  // SLLI / SRLI / SRAI: the shift chip, shift amount as the immediate operand
            in.imm = rs2;
            in.flags |= FL(F_ALU);
            if (f3 == 1 && f7 == 0x00) in.alu_op = ALU_SLL;
            else if (f3 == 5 && f7 == 0x00) in.alu_op = ALU_SRL;
            else if (f3 == 5 && f7 == 0x20) in.alu_op = ALU_SRA;
            else in.supported = 0;
        }
        break;
    }
    case 0x33: {
        set_rd(); set_rs1(); set_rs2();
        int fam = -1;
        if (f7 == 0x00 && (f3 == 1 || f3 == 5)) { fam = F_ALU; in.alu_op = f3 == 1 ? ALU_SLL : ALU_SRL; }
        else if (f7 == 0x20 && f3 == 5) { fam = F_ALU; in.alu_op = ALU_SRA; }
        else if (f7 == 0x00) { static const int m0[8] = {F_ADD, -1, F_SLT, F_SLTU, F_XOR, -1, F_OR, F_AND}; fam = m0[f3]; }
        else if (f7 == 0x20) { fam = f3 == 0 ? F_SUB : -1; }
        else if (f7 == 0x01) {  // M extension: MUL / MULHU in the cpu chip, the rest in the muldiv chip
            static const uint32_t mop[8] = {0, ALU_MULH, ALU_MULHSU, 0, ALU_DIV, ALU_DIVU, ALU_REM, ALU_REMU};
            if (f3 == 0) fam = F_MUL;
            else if (f3 == 3) fam = F_MULHU;
            else { fam = F_ALU; in.alu_op = mop[f3]; }
        }
        if (fam < 0) in.supported = 0; else in.flags |= FL(fam);
        break;
    }
    case 0x73:
        if (w == 0x00000073u) {  // ecall: a (t0) <- advice, b = t0, c = a0
            in.rd = 5; in.rs1 = 5; in.rs2 = 10;
            in.flags |= FL(F_ECALL) | FL(F_RD_EN) | FL(F_RS1_EN) | FL(F_RS2_EN);
        } else in.supported = 0;
        break;
    case 0x0f: in.supported = 0; break;  // fence
    default: in.supported = 0; break;
    }
    // dispatch code of the interpreter
    const uint32_t fl = in.flags;
    const bool immf = fl & FL(F_IMM_C);
    uint8_t k = K_UNSUP;
    if (!in.supported) k = K_UNSUP;
    else if (fl & FL(F_ADD)) k = immf ? K_ADDI : K_ADD;
    else if (fl & FL(F_SUB)) k = K_SUB;
    else if (fl & FL(F_AND)) k = immf ? K_ANDI : K_AND;
    else if (fl & FL(F_OR)) k = immf ? K_ORI : K_OR;
    else if (fl & FL(F_XOR)) k = immf ? K_XORI : K_XOR;
    else if (fl & FL(F_SLT)) k = immf ? K_SLTI : K_SLT;
    else if (fl & FL(F_SLTU)) k = immf ? K_SLTIU : K_SLTU;
    else if (fl & FL(F_MUL)) k = K_MUL;
    else if (fl & FL(F_MULHU)) k = K_MULHU;
    else if (fl & FL(F_LUI)) k = K_LUI;
    else if (fl & FL(F_JAL)) k = K_JAL;
    else if (fl & FL(F_JALR)) k = K_JALR;
    else if (fl & FL(F_BEQ)) k = K_BEQ;
    else if (fl & FL(F_BNE)) k = K_BNE;
    else if (fl & FL(F_BLT)) k = K_BLT;
    else if (fl & FL(F_BGE)) k = K_BGE;
    else if (fl & FL(F_BLTU)) k = K_BLTU;
    else if (fl & FL(F_BGEU)) k = K_BGEU;
    else if (fl & FL(F_LW)) k = K_LW;
    else if (fl & FL(F_LB)) k = K_LB;
    else if (fl & FL(F_LBU)) k = K_LBU;
    else if (fl & FL(F_LH)) k = K_LH;
    else if (fl & FL(F_LHU)) k = K_LHU;
    else if (fl & FL(F_SW)) k = K_SW;
    else if (fl & FL(F_SB)) k = K_SB;
    else if (fl & FL(F_SH)) k = K_SH;
    else if (fl & FL(F_ALU)) k = immf ? K_ALU_I : K_ALU_R;
    else if (fl & FL(F_ECALL)) k = K_ECALL;
    in.kind = k;
    return in;
}

// ------------------------------------------------------------------ ELF
static uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

bool load_elf(const uint8_t *elf, size_t n, Program *out, std::string *err) {
    auto bad = [&](const char *m) { if (err) *err = m; return false; };
    if (!elf || n < 52 || memcmp(elf, "\x7f" "ELF", 4) != 0) return bad("not an ELF file");
    if (elf[4] != 1 || elf[5] != 1) return bad("not ELF32 little-endian");
    if (rd16(elf + 18) != 243) return bad("not a RISC-V ELF");
    Program p;
    p.entry = rd32(elf + 24);
    uint32_t phoff = rd32(elf + 28), phentsize = rd16(elf + 42), phnum = rd16(elf + 44);
    if (phentsize < 32 || (uint64_t)phoff + (uint64_t)phentsize * phnum > n) return bad("bad program headers");
    std::map<uint32_t, uint32_t> image;
    bool have_text = false;
    for (uint32_t r = 0; r < 32; r++) image[REG_BASE + r] = 0;  // registers: words REG_BASE + 0..31 of the memory argument, initial value 0
    for (uint32_t i = 0; i < phnum; i++) {
        const uint8_t *ph = elf + phoff + (size_t)i * phentsize;
        if (rd32(ph) != 1) continue;  // PT_LOAD
        uint32_t off = rd32(ph + 4), vaddr = rd32(ph + 8), filesz = rd32(ph + 16), memsz = rd32(ph + 20), flags = rd32(ph + 24);
        if ((uint64_t)off + filesz > n || filesz > memsz) return bad("segment outside file");
        if (vaddr % 4 || vaddr < 32 || (uint64_t)vaddr + memsz > ADDR_LIMIT) return bad("segment address not word aligned or out of range (< 2^30)");
        for (uint32_t k = 0; k < memsz; k += 4) {
            uint32_t w = 0;
            for (uint32_t b = 0; b < 4; b++)
                if (k + b < filesz) w |= (uint32_t)elf[off + k + b] << (8 * b);
            image[vaddr + k] = w;
        }
        if (flags & 1) {  // executable
            if (have_text) return bad("more than one executable segment");
            have_text = true;
            p.text_base = vaddr;
            uint32_t words = (filesz + 3) / 4;
            p.instrs.resize(words + 1);
            for (uint32_t k = 0; k < words; k++) p.instrs[k] = decode(image[vaddr + 4 * k], vaddr + 4 * k);
            // sentinel: where sequential flow past the end and every static target outside the text land
            Instr oob{};
            oob.pc = vaddr + 4 * words; oob.kind = K_OOB; oob.supported = 0; oob.tgt_idx = words;
            p.instrs[words] = oob;
            for (uint32_t k = 0; k < words; k++) {
                Instr &in = p.instrs[k];
                const uint32_t t = in.tgt - vaddr;
                in.tgt_idx = (in.tgt >= vaddr && t % 4 == 0 && t / 4 < words) ? t / 4 : words;
                in.tgt_raw = in.tgt;
                // a static target outside the text traps in the executor; in the program table it becomes a value that is
                // no row's pc and not HALT_PC (a raw 32-bit target could alias a valid one mod p)
                const bool cf = in.flags & (FL(F_JAL) | FL(F_BEQ) | FL(F_BNE) | FL(F_BLT) | FL(F_BGE) | FL(F_BLTU) | FL(F_BGEU));
                if (cf && in.tgt_idx == words) in.tgt = BAD_PC;
            }
        }
    }
    if (!have_text) return bad("no executable segment");
    if (p.entry < p.text_base || p.entry >= p.text_base + 4 * (p.instrs.size() - 1) || p.entry % 4) return bad("entry point outside text");
    p.image.assign(image.begin(), image.end());
    *out = std::move(p);
    return true;
}

// ------------------------------------------------------------------ interpreter
static constexpr uint32_t FL_TOUCHED = 1, FL_IMG = 2;

Vm::Vm(const Program &p, const std::vector<std::vector<uint8_t>> *in, uint32_t ls)
    : prog(p), pc(p.entry), log_shard(ls), stdin_bufs(in), holders(N_PAGES), raw(N_PAGES, nullptr), own(N_PAGES, 0) {
    for (auto &c : regs) c = Cell{0, FL_IMG, 0};
    for (auto &kv : prog.image)
        if (kv.first < REG_BASE) { Cell &c = at(kv.first); c.val = kv.second; c.flags = FL_IMG; }
}

Vm::Vm(const Program &p, const std::vector<std::vector<uint8_t>> *in, uint32_t ls, const Snapshot &s)
    : prog(p), pc(s.pc), shard(s.shard), log_shard(ls), cycles(s.cycles), collect_output(false), stdin_bufs(in), next_input(s.next_input),
      holders(N_PAGES), raw(N_PAGES, nullptr), own(N_PAGES, 0) {
    curve_index = s.curve_index;
    for (int i = 0; i < 32; i++) regs[i] = s.regs[i];
    for (auto &kv : s.pages) { holders[kv.first] = kv.second; raw[kv.first] = kv.second.get(); alloc_pages.push_back(kv.first); }
}

void Vm::make_own(uint32_t pg) {
    std::shared_ptr<Page> np;
    if (holders[pg]) np = std::make_shared<Page>(*holders[pg]);   // shared with a snapshot: copy on first access
    else { np = std::make_shared<Page>(); memset(np.get(), 0, sizeof(Page)); alloc_pages.push_back(pg); }
    holders[pg] = np;
    raw[pg] = np.get();
    own[pg] = 1;
    owned_pages.push_back(pg);
}

Snapshot Vm::snapshot() {
    Snapshot s;
    for (int i = 0; i < 32; i++) s.regs[i] = regs[i];
    s.pc = pc; s.shard = shard; s.cycles = cycles; s.next_input = next_input; s.curve_index = curve_index;
    s.pages.reserve(alloc_pages.size());
    for (uint32_t pg : alloc_pages) s.pages.emplace_back(pg, holders[pg]);
    for (uint32_t pg : owned_pages) own[pg] = 0;   // from now on shared with the snapshot
    owned_pages.clear();
    return s;
}

void Vm::trap(const std::string &m) {
    char b[24];
    snprintf(b, sizeof b, " at pc 0x%x", pc);
    error = m + b;
}

bool Vm::next_shard() {
    if (shard >= MAX_SHARDS) { error = "execution longer than 65535 shards"; return false; }
    shard++;
    in_shard = 0;
    return true;
}

void Vm::run_shard(bool trace, ShardOut *out, uint64_t max_total_cycles) {
    const uint64_t left_in_shard = ((uint64_t)1 << log_shard) - in_shard;
    const uint64_t left_total = max_total_cycles > cycles ? max_total_cycles - cycles : 0;
    const uint64_t budget = left_in_shard < left_total ? left_in_shard : left_total;
    if (trace) {
        out->index = shard;
        out->start_pc = pc;
        out->n_recs = 0;
        out->alu.clear();
        out->sha_ext.clear();
        out->sha_cmp.clear();
        out->big.clear();
        run<true>(out, budget);
        out->next_pc = pc;   // HALT_PC after HALT
    } else {
        run<false>(nullptr, budget);
    }
    if (error.empty() && !halted && cycles >= max_total_cycles) error = "cycle limit reached before HALT";
}

// Threaded interpreter: every instruction form has its own handler that ends in its own indirect jump to the next
// handler (one predictor entry per form instead of one shared dispatch branch), the text is walked by instruction
// pointer (static targets are pre-resolved to indices, a K_OOB sentinel after the last instruction catches control
// leaving the text), and an access stamps its (shard, clk) pair with one 64-bit store.
template <bool TRACE>
void Vm::run(ShardOut *out, uint64_t budget) {
    const Instr *const code = prog.instrs.data();
    const uint32_t ninstr = (uint32_t)prog.instrs.size() - 1, text_base = prog.text_base;
    const uint64_t sh64 = (uint64_t)shard << 32;
    CycleRec *recs = TRACE ? out->recs + out->n_recs : nullptr;
    uint64_t done = 0;
    uint32_t clk = 4 * (in_shard + 1);
    const Instr *ip;
    {
        const uint32_t idx0 = (pc - text_base) >> 2;
        ip = code + ((pc >= text_base && !(pc & 3) && idx0 < ninstr) ? idx0 : ninstr);
    }
    const Instr *nip = ip, *prev = nullptr;
    uint32_t jump_pc = 0;
    CycleRec rec;
    uint32_t sha = 0, shb = 0, shc = 0, shm = 0, a = 0, b = 0, c = 0, addr = 0, s8 = 0;
    Cell *cellp = nullptr;
    const char *why = nullptr;
    static const void *const handlers[N_KINDS] = {
        &&L_UNSUP, &&L_ADD, &&L_SUB, &&L_AND, &&L_OR, &&L_XOR, &&L_SLT, &&L_SLTU, &&L_MUL, &&L_MULHU,
        &&L_ADDI, &&L_ANDI, &&L_ORI, &&L_XORI, &&L_SLTI, &&L_SLTIU,
        &&L_LUI, &&L_JAL, &&L_JALR, &&L_BEQ, &&L_BNE, &&L_BLT, &&L_BGE, &&L_BLTU, &&L_BGEU,
        &&L_LW, &&L_LB, &&L_LBU, &&L_LH, &&L_LHU, &&L_SW, &&L_SB, &&L_SH, &&L_ALU, &&L_ALU, &&L_ECALL, &&L_OOB};
    // ports in time order: c (rs2) at clk, b (rs1) at clk + 1, memory at clk + 2, a (rd) at clk + 3; each access leaves
    // its (shard, clk) on the word and, in trace mode, records the pair it found there
#define IN (*ip)
#define PORT_C() do { Cell &r2_ = regs[IN.rs2]; c = r2_.val; if (TRACE) { rec.pc_ts = r2_.ts(); shc = r2_.sh(); } r2_.tsh = sh64 | clk; } while (0)
#define PORT_B() do { Cell &r1_ = regs[IN.rs1]; b = r1_.val; if (TRACE) { rec.pb_ts = r1_.ts(); shb = r1_.sh(); } r1_.tsh = sh64 | (clk + 1); } while (0)
#define PORT_A(v) do { if (IN.rd) { Cell &rd_ = regs[IN.rd]; if (TRACE) { rec.pa_prev = rd_.val; rec.pa_ts = rd_.ts(); sha = rd_.sh(); } rd_.val = (v); rd_.tsh = sh64 | (clk + 3); } } while (0)
#define MEM(align_mask, what) do { addr = b + IN.off; \
        if (__builtin_expect(addr < 32 || addr >= ADDR_LIMIT, 0)) { why = "memory access out of range"; goto trapped; } \
        if (__builtin_expect(addr & (align_mask), 0)) { why = what; goto trapped; } \
        cellp = &at(addr & ~3u); \
        if (!(cellp->flags & (FL_TOUCHED | FL_IMG))) { cellp->flags |= FL_TOUCHED; if (collect_output) first_touch.emplace_back(addr & ~3u, cellp->val); } \
        if (TRACE) { rec.m_prev = cellp->val; rec.m_ts = cellp->ts(); shm = cellp->sh(); } \
        cellp->tsh = sh64 | (clk + 2); s8 = 8 * (addr & 3); } while (0)
#define BEGIN() do { if (TRACE) { rec.idx = (uint32_t)(ip - code); rec.pa_prev = rec.pa_ts = rec.pb_ts = rec.pc_ts = rec.m_prev = rec.m_ts = 0; sha = shb = shc = shm = 0; } } while (0)
#define NEXT() do { \
        if (TRACE) { rec.a = a; rec.b = b; rec.c = c; rec.sh_ab = sha | (shb << 16); rec.sh_cm = shc | (shm << 16); recs[done] = rec; } \
        done++; clk += 4; prev = ip; ip = nip; \
        if (__builtin_expect(done >= budget, 0)) goto out_of_budget; \
        goto *handlers[ip->kind]; } while (0)
#define SEQ() do { nip = ip + 1; NEXT(); } while (0)
#define RRR(expr) do { BEGIN(); PORT_C(); PORT_B(); a = (expr); PORT_A(a); SEQ(); } while (0)
#define RRI(expr) do { BEGIN(); c = IN.imm; PORT_B(); a = (expr); PORT_A(a); SEQ(); } while (0)
#define BRANCH(cond) do { BEGIN(); PORT_C(); PORT_B(); a = 0; nip = (cond) ? code + IN.tgt_idx : ip + 1; NEXT(); } while (0)
    if (budget == 0) goto out_of_budget;
    goto *handlers[ip->kind];
L_ADD: RRR(b + c);
L_SUB: RRR(b - c);
L_AND: RRR(b & c);
L_OR: RRR(b | c);
L_XOR: RRR(b ^ c);
L_SLT: RRR((uint32_t)((int32_t)b < (int32_t)c));
L_SLTU: RRR((uint32_t)(b < c));
L_MUL: RRR(b * c);
L_MULHU: RRR((uint32_t)(((uint64_t)b * c) >> 32));
L_ADDI: RRI(b + c);
L_ANDI: RRI(b & c);
L_ORI: RRI(b | c);
L_XORI: RRI(b ^ c);
L_SLTI: RRI((uint32_t)((int32_t)b < (int32_t)c));
L_SLTIU: RRI((uint32_t)(b < c));
L_LUI: BEGIN(); b = c = 0; a = IN.imm; PORT_A(a); SEQ();
L_JAL: BEGIN(); b = c = 0; a = IN.pc + 4; PORT_A(a); nip = code + IN.tgt_idx; NEXT();
L_JALR: {
    BEGIN(); c = 0; PORT_B();
    a = IN.pc + 4;
    const uint32_t t = b + IN.off;
    if (__builtin_expect(t >= ADDR_LIMIT, 0)) { why = "jump target out of range"; goto trapped; }
    const uint32_t tp = t & ~1u, ti = (tp - text_base) >> 2;
    PORT_A(a);
    nip = code + ((tp >= text_base && !(tp & 3) && ti < ninstr) ? ti : ninstr);
    if (nip == code + ninstr) jump_pc = tp;   // (reported by the sentinel's trap)
    NEXT();
}
L_BEQ: BRANCH(b == c);
L_BNE: BRANCH(b != c);
L_BLT: BRANCH((int32_t)b < (int32_t)c);
L_BGE: BRANCH((int32_t)b >= (int32_t)c);
L_BLTU: BRANCH(b < c);
L_BGEU: BRANCH(b >= c);
L_LW: BEGIN(); c = 0; PORT_B(); MEM(3, "misaligned word access"); a = cellp->val; PORT_A(a); SEQ();
L_LB: BEGIN(); c = 0; PORT_B(); MEM(0, ""); a = sext((cellp->val >> s8) & 0xff, 8); PORT_A(a); SEQ();
L_LBU: BEGIN(); c = 0; PORT_B(); MEM(0, ""); a = (cellp->val >> s8) & 0xff; PORT_A(a); SEQ();
L_LH: BEGIN(); c = 0; PORT_B(); MEM(1, "misaligned halfword access"); a = sext((cellp->val >> s8) & 0xffff, 16); PORT_A(a); SEQ();
L_LHU: BEGIN(); c = 0; PORT_B(); MEM(1, "misaligned halfword access"); a = (cellp->val >> s8) & 0xffff; PORT_A(a); SEQ();
L_SW: BEGIN(); PORT_C(); PORT_B(); MEM(3, "misaligned word access"); a = 0; cellp->val = c; SEQ();
L_SB: BEGIN(); PORT_C(); PORT_B(); MEM(0, ""); a = 0; cellp->val = (cellp->val & ~(0xffu << s8)) | ((c & 0xff) << s8); SEQ();
L_SH: BEGIN(); PORT_C(); PORT_B(); MEM(1, "misaligned halfword access"); a = 0; cellp->val = (cellp->val & ~(0xffffu << s8)) | ((c & 0xffff) << s8); SEQ();
L_ALU: {
    BEGIN();
    if (IN.kind == K_ALU_R) PORT_C(); else c = IN.imm;
    PORT_B();
    const uint32_t s5 = c & 31;
    const int32_t sb = (int32_t)b, sc = (int32_t)c;
    switch (IN.alu_op) {
    case ALU_SLL: a = b << s5; break;
    case ALU_SRL: a = b >> s5; break;
    case ALU_SRA: a = (uint32_t)(sb >> s5); break;
    case ALU_MULH: a = (uint32_t)(((int64_t)sb * (int64_t)sc) >> 32); break;
    case ALU_MULHSU: a = (uint32_t)(((int64_t)sb * (int64_t)(uint64_t)c) >> 32); break;
    // RISC-V: x / 0 = all ones, x % 0 = x; -2^31 / -1 = -2^31 remainder 0
    case ALU_DIV: a = c == 0 ? 0xffffffffu : (b == 0x80000000u && c == 0xffffffffu) ? b : (uint32_t)(sb / sc); break;
    case ALU_DIVU: a = c == 0 ? 0xffffffffu : b / c; break;
    case ALU_REM: a = c == 0 ? b : (b == 0x80000000u && c == 0xffffffffu) ? 0u : (uint32_t)(sb % sc); break;
    default: a = c == 0 ? b : b % c; break;  // ALU_REMU
    }
    if (TRACE) out->alu.push_back(AluEvent{IN.alu_op, a, b, c});
    PORT_A(a);
    SEQ();
}
L_ECALL: {
    // b = t0 (id), c = a0; a1 / a2 are read without a port except by COMMIT
    BEGIN(); PORT_C(); PORT_B();
    const uint32_t a1 = regs[11].val, a2 = regs[12].val;
    a = b;  // t0 unchanged unless the call returns a value
    nip = ip + 1;
    switch (b) {
    case 0x00:
        if (c >> 24) { why = "HALT with an exit code of 2^24 or more"; goto trapped; }
        halted = true; exit_code = (int)c; break;
    case 0x02: {  // WRITE(fd = a0, ptr = a1, len = a2): fd 3 = the public-value stream
        if ((uint64_t)a1 + a2 > ADDR_LIMIT) { why = "WRITE buffer out of range"; goto trapped; }
        if (!collect_output) break;
        std::vector<uint8_t> &dst = c == 3 ? public_values : stdout_bytes;
        for (uint32_t k = 0; k < a2; k++) {
            const uint32_t ad = a1 + k;
            const Cell *cell = ad >= 32 ? peek(ad & ~3u) : nullptr;
            dst.push_back(cell ? (uint8_t)(cell->val >> (8 * (ad & 3))) : 0);
        }
        break;
    }
    case SYS_COMMIT: {  // COMMIT(a0 = index, a1 = word): a1 is read through the memory port
        Cell &r11 = regs[REG_A1];
        if (TRACE) { rec.m_prev = r11.val; rec.m_ts = r11.ts(); shm = r11.sh(); }
        r11.tsh = sh64 | (clk + 2);
        if (c < 8) { committed[c] = a1; committed_mask |= 1u << c; }
        break;
    }
    case SYS_SHA_EXTEND: {  // SHA_EXTEND(a0 = w): w[16..63] of the SHA-256 message schedule, in place (precompile chip sha_extend)
        if (a1 != 0) { why = "SHA_EXTEND with a1 != 0"; goto trapped; }
        if (c % 4 || c < 32 || (uint64_t)c + 256 > ADDR_LIMIT) { why = "SHA_EXTEND pointer misaligned or out of range"; goto trapped; }
        {   // a1 (x11) is read through the memory port, as for COMMIT
            Cell &r11 = regs[REG_A1];
            if (TRACE) { rec.m_prev = r11.val; rec.m_ts = r11.ts(); shm = r11.sh(); }
            r11.tsh = sh64 | (clk + 2);
        }
        ShaExtEvent *ev = nullptr;
        if (TRACE) { out->sha_ext.emplace_back(); ev = &out->sha_ext.back(); ev->clk = clk; ev->ptr = c; }
        uint32_t w[64];
        for (uint32_t k = 0; k < 64; k++) {   // every word of the array is accessed exactly once, at (shard, clk + 2)
            Cell &cell = at(c + 4 * k);
            if (!(cell.flags & (FL_TOUCHED | FL_IMG))) { cell.flags |= FL_TOUCHED; if (collect_output) first_touch.emplace_back(c + 4 * k, cell.val); }
            if (k < 16) w[k] = cell.val;
            else {
                const uint32_t x = w[k - 15], y = w[k - 2];
                const uint32_t s0 = ((x >> 7) | (x << 25)) ^ ((x >> 18) | (x << 14)) ^ (x >> 3);
                const uint32_t s1 = ((y >> 17) | (y << 15)) ^ ((y >> 19) | (y << 13)) ^ (y >> 10);
                w[k] = s1 + w[k - 7] + s0 + w[k - 16];
            }
            if (TRACE) { ev->old[k] = cell.val; ev->prev_ts[k] = cell.ts(); ev->prev_sh[k] = (uint16_t)cell.sh(); ev->w[k] = w[k]; }
            cell.val = w[k];
            cell.tsh = sh64 | (clk + 2);
        }
        break;
    }
    case SYS_SHA_COMPRESS: {  // SHA_COMPRESS(a0 = w, a1 = state): the SHA-256 compression function, state updated in place
        if (c % 4 || c < 32 || (uint64_t)c + 256 > ADDR_LIMIT || a1 % 4 || a1 < 32 || (uint64_t)a1 + 32 > ADDR_LIMIT) {
            why = "SHA_COMPRESS pointer misaligned or out of range"; goto trapped;
        }
        if (a1 + 32 > c && c + 256 > a1) { why = "SHA_COMPRESS arrays overlap"; goto trapped; }
        {
            Cell &r11 = regs[REG_A1];
            if (TRACE) { rec.m_prev = r11.val; rec.m_ts = r11.ts(); shm = r11.sh(); }
            r11.tsh = sh64 | (clk + 2);
        }
        ShaCmpEvent *ev = nullptr;
        if (TRACE) { out->sha_cmp.emplace_back(); ev = &out->sha_cmp.back(); ev->clk = clk; ev->w_ptr = c; ev->h_ptr = a1; }
        uint32_t w[64], hs[8];
        auto touch = [&](uint32_t ad) -> Cell & {
            Cell &cell = at(ad);
            if (!(cell.flags & (FL_TOUCHED | FL_IMG))) { cell.flags |= FL_TOUCHED; if (collect_output) first_touch.emplace_back(ad, cell.val); }
            return cell;
        };
        for (uint32_t k = 0; k < 8; k++) {    // the state is read at clk + 2 ...
            Cell &cell = touch(a1 + 4 * k);
            hs[k] = cell.val;
            if (TRACE) { ev->hs[k] = cell.val; ev->h_ts[k] = cell.ts(); ev->h_sh[k] = (uint16_t)cell.sh(); }
            cell.tsh = sh64 | (clk + 2);
        }
        for (uint32_t k = 0; k < 64; k++) {
            Cell &cell = touch(c + 4 * k);
            w[k] = cell.val;
            if (TRACE) { ev->w[k] = cell.val; ev->w_ts[k] = cell.ts(); ev->w_sh[k] = (uint16_t)cell.sh(); }
            cell.tsh = sh64 | (clk + 2);
        }
        uint32_t v[8];
        for (int k = 0; k < 8; k++) v[k] = hs[k];
        for (uint32_t i = 0; i < 64; i++) {
            const uint32_t e_ = v[4], a_ = v[0];
            const uint32_t S1 = ((e_ >> 6) | (e_ << 26)) ^ ((e_ >> 11) | (e_ << 21)) ^ ((e_ >> 25) | (e_ << 7));
            const uint32_t chv = (e_ & v[5]) ^ (~e_ & v[6]);
            const uint32_t t1 = v[7] + S1 + chv + SHA256_K[i] + w[i];
            const uint32_t S0 = ((a_ >> 2) | (a_ << 30)) ^ ((a_ >> 13) | (a_ << 19)) ^ ((a_ >> 22) | (a_ << 10));
            const uint32_t mj = (a_ & v[1]) ^ (a_ & v[2]) ^ (v[1] & v[2]);
            for (int k = 7; k > 0; k--) v[k] = v[k - 1];
            v[4] += t1;
            v[0] = t1 + S0 + mj;
        }
        for (uint32_t k = 0; k < 8; k++) {    // ... and written back at clk + 3
            Cell &cell = at(a1 + 4 * k);
            cell.val = hs[k] + v[k];
            cell.tsh = sh64 | (clk + 3);
        }
        break;
    }
    case 0x1a: break;  // COMMIT_DEFERRED_PROOFS: no-op (no recursion in core proofs)
    case 0xf0: a = next_input < stdin_bufs->size() ? (uint32_t)(*stdin_bufs)[next_input].size() : 0; break;
    case 0xf1: {  // HINT_READ(ptr = a0, len = a1): the words become initial memory (must be untouched so far)
        if (next_input >= stdin_bufs->size()) { why = "HINT_READ with no input left"; goto trapped; }
        const auto &buf = (*stdin_bufs)[next_input++];
        if (a1 != buf.size()) { why = "HINT_READ length mismatch"; goto trapped; }
        if (c % 4 || c < 32 || (uint64_t)c + a1 > ADDR_LIMIT) { why = "HINT_READ pointer misaligned or out of range"; goto trapped; }
        for (uint32_t k = 0; k < a1; k += 4) {
            Cell &cell = at(c + k);
            if (cell.flags || cell.tsh) { why = "HINT_READ into the program image or into memory that was already accessed"; goto trapped; }
            uint32_t wv = 0;
            for (uint32_t q = 0; q < 4 && k + q < a1; q++) wv |= (uint32_t)buf[k + q] << (8 * q);
            cell.val = wv;
        }
        break;
    }
    default: {
        // field / curve precompiles: a0 = operand replaced by the result, a1 = second operand (0 for DOUBLE)
        BigOpInfo bi;
        if (!bigop_info(b, &bi)) { why = "unknown syscall"; goto trapped; }
        if (c % 4 || c < 32 || (uint64_t)c + 4 * bi.words_a > ADDR_LIMIT) { why = "precompile operand pointer misaligned or out of range"; goto trapped; }
        if (bi.words_b ? (a1 % 4 || a1 < 32 || (uint64_t)a1 + 4 * bi.words_b > ADDR_LIMIT) : a1 != 0) {
            why = bi.words_b ? "precompile operand pointer misaligned or out of range" : "DOUBLE precompile with a1 != 0"; goto trapped;
        }
        BigOpEvent lev, *ev = &lev;
        auto touch = [&](uint32_t ad) -> Cell & {
            Cell &cell = at(ad);
            if (!(cell.flags & (FL_TOUCHED | FL_IMG))) { cell.flags |= FL_TOUCHED; if (collect_output) first_touch.emplace_back(ad, cell.val); }
            return cell;
        };
        // values first (nothing is stamped if the call traps), then the accesses: b at clk + 2, a at clk + 3
        for (int k = 0; k < bi.words_b; k++) { const Cell *cl = peek(a1 + 4 * k); lev.b[k] = cl ? cl->val : 0; }
        for (int k = 0; k < bi.words_a; k++) { const Cell *cl = peek(c + 4 * k); lev.a[k] = cl ? cl->val : 0; }
        if (curve_log && (bi.chip == RV32_CHIP_BLS_G1 || bi.chip == RV32_CHIP_SECP_K1)) {
            // (results are a function of the operands: whoever computes them, the fast pass or this one, they are the same)
            const size_t idx = curve_index++;
            if (TRACE) {
                if (!curve_log->fetch(idx, lev.r, lev.lam)) why = bigop_compute(b, lev.a, lev.b, lev.r, lev.lam);
            } else {
                why = bigop_compute(b, lev.a, lev.b, lev.r, lev.lam);
                if (!why) curve_log->append(idx, lev.r, lev.lam);
            }
        } else {
            why = bigop_compute(b, lev.a, lev.b, lev.r, lev.lam);
        }
        if (why) goto trapped;
        {
            Cell &r11 = regs[REG_A1];
            if (TRACE) { rec.m_prev = r11.val; rec.m_ts = r11.ts(); shm = r11.sh(); }
            r11.tsh = sh64 | (clk + 2);
        }
        if (TRACE) { out->big.emplace_back(lev); ev = &out->big.back(); ev->code = b; ev->clk = clk; ev->a_ptr = c; ev->b_ptr = a1; }
        for (int k = 0; k < bi.words_b; k++) {
            Cell &cell = touch(a1 + 4 * k);
            if (TRACE) { ev->b_ts[k] = cell.ts(); ev->b_sh[k] = (uint16_t)cell.sh(); }
            cell.tsh = sh64 | (clk + 2);
        }
        for (int k = 0; k < bi.words_a; k++) {
            Cell &cell = touch(c + 4 * k);
            if (TRACE) { ev->a_ts[k] = cell.ts(); ev->a_sh[k] = (uint16_t)cell.sh(); }
            cell.val = lev.r[k];
            cell.tsh = sh64 | (clk + 3);
        }
        break;
    }
    }
    PORT_A(a);
    if (halted) {
        if (TRACE) { rec.a = a; rec.b = b; rec.c = c; rec.sh_ab = sha | (shb << 16); rec.sh_cm = shc | (shm << 16); recs[done] = rec; }
        done++;
        pc = HALT_PC;
        goto finished;
    }
    NEXT();
}
L_UNSUP: {
    // FENCE retires as a no-op (no chip: the run cannot be proven); everything else without a chip is illegal
    if ((IN.raw & 0x7f) != 0x0f) { why = "illegal instruction"; goto trapped; }
    if (!unsupported) { char bb[64]; snprintf(bb, sizeof bb, "instruction 0x%08x at pc 0x%x", IN.raw, IN.pc); unsupported_what = bb; }
    unsupported = true;
    if (TRACE) { why = "instruction without a chip in a traced run"; goto trapped; }
    done++; clk += 4; ip = ip + 1;
    if (done >= budget) goto out_of_budget;
    goto *handlers[ip->kind];
}
L_OOB:
    // control left the text: report the address it went to (a JALR recorded it; a static jump's target is in `prev`)
    if (jump_pc) pc = jump_pc;
    else if (prev && prev->tgt_idx == ninstr && (prev->kind == K_JAL || (prev->kind >= K_BEQ && prev->kind <= K_BGEU && prev + 1 != ip))) pc = prev->tgt_raw;
    else pc = ip->pc;
    trap("pc outside text");
    goto finished;
trapped:
    pc = ip->pc;
    trap(why);
    goto finished;
out_of_budget:
    pc = ip->pc;
finished:
#undef IN
#undef PORT_A
#undef PORT_B
#undef PORT_C
#undef MEM
#undef BEGIN
#undef NEXT
#undef SEQ
#undef RRR
#undef RRI
#undef BRANCH
    cycles += done;
    in_shard += (uint32_t)done;
    if (TRACE) out->n_recs += done;
}

std::vector<MemInitRow> Vm::mem_rows() const {
    std::vector<MemInitRow> rows;
    rows.reserve(prog.image.size() + first_touch.size());
    for (auto &kv : prog.image) {
        MemInitRow r{kv.first, kv.second, kv.second, 0, 0, 1};
        const Cell *c = kv.first >= REG_BASE ? &regs[kv.first - REG_BASE] : peek(kv.first);
        if (c && c->tsh) { r.f = c->val; r.fts = c->ts(); r.fsh = c->sh(); }
        rows.push_back(r);
    }
    for (auto &ft : first_touch) {
        const Cell *c = peek(ft.first);
        rows.push_back(MemInitRow{ft.first, ft.second, c->val, c->ts(), c->sh(), 0});
    }
    std::sort(rows.begin(), rows.end(), [](const MemInitRow &x, const MemInitRow &y) { return x.addr < y.addr; });
    return rows;
}

void execute(const Program &prog, const std::vector<std::vector<uint8_t>> &stdin_bufs, bool trace, uint64_t max_cycles,
             uint32_t log_shard, ExecResult *res) {
    ExecResult &R = *res;
    R = ExecResult();
    Vm vm(prog, &stdin_bufs, log_shard);
    ShardOut out;
    std::vector<CycleRec> buf;
    if (trace) buf.resize((size_t)1 << log_shard);
    out.recs = buf.data();
    for (;;) {
        vm.run_shard(trace, &out, max_cycles);
        if (trace && vm.error.empty()) {
            R.shards.emplace_back();
            ShardRec &S = R.shards.back();
            S.index = out.index; S.start_pc = out.start_pc; S.next_pc = out.next_pc;
            S.recs.assign(out.recs, out.recs + out.n_recs);
            S.alu = out.alu;
            S.sha_ext = out.sha_ext;
            S.sha_cmp = out.sha_cmp;
            S.big = out.big;
        }
        if (vm.halted || !vm.error.empty()) break;
        if (!vm.next_shard()) break;
    }
    R.exit_code = vm.exit_code; R.halted = vm.halted; R.cycles = vm.cycles;
    R.unsupported = vm.unsupported; R.unsupported_what = vm.unsupported_what;
    R.public_values = std::move(vm.public_values); R.stdout_bytes = std::move(vm.stdout_bytes);
    for (int i = 0; i < 8; i++) R.committed[i] = vm.committed[i];
    R.committed_mask = vm.committed_mask;
    R.error = vm.error;
    if (trace && vm.halted) R.mem_rows = vm.mem_rows();
}

// ------------------------------------------------------------------ preprocessed traces
static uint32_t ceil_log2(size_t n) { uint32_t l = 0; while (((size_t)1 << l) < n) l++; return l; }

void build_prep(const Program &prog, HostPrep *out) {
    HostPrep &H = *out;
    for (auto &l : H.log_n) l = 0;
    // program: one row per provable instruction, padded by repeating the first row
    std::vector<const Instr *> rows;
    for (auto &in : prog.instrs) if (in.supported) rows.push_back(&in);
    if (rows.empty()) rows.push_back(&prog.instrs[0]);
    uint32_t lp = ceil_log2(rows.size());
    size_t np = (size_t)1 << lp;
    H.log_n[RV32_CHIP_PROGRAM] = lp;
    auto &P0 = H.prep[RV32_CHIP_PROGRAM];
    P0.assign((size_t)RV32_PROGRAM_PREP_W * np, 0);
    for (size_t r = 0; r < np; r++) {
        const Instr &in = *rows[r < rows.size() ? r : 0];
        auto put = [&](int col, uint32_t v) { P0[(size_t)col * np + r] = v; };
        const ColFlags cf = column_flags(in);
        put(RV32_PROGRAM_P_pc, in.pc); put(RV32_PROGRAM_P_rd, in.rd); put(RV32_PROGRAM_P_rs1, in.rs1); put(RV32_PROGRAM_P_rs2, in.rs2);
        for (int i = 0; i < 4; i++) put(RV32_PROGRAM_P_imm_0 + i, (cf.imm >> (8 * i)) & 0xff);
        put(RV32_PROGRAM_P_aux, in.tgt + in.alu_op);
        put(RV32_PROGRAM_P_bit_op, cf.bit_op); put(RV32_PROGRAM_P_cmp_signed, cf.cmp_signed);
        put(RV32_PROGRAM_P_rd_en, cf.rd_en); put(RV32_PROGRAM_P_rs1_en, cf.rs1_en); put(RV32_PROGRAM_P_rs2_en, cf.rs2_en); put(RV32_PROGRAM_P_imm_c, cf.imm_c);
        put(RV32_PROGRAM_P_is_add, cf.is_add); put(RV32_PROGRAM_P_is_sub, cf.is_sub); put(RV32_PROGRAM_P_is_bit, cf.is_bit); put(RV32_PROGRAM_P_is_set, cf.is_set);
        put(RV32_PROGRAM_P_is_mul, cf.is_mul); put(RV32_PROGRAM_P_is_mulhu, cf.is_mulhu); put(RV32_PROGRAM_P_is_lui, cf.is_lui); put(RV32_PROGRAM_P_is_jal, cf.is_jal);
        put(RV32_PROGRAM_P_is_jalr, cf.is_jalr); put(RV32_PROGRAM_P_is_beq, cf.is_beq); put(RV32_PROGRAM_P_is_bne, cf.is_bne); put(RV32_PROGRAM_P_is_brlt, cf.is_brlt);
        put(RV32_PROGRAM_P_is_brge, cf.is_brge); put(RV32_PROGRAM_P_is_lw, cf.is_lw); put(RV32_PROGRAM_P_is_sw, cf.is_sw); put(RV32_PROGRAM_P_is_ecall, cf.is_ecall);
        put(RV32_PROGRAM_P_is_lb, cf.is_lb); put(RV32_PROGRAM_P_is_lbu, cf.is_lbu); put(RV32_PROGRAM_P_is_lh, cf.is_lh); put(RV32_PROGRAM_P_is_lhu, cf.is_lhu);
        put(RV32_PROGRAM_P_is_sb, cf.is_sb); put(RV32_PROGRAM_P_is_sh, cf.is_sh); put(RV32_PROGRAM_P_is_alu, cf.is_alu);
    }
    // byte table
    H.log_n[RV32_CHIP_BYTE] = 16;
    auto &B0 = H.prep[RV32_CHIP_BYTE];
    const size_t nb = 65536;
    B0.assign((size_t)RV32_BYTE_PREP_W * nb, 0);
    for (uint32_t r = 0; r < nb; r++) {
        uint32_t b = r >> 8, c = r & 0xff;
        B0[(size_t)RV32_BYTE_P_b * nb + r] = b; B0[(size_t)RV32_BYTE_P_c * nb + r] = c;
        B0[(size_t)RV32_BYTE_P_and * nb + r] = b & c; B0[(size_t)RV32_BYTE_P_or * nb + r] = b | c;
        B0[(size_t)RV32_BYTE_P_xor * nb + r] = b ^ c; B0[(size_t)RV32_BYTE_P_ltu * nb + r] = b < c;
        B0[(size_t)RV32_BYTE_P_msb * nb + r] = b >> 7;
        B0[(size_t)RV32_BYTE_P_addr * nb + r] = (b & 3) + 4 * (c >= (ADDR_LIMIT >> 24));
    }
    // memory image
    uint32_t li = ceil_log2(prog.image.size());
    size_t ni = (size_t)1 << li;
    H.log_n[RV32_CHIP_MEM_IMAGE] = li;
    auto &I0 = H.prep[RV32_CHIP_MEM_IMAGE];
    I0.assign((size_t)RV32_MEM_IMAGE_PREP_W * ni, 0);
    for (size_t r = 0; r < prog.image.size(); r++) {
        I0[(size_t)RV32_MEM_IMAGE_P_addr * ni + r] = prog.image[r].first;
        for (int i = 0; i < 4; i++) I0[(size_t)(RV32_MEM_IMAGE_P_v_0 + i) * ni + r] = (prog.image[r].second >> (8 * i)) & 0xff;
        I0[(size_t)RV32_MEM_IMAGE_P_is_real * ni + r] = 1;
    }
}

// ------------------------------------------------------------------ host trace assembly (debug / tests / first path)
namespace {
struct HostSink {
    uint32_t *cpu; size_t n, row;
    uint32_t *byte_mult; uint32_t *prog_mult;  // byte_mult[op][65536], prog_mult[idx]
    void put(int col, uint32_t v) { cpu[(size_t)col * n + row] = v; }
    void byte(int op, uint32_t table_row) { byte_mult[(size_t)op * 65536 + table_row]++; }
    void prog(uint32_t idx) { prog_mult[idx]++; }
    void fence(uint32_t &) {}
};
}  // namespace

std::vector<uint32_t> program_row_map(const Program &prog) {
    std::vector<uint32_t> m(prog.instrs.size(), 0);
    uint32_t r = 0;
    for (size_t i = 0; i < prog.instrs.size(); i++)
        if (prog.instrs[i].supported) m[i] = r++;
    return m;
}

bool build_aux_host(const ShardMeta &S, const std::vector<AluEvent> &alu, const std::vector<ShaExtEvent> &sha_ext,
                    const std::vector<ShaCmpEvent> &sha_cmp, const std::vector<BigOpEvent> &big, const std::vector<MemInitRow> *mem_rows, int exit_code,
                    const HostPrep &prep, HostTraces *out, std::string *err, BigOpBatches *device_rows) {
    HostTraces &T = *out;
    if (S.n_recs == 0) { if (err) *err = "no cycles to prove"; return false; }
    const bool last = mem_rows != nullptr;
    const uint32_t lc = ceil_log2(S.n_recs);
    if (lc > 22) { if (err) *err = "shard longer than 2^22 cycles"; return false; }
    for (int c = 0; c < N_CHIPS; c++) { T.present[c] = true; T.main[c].clear(); }
    T.log_n[RV32_CHIP_CPU] = lc;
    static const bool time_aux = getenv("DVT_TIME_PREPARE") != nullptr;
    auto t_aux = std::chrono::steady_clock::now();
    auto lap_aux = [&](const char *what) {
        if (!time_aux) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[aux] %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_aux).count());
        t_aux = now;
    };
    std::vector<uint32_t> byte_mult((size_t)N_BYTE_OPS * 65536, 0);
    HostSink sink{nullptr, 0, 0, byte_mult.data(), nullptr};
    T.present[RV32_CHIP_MEM_INIT] = last;
    T.log_n[RV32_CHIP_MEM_INIT] = 0;
    if (last) {
        const uint32_t lm = ceil_log2(mem_rows->size());
        const size_t nm = (size_t)1 << lm;
        T.log_n[RV32_CHIP_MEM_INIT] = lm;
        if (device_rows) {
            device_rows->mem_rows = mem_rows;      // K0 of this chip runs on the GPU (launch_k0_mem_init_rows)
        } else {
            auto &M = T.main[RV32_CHIP_MEM_INIT];
            M.assign((size_t)RV32_MEM_INIT_MAIN_W * nm, 0);
            HostSink ms{M.data(), nm, 0, byte_mult.data(), nullptr};
            for (size_t r = 0; r < mem_rows->size(); r++) {
                ms.row = r;
                fill_mem_init_row(mem_rows->data(), r, ms);
            }
        }
    }
    lap_aux("byte counts zeroed + mem_init rows");
    // shift chip: one row per SLL/SRL/SRA of this shard (absent when the shard does not shift)
    std::vector<AluEvent> shifts, muldivs;
    for (auto &e : alu) (e.op <= ALU_SRA ? shifts : muldivs).push_back(e);
    T.present[RV32_CHIP_SHIFT] = !shifts.empty();
    T.log_n[RV32_CHIP_SHIFT] = 0;
    if (!shifts.empty()) {
        const uint32_t ls = ceil_log2(shifts.size());
        const size_t ns = (size_t)1 << ls;
        T.log_n[RV32_CHIP_SHIFT] = ls;
        if (device_rows) {
            device_rows->shifts = std::move(shifts);     // K0 of this chip runs on the GPU (launch_k0_shift_rows)
        } else {
            auto &H = T.main[RV32_CHIP_SHIFT];
            H.assign((size_t)RV32_SHIFT_MAIN_W * ns, 0);
            HostSink hs{H.data(), ns, 0, byte_mult.data(), nullptr};
            for (size_t r = 0; r < shifts.size(); r++) {
                hs.row = r;
                fill_shift_row(shifts[r], hs);
            }
        }
    }
    lap_aux("shift rows");
    // muldiv chip: one row per MULH/MULHSU/DIV/DIVU/REM/REMU of this shard (absent when there is none);
    // witness as tools/airgen/rv32.py:build_muldiv lays it out
    T.present[RV32_CHIP_MULDIV] = !muldivs.empty();
    T.log_n[RV32_CHIP_MULDIV] = 0;
    if (!muldivs.empty()) {
        const uint32_t ls = ceil_log2(muldivs.size());
        const size_t ns = (size_t)1 << ls;
        T.log_n[RV32_CHIP_MULDIV] = ls;
        auto &H = T.main[RV32_CHIP_MULDIV];
        H.assign((size_t)RV32_MULDIV_MAIN_W * ns, 0);
        for (size_t row = 0; row < muldivs.size(); row++) {
            const AluEvent &e = muldivs[row];
            auto put = [&](int col, uint32_t v) { H[(size_t)col * ns + row] = v % P; };
            auto B = [](uint32_t w, int i) -> uint32_t { return (w >> (8 * i)) & 0xffu; };
            const bool is_mul = e.op == ALU_MULH || e.op == ALU_MULHSU, is_sdr = e.op == ALU_DIV || e.op == ALU_REM;
            const bool is_dr = !is_mul;
            static const int flag_col[10] = {0, 0, 0, 0, RV32_MULDIV_is_mulh, RV32_MULDIV_is_mulhsu, RV32_MULDIV_is_div, RV32_MULDIV_is_divu,
                                             RV32_MULDIV_is_rem, RV32_MULDIV_is_remu};
            put(RV32_MULDIV_is_real, 1);
            put(flag_col[e.op], 1);
            // quotient / remainder (divisions), X = b (multiplications)
            uint32_t q = e.b, r = 0;
            const bool c0 = is_dr && e.c == 0, ovf = is_sdr && e.b == 0x80000000u && e.c == 0xffffffffu;
            if (is_dr) {
                if (c0) { q = 0xffffffffu; r = e.b; }
                else if (ovf) { q = e.b; r = 0; }
                else if (is_sdr) { q = (uint32_t)((int32_t)e.b / (int32_t)e.c); r = (uint32_t)((int32_t)e.b % (int32_t)e.c); }
                else { q = e.b / e.c; r = e.b % e.c; }
            }
            const uint32_t mx = q >> 31, my = e.c >> 31, mr = r >> 31, mb = e.b >> 31;
            const uint32_t sx = mx & (uint32_t)(is_mul || is_sdr), sy = my & (uint32_t)(e.op == ALU_MULH || is_sdr);
            const uint32_t sr = mr & (uint32_t)is_sdr, sb = mb & (uint32_t)is_sdr;
            for (int i = 0; i < 4; i++) {
                put(RV32_MULDIV_a_0 + i, B(e.a, i)); put(RV32_MULDIV_b_0 + i, B(e.b, i)); put(RV32_MULDIV_c_0 + i, B(e.c, i));
                put(RV32_MULDIV_q_0 + i, B(q, i)); put(RV32_MULDIV_r_0 + i, B(r, i));
            }
            put(RV32_MULDIV_mx, mx); put(RV32_MULDIV_my, my); put(RV32_MULDIV_mr, mr); put(RV32_MULDIV_mb, mb);
            put(RV32_MULDIV_sx, sx); put(RV32_MULDIV_sy, sy); put(RV32_MULDIV_sr, sr); put(RV32_MULDIV_sb, sb);
            sink.byte(B_MSB - 1, B(q, 3) << 8); sink.byte(B_MSB - 1, B(e.c, 3) << 8);
            sink.byte(B_MSB - 1, B(r, 3) << 8); sink.byte(B_MSB - 1, B(e.b, 3) << 8);
            // unsigned product bytes and carries
            uint32_t prod[8], carry = 0;
            for (int k = 0; k < 8; k++) {
                uint32_t t = carry;
                for (int i = 0; i < 4; i++) if (k - i >= 0 && k - i < 4) t += B(q, i) * B(e.c, k - i);
                prod[k] = t & 0xff; carry = t >> 8;
                put(RV32_MULDIV_prod_0 + k, prod[k]); put(RV32_MULDIV_mcy_0 + k, carry);
                sink.byte(B_U16 - 1, carry);
            }
            for (int k = 0; k < 4; k++) sink.byte(B_RANGE - 1, (prod[2 * k] << 8) | prod[2 * k + 1]);
            // high word of the signed product with borrows 0..2
            uint32_t h[4], bin = 0;
            for (int i = 0; i < 4; i++) {
                int32_t t = (int32_t)prod[4 + i] - (int32_t)(sx * B(e.c, i)) - (int32_t)(sy * B(q, i)) - (int32_t)bin;
                uint32_t bo = 0;
                while (t < 0) { t += 256; bo++; }
                h[i] = (uint32_t)t; bin = bo;
                put(RV32_MULDIV_h_0 + i, h[i]); put(RV32_MULDIV_bw_0 + i, bo);
            }
            sink.byte(B_RANGE - 1, (h[0] << 8) | h[1]); sink.byte(B_RANGE - 1, (h[2] << 8) | h[3]);
            sink.byte(B_RANGE - 1, (B(q, 0) << 8) | B(q, 1)); sink.byte(B_RANGE - 1, (B(q, 2) << 8) | B(q, 3));
            sink.byte(B_RANGE - 1, (B(r, 0) << 8) | B(r, 1)); sink.byte(B_RANGE - 1, (B(r, 2) << 8) | B(r, 3));
            uint32_t dl0 = 0, dl1 = 0;
            if (is_dr) {
                put(RV32_MULDIV_is_c0, c0); put(RV32_MULDIV_is_ovf, ovf);
                const uint32_t csum = B(e.c, 0) + B(e.c, 1) + B(e.c, 2) + B(e.c, 3);
                if (csum) put(RV32_MULDIV_cinv, inv(Fp::from_canonical(csum)).canonical());
                if (!ovf) {  // 64-bit sum P + R' = B' in 16-bit limbs
                    const uint32_t Pl[4] = {prod[0] | (prod[1] << 8), prod[2] | (prod[3] << 8), h[0] | (h[1] << 8), h[2] | (h[3] << 8)};
                    const uint32_t Rl[4] = {r & 0xffff, r >> 16, 65535 * sr, 65535 * sr};
                    uint32_t cy = 0;
                    for (int k = 0; k < 4; k++) { cy = (Pl[k] + Rl[k] + cy) >> 16; put(RV32_MULDIV_dcy_0 + k, cy); }
                }
                if (!c0) {  // |c| - |r| - 1 in two limbs, low-limb carry e0 = ea + 2 eb - 1
                    const int64_t sc_ = 1 - 2 * (int64_t)sy, sr_ = 1 - 2 * (int64_t)sr;
                    const int64_t t0 = sc_ * (e.c & 0xffff) - sr_ * (r & 0xffff) - 1;
                    int64_t e0 = 0;
                    while (t0 + 65536 * e0 < 0) e0++;
                    while (t0 + 65536 * e0 > 65535) e0--;
                    dl0 = (uint32_t)(t0 + 65536 * e0);
                    dl1 = (uint32_t)(sc_ * (e.c >> 16) - sr_ * (r >> 16) + 65536 * ((int64_t)sy - (int64_t)sr) - e0);
                    put(RV32_MULDIV_ea, (uint32_t)((e0 + 1) & 1)); put(RV32_MULDIV_eb, (uint32_t)((e0 + 1) >> 1));
                } else {
                    put(RV32_MULDIV_ea, 1);  // e0 = 0 (unconstrained here; any boolean pair is fine)
                }
            } else {
                put(RV32_MULDIV_ea, 1);
            }
            put(RV32_MULDIV_dl_0, dl0); put(RV32_MULDIV_dl_1, dl1);
            sink.byte(B_U16 - 1, dl0); sink.byte(B_U16 - 1, dl1);
        }
    }
    lap_aux("muldiv rows");
    // sha_extend chip: 64 rows per SHA_EXTEND call (absent when the shard makes none)
    T.present[RV32_CHIP_SHA_EXTEND] = !sha_ext.empty();
    T.log_n[RV32_CHIP_SHA_EXTEND] = 0;
    if (!sha_ext.empty()) {
        const uint32_t lx = ceil_log2(sha_ext.size() * 64);
        const size_t nx = (size_t)1 << lx;
        T.log_n[RV32_CHIP_SHA_EXTEND] = lx;
        auto &H = T.main[RV32_CHIP_SHA_EXTEND];
        H.assign((size_t)RV32_SHA_EXTEND_MAIN_W * nx, 0);
        for (size_t e = 0; e < sha_ext.size(); e++) {
            const ShaExtEvent &ev = sha_ext[e];
            for (uint32_t j = 0; j < 64; j++) {
                const size_t row = e * 64 + j;
                auto put = [&](int col, uint32_t v) { H[(size_t)col * nx + row] = v; };
                auto put_word = [&](int col0, uint32_t v) { for (int i = 0; i < 4; i++) put(col0 + i, (v >> (8 * i)) & 0xff); };
                put(RV32_SHA_EXTEND_is_real, 1);
                put(RV32_SHA_EXTEND_is_first, j == 0); put(RV32_SHA_EXTEND_is_last, j == 63);
                put(RV32_SHA_EXTEND_is_load, j < 16); put(RV32_SHA_EXTEND_is_e, j == 15);
                put(RV32_SHA_EXTEND_j, j);
                if (j != 63) put(RV32_SHA_EXTEND_j_inv, inv(Fp::from_canonical(j) - Fp::from_canonical(63)).canonical());
                put(RV32_SHA_EXTEND_clk, ev.clk);
                put_word(RV32_SHA_EXTEND_p_0, ev.ptr);
                // window: W[k] = w[j - 16 + k] (zero before the array's start)
                for (uint32_t k = 0; k < 16; k++) put_word(RV32_SHA_EXTEND_w0_0 + 4 * k, j + k >= 16 ? ev.w[j + k - 16] : 0u);
                const uint32_t x = j + 1 >= 16 ? ev.w[j + 1 - 16] : 0u, y = j + 14 >= 16 ? ev.w[j + 14 - 16] : 0u;
                for (int k = 0; k < 32; k++) { put(RV32_SHA_EXTEND_xb_0 + k, (x >> k) & 1); put(RV32_SHA_EXTEND_yb_0 + k, (y >> k) & 1); }
                const uint32_t s0 = ((x >> 7) | (x << 25)) ^ ((x >> 18) | (x << 14)) ^ (x >> 3);
                const uint32_t s1 = ((y >> 17) | (y << 15)) ^ ((y >> 19) | (y << 13)) ^ (y >> 10);
                put(RV32_SHA_EXTEND_s0_0, s0 & 0xffff); put(RV32_SHA_EXTEND_s0_1, s0 >> 16);
                put(RV32_SHA_EXTEND_s1_0, s1 & 0xffff); put(RV32_SHA_EXTEND_s1_1, s1 >> 16);
                const uint32_t nw = ev.w[j];
                put_word(RV32_SHA_EXTEND_nw_0, nw);
                put_word(RV32_SHA_EXTEND_old_0, ev.old[j]);
                if (j >= 16) {
                    const uint32_t w0 = ev.w[j - 16], w9 = ev.w[j - 7];
                    const uint32_t lo = (w0 & 0xffff) + (s0 & 0xffff) + (w9 & 0xffff) + (s1 & 0xffff);
                    const uint32_t c_lo = lo >> 16;
                    const uint32_t hi = (w0 >> 16) + (s0 >> 16) + (w9 >> 16) + (s1 >> 16) + c_lo;
                    const uint32_t c_hi = hi >> 16;
                    put(RV32_SHA_EXTEND_cy_0, c_lo & 1); put(RV32_SHA_EXTEND_cy_1, c_lo >> 1);
                    put(RV32_SHA_EXTEND_cy_2, c_hi & 1); put(RV32_SHA_EXTEND_cy_3, c_hi >> 1);
                    sink.byte(B_RANGE - 1, ((nw & 0xff) << 8) | ((nw >> 8) & 0xff));
                    sink.byte(B_RANGE - 1, (((nw >> 16) & 0xff) << 8) | (nw >> 24));
                }
                // the access: previous (shard, clk) of the word, gap to (shard, clk + 2)
                const uint32_t psh = ev.prev_sh[j], pts = ev.prev_ts[j];
                const uint32_t d = psh == S.index ? ev.clk + 2 - pts - 1 : S.index - psh - 1;
                put(RV32_SHA_EXTEND_m_sh, psh); put(RV32_SHA_EXTEND_m_ts, pts); put(RV32_SHA_EXTEND_m_same, psh == S.index);
                put(RV32_SHA_EXTEND_m_lo, d & 0xffff); put(RV32_SHA_EXTEND_m_hi, d >> 16);
                sink.byte(B_U16 - 1, d & 0xffff);
                sink.byte(B_RANGE - 1, (d >> 16) << 8);
                if (j == 0) sink.byte(B_ADDR - 1, ((ev.ptr & 0xff) << 8) | (ev.ptr >> 24));
            }
        }
    }
    // sha_compress chip: 80 rows per SHA_COMPRESS call
    T.present[RV32_CHIP_SHA_COMPRESS] = !sha_cmp.empty();
    T.log_n[RV32_CHIP_SHA_COMPRESS] = 0;
    if (!sha_cmp.empty()) {
        const uint32_t lx = ceil_log2(sha_cmp.size() * 80);
        const size_t nx = (size_t)1 << lx;
        T.log_n[RV32_CHIP_SHA_COMPRESS] = lx;
        auto &H = T.main[RV32_CHIP_SHA_COMPRESS];
        H.assign((size_t)RV32_SHA_COMPRESS_MAIN_W * nx, 0);
        auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
        for (size_t e = 0; e < sha_cmp.size(); e++) {
            const ShaCmpEvent &ev = sha_cmp[e];
            uint32_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // working variables a..h at the START of the row
            for (uint32_t j = 0; j < 80; j++) {
                const size_t row = e * 80 + j;
                const uint32_t g = j >> 3, o = j & 7;
                auto put = [&](int col, uint32_t x) { H[(size_t)col * nx + row] = x; };
                auto put_word = [&](int col0, uint32_t x) { for (int i = 0; i < 4; i++) put(col0 + i, (x >> (8 * i)) & 0xff); };
                auto put_bits = [&](int col0, uint32_t x) { for (int k = 0; k < 32; k++) put(col0 + k, (x >> k) & 1); };
                put(RV32_SHA_COMPRESS_is_real, 1); put(RV32_SHA_COMPRESS_is_first, j == 0); put(RV32_SHA_COMPRESS_is_last, j == 79);
                put(RV32_SHA_COMPRESS_oc_0 + o, 1); put(RV32_SHA_COMPRESS_gr_0 + g, 1);
                put(RV32_SHA_COMPRESS_clk, ev.clk);
                put_word(RV32_SHA_COMPRESS_wp_0, ev.w_ptr); put_word(RV32_SHA_COMPRESS_hp_0, ev.h_ptr);
                put_bits(RV32_SHA_COMPRESS_ab_0, v[0]); put_bits(RV32_SHA_COMPRESS_bb_0, v[1]); put_bits(RV32_SHA_COMPRESS_cb_0, v[2]);
                put_bits(RV32_SHA_COMPRESS_eb_0, v[4]); put_bits(RV32_SHA_COMPRESS_fb_0, v[5]); put_bits(RV32_SHA_COMPRESS_gb_0, v[6]);
                put(RV32_SHA_COMPRESS_d_0, v[3] & 0xffff); put(RV32_SHA_COMPRESS_d_1, v[3] >> 16);
                put(RV32_SHA_COMPRESS_h_0, v[7] & 0xffff); put(RV32_SHA_COMPRESS_h_1, v[7] >> 16);
                const uint32_t S1 = rotr(v[4], 6) ^ rotr(v[4], 11) ^ rotr(v[4], 25), S0 = rotr(v[0], 2) ^ rotr(v[0], 13) ^ rotr(v[0], 22);
                const uint32_t mj = (v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]), chv = (v[4] & v[5]) ^ (~v[4] & v[6]);
                put(RV32_SHA_COMPRESS_s1_0, S1 & 0xffff); put(RV32_SHA_COMPRESS_s1_1, S1 >> 16);
                put(RV32_SHA_COMPRESS_s0_0, S0 & 0xffff); put(RV32_SHA_COMPRESS_s0_1, S0 >> 16);
                put(RV32_SHA_COMPRESS_mj_0, mj & 0xffff); put(RV32_SHA_COMPRESS_mj_1, mj >> 16);
                // the access of this row
                uint32_t addr, before, after, psh, pts, ts = ev.clk + 2;
                if (g == 0) { addr = ev.h_ptr + 4 * (7 - o); before = after = ev.hs[7 - o]; psh = ev.h_sh[7 - o]; pts = ev.h_ts[7 - o]; }
                else if (g <= 8) { const uint32_t i = 8 * (g - 1) + o; addr = ev.w_ptr + 4 * i; before = after = ev.w[i]; psh = ev.w_sh[i]; pts = ev.w_ts[i]; }
                else { addr = ev.h_ptr + 4 * (7 - o); before = ev.hs[7 - o]; after = before + v[7]; psh = (uint16_t)S.index; pts = ev.clk + 2; ts = ev.clk + 3; }
                put(RV32_SHA_COMPRESS_maddr, addr);
                put_word(RV32_SHA_COMPRESS_mv_0, after); put_word(RV32_SHA_COMPRESS_mo_0, before);
                const uint32_t dgap = psh == S.index ? ts - pts - 1 : S.index - psh - 1;
                put(RV32_SHA_COMPRESS_m_sh, psh); put(RV32_SHA_COMPRESS_m_ts, pts); put(RV32_SHA_COMPRESS_m_same, psh == S.index);
                put(RV32_SHA_COMPRESS_m_lo, dgap & 0xffff); put(RV32_SHA_COMPRESS_m_hi, dgap >> 16);
                sink.byte(B_U16 - 1, dgap & 0xffff);
                sink.byte(B_RANGE - 1, (dgap >> 16) << 8);
                if (j == 0) {
                    sink.byte(B_ADDR - 1, ((ev.w_ptr & 0xff) << 8) | (ev.w_ptr >> 24));
                    sink.byte(B_ADDR - 1, ((ev.h_ptr & 0xff) << 8) | (ev.h_ptr >> 24));
                }
                // next row's variables
                uint32_t nv[8];
                for (int k = 7; k > 0; k--) nv[k] = v[k - 1];
                nv[0] = 0;
                if (g == 0) nv[0] = before;
                else if (g <= 8) {
                    const uint32_t i = 8 * (g - 1) + o, K = SHA256_K[i], wv = ev.w[i];
                    // e' = d + T1, a' = T1 + T2 in 16-bit halves: the carries are witnesses
                    const uint32_t lo_t = (v[7] & 0xffff) + (S1 & 0xffff) + (chv & 0xffff) + (K & 0xffff) + (wv & 0xffff);
                    const uint32_t hi_t = (v[7] >> 16) + (S1 >> 16) + (chv >> 16) + (K >> 16) + (wv >> 16);
                    const uint32_t e_lo = lo_t + (v[3] & 0xffff), ce_lo = e_lo >> 16, e_hi = hi_t + (v[3] >> 16) + ce_lo, ce_hi = e_hi >> 16;
                    const uint32_t a_lo = lo_t + (S0 & 0xffff) + (mj & 0xffff), ca_lo = a_lo >> 16;
                    const uint32_t a_hi = hi_t + (S0 >> 16) + (mj >> 16) + ca_lo, ca_hi = a_hi >> 16;
                    for (int k = 0; k < 3; k++) {
                        put(RV32_SHA_COMPRESS_ce_0 + k, (ce_lo >> k) & 1); put(RV32_SHA_COMPRESS_ce_0 + 3 + k, (ce_hi >> k) & 1);
                        put(RV32_SHA_COMPRESS_ca_0 + k, (ca_lo >> k) & 1); put(RV32_SHA_COMPRESS_ca_0 + 3 + k, (ca_hi >> k) & 1);
                    }
                    nv[4] = (e_lo & 0xffff) | (e_hi << 16);
                    nv[0] = (a_lo & 0xffff) | (a_hi << 16);
                } else {
                    const uint32_t lo = (before & 0xffff) + (v[7] & 0xffff), hi2 = (before >> 16) + (v[7] >> 16) + (lo >> 16);
                    put(RV32_SHA_COMPRESS_cf_0, lo >> 16); put(RV32_SHA_COMPRESS_cf_1, hi2 >> 16);
                    sink.byte(B_RANGE - 1, ((after & 0xff) << 8) | ((after >> 8) & 0xff));
                    sink.byte(B_RANGE - 1, (((after >> 16) & 0xff) << 8) | (after >> 24));
                }
                for (int k = 0; k < 8; k++) v[k] = nv[k];
            }
        }
    }
    lap_aux("sha rows");
    // field / curve precompile chips: one row per call
    if (!build_bigop_traces(big, S.index, &T, byte_mult.data(), err, device_rows)) return false;
    const uint32_t lp = prep.log_n[RV32_CHIP_PROGRAM];
    T.log_n[RV32_CHIP_PROGRAM] = lp;
    T.main[RV32_CHIP_PROGRAM].assign((size_t)1 << lp, 0);
    T.log_n[RV32_CHIP_BYTE] = 16;
    T.main[RV32_CHIP_BYTE] = byte_mult;  // [N_BYTE_OPS][65536] already column-major in op order
    T.log_n[RV32_CHIP_MEM_IMAGE] = prep.log_n[RV32_CHIP_MEM_IMAGE];
    T.main[RV32_CHIP_MEM_IMAGE].assign((size_t)1 << T.log_n[RV32_CHIP_MEM_IMAGE], 0);
    T.pubs = {S.start_pc % P, S.next_pc % P, last ? (uint32_t)exit_code % P : 0u, S.index, last ? 1u : 0u};
    lap_aux("precompile shapes, program / byte / mem_image columns");
    return true;
}

bool build_traces_host(const Program &prog, const ExecResult &res, size_t shard_pos, const HostPrep &prep, HostTraces *out, std::string *err) {
    HostTraces &T = *out;
    if (shard_pos >= res.shards.size()) { if (err) *err = "no such shard"; return false; }
    if (res.unsupported) { if (err) *err = "unsupported " + res.unsupported_what; return false; }
    const ShardRec &S = res.shards[shard_pos];
    const bool last = shard_pos + 1 == res.shards.size();
    if (!build_aux_host(ShardMeta{S.index, S.start_pc, S.next_pc, S.recs.size()}, S.alu, S.sha_ext, S.sha_cmp, S.big, last ? &res.mem_rows : nullptr, res.exit_code, prep, out, err)) return false;
    const size_t nc = (size_t)1 << T.log_n[RV32_CHIP_CPU];
    T.main[RV32_CHIP_CPU].assign((size_t)RV32_CPU_MAIN_W * nc, 0);
    std::vector<uint32_t> prog_idx_mult(prog.instrs.size(), 0);
    HostSink sink{T.main[RV32_CHIP_CPU].data(), nc, 0, T.main[RV32_CHIP_BYTE].data(), prog_idx_mult.data()};
    for (size_t r = 0; r < S.recs.size(); r++) {
        sink.row = r;
        const uint32_t next_pc = r + 1 < S.recs.size() ? prog.instrs[S.recs[r + 1].idx].pc : S.next_pc;
        fill_cpu_row(S.recs[r], prog.instrs[S.recs[r].idx], (uint32_t)r, S.index, next_pc, sink);
    }
    // program multiplicities follow the preprocessed row order (provable instructions only)
    std::vector<uint32_t> rowmap = program_row_map(prog);
    for (size_t i = 0; i < prog.instrs.size(); i++)
        if (prog.instrs[i].supported) T.main[RV32_CHIP_PROGRAM][rowmap[i]] = prog_idx_mult[i];
    return true;
}

}  // namespace rv32
}  // namespace dvt
