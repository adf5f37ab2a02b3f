// RV32IM ELF loader, decoder, interpreter and host-side trace assembly (see rv32.h).
#include <algorithm>
#include <cstring>
#include <map>
#include <unordered_map>

#include "rv32.h"

namespace dvt {
namespace rv32 {

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
    case 0x6f: set_rd(); in.imm = pc + 4; in.tgt = pc + imm_j; in.flags |= FL(F_JAL); break;
    case 0x67:
        if (f3 != 0) { in.supported = 0; break; }
        set_rd(); set_rs1(); in.imm = pc + 4; in.off = imm_i; in.flags |= FL(F_JALR);
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
    for (uint32_t r = 0; r < 32; r++) image[r] = 0;  // registers live at word addresses 0..31, initial value 0
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
            p.instrs.resize(words);
            for (uint32_t k = 0; k < words; k++) p.instrs[k] = decode(image[vaddr + 4 * k], vaddr + 4 * k);
        }
    }
    if (!have_text) return bad("no executable segment");
    if (p.entry < p.text_base || p.entry >= p.text_base + 4 * p.instrs.size() || p.entry % 4) return bad("entry point outside text");
    p.image.assign(image.begin(), image.end());
    *out = std::move(p);
    return true;
}

// ------------------------------------------------------------------ interpreter
namespace {
struct Cell { uint32_t val, ts, sh; uint8_t touched, img; uint32_t init; };
struct Memory {
    static constexpr uint32_t PAGE_BITS = 10;
    std::unordered_map<uint32_t, std::vector<Cell>> pages;
    Cell &at(uint32_t addr) {  // addr: word-aligned byte address
        uint32_t w = addr >> 2, pg = w >> PAGE_BITS;
        auto it = pages.find(pg);
        if (it == pages.end()) it = pages.emplace(pg, std::vector<Cell>(1u << PAGE_BITS, Cell{0, 0, 0, 0, 0, 0})).first;
        return it->second[w & ((1u << PAGE_BITS) - 1)];
    }
};
}  // namespace

void execute(const Program &prog, const std::vector<std::vector<uint8_t>> &stdin_bufs, bool trace, uint64_t max_cycles,
             uint32_t log_shard, ExecResult *res) {
    ExecResult &R = *res;
    R = ExecResult();
    Memory mem;  // guest memory, keyed by byte address; registers are kept separately (addresses 0..31 of the AIR)
    Cell regs[32];
    for (auto &c : regs) c = Cell{0, 0, 0, 0, 0, 0};
    for (auto &kv : prog.image)
        if (kv.first >= 32) { Cell &c = mem.at(kv.first); c.val = c.init = kv.second; c.img = 1; }
    size_t next_input = 0;
    uint32_t pc = prog.entry;
    const size_t ninstr = prog.instrs.size();
    const uint64_t shard_cycles = (uint64_t)1 << log_shard;
    uint32_t shard = 1;   // current shard index (timestamps are (shard, clk))
    uint32_t in_shard = 0;  // instructions retired in the current shard
    uint32_t pv_count = 0;  // public-value words committed so far
    if (trace) { R.shards.emplace_back(); R.shards.back().index = 1; R.shards.back().start_pc = pc; }
    auto trap = [&](const std::string &m) { R.error = m + " at pc 0x" + [&] { char b[16]; snprintf(b, sizeof b, "%x", pc); return std::string(b); }(); };
    while (R.cycles < max_cycles) {
        if (pc < prog.text_base || pc % 4 || (pc - prog.text_base) / 4 >= ninstr) { trap("pc outside text"); return; }
        if (trace && in_shard == shard_cycles) {  // cut a shard
            R.shards.back().next_pc = pc;
            R.shards.back().pv_end = pv_count;
            shard++;
            in_shard = 0;
            R.shards.emplace_back();
            R.shards.back().index = shard;
            R.shards.back().start_pc = pc;
            R.shards.back().pv_start = pv_count;
        }
        const uint32_t idx = (pc - prog.text_base) / 4;
        const Instr &in = prog.instrs[idx];
        const uint32_t w = in.raw, fl = in.flags;
        const uint32_t clk = 4 * (in_shard + 1);
        CycleRec rec{};
        rec.idx = idx;
        rec.pv_idx = pv_count;
        uint32_t next_pc = pc + 4, a = 0, b = 0, c = 0;
        if (!in.supported) {
            // executes but cannot be proven: plain interpreter semantics
            if (!R.unsupported) { char bb[64]; snprintf(bb, sizeof bb, "instruction 0x%08x at pc 0x%x", w, pc); R.unsupported_what = bb; }
            R.unsupported = true;
            const uint32_t op = w & 0x7f, rd = bits(w, 11, 7), f3 = bits(w, 14, 12), rs1 = bits(w, 19, 15), rs2 = bits(w, 24, 20), f7 = bits(w, 31, 25);
            uint32_t x = regs[rs1].val, y = regs[rs2].val, out = 0;
            bool wr = true;
            if (op == 0x13 || op == 0x33) {
                uint32_t sh = (op == 0x13 ? rs2 : y) & 31;
                if (op == 0x33 && f7 == 0x01) {
                    int64_t sx = (int32_t)x, sy = (int32_t)y;
                    switch (f3) {
                    case 1: out = (uint32_t)((sx * sy) >> 32); break;
                    case 2: out = (uint32_t)((sx * (int64_t)(uint64_t)y) >> 32); break;
                    case 4: out = y == 0 ? 0xffffffffu : (x == 0x80000000u && y == 0xffffffffu) ? x : (uint32_t)((int32_t)x / (int32_t)y); break;
                    case 5: out = y == 0 ? 0xffffffffu : x / y; break;
                    case 6: out = y == 0 ? x : (x == 0x80000000u && y == 0xffffffffu) ? 0 : (uint32_t)((int32_t)x % (int32_t)y); break;
                    case 7: out = y == 0 ? x : x % y; break;
                    default: trap("illegal instruction"); return;
                    }
                } else if (f3 == 1) out = x << sh;
                else if (f3 == 5) out = (bits(w, 30, 30)) ? (uint32_t)((int32_t)x >> sh) : x >> sh;
                else { trap("illegal instruction"); return; }
            } else if (op == 0x03 || op == 0x23) {
                uint32_t addr = x + (op == 0x03 ? sext(bits(w, 31, 20), 12) : sext((bits(w, 31, 25) << 5) | bits(w, 11, 7), 12));
                if (addr < 32 || addr >= ADDR_LIMIT) { trap("memory access out of range"); return; }
                Cell &cell = mem.at(addr & ~3u);
                uint32_t sh = 8 * (addr & 3);
                if (op == 0x03) {
                    uint32_t v = cell.val >> sh;
                    switch (f3) {
                    case 0: out = sext(v & 0xff, 8); break;
                    case 1: if (addr & 1) { trap("misaligned load"); return; } out = sext(v & 0xffff, 16); break;
                    case 4: out = v & 0xff; break;
                    case 5: if (addr & 1) { trap("misaligned load"); return; } out = v & 0xffff; break;
                    default: trap("illegal load"); return;
                    }
                } else {
                    wr = false;
                    if (f3 == 0) cell.val = (cell.val & ~(0xffu << sh)) | ((y & 0xff) << sh);
                    else if (f3 == 1) { if (addr & 1) { trap("misaligned store"); return; } cell.val = (cell.val & ~(0xffffu << sh)) | ((y & 0xffff) << sh); }
                    else { trap("illegal store"); return; }
                }
            } else if (op == 0x0f) { wr = false; }
            else { trap("illegal instruction"); return; }
            if (wr && rd) regs[rd].val = out;
            pc = next_pc;
            R.cycles++;
            in_shard++;
            continue;
        }
        // ---- provable instruction: accesses in port order c (rs2), b (rs1), memory, a (rd)
        if (fl & FL(F_RS2_EN)) { Cell &r2 = regs[in.rs2]; c = r2.val; rec.pc_ts = r2.ts; rec.pc_sh = r2.sh; r2.ts = clk; r2.sh = shard; r2.touched = 1; }
        if (fl & FL(F_IMM_C)) c = in.imm;
        if (fl & FL(F_RS1_EN)) { Cell &r1 = regs[in.rs1]; b = r1.val; rec.pb_ts = r1.ts; rec.pb_sh = r1.sh; r1.ts = clk + 1; r1.sh = shard; r1.touched = 1; }
        if (fl & (FL(F_ADD))) a = b + c;
        else if (fl & FL(F_SUB)) a = b - c;
        else if (fl & FL(F_AND)) a = b & c;
        else if (fl & FL(F_OR)) a = b | c;
        else if (fl & FL(F_XOR)) a = b ^ c;
        else if (fl & FL(F_SLT)) a = (int32_t)b < (int32_t)c;
        else if (fl & FL(F_SLTU)) a = b < c;
        else if (fl & FL(F_MUL)) a = b * c;
        else if (fl & FL(F_MULHU)) a = (uint32_t)(((uint64_t)b * c) >> 32);
        else if (fl & FL(F_LUI)) a = in.imm;
        else if (fl & FL(F_ALU)) {
            const uint32_t sh = c & 31;
            const int32_t sb = (int32_t)b, sc = (int32_t)c;
            switch (in.alu_op) {
            case ALU_SLL: a = b << sh; break;
            case ALU_SRL: a = b >> sh; break;
            case ALU_SRA: a = (uint32_t)(sb >> sh); break;
            case ALU_MULH: a = (uint32_t)(((int64_t)sb * (int64_t)sc) >> 32); break;
            case ALU_MULHSU: a = (uint32_t)(((int64_t)sb * (int64_t)(uint64_t)c) >> 32); break;
            // RISC-V: x / 0 = all ones, x % 0 = x; -2^31 / -1 = -2^31 remainder 0
            case ALU_DIV: a = c == 0 ? 0xffffffffu : (b == 0x80000000u && c == 0xffffffffu) ? b : (uint32_t)(sb / sc); break;
            case ALU_DIVU: a = c == 0 ? 0xffffffffu : b / c; break;
            case ALU_REM: a = c == 0 ? b : (b == 0x80000000u && c == 0xffffffffu) ? 0u : (uint32_t)(sb % sc); break;
            default: a = c == 0 ? b : b % c; break;  // ALU_REMU
            }
            if (trace) R.shards.back().alu.push_back(AluEvent{in.alu_op, a, b, c});
        }
        else if (fl & FL(F_JAL)) { a = in.imm; next_pc = in.tgt; }
        else if (fl & FL(F_JALR)) {
            a = in.imm;
            uint32_t t = b + in.off;
            if (t >= ADDR_LIMIT) { trap("jump target out of range"); return; }
            next_pc = t & ~1u;
        } else if (fl & (FL(F_BEQ) | FL(F_BNE) | FL(F_BLT) | FL(F_BGE) | FL(F_BLTU) | FL(F_BGEU))) {
            bool t = (fl & FL(F_BEQ)) ? b == c : (fl & FL(F_BNE)) ? b != c : (fl & FL(F_BLT)) ? (int32_t)b < (int32_t)c
                   : (fl & FL(F_BGE)) ? (int32_t)b >= (int32_t)c : (fl & FL(F_BLTU)) ? b < c : b >= c;
            if (t) next_pc = in.tgt;
        } else if (fl & (FL(F_LW) | FL(F_SW) | FL(F_LB) | FL(F_LBU) | FL(F_LH) | FL(F_LHU) | FL(F_SB) | FL(F_SH))) {
            const uint32_t addr = b + in.off;
            if (addr < 32 || addr >= ADDR_LIMIT) { trap("memory access out of range"); return; }
            if ((fl & (FL(F_LW) | FL(F_SW))) && (addr & 3)) { trap("misaligned word access"); return; }
            if ((fl & (FL(F_LH) | FL(F_LHU) | FL(F_SH))) && (addr & 1)) { trap("misaligned halfword access"); return; }
            Cell &cell = mem.at(addr & ~3u);
            if (!cell.touched) { cell.touched = 1; cell.init = cell.val; }
            rec.m_prev = cell.val;
            rec.m_ts = cell.ts;
            rec.m_sh = cell.sh;
            const uint32_t sh8 = 8 * (addr & 3);
            if (fl & FL(F_SW)) cell.val = c;
            else if (fl & FL(F_SB)) cell.val = (cell.val & ~(0xffu << sh8)) | ((c & 0xff) << sh8);
            else if (fl & FL(F_SH)) cell.val = (cell.val & ~(0xffffu << sh8)) | ((c & 0xffff) << sh8);
            else if (fl & FL(F_LW)) a = cell.val;
            else if (fl & FL(F_LB)) a = sext((cell.val >> sh8) & 0xff, 8);
            else if (fl & FL(F_LBU)) a = (cell.val >> sh8) & 0xff;
            else if (fl & FL(F_LH)) a = sext((cell.val >> sh8) & 0xffff, 16);
            else a = (cell.val >> sh8) & 0xffff;
            rec.m_val = cell.val;
            cell.ts = clk + 2;
            cell.sh = shard;
        } else if (fl & FL(F_ECALL)) {
            // b = t0 (id), c = a0
            uint32_t a1 = regs[11].val, a2 = regs[12].val;
            a = b;  // t0 unchanged unless the call returns a value
            switch (b) {
            case 0x00: R.halted = true; R.exit_code = (int)c; next_pc = 0; break;
            case 0x02: {  // WRITE(fd = a0, ptr = a1, len = a2)
                for (uint32_t k = 0; k < a2; k++) {
                    uint32_t ad = a1 + k;
                    if (ad < 32 || ad >= ADDR_LIMIT) { trap("WRITE buffer out of range"); return; }
                    uint8_t by = (uint8_t)(mem.at(ad & ~3u).val >> (8 * (ad & 3)));
                    R.stdout_bytes.push_back(by);  // (fd 3 included: only COMMIT words are public values)
                }
                break;
            }
            case 0x10:  // COMMIT(a0 = word): the next public-value word (bound by the proof)
                for (int q = 0; q < 4; q++) R.public_values.push_back((uint8_t)(c >> (8 * q)));
                pv_count++;
                break;
            case 0x1a: break;  // COMMIT_DEFERRED_PROOFS: no-op (no recursion in core proofs)
            case 0xf0: a = next_input < stdin_bufs.size() ? (uint32_t)stdin_bufs[next_input].size() : 0; break;
            case 0xf1: {  // HINT_READ(ptr = a0, len = a1): the words become initial memory (must be untouched so far)
                if (next_input >= stdin_bufs.size()) { trap("HINT_READ with no input left"); return; }
                const auto &buf = stdin_bufs[next_input++];
                if (a1 != buf.size()) { trap("HINT_READ length mismatch"); return; }
                if (c % 4 || c < 32 || (uint64_t)c + a1 > ADDR_LIMIT) { trap("HINT_READ pointer misaligned or out of range"); return; }
                for (uint32_t k = 0; k < a1; k += 4) {
                    Cell &cell = mem.at(c + k);
                    if (cell.touched || cell.sh || cell.img) { trap("HINT_READ into the program image or into memory that was already accessed"); return; }
                    uint32_t wv = 0;
                    for (uint32_t q = 0; q < 4 && k + q < a1; q++) wv |= (uint32_t)buf[k + q] << (8 * q);
                    cell.val = wv;
                }
                break;
            }
            default: trap("unknown syscall"); return;
            }
        }
        if (fl & FL(F_RD_EN)) { Cell &rdc = regs[in.rd]; rec.pa_prev = rdc.val; rec.pa_ts = rdc.ts; rec.pa_sh = rdc.sh; rdc.val = a; rdc.ts = clk + 3; rdc.sh = shard; rdc.touched = 1; }
        rec.a = a; rec.b = b; rec.c = c; rec.next_pc = next_pc;
        if (trace) R.shards.back().recs.push_back(rec);
        R.cycles++;
        in_shard++;
        pc = next_pc;
        if (R.halted) break;
    }
    if (!R.halted && R.error.empty()) R.error = "cycle limit reached before HALT";
    if (trace) {
        R.shards.back().next_pc = pc;  // 0 after HALT
        R.shards.back().pv_end = pv_count;
        // one mem_init row per image word and per touched non-image word, sorted by address
        std::map<uint32_t, MemInitRow> rows;
        for (auto &kv : prog.image) {
            MemInitRow r{kv.first, kv.second, kv.second, 0, 0, 1};
            if (kv.first < 32) { r.f = regs[kv.first].val; r.fts = regs[kv.first].ts; r.fsh = regs[kv.first].sh; }
            rows[kv.first] = r;
        }
        for (auto &pg : mem.pages)
            for (uint32_t k = 0; k < (1u << Memory::PAGE_BITS); k++) {
                const Cell &cell = pg.second[k];
                uint32_t addr = ((pg.first << Memory::PAGE_BITS) | k) << 2;
                auto it = rows.find(addr);
                if (it != rows.end()) {
                    if (cell.touched) { it->second.f = cell.val; it->second.fts = cell.ts; it->second.fsh = cell.sh; }
                } else if (cell.touched) rows[addr] = MemInitRow{addr, cell.init, cell.val, cell.ts, cell.sh, 0};
            }
        for (auto &kv : rows) R.mem_rows.push_back(kv.second);
    }
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
        put(RV32_PROGRAM_P_pc, in.pc); put(RV32_PROGRAM_P_rd, in.rd); put(RV32_PROGRAM_P_rs1, in.rs1); put(RV32_PROGRAM_P_rs2, in.rs2);
        for (int i = 0; i < 4; i++) { put(RV32_PROGRAM_P_imm_0 + i, (in.imm >> (8 * i)) & 0xff); put(RV32_PROGRAM_P_off_0 + i, (in.off >> (8 * i)) & 0xff); }
        put(RV32_PROGRAM_P_aux, in.tgt + in.alu_op);
        for (uint32_t k = 0; k < N_FLAGS; k++) put(RV32_PROGRAM_P_rd_en + k, (in.flags >> k) & 1);
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

bool build_aux_host(const Program &prog, const ExecResult &res, size_t shard_pos, const HostPrep &prep, HostTraces *out, std::string *err) {
    HostTraces &T = *out;
    if (shard_pos >= res.shards.size() || res.shards[shard_pos].recs.empty()) { if (err) *err = "no cycles to prove"; return false; }
    if (res.unsupported) { if (err) *err = "unsupported " + res.unsupported_what; return false; }
    const ShardRec &S = res.shards[shard_pos];
    const bool last = shard_pos + 1 == res.shards.size();
    const uint32_t lc = ceil_log2(S.recs.size());
    if (lc > 22) { if (err) *err = "shard longer than 2^22 cycles"; return false; }
    for (int c = 0; c < N_CHIPS; c++) { T.present[c] = true; T.main[c].clear(); }
    T.log_n[RV32_CHIP_CPU] = lc;
    std::vector<uint32_t> byte_mult((size_t)N_BYTE_OPS * 65536, 0);
    HostSink sink{nullptr, 0, 0, byte_mult.data(), nullptr};
    T.present[RV32_CHIP_MEM_INIT] = last;
    T.log_n[RV32_CHIP_MEM_INIT] = 0;
    if (last) {
        const uint32_t lm = ceil_log2(res.mem_rows.size());
        const size_t nm = (size_t)1 << lm;
        T.log_n[RV32_CHIP_MEM_INIT] = lm;
        auto &M = T.main[RV32_CHIP_MEM_INIT];
        M.assign((size_t)RV32_MEM_INIT_MAIN_W * nm, 0);
        uint32_t prev_addr = 0;
        for (size_t r = 0; r < res.mem_rows.size(); r++) {
            const MemInitRow &m = res.mem_rows[r];
            auto put = [&](int col, uint32_t v) { M[(size_t)col * nm + r] = v; };
            put(RV32_MEM_INIT_addr, m.addr); put(RV32_MEM_INIT_fts, m.fts); put(RV32_MEM_INIT_fsh, m.fsh);
            put(RV32_MEM_INIT_is_img, m.is_img); put(RV32_MEM_INIT_is_real, 1);
            uint32_t d = r ? m.addr - prev_addr - 1 : 0;
            for (int i = 0; i < 4; i++) {
                put(RV32_MEM_INIT_v_0 + i, (m.v >> (8 * i)) & 0xff);
                put(RV32_MEM_INIT_f_0 + i, (m.f >> (8 * i)) & 0xff);
                put(RV32_MEM_INIT_d_0 + i, (d >> (8 * i)) & 0xff);
            }
            sink.byte(B_RANGE - 1, ((d & 0xff) << 8) | ((d >> 8) & 0xff));
            sink.byte(B_RANGE - 1, (((d >> 16) & 0xff) << 8) | (d >> 24));
            sink.byte(B_LTU - 1, ((d >> 24) << 8) | 0x40);
            if (!m.is_img) {
                sink.byte(B_RANGE - 1, ((m.v & 0xff) << 8) | ((m.v >> 8) & 0xff));
                sink.byte(B_RANGE - 1, (((m.v >> 16) & 0xff) << 8) | (m.v >> 24));
            }
            prev_addr = m.addr;
        }
    }
    // shift chip: one row per SLL/SRL/SRA of this shard (absent when the shard does not shift)
    std::vector<AluEvent> shifts, muldivs;
    for (auto &e : S.alu) (e.op <= ALU_SRA ? shifts : muldivs).push_back(e);
    T.present[RV32_CHIP_SHIFT] = !shifts.empty();
    T.log_n[RV32_CHIP_SHIFT] = 0;
    if (!shifts.empty()) {
        const uint32_t ls = ceil_log2(shifts.size());
        const size_t ns = (size_t)1 << ls;
        T.log_n[RV32_CHIP_SHIFT] = ls;
        auto &H = T.main[RV32_CHIP_SHIFT];
        H.assign((size_t)RV32_SHIFT_MAIN_W * ns, 0);
        for (size_t r = 0; r < shifts.size(); r++) {
            const AluEvent &e = shifts[r];
            auto put = [&](int col, uint32_t v) { H[(size_t)col * ns + r] = v; };
            auto B = [](uint32_t w, int i) -> uint32_t { return (w >> (8 * i)) & 0xffu; };
            const uint32_t sh = e.c & 31, q = sh >> 3, rb = sh & 7, m = 1u << rb, mi = 1u << (8 - rb);
            const bool left = e.op == ALU_SLL;
            const uint32_t sgn = e.op == ALU_SRA ? e.b >> 31 : 0;
            put(RV32_SHIFT_is_real, 1);
            put(left ? RV32_SHIFT_is_sll : e.op == ALU_SRL ? RV32_SHIFT_is_srl : RV32_SHIFT_is_sra, 1);
            put(RV32_SHIFT_sh, sh);
            if (q) put(RV32_SHIFT_q_0 + q - 1, 1);
            put(RV32_SHIFT_r_0 + rb, 1);
            put(RV32_SHIFT_sgn, sgn);
            uint32_t lo[4], hi[4];
            for (int i = 0; i < 4; i++) {
                put(RV32_SHIFT_a_0 + i, B(e.a, i)); put(RV32_SHIFT_b_0 + i, B(e.b, i)); put(RV32_SHIFT_c_0 + i, B(e.c, i));
                if (left) { uint32_t pr = B(e.b, i) * m; lo[i] = pr & 0xff; hi[i] = pr >> 8; }
                else { hi[i] = B(e.b, i) >> rb; lo[i] = B(e.b, i) & (m - 1); }
                put(RV32_SHIFT_lo_0 + i, lo[i]); put(RV32_SHIFT_hi_0 + i, hi[i]);
            }
            for (int i = 0; i < 4; i++) {
                uint32_t t = left ? lo[i] + (i ? hi[i - 1] : 0) : hi[i] + (i < 3 ? lo[i + 1] * mi : sgn * (256 - mi));
                put(RV32_SHIFT_t_0 + i, t);
                if (!left) sink.byte(B_LTU - 1, (lo[i] << 8) | m);
            }
            sink.byte(B_AND - 1, (B(e.c, 0) << 8) | 31);
            sink.byte(B_RANGE - 1, (lo[0] << 8) | lo[1]); sink.byte(B_RANGE - 1, (lo[2] << 8) | lo[3]);
            sink.byte(B_RANGE - 1, (hi[0] << 8) | hi[1]); sink.byte(B_RANGE - 1, (hi[2] << 8) | hi[3]);
            if (e.op == ALU_SRA) sink.byte(B_MSB - 1, B(e.b, 3) << 8);
        }
    }
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
    const uint32_t lp = prep.log_n[RV32_CHIP_PROGRAM];
    T.log_n[RV32_CHIP_PROGRAM] = lp;
    T.main[RV32_CHIP_PROGRAM].assign((size_t)1 << lp, 0);
    T.log_n[RV32_CHIP_BYTE] = 16;
    T.main[RV32_CHIP_BYTE] = byte_mult;  // [7][65536] already column-major in op order
    T.log_n[RV32_CHIP_MEM_IMAGE] = prep.log_n[RV32_CHIP_MEM_IMAGE];
    T.main[RV32_CHIP_MEM_IMAGE].assign((size_t)1 << T.log_n[RV32_CHIP_MEM_IMAGE], 0);
    T.pubs = {S.start_pc % P, S.next_pc % P, last ? (uint32_t)res.exit_code % P : 0u, S.index, last ? 1u : 0u, S.pv_start, S.pv_end};
    return true;
}

bool build_traces_host(const Program &prog, const ExecResult &res, size_t shard_pos, const HostPrep &prep, HostTraces *out, std::string *err) {
    HostTraces &T = *out;
    if (!build_aux_host(prog, res, shard_pos, prep, out, err)) return false;
    const ShardRec &S = res.shards[shard_pos];
    const size_t nc = (size_t)1 << T.log_n[RV32_CHIP_CPU];
    T.main[RV32_CHIP_CPU].assign((size_t)RV32_CPU_MAIN_W * nc, 0);
    std::vector<uint32_t> prog_idx_mult(prog.instrs.size(), 0);
    HostSink sink{T.main[RV32_CHIP_CPU].data(), nc, 0, T.main[RV32_CHIP_BYTE].data(), prog_idx_mult.data()};
    for (size_t r = 0; r < S.recs.size(); r++) {
        sink.row = r;
        fill_cpu_row(S.recs[r], prog.instrs[S.recs[r].idx], (uint32_t)r, S.index, sink);
    }
    for (size_t r = S.recs.size(); r < nc; r++) T.main[RV32_CHIP_CPU][(size_t)RV32_CPU_pv_idx * nc + r] = S.pv_end;
    // program multiplicities follow the preprocessed row order (provable instructions only)
    std::vector<uint32_t> rowmap = program_row_map(prog);
    for (size_t i = 0; i < prog.instrs.size(); i++)
        if (prog.instrs[i].supported) T.main[RV32_CHIP_PROGRAM][rowmap[i]] = prog_idx_mult[i];
    return true;
}

}  // namespace rv32
}  // namespace dvt
