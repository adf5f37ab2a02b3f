"""RV32IM core machine (original design; the reference delegates its AIRs to the
absent sp1-core-machine crate — SURVEY.md sections 0.1 and 8(a)).

Chips
  program   preprocessed decoded instruction table, one row per instruction word
  byte      preprocessed 2^16-row byte-pair table: AND/OR/XOR/LTU/MSB/range lookups
  cpu       one row per executed instruction: fetch (program lookup), register
            ports (registers live in the memory argument at addresses REG_BASE + 0..31),
            and every instruction family in a shared ("union") column block
  mem_image preprocessed initial memory image (registers = 0, ELF segments)
  mem_init  one row per initialised address, sorted: initial value at timestamp 0,
            final value/timestamp; image words are bound to mem_image
  shift     SLL/SRL/SRA rows, fed by the cpu chip over the alu bus (only in shards that shift)
  muldiv    MULH/MULHSU/DIV/DIVU/REM/REMU rows, same bus (only in shards that use them)
  sha_extend  the SHA_EXTEND precompile (SHA-256 message schedule, 64 rows per call), fed over the sys bus
  sha_compress  the SHA_COMPRESS precompile (SHA-256 compression function, 80 rows per call), same bus
  fp_op     BLS12-381 base-field add / sub / mul precompiles (one row per call), same bus
  fp2_op    BLS12-381 Fp2 add / sub / mul precompiles (one row per call)
  bls_g1    BLS12-381 G1 affine add / double precompiles (one row per call)
  secp_k1   secp256k1 affine add / double precompiles (one row per call; the same short-Weierstrass a = 0 template)
  u256_mul  UINT256_MUL precompile: x := x * y mod m for 256-bit numbers, the modulus read from memory (one row per call)

Memory consistency is an offline-checking LogUp multiset over tuples
(addr, byte0..3, timestamp): every access consumes the previous tuple of its
address and produces a new one with a strictly larger timestamp.
Words are 4 byte limbs; all limbs written to a register or to memory are
range-checked through the byte table.  Addresses are < 2^30.

Timestamps are pairs (shard, clk), shards numbered from 1; inside a shard instruction
i (0-based) has clk = 4(i+1); it reads rs2 at clk, rs1 at clk+1, touches memory at
clk+2 and writes rd at clk+3.  A long execution is cut into shards that are proven
independently with COMMON LogUp challenges (derived from all shards' main
commitments), so the memory bus balances across shards; the mem_init table (initial
tuples at (0,0), final tuples) is part of the last shard only.

DSL conventions: `sel * (...)` gates a family's constraints by its (program-table
supplied, hence trusted and mutually exclusive) selector.
"""
from .dsl import Chip, Expr, Machine, esum, word

BUSES = {"program": 1, "byte": 2, "mem": 3, "image": 4, "sys": 5, "alu": 6}
ALU_SLL, ALU_SRL, ALU_SRA = 1, 2, 3
ALU_MULH, ALU_MULHSU, ALU_DIV, ALU_DIVU, ALU_REM, ALU_REMU = 4, 5, 6, 7, 8, 9
SYS_COMMIT = 0x10
SYS_HINT_LEN = 0xF0
SYS_SHA_EXTEND = 0x00300105     # SP1's syscall code: byte 0 = id, byte 1 = 1 "has a precompile table", byte 2 = extra cycles
SYS_SHA_COMPRESS = 0x00010106
SHA_K = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
REG_A1 = 11
# field / curve precompiles: SP1's syscall numbers as best recalled [EXTERNAL, unverified: the
# sp1-core-executor crate is absent]; byte 1 = 1 "has a table" is what the cpu chip keys on, the chips receive the full code
SYS_SECP256K1_ADD, SYS_SECP256K1_DOUBLE = 0x0001010A, 0x0000010B
SYS_BLS12381_ADD, SYS_BLS12381_DOUBLE = 0x0001011E, 0x0000011F
SYS_UINT256_MUL = 0x0001011D
SYS_BLS12381_FP_ADD, SYS_BLS12381_FP_SUB, SYS_BLS12381_FP_MUL = 0x00010120, 0x00010121, 0x00010122
SYS_BLS12381_FP2_ADD, SYS_BLS12381_FP2_SUB, SYS_BLS12381_FP2_MUL = 0x00010123, 0x00010124, 0x00010125
BLS12381_P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
SECP256K1_P = (1 << 256) - (1 << 32) - 977

# byte-table opcodes
B_AND, B_OR, B_XOR, B_LTU, B_MSB, B_RANGE, B_U16, B_ADDR = 1, 2, 3, 4, 5, 6, 7, 8
# B_ADDR [8, r, b, c]: r = (b & 3) + 4 * (c >= ADDR_TOP_BYTE).  A sender that puts a value 0..3 into r gets, in ONE lookup,
# the byte offset b & 3 of an address whose low byte is b, the bound c < ADDR_TOP_BYTE on its top byte, and the range
# check of both bytes.

FLAGS = [
    "rd_en", "rs1_en", "rs2_en", "imm_c",
    "is_add", "is_sub",
    "is_bit",             # AND / OR / XOR: the byte-table opcode is the value column bit_op
    "is_set",             # SLT / SLTU: signedness is the value column cmp_signed
    "is_mul", "is_mulhu",
    "is_lui", "is_jal", "is_jalr", "is_beq", "is_bne",
    "is_brlt", "is_brge",  # BLT / BLTU and BGE / BGEU: signedness in cmp_signed
    "is_lw", "is_sw", "is_ecall",
    "is_lb", "is_lbu", "is_lh", "is_lhu", "is_sb", "is_sh",
    "is_alu",   # the result comes from another chip over the "alu" bus (alu_op selects it)
]
VALUE_FLAGS = ["bit_op", "cmp_signed"]   # decode-time values that are not 0/1 selectors of a family (0 outside their family)
# instruction tuple on the program bus: pc, rd, rs1, rs2, imm[4], aux, bit_op, cmp_signed, flags...
# (imm = the immediate operand, the LUI / AUIPC constant, or the address offset of loads / stores / JALR: never two of
#  them; aux = branch / jump target for the control-flow families, alu-bus opcode for is_alu rows: never both)
N_INSTR_FIELDS = 1 + 3 + 4 + 1 + len(VALUE_FLAGS) + len(FLAGS)
LINK_TOP_BYTE = 0x78      # a link value pc + 4 < 0x38000004; its alias pc + 4 + p has a top byte >= 0x78

PUB_START_PC, PUB_NEXT_PC, PUB_EXIT_CODE, PUB_SHARD, PUB_IS_LAST = 0, 1, 2, 3, 4
N_PUB = 5
UNION_W = 26
# every guest address (and every jump target) is below ADDR_TOP_BYTE << 24.  0x38000000 rather than 2^30 so that the sum
# of an address and an address gap (both below it) stays below p = 2^31 - 2^27 + 1: the mem_init table's "strictly
# increasing" check then holds over the integers, not only mod p.
ADDR_TOP_BYTE = 0x38
# The 32 registers are words of the memory argument at REG_BASE + r: ABOVE every address a load, a store or a precompile
# can form (those are four range-checked bytes with a top byte below ADDR_TOP_BYTE), so no guest access can alias a
# register (x0 stays 0 whatever pointer the guest dereferences) and nothing has to compare an address with 32.  The 8 MiB gap
# above ADDR_TOP_BYTE << 24 keeps the multi-word accesses of a precompile whose pointer sits just below the limit (the executor
# traps on those) away from the registers as well: they land on ordinary words the mem_init table may hold.
REG_BASE = (ADDR_TOP_BYTE << 24) + (1 << 23)
# next_pc of a HALT row, and therefore of the last shard: a value no other row can produce (a sequential pc, a static
# target and a JALR target are all below REG_BASE; p - 1, the JALR "target" 0 - 1, is not 2^30 either), so "the
# execution halted" is bound to a HALT row, not to control flow that happens to reach address 0.
HALT_PC = 1 << 30
# aux (target) field of a JAL / branch whose static target lies outside the text (the executor traps there): an odd value,
# never the pc of a program row, never HALT_PC
BAD_PC = 1


def build_program():
    ch = Chip("program")
    fields = [ch.prep("pc"), ch.prep("rd"), ch.prep("rs1"), ch.prep("rs2")]
    fields += ch.preps("imm", 4) + [ch.prep("aux")] + [ch.prep(f) for f in VALUE_FLAGS]
    fields += [ch.prep(f) for f in FLAGS]
    mult = ch.col("mult")
    ch.receive("program", fields, mult)
    return ch


def build_byte():
    ch = Chip("byte")
    b, c = ch.prep("b"), ch.prep("c")
    r_and, r_or, r_xor, r_ltu, r_msb = ch.prep("and"), ch.prep("or"), ch.prep("xor"), ch.prep("ltu"), ch.prep("msb")
    r_addr = ch.prep("addr")
    m = {k: ch.col("mult_" + k) for k in ("and", "or", "xor", "ltu", "msb", "range", "u16", "addr")}
    ch.receive("byte", [B_AND, r_and, b, c], m["and"])
    ch.receive("byte", [B_OR, r_or, b, c], m["or"])
    ch.receive("byte", [B_XOR, r_xor, b, c], m["xor"])
    ch.receive("byte", [B_LTU, r_ltu, b, c], m["ltu"])
    ch.receive("byte", [B_MSB, r_msb, b, c], m["msb"])      # senders use c = 0
    ch.receive("byte", [B_RANGE, 0, b, c], m["range"])
    ch.receive("byte", [B_U16, 0, 256 * b + c, 0], m["u16"])
    ch.receive("byte", [B_ADDR, r_addr, b, c], m["addr"])
    return ch


def build_cpu():
    ch = Chip("cpu")
    clk, pc, next_pc = ch.col("clk"), ch.col("pc"), ch.col("next_pc")
    rd, rs1, rs2 = ch.col("rd"), ch.col("rs1"), ch.col("rs2")
    imm, aux = ch.cols("imm", 4), ch.col("aux")
    off = imm                 # loads / stores / JALR: the program table's immediate field is the address offset
    tgt = alu_op = aux
    bit_op, cmp_signed = ch.col("bit_op"), ch.col("cmp_signed")
    # Flags that are linear in other flags are expressions, not columns (96 main columns = 12 sponge blocks exactly):
    #   is_real = sum of the family flags (a real row belongs to exactly one family; the program table, whose rows
    #             the fetch lookup must match with this very multiplicity, guarantees it);
    #   rs1_en  = every family that reads rs1;
    #   rs2_en  = every family that has a c operand, minus the immediate forms (c = rs2 or c = imm, never both).
    FAMILY = [f for f in FLAGS if f not in ("rd_en", "rs1_en", "rs2_en", "imm_c")]
    F = {f: ch.col(f) for f in FLAGS if f not in ("rs1_en", "rs2_en")}
    NO_RS1 = ("is_lui", "is_jal")
    NO_C = ("is_lui", "is_jal", "is_jalr", "is_lw", "is_lb", "is_lbu", "is_lh", "is_lhu")
    is_real = esum(F[f] for f in FAMILY)
    is_real_next = esum(F[f].next() for f in FAMILY)
    F["rs1_en"] = esum(F[f] for f in FAMILY if f not in NO_RS1)
    F["rs2_en"] = esum(F[f] for f in FAMILY if f not in NO_C) - F["imm_c"]
    a, b, c = ch.cols("a", 4), ch.cols("b", 4), ch.cols("c", 4)
    # register ports: previous timestamp + 24-bit difference (16 + 8 bit limbs)
    pb_ts, pb_lo, pb_hi = ch.col("pb_ts"), ch.col("pb_lo"), ch.col("pb_hi")
    pc_ts, pc_lo, pc_hi = ch.col("pc_ts"), ch.col("pc_lo"), ch.col("pc_hi")
    # shard of the previous access of each port and "same shard" flags (timestamps are (shard, clk) pairs)
    pb_sh, pb_same, pc_sh, pc_same = ch.col("pb_sh"), ch.col("pb_same"), ch.col("pc_sh"), ch.col("pc_same")
    pa_sh, pa_same = ch.col("pa_sh"), ch.col("pa_same")
    shard = ch.pub(PUB_SHARD)
    pa_prev = ch.cols("pa_prev", 4)
    pa_ts, pa_lo, pa_hi = ch.col("pa_ts"), ch.col("pa_lo"), ch.col("pa_hi")
    U = ch.cols("u", UNION_W)
    sys_m = ch.col("sys_m")

    # ---------------- row bookkeeping
    ch.assert_bool(is_real)
    ch.assert_zero(is_real_next * (1 - is_real), "trans")            # real rows first
    ch.assert_eq(is_real, 1, "first")
    ch.assert_eq(pc, ch.pub(PUB_START_PC), "first")
    ch.assert_eq(clk, 4, "first")
    ch.assert_zero(is_real_next * (clk.next() - clk - 4), "trans")
    ch.assert_zero(is_real_next * (pc.next() - next_pc), "trans")
    ch.assert_zero((is_real - is_real_next) * (next_pc - ch.pub(PUB_NEXT_PC)), "trans")
    ch.assert_zero(is_real * (next_pc - ch.pub(PUB_NEXT_PC)), "last")
    # padding rows do nothing: EVERY flag column is zero there, not only their sum (two cancelling flags, e.g.
    # is_ecall = 1 with is_lui = -1, would leave is_real = 0 while the family's interactions still fire);
    # on real rows the program-table lookup pins all of them
    for f in ["rd_en", "imm_c"] + FAMILY:
        ch.assert_zero((1 - is_real) * F[f])
    ch.assert_zero((1 - is_real) * bit_op)
    ch.assert_zero((1 - is_real) * cmp_signed)

    # ---------------- fetch
    ch.send("program", [pc, rd, rs1, rs2] + imm + [aux, bit_op, cmp_signed] + [F[f] for f in FLAGS], is_real)
    # families that live in their own chips (shifts): the row only ships (op, a, b, c) over the alu bus;
    # the receiving chip constrains a and range-checks its bytes
    ch.send("alu", [alu_op] + a + b + c, F["is_alu"])

    # ---------------- register ports (memory bus, addresses REG_BASE + 0..31)
    def port(addr, prev_val, val, prev_sh, same, prev_ts, ts, lo, hi, en):
        # consume (addr, value, shard', clk'), produce (addr, value, shard, clk) with (shard', clk') < (shard, clk):
        # same shard -> clk - clk' - 1 is a 24-bit number; earlier shard -> shard - shard' - 1 is
        ch.receive("mem", [addr] + prev_val + [prev_sh, prev_ts], en)
        ch.send("mem", [addr] + val + [shard, ts], en)
        ch.assert_zero(en * (same * (same - 1)))
        ch.assert_zero(en * (same * (shard - prev_sh)))
        ch.assert_zero(en * (same * (ts - prev_ts - 1) + (1 - same) * (shard - prev_sh - 1) - lo - 65536 * hi))
        ch.send("byte", [B_U16, 0, lo, 0], en)

    port(rs2 + REG_BASE, c, c, pc_sh, pc_same, pc_ts, clk, pc_lo, pc_hi, F["rs2_en"])
    port(rs1 + REG_BASE, b, b, pb_sh, pb_same, pb_ts, clk + 1, pb_lo, pb_hi, F["rs1_en"])
    port(rd + REG_BASE, pa_prev, a, pa_sh, pa_same, pa_ts, clk + 3, pa_lo, pa_hi, F["rd_en"])
    for i in range(4):
        ch.assert_zero(F["imm_c"] * (c[i] - imm[i]))

    # ---------------- families
    sel_addsub = F["is_add"] + F["is_sub"]
    sel_bit = F["is_bit"]
    sel_branch = F["is_beq"] + F["is_bne"] + F["is_brlt"] + F["is_brge"]
    sel_signed = cmp_signed        # (a 0/1 value of the program table, 1 only on SLT / BLT / BGE rows)
    sel_cmp = F["is_set"] + sel_branch
    sel_link = F["is_jal"] + F["is_jalr"]
    sel_mul = F["is_mul"] + F["is_mulhu"]
    sel_loadsub = F["is_lb"] + F["is_lbu"] + F["is_lh"] + F["is_lhu"]
    sel_mem = F["is_lw"] + F["is_sw"] + sel_loadsub + F["is_sb"] + F["is_sh"]
    sel_adder = sel_mem + F["is_jalr"]

    # ADD / SUB : u[0..3] = carries
    cy = U[0:4]
    for i in range(4):
        cin = cy[i - 1] if i else Expr.const(0)
        ch.assert_zero(F["is_add"] * (b[i] + c[i] + cin - a[i] - 256 * cy[i]))
        ch.assert_zero(F["is_sub"] * (a[i] + c[i] + cin - b[i] - 256 * cy[i]))
        ch.assert_zero(sel_addsub * (cy[i] * (cy[i] - 1)))

    # ---- byte-table lookups of the mutually exclusive families share interactions ("slots").
    # A slot is one interaction [OP, R, B, C] whose entries are affine: OP = sum of (table opcode x family selector),
    # R / B / C = a union column (plus a constant x selector where a family needs a constant there), multiplicity = the
    # sum of the selectors.  Families put their operands into the slot's columns (copying them from a / b / c where the
    # operands live outside the union block); a column a family does not use is 0 on its rows, which is also what the
    # table forces for the U16 entries [7, 0, v, 0].  The MUL family's seven carry checks provide the slots:
    #   u[4..7]  : AND / OR / XOR byte i  (R = u[11+i] = a_i, B = u[4+i] = b_i, C = u[19+i] = c_i; not u[18], which the
    #              timestamp range check of every row reads as the memory family's high limb)
    #   u[8]     : the memory family's timestamp limb (below)
    #   u[9]     : comparator, sign bit of c's top byte (R = u[21], B = u[9] = c_3)
    #   u[10]    : comparator, the byte comparison itself (R = u[19] = lt, B = u[10] = b_cmp, C = u[20] = c_cmp)
    x, mcy = U[0:4], U[4:11] + [Expr.const(0)]

    # AND / OR / XOR : four byte lookups on copies of the operand bytes
    op_bit = bit_op
    for i in range(4):
        ch.assert_zero(sel_bit * (U[11 + i] - a[i]))
        ch.assert_zero(sel_bit * (U[4 + i] - b[i]))
        ch.assert_zero(sel_bit * (U[19 + i] - c[i]))
        ch.send("byte", [op_bit + B_U16 * sel_mul, U[11 + i], U[4 + i], U[19 + i]], sel_bit + sel_mul)

    # comparator (SLT, SLTU, branches): u[0..3] differing-byte flags, u[4] 1/(b_cmp - c_cmp), u[10] b_cmp, u[20] c_cmp,
    # u[19] lt, u[9] copy of c_3 and u[21] its top bit, u[24] copy of b_3 and u[25] its top bit (the slot of the
    # sub-word loads' sign byte)
    df, inv_d, b_cmp, c_cmp, lt = U[0:4], U[4], U[10], U[20], U[19]
    c3c, msb_c, b3c, msb_b = U[9], U[21], U[24], U[25]
    bt = b[3] + 128 * sel_signed - 256 * msb_b       # top bytes with the sign bit flipped when signed
    ct = c[3] + 128 * sel_signed - 256 * msb_c
    bb = b[0:3] + [bt]
    cc = c[0:3] + [ct]
    any_df = esum(df)
    for i in range(4):
        ch.assert_zero(sel_cmp * (df[i] * (df[i] - 1)))
        ch.assert_zero(sel_cmp * ((1 - esum(df[i:])) * (bb[i] - cc[i])))   # equal above the flagged byte
    ch.assert_zero(sel_cmp * (any_df * (any_df - 1)))
    ch.assert_zero(sel_cmp * (b_cmp - esum(df[i] * bb[i] for i in range(4))))
    ch.assert_zero(sel_cmp * (c_cmp - esum(df[i] * cc[i] for i in range(4))))
    ch.assert_zero(sel_cmp * ((b_cmp - c_cmp) * inv_d - any_df))
    ch.assert_zero((sel_cmp - sel_signed) * msb_b)
    ch.assert_zero((sel_cmp - sel_signed) * msb_c)
    ch.assert_zero(sel_signed * (b3c - b[3]))
    ch.assert_zero(sel_signed * (c3c - c[3]))
    # (MUL: carry 6 in u[10]; JAL / JALR: the top byte of the link value is below 0x78, see the link constraints below)
    ch.send("byte", [B_LTU * (sel_cmp + sel_link) + B_U16 * sel_mul, lt, b_cmp, c_cmp], sel_cmp + sel_mul + sel_link)
    ch.send("byte", [B_MSB * sel_signed + B_U16 * sel_mul, msb_c, c3c, 0], sel_signed + sel_mul)        # (MUL: carry 5 in u[9])
    # (the sign of b's top byte goes through the slot [MSB, u[25], u[24], 0] of the sub-word loads, below)
    sel_set = F["is_set"]
    ch.assert_zero(sel_set * (a[0] - lt))
    for i in range(1, 4):
        ch.assert_zero(sel_set * a[i])
    is_eq = 1 - any_df
    taken = F["is_beq"] * is_eq + F["is_bne"] * (1 - is_eq) + F["is_brlt"] * lt + F["is_brge"] * (1 - lt)
    ch.assert_zero(sel_branch * (next_pc - pc - 4) - taken * (tgt - pc - 4))

    # MUL / MULHU : the half of the 64-bit product that is NOT the result lives in u[0..3] (x), the result half is `a`
    # itself; u[4..10] = the carries out of bytes 0..6 (the carry out of byte 7 of a 64-bit product is the constant 0).
    # The range check of x is the adder family's range check of u[0..3], the range check of `a` is the common one,
    # every carry check is one of the shared slots above.
    for k in range(8):
        terms = esum(b[i] * c[k - i] for i in range(4) if 0 <= k - i < 4)
        cin = mcy[k - 1] if k else Expr.const(0)
        pk = F["is_mul"] * a[k] + F["is_mulhu"] * x[k] if k < 4 else F["is_mul"] * x[k - 4] + F["is_mulhu"] * a[k - 4]
        ch.assert_zero(sel_mul * (terms + cin - 256 * mcy[k]) - pk)

    # LUI / AUIPC : a := imm   (the pc-relative constant is folded at decode time)
    for i in range(4):
        ch.assert_zero(F["is_lui"] * (a[i] - imm[i]))
    # JAL / JALR : a := pc + 4.  The bytes of a are range-checked (below, with the arithmetic families) and its top byte is
    # below 0x78 through the comparator's lookup slot (u[19] = 1 = "u[10] < u[20]", u[10] = a_3, u[20] = 0x78), so the
    # equality mod p is an equality of 32-bit values: pc + 4 < 0x38000004 and the alias pc + 4 + p starts with a byte >= 0x78
    ch.assert_zero(sel_link * (word(a) - pc - 4))
    ch.assert_zero(sel_link * (lt - 1))
    ch.assert_zero(sel_link * (b_cmp - a[3]))
    ch.assert_zero(sel_link * (c_cmp - LINK_TOP_BYTE))
    ch.assert_zero(F["is_jal"] * (next_pc - tgt))

    # address adder (LW, SW, JALR): u[0..3] sum bytes, u[4..7] carries
    s, acy = U[0:4], U[4:8]
    for i in range(4):
        cin = acy[i - 1] if i else Expr.const(0)
        ch.assert_zero(sel_adder * (b[i] + off[i] + cin - s[i] - 256 * acy[i]))
        ch.assert_zero(sel_adder * (acy[i] * (acy[i] - 1)))
    # range of the four sum bytes, address / target < 0x38000000, and the byte offset (low two address bits) in TWO lookups:
    # [RANGE, 0, s1, s2] and [ADDR, offset, s0, s3] (byte-table op B_ADDR).  `offset` = o1 + 2 o2 + 3 o3, a one-hot value in
    # 0..3 on every adder row (JALR included: with free cells the entry (s0 & 3) + 4 would pass for a top byte >= 0x38 and
    # the target bound would not hold); on MUL / MULHU rows (the bytes of x, plain range checks) the entry must be 0 and
    # the cells are.
    o_val = U[21] + 2 * U[22] + 3 * U[23]
    ch.send("byte", [B_RANGE, 0, s[1], s[2]], sel_adder + sel_mul)
    ch.send("byte", [B_ADDR * sel_adder + B_RANGE * sel_mul, o_val, s[0], s[3]], sel_adder + sel_mul)
    # JALR: u[8] = low bit cleared from the target
    jl = U[8]
    ch.assert_zero(F["is_jalr"] * (jl * (jl - 1)))
    ch.assert_zero(F["is_jalr"] * (next_pc - word(s) + jl))
    # loads / stores: u[8] low limb of the timestamp difference, u[9..12] memory word after, u[13..16] before, u[17] prev clk,
    # u[18] hi limb, u[19] prev shard, u[20] same-shard flag, u[21..23] byte-offset one-hot (offset 0 = none set),
    # u[24] the byte whose sign extends a sub-word load, u[25] its sign bit.
    # The access always moves the whole aligned word on the memory bus; sub-word forms select / patch bytes.
    m_lo, mv, mp, m_ts, m_hi, m_sh, m_same = U[8], U[9:13], U[13:17], U[17], U[18], U[19], U[20]
    o1, o2, o3, sb, sgn = U[21], U[22], U[23], U[24], U[25]
    o0 = 1 - o1 - o2 - o3
    oh = [o0, o1, o2, o3]
    for x in (o1, o2, o3):
        ch.assert_zero(sel_adder * (x * (x - 1)))
    ch.assert_zero(sel_adder * ((o1 + o2 + o3) * (o1 + o2 + o3 - 1)))
    # (offset = the low two address bits: the B_ADDR lookup above)
    ch.assert_zero((F["is_lw"] + F["is_sw"]) * (o1 + o2 + o3))                      # word access: aligned
    ch.assert_zero((F["is_lh"] + F["is_lhu"] + F["is_sh"]) * (o1 + o3))              # halfword access: even
    maddr = word(s) - o_val
    # (COMMIT and precompile ecalls read their second argument, register a1, through this port: sys_m rows pin the address to 11)
    sel_port = sel_mem + sys_m
    ch.receive("mem", [maddr] + mp + [m_sh, m_ts], sel_port)
    ch.send("mem", [maddr] + mv + [shard, clk + 2], sel_port)
    ch.assert_zero(sel_port * (m_same * (m_same - 1)))
    ch.assert_zero(sel_port * (m_same * (shard - m_sh)))
    ch.assert_zero(sel_port * (m_same * (clk + 2 - m_ts - 1) + (1 - m_same) * (shard - m_sh - 1) - m_lo - 65536 * m_hi))
    ch.send("byte", [B_U16, 0, m_lo, 0], sel_port + sel_mul)            # (MUL / MULHU: carry 4, also in u[8])
    sel_load = F["is_lw"] + sel_loadsub
    for i in range(4):
        ch.assert_zero(F["is_lw"] * (a[i] - mv[i]))
        ch.assert_zero(sel_load * (mp[i] - mv[i]))                                   # loads leave memory unchanged
        ch.assert_zero(F["is_sw"] * (mv[i] - c[i]))
        ch.assert_zero(F["is_sb"] * (mv[i] - mp[i] - oh[i] * (c[0] - mp[i])))        # patch byte `offset`
    ch.assert_zero(F["is_sh"] * (mv[0] - mp[0] - o0 * (c[0] - mp[0])))
    ch.assert_zero(F["is_sh"] * (mv[1] - mp[1] - o0 * (c[1] - mp[1])))
    ch.assert_zero(F["is_sh"] * (mv[2] - mp[2] - o2 * (c[0] - mp[2])))
    ch.assert_zero(F["is_sh"] * (mv[3] - mp[3] - o2 * (c[1] - mp[3])))
    sel_byte, sel_half, sel_sext = F["is_lb"] + F["is_lbu"], F["is_lh"] + F["is_lhu"], F["is_lb"] + F["is_lh"]
    ch.assert_zero(sel_byte * (a[0] - esum(oh[i] * mp[i] for i in range(4))))
    ch.assert_zero(sel_half * (a[0] - o0 * mp[0] - o2 * mp[2]))
    ch.assert_zero(sel_half * (a[1] - o0 * mp[1] - o2 * mp[3]))
    ch.assert_zero(F["is_lb"] * (sb - a[0]))
    ch.assert_zero(F["is_lh"] * (sb - a[1]))
    ch.send("byte", [B_MSB, sgn, sb, 0], sel_sext + sel_signed)         # (signed compares: u[24] = b_3, u[25] = its top bit)
    ch.assert_zero((F["is_lbu"] + F["is_lhu"]) * sgn)
    ch.assert_zero(sel_byte * (a[1] - 255 * sgn))
    for i in (2, 3):
        ch.assert_zero(sel_loadsub * (a[i] - 255 * sgn))
    # the 8-bit high limbs of the four timestamp differences, two per lookup
    ch.send("byte", [B_RANGE, 0, pb_hi, pc_hi], is_real)
    ch.send("byte", [B_RANGE, 0, pa_hi, m_hi], is_real)      # (u[16..19] belong to the memory family only)

    # range check of a for the families that compute it arithmetically
    sel_range_a = sel_addsub + sel_mul + F["is_ecall"] + sel_link
    ch.send("byte", [B_RANGE, 0, a[0], a[1]], sel_range_a)
    ch.send("byte", [B_RANGE, 0, a[2], a[3]], sel_range_a)

    # everything that is not a branch / jump / ecall falls through
    sel_seq = is_real - sel_branch - F["is_jal"] - F["is_jalr"] - F["is_ecall"]
    ch.assert_zero(sel_seq * (next_pc - pc - 4))

    # ECALL: b = t0 (syscall id), c = a0; a = new t0 (advice).  id 0 = HALT(exit code a0).
    # (witness cells in u[4..7]: u[0..3] hold the address bytes of the a1 read of COMMIT rows)
    # The call id is compared through idw = b0 + 256 b1 + 65536 b2 + 2^22 b3 (at most 1.09e9 < p: no reduction mod p), NOT
    # through word(b): t0 = id + p is a valid 32-bit value that word(b) cannot tell from id.  For a target id below 2^8,
    # idw = id holds only for the bytes (id, 0, 0, 0): the higher terms are multiples of 256 and b0 - id lies in (-256, 256).
    idw = b[0] + 256 * b[1] + 65536 * b[2] + (1 << 22) * b[3]
    is_halt, id_inv = U[4], U[5]
    ec = F["is_ecall"]
    ch.assert_zero(ec * (is_halt * (is_halt - 1)))
    ch.assert_zero(ec * (is_halt * idw))
    ch.assert_zero(ec * (idw * id_inv - (1 - is_halt)))
    ch.assert_zero(ec * (next_pc - (1 - is_halt) * (pc + 4) - is_halt * HALT_PC))
    ch.assert_zero(ec * (is_halt * (word(c) - ch.pub(PUB_EXIT_CODE))))
    ch.assert_zero(ec * (is_halt * c[3]))        # exit codes are below 2^24: word(c) is the exit code itself, not a residue
    # only HINT_LEN (id 0xF0) returns a value in t0 (advice: the length of the next stdin buffer, private input like the
    # buffer itself); every other call leaves t0 unchanged.  u[24] = "is HINT_LEN", u[25] = 1 / (id - 0xF0)
    is_hl, hl_inv = U[24], U[25]
    ch.assert_zero(ec * (is_hl * (is_hl - 1)))
    ch.assert_zero(ec * (is_hl * (idw - SYS_HINT_LEN)))
    ch.assert_zero(ec * ((idw - SYS_HINT_LEN) * hl_inv - (1 - is_hl)))
    for i in range(4):
        ch.assert_zero(ec * ((1 - is_hl) * (a[i] - b[i])))
    # COMMIT (id 0x10, a0 = index, a1 = word) — SP1's syscall contract: the guest hashes the bytes
    # it wrote to fd 3 with SHA-256 and commits the eight digest words, COMMIT(k, digest word k), before HALT
    # (reference crates/finalization_prove/src/main.rs:26-32 via sp1_zkvm::io::commit).
    # PRECOMPILES: SP1 syscall codes carry "this call has a table" in byte 1; such rows (is_pre) hand (code, a0, a1, clk,
    # shard) to the precompile's own chip, which does the memory accesses of the call at (shard, clk + 2).
    # Both kinds of row ("sys rows", multiplicity sys_m) read a1 = x11 through the memory port (address pinned to 11 by ONE
    # affine constraint on the address expression, value unchanged) and send
    #     [t0 bytes, a0 bytes, a1 bytes, u_clk, u_sh]        on the "sys" bus,
    # u_clk = u_sh = 0 on COMMIT rows (the VERIFIER supplies the receiving side from SHA-256 of the claimed public-value
    # bytes, so they are bound to the proof), u_clk = clk and u_sh = shard on precompile rows.
    # Witness cells (free on sys rows because only the address EXPRESSION is pinned): u[1] is_pre, u[2] 1 / (t0 byte 1 - 1),
    # u[3] u_clk, u[21] u_sh; u[0] balances the address expression.
    is_commit, cm_inv = U[6], U[7]
    ch.assert_zero(ec * (is_commit * (is_commit - 1)))
    ch.assert_zero(ec * (is_commit * (idw - SYS_COMMIT)))
    ch.assert_zero(ec * ((idw - SYS_COMMIT) * cm_inv - (1 - is_commit)))
    is_pre, pre_inv, u_clk, u_sh = U[1], U[2], U[3], U[21]
    ch.assert_zero(ec * (is_pre * (is_pre - 1)))
    ch.assert_zero(ec * (is_pre * (b[1] - 1)))
    ch.assert_zero(ec * ((b[1] - 1) * pre_inv - (1 - is_pre)))
    ch.assert_zero(sys_m - ec * (is_commit + is_pre))
    ch.assert_zero(sys_m * (maddr - (REG_BASE + REG_A1)))
    for i in range(4):
        ch.assert_zero(sys_m * (mv[i] - mp[i]))
    ch.assert_zero(ec * (is_pre * (u_clk - clk)))
    ch.assert_zero(ec * (is_pre * (u_sh - shard)))
    ch.assert_zero(ec * (is_commit * u_clk))
    ch.assert_zero(ec * (is_commit * u_sh))
    ch.send("sys", b + c + mv + [u_clk, u_sh], sys_m)
    ch.quotient_parts = 3
    ch.logup_parts = 4
    return ch


def build_shift():
    """SLL / SRL / SRA rows received from the cpu chip over the alu bus.  shift = 8*q + r: the bit shift by r
    splits every byte with the multiplier m = 2^r, the byte shift by q moves whole bytes."""
    ch = Chip("shift")
    is_real = ch.col("is_real")
    is_sll, is_srl, is_sra = ch.col("is_sll"), ch.col("is_srl"), ch.col("is_sra")
    a, b, c = ch.cols("a", 4), ch.cols("b", 4), ch.cols("c", 4)
    sh = ch.col("sh")
    q = ch.cols("q", 3)          # byte shift 1..3 (0 = none set)
    r = ch.cols("r", 8)          # one-hot bit shift 0..7
    lo, hi, t = ch.cols("lo", 4), ch.cols("hi", 4), ch.cols("t", 4)
    sgn = ch.col("sgn")
    ch.assert_bool(is_real)
    for x in (is_sll, is_srl, is_sra):
        ch.assert_bool(x)
    ch.assert_eq(is_sll + is_srl + is_sra, is_real)
    for x in q + r:
        ch.assert_bool(x)
    ch.assert_eq(esum(r), is_real)
    qs = esum(q)
    ch.assert_zero(qs * (qs - 1))
    q0 = 1 - qs
    qq = [q0] + q
    ch.assert_eq(sh, 8 * (q[0] + 2 * q[1] + 3 * q[2]) + esum(k * r[k] for k in range(8)))
    m = esum((1 << k) * r[k] for k in range(8))            # 2^r
    mi = esum((1 << (8 - k)) * r[k] for k in range(8))     # 2^(8-r)
    sel_r = is_srl + is_sra
    ch.receive("alu", [ALU_SLL * is_sll + ALU_SRL * is_srl + ALU_SRA * is_sra] + a + b + c, is_real)
    ch.send("byte", [B_AND, sh, c[0], 31], is_real)        # shift amount = low five bits of c
    ch.send("byte", [B_RANGE, 0, lo[0], lo[1]], is_real)
    ch.send("byte", [B_RANGE, 0, lo[2], lo[3]], is_real)
    ch.send("byte", [B_RANGE, 0, hi[0], hi[1]], is_real)
    ch.send("byte", [B_RANGE, 0, hi[2], hi[3]], is_real)
    ch.send("byte", [B_MSB, sgn, b[3], 0], is_sra)
    ch.assert_zero((is_sll + is_srl) * sgn)
    for i in range(4):
        # left: b_i * 2^r = lo_i + 256 hi_i, shifted byte t_i = lo_i + hi_(i-1)
        ch.assert_zero(is_sll * (b[i] * m - lo[i] - 256 * hi[i]))
        ch.assert_zero(is_sll * (t[i] - lo[i] - (hi[i - 1] if i else Expr.const(0))))
        # right: b_i = hi_i * 2^r + lo_i with lo_i < 2^r, shifted byte t_i = hi_i + lo_(i+1) * 2^(8-r);
        # above the top byte sit the sign bits: lo_4 = sgn * (2^r - 1)
        ch.assert_zero(sel_r * (b[i] - hi[i] * m - lo[i]))
        ch.send("byte", [B_LTU, 1, lo[i], m], sel_r)
        # (lo_4 * 2^(8-r) = sgn * (2^r - 1) * 2^(8-r) = sgn * (256 - 2^(8-r)) since 2^r * 2^(8-r) = 256)
        up = lo[i + 1] * mi if i < 3 else sgn * (256 - mi)
        ch.assert_zero(sel_r * (t[i] - hi[i] - up))
    fill = 255 * sgn
    for i in range(4):
        left = esum(qq[k] * t[i - k] for k in range(4) if i - k >= 0)
        right = esum(qq[k] * (t[i + k] if i + k < 4 else fill) for k in range(4))
        ch.assert_zero(is_sll * (a[i] - left))
        ch.assert_zero(sel_r * (a[i] - right))
    ch.quotient_parts = 2
    return ch


def build_muldiv():
    """MULH / MULHSU / DIV / DIVU / REM / REMU rows received from the cpu chip over the alu bus.
    One relation serves all six: X * Y (+ R) = B as two's-complement 64-bit numbers, with
      multiplications:  X = b, Y = c, result a = high word of the product
      divisions:        X = quotient q, Y = c, R = remainder r, B = dividend b;  a = q or r.
    The unsigned 64-bit product of the words is built from bytes (as the cpu chip's MUL family does); sign extension
    of X (flag sx) and Y (sy) only changes the high word: hi = prod_hi - sx * Y - sy * X (mod 2^32).
    Division: q * c + r = b holds over the integers (|q c| < 2^62, no wrap), |r| < |c| and sign(r) = sign(b) pin q and r;
    RISC-V's two special cases are flags: c = 0 (q = all ones, r = b falls out of the equation) and the signed overflow
    -2^31 / -1 (q = -2^31, r = 0, the equation is waived)."""
    ch = Chip("muldiv")
    is_real = ch.col("is_real")
    names = ["mulh", "mulhsu", "div", "divu", "rem", "remu"]
    f = {n: ch.col("is_" + n) for n in names}
    a, b, c = ch.cols("a", 4), ch.cols("b", 4), ch.cols("c", 4)
    q, r = ch.cols("q", 4), ch.cols("r", 4)
    prod, mcy = ch.cols("prod", 8), ch.cols("mcy", 8)
    h, bw = ch.cols("h", 4), ch.cols("bw", 4)
    mx, my, mr, mb = ch.col("mx"), ch.col("my"), ch.col("mr"), ch.col("mb")      # top bits of q3, c3, r3, b3
    sx, sy, sr, sb = ch.col("sx"), ch.col("sy"), ch.col("sr"), ch.col("sb")      # ... where the operation reads them as signs
    is_c0, cinv, is_ovf = ch.col("is_c0"), ch.col("cinv"), ch.col("is_ovf")
    dl, ea, eb = ch.cols("dl", 2), ch.col("ea"), ch.col("eb")
    dcy = ch.cols("dcy", 4)

    ch.assert_bool(is_real)
    for n in names:
        ch.assert_bool(f[n])
    ch.assert_eq(esum(f.values()), is_real)
    codes = dict(mulh=ALU_MULH, mulhsu=ALU_MULHSU, div=ALU_DIV, divu=ALU_DIVU, rem=ALU_REM, remu=ALU_REMU)
    ch.receive("alu", [esum(codes[n] * f[n] for n in names)] + a + b + c, is_real)
    is_mul = f["mulh"] + f["mulhsu"]
    is_dr = f["div"] + f["divu"] + f["rem"] + f["remu"]
    is_sdr = f["div"] + f["rem"]

    # operands and results
    for i in range(4):
        ch.assert_zero(is_mul * (q[i] - b[i]))                   # X = b for multiplications
        ch.assert_zero(is_mul * (a[i] - h[i]))
        ch.assert_zero((f["div"] + f["divu"]) * (a[i] - q[i]))
        ch.assert_zero((f["rem"] + f["remu"]) * (a[i] - r[i]))
    # signs: top bits from the byte table, used only where the operation is signed
    ch.send("byte", [B_MSB, mx, q[3], 0], is_real)
    ch.send("byte", [B_MSB, my, c[3], 0], is_real)
    ch.send("byte", [B_MSB, mr, r[3], 0], is_real)
    ch.send("byte", [B_MSB, mb, b[3], 0], is_real)
    ch.assert_eq(sx, mx * (is_mul + is_sdr))
    ch.assert_eq(sy, my * (f["mulh"] + is_sdr))
    ch.assert_eq(sr, mr * is_sdr)
    ch.assert_eq(sb, mb * is_sdr)

    # unsigned product X * Y: 8 bytes + carries
    for k in range(8):
        terms = esum(q[i] * c[k - i] for i in range(4) if 0 <= k - i < 4)
        cin = mcy[k - 1] if k else Expr.const(0)
        ch.assert_zero(terms + cin - prod[k] - 256 * mcy[k])
    for k in range(4):
        ch.send("byte", [B_RANGE, 0, prod[2 * k], prod[2 * k + 1]], is_real)
    for k in range(8):
        ch.send("byte", [B_U16, 0, mcy[k], 0], is_real)
    # high word of the signed product: h = prod_hi - sx * Y - sy * X (mod 2^32), borrows 0..2
    for i in range(4):
        bin_ = bw[i - 1] if i else Expr.const(0)
        ch.assert_zero(prod[4 + i] - sx * c[i] - sy * q[i] - bin_ + 256 * bw[i] - h[i])
        ch.assert_zero(bw[i] * (bw[i] - 1) * (bw[i] - 2))
    ch.send("byte", [B_RANGE, 0, h[0], h[1]], is_real)
    ch.send("byte", [B_RANGE, 0, h[2], h[3]], is_real)
    ch.send("byte", [B_RANGE, 0, q[0], q[1]], is_real)
    ch.send("byte", [B_RANGE, 0, q[2], q[3]], is_real)
    ch.send("byte", [B_RANGE, 0, r[0], r[1]], is_real)
    ch.send("byte", [B_RANGE, 0, r[2], r[3]], is_real)

    # ---- divisions
    csum = esum(c)
    ch.assert_bool(is_c0)
    ch.assert_bool(is_ovf)
    ch.assert_zero(is_c0 * (1 - is_dr))
    ch.assert_zero(is_ovf * (1 - is_sdr))
    ch.assert_zero(is_c0 * csum)
    ch.assert_zero(is_dr * (csum * cinv - (1 - is_c0)))
    for i in range(4):
        ch.assert_zero(is_c0 * (q[i] - 255))                     # x / 0 = all ones
        ch.assert_zero(is_ovf * (c[i] - 255))                    # -2^31 / -1 = -2^31 remainder 0
        ch.assert_zero(is_ovf * (b[i] - (128 if i == 3 else 0)))
        ch.assert_zero(is_ovf * (q[i] - b[i]))
        ch.assert_zero(is_ovf * r[i])
    # q * c + r = b as 64-bit two's-complement numbers, in 16-bit limbs (the field holds < 2^31)
    P = [prod[0] + 256 * prod[1], prod[2] + 256 * prod[3], h[0] + 256 * h[1], h[2] + 256 * h[3]]
    R = [r[0] + 256 * r[1], r[2] + 256 * r[3], 65535 * sr, 65535 * sr]
    Bv = [b[0] + 256 * b[1], b[2] + 256 * b[3], 65535 * sb, 65535 * sb]
    for k in range(4):
        cin = dcy[k - 1] if k else Expr.const(0)
        ch.assert_zero((is_dr - is_ovf) * (P[k] + R[k] + cin - Bv[k] - 65536 * dcy[k]))
        ch.assert_bool(dcy[k])
    # sign(r) = sign(b) unless r = 0
    for i in range(4):
        ch.assert_zero(is_dr * ((sr - sb) * r[i]))
    # |r| < |c|:  |c| - |r| - 1 = dl (two 16-bit limbs), |v| = (1 - 2 s_v) v + s_v 2^32
    # (low-limb carry e0 = ea + 2 eb - 1 in {-1, 0, 1, 2}: the low limbs span [-131071, 131069])
    sc_, sr_ = 1 - 2 * sy, 1 - 2 * sr
    sel_lt = is_dr - is_c0
    e0 = ea + 2 * eb - 1
    ch.assert_bool(ea)
    ch.assert_bool(eb)
    ch.assert_zero(sel_lt * (sc_ * (c[0] + 256 * c[1]) - sr_ * (r[0] + 256 * r[1]) - 1 + 65536 * e0 - dl[0]))
    ch.assert_zero(sel_lt * (sc_ * (c[2] + 256 * c[3]) - sr_ * (r[2] + 256 * r[3]) + 65536 * (sy - sr) - e0 - dl[1]))
    ch.send("byte", [B_U16, 0, dl[0], 0], is_real)
    ch.send("byte", [B_U16, 0, dl[1], 0], is_real)
    ch.quotient_parts = 2
    ch.logup_parts = 2
    return ch


def build_sha_extend():
    """SHA_EXTEND precompile (SP1 syscall 0x00_30_01_05, a0 = pointer to a 64-word array, a1 = 0): w[16..63] of the SHA-256
    message schedule, w[i] = s1(w[i-2]) + w[i-7] + s0(w[i-15]) + w[i-16], computed in place.
    64 rows per call: rows j = 0..15 READ w[j] into a 16-word window carried from row to row, rows j = 16..63 compute the
    next word from the window and WRITE it; every row makes exactly one memory access, at (shard, clk + 2) of the ECALL
    (64 distinct addresses, so one timestamp serves the whole call and the cpu clock does not skip).
    The first row of a call receives the cpu chip's tuple from the sys bus."""
    ch = Chip("sha_extend")
    shard = ch.pub(PUB_SHARD)
    is_real, is_first, is_last, is_load, is_e = ch.col("is_real"), ch.col("is_first"), ch.col("is_last"), ch.col("is_load"), ch.col("is_e")
    j, j_inv, clk = ch.col("j"), ch.col("j_inv"), ch.col("clk")
    p = ch.cols("p", 4)                                   # w_ptr
    W = [ch.cols(f"w{k}", 4) for k in range(16)]            # window: row j holds w[j-16+k] in W[k] (garbage before it fills)
    nw, old = ch.cols("nw", 4), ch.cols("old", 4)          # the word this row reads / writes, the memory word it replaces
    xb, yb = ch.cols("xb", 32), ch.cols("yb", 32)          # bits of W[1] (s0 input) and of W[14] (s1 input)
    s0, s1 = ch.cols("s0", 2), ch.cols("s1", 2)            # 16-bit halves of s0(W[1]), s1(W[14])
    cy = ch.cols("cy", 4)                                  # two carry bits per half
    m_sh, m_ts, m_same, m_lo, m_hi = ch.col("m_sh"), ch.col("m_ts"), ch.col("m_same"), ch.col("m_lo"), ch.col("m_hi")
    nr = is_real.next()
    # ---- row structure
    for f in (is_real, is_first, is_last, is_load, is_e):
        ch.assert_bool(f)
    ch.assert_zero(nr * (1 - is_real), "trans")                       # real rows first
    ch.assert_eq(is_first, is_real, "first")
    ch.assert_zero(nr * (is_first.next() - is_last), "trans")         # a call starts right after the previous one ends
    ch.assert_zero((is_real - nr) * (1 - is_last), "trans")           # ... and the table ends with a complete call
    ch.assert_zero(is_real * (1 - is_last), "last")
    for f in (is_first, is_last, is_load, is_e):
        ch.assert_zero((1 - is_real) * f)
    ch.assert_zero(is_first * j)
    ch.assert_zero(is_last * (j - 63))
    ch.assert_zero(is_real * ((j - 63) * j_inv - (1 - is_last)))
    inner = is_real - is_last           # 1 iff this row and the next belong to the same call (is_last implies is_real)
    ch.assert_zero(inner * (j.next() - j - 1), "trans")
    ch.assert_zero(inner * (clk.next() - clk), "trans")
    for i in range(4):
        ch.assert_zero(inner * (p[i].next() - p[i]), "trans")
    # is_load = 1 on rows 0..15: starts at 1, ends at 0, drops exactly where is_e marks row 15
    ch.assert_zero(is_first * (1 - is_load))
    ch.assert_zero(is_last * is_load)
    ch.assert_zero(is_e * (j - 15))
    ch.assert_zero(inner * (is_load - is_load.next() - is_e), "trans")
    # ---- the call: (code bytes, a0 bytes, a1 = 0, clk, shard) from the cpu chip; pointer word-aligned and below 0x38000000
    code = [(SYS_SHA_EXTEND >> (8 * i)) & 0xFF for i in range(4)]
    ch.receive("sys", code + p + [0, 0, 0, 0] + [clk, shard], is_first)
    ch.send("byte", [B_ADDR, 0, p[0], p[3]], is_first)
    # ---- window
    for k in range(15):
        for i in range(4):
            ch.assert_zero(inner * (W[k][i].next() - W[k + 1][i]), "trans")
    for i in range(4):
        ch.assert_zero(inner * (W[15][i].next() - nw[i]), "trans")
    # ---- s0(x) = rotr7 ^ rotr18 ^ shr3 of x = W[1], s1(y) = rotr17 ^ rotr19 ^ shr10 of y = W[14], from bits
    for bits, wd in ((xb, W[1]), (yb, W[14])):
        for t in bits:
            ch.assert_bool(t)
        for i in range(4):
            ch.assert_eq(wd[i], esum((1 << k) * bits[8 * i + k] for k in range(8)))

    def xor3(a, b_, c_):
        return a + b_ + c_ - 2 * (a * b_ + a * c_ + b_ * c_) + 4 * (a * b_ * c_)

    def xor2(a, b_):
        return a + b_ - 2 * (a * b_)

    def sigma(bits, r1, r2, sh):
        out = []
        for k in range(32):
            a_, b_ = bits[(k + r1) % 32], bits[(k + r2) % 32]
            out.append(xor3(a_, b_, bits[k + sh]) if k + sh < 32 else xor2(a_, b_))
        return out

    g0, g1 = sigma(xb, 7, 18, 3), sigma(yb, 17, 19, 10)
    for h in range(2):
        ch.assert_eq(s0[h], esum((1 << k) * g0[16 * h + k] for k in range(16)))
        ch.assert_eq(s1[h], esum((1 << k) * g1[16 * h + k] for k in range(16)))
    # ---- compute rows: nw = s1 + w[j-7] + s0 + w[j-16] mod 2^32, in 16-bit halves (carries 0..3 as two bits each)
    for t in cy:
        ch.assert_bool(t)
    half = lambda wd, h: wd[2 * h] + 256 * wd[2 * h + 1]
    c_lo, c_hi = cy[0] + 2 * cy[1], cy[2] + 2 * cy[3]
    ch.assert_zero((1 - is_load) * (half(W[0], 0) + s0[0] + half(W[9], 0) + s1[0] - half(nw, 0) - 65536 * c_lo))
    ch.assert_zero((1 - is_load) * (half(W[0], 1) + s0[1] + half(W[9], 1) + s1[1] + c_lo - half(nw, 1) - 65536 * c_hi))
    ch.send("byte", [B_RANGE, 0, nw[0], nw[1]], is_real - is_load)
    ch.send("byte", [B_RANGE, 0, nw[2], nw[3]], is_real - is_load)
    # ---- the row's memory access: word j of the array, read (unchanged) on load rows, written on compute rows
    addr = word(p) + 4 * j
    for i in range(4):
        ch.assert_zero(is_load * (old[i] - nw[i]))
    ch.receive("mem", [addr] + old + [m_sh, m_ts], is_real)
    ch.send("mem", [addr] + nw + [shard, clk + 2], is_real)
    ch.assert_zero(is_real * (m_same * (m_same - 1)))
    ch.assert_zero(is_real * (m_same * (shard - m_sh)))
    ch.assert_zero(is_real * (m_same * (clk + 2 - m_ts - 1) + (1 - m_same) * (shard - m_sh - 1) - m_lo - 65536 * m_hi))
    ch.send("byte", [B_U16, 0, m_lo, 0], is_real)
    ch.send("byte", [B_RANGE, 0, m_hi, 0], is_real)
    ch.quotient_parts = 2
    return ch


def build_sha_compress():
    """SHA_COMPRESS precompile (SP1 syscall 0x00_01_01_06, a0 = pointer to the 64 schedule words, a1 = pointer to the
    8 state words): the SHA-256 compression function, state updated in place.
    80 rows per call, row j = 8 g + o (group g = 0..9, position o = 0..7 as one-hot flags):
      g = 0      row o READS state word 7 - o into the working variable a,
      g = 1..8   round i = 8 (g - 1) + o: READS w[i]; a' = T1 + T2, e' = d + T1,
      g = 9      row o does state[7 - o] += h (read-modify-write, one timestamp later than the reads).
    On EVERY row the working variables rotate a -> b -> ... -> h, so the eight loaded words line up as (a, ..., h) after
    group 0 and pass through h, in the order 7, 6, ..., 0, in group 9.  a, b, c, e, f, g are kept as bits (Sigma / Ch / Maj), d and
    h as 16-bit halves.  One memory access per row: the reads at (shard, clk + 2) of the ECALL, the write-back at clk + 3."""
    ch = Chip("sha_compress")
    shard = ch.pub(PUB_SHARD)
    is_real, is_first = ch.col("is_real"), ch.col("is_first")
    oc, gr = ch.cols("oc", 8), ch.cols("gr", 10)
    clk = ch.col("clk")
    wp, hp = ch.cols("wp", 4), ch.cols("hp", 4)
    A, Bb, Cc = ch.cols("ab", 32), ch.cols("bb", 32), ch.cols("cb", 32)
    E, Fb, G = ch.cols("eb", 32), ch.cols("fb", 32), ch.cols("gb", 32)
    d, h = ch.cols("d", 2), ch.cols("h", 2)
    S1, S0, MJ = ch.cols("s1", 2), ch.cols("s0", 2), ch.cols("mj", 2)
    ce, ca, cf = ch.cols("ce", 6), ch.cols("ca", 6), ch.cols("cf", 2)      # carries of e' (3 bits per half), a', the write-back
    maddr = ch.col("maddr")
    mv, mo = ch.cols("mv", 4), ch.cols("mo", 4)          # memory word after / before the access
    m_sh, m_ts, m_same, m_lo, m_hi = ch.col("m_sh"), ch.col("m_ts"), ch.col("m_same"), ch.col("m_lo"), ch.col("m_hi")
    nr = is_real.next()
    is_init, is_fin = gr[0], gr[9]
    is_round = esum(gr[1:9])
    is_last = ch.col("is_last")
    # ---- row structure: position and group one-hot, position advances every row, group when position wraps
    for f in [is_real, is_first, is_last] + oc + gr:
        ch.assert_bool(f)
    ch.assert_eq(esum(oc), is_real)
    ch.assert_eq(esum(gr), is_real)
    ch.assert_zero(nr * (1 - is_real), "trans")
    ch.assert_eq(is_first, is_real, "first")
    ch.assert_zero(is_first - gr[0] * oc[0])
    ch.assert_zero(is_last - gr[9] * oc[7])
    ch.assert_zero(nr * (is_first.next() - is_last), "trans")        # a call starts right after the previous one ends
    ch.assert_zero((is_real - nr) * (1 - is_last), "trans")          # ... and the table ends with a complete call
    ch.assert_zero(is_real * (1 - is_last), "last")
    inner = is_real - is_last
    for o in range(8):
        ch.assert_zero(nr * (oc[(o + 1) % 8].next() - oc[o]), "trans")
    # (no gating by "the next row is real": group 0 is entered through the next row's is_first, not by wrapping from
    #  group 9, so the equations also hold from the last call into the padding, where every flag is 0)
    ch.assert_zero(gr[0].next() - gr[0] * (1 - oc[7]) - is_first.next(), "trans")
    for g in range(1, 10):
        ch.assert_zero(gr[g].next() - gr[g] * (1 - oc[7]) - gr[g - 1] * oc[7], "trans")
    ch.assert_zero(inner * (clk.next() - clk), "trans")
    for i in range(4):
        ch.assert_zero(inner * (wp[i].next() - wp[i]), "trans")
        ch.assert_zero(inner * (hp[i].next() - hp[i]), "trans")
    # ---- the call from the cpu chip; both pointers word-aligned and below 0x38000000
    code = [(SYS_SHA_COMPRESS >> (8 * i)) & 0xFF for i in range(4)]
    ch.receive("sys", code + wp + hp + [clk, shard], is_first)
    ch.send("byte", [B_ADDR, 0, wp[0], wp[3]], is_first)
    ch.send("byte", [B_ADDR, 0, hp[0], hp[3]], is_first)
    # ---- bits
    for t in A + Bb + Cc + E + Fb + G + ce + ca + cf:
        ch.assert_bool(t)

    def halves(bits):
        return [esum((1 << k) * bits[16 * hh + k] for k in range(16)) for hh in range(2)]

    def xor3(a, b_, c_):
        return a + b_ + c_ - 2 * (a * b_ + a * c_ + b_ * c_) + 4 * (a * b_ * c_)

    def rot3(bits, r1, r2, r3):
        return [xor3(bits[(k + r1) % 32], bits[(k + r2) % 32], bits[(k + r3) % 32]) for k in range(32)]

    sig1, sig0 = rot3(E, 6, 11, 25), rot3(A, 2, 13, 22)
    maj = [A[k] * Bb[k] + A[k] * Cc[k] + Bb[k] * Cc[k] - 2 * (A[k] * Bb[k] * Cc[k]) for k in range(32)]
    chh = [E[k] * Fb[k] + (1 - E[k]) * G[k] for k in range(32)]            # degree 2: used inline
    for hh in range(2):
        ch.assert_eq(S1[hh], halves(sig1)[hh])
        ch.assert_eq(S0[hh], halves(sig0)[hh])
        ch.assert_eq(MJ[hh], halves(maj)[hh])
    CH = halves(chh)
    # round constant of this row: K[8 (g - 1) + o] on round rows
    K = [esum(gr[g] * esum(oc[o] * ((SHA_K[8 * (g - 1) + o] >> (16 * hh)) & 0xFFFF) for o in range(8)) for g in range(1, 9)) for hh in range(2)]
    x = [mv[0] + 256 * mv[1], mv[2] + 256 * mv[3]]                         # the word this row reads (state word or w[i])
    xo = [mo[0] + 256 * mo[1], mo[2] + 256 * mo[3]]
    # ---- rotation of the working variables (every row of a call but its last)
    an, en = halves([t.next() for t in A]), halves([t.next() for t in E])
    for k in range(32):
        ch.assert_zero(inner * (Bb[k].next() - A[k]), "trans")
        ch.assert_zero(inner * (Cc[k].next() - Bb[k]), "trans")
        ch.assert_zero(inner * (Fb[k].next() - E[k]), "trans")
        ch.assert_zero(inner * (G[k].next() - Fb[k]), "trans")
    hc, hg = halves(Cc), halves(G)
    for hh in range(2):
        ch.assert_zero(inner * (d[hh].next() - hc[hh]), "trans")
        ch.assert_zero(inner * (h[hh].next() - hg[hh]), "trans")
    # e' = d + T1 on round rows, e' = d otherwise;  T1 = h + Sigma1(e) + Ch(e, f, g) + K + w
    # a' = T1 + T2 on round rows, a' = the word read on load rows (free in the write-back group); T2 = Sigma0(a) + Maj(a, b, c)
    # (is_round x the degree-2 Ch / K expressions is degree 3, so these equations carry no transition selector: they hold on
    #  every row, the cyclic last -> first one included, because the table's last row is padding or a call's last row and
    #  is_round = is_init = 0 there)
    ce_lo, ce_hi = ce[0] + 2 * ce[1] + 4 * ce[2], ce[3] + 2 * ce[4] + 4 * ce[5]
    ca_lo, ca_hi = ca[0] + 2 * ca[1] + 4 * ca[2], ca[3] + 2 * ca[4] + 4 * ca[5]
    t1 = [h[hh] + S1[hh] + x[hh] for hh in range(2)]                       # the linear part of T1
    ch.assert_zero(is_round * (t1[0] + CH[0] + K[0] + d[0] - en[0] - 65536 * ce_lo) + is_init * (d[0] - en[0]))
    ch.assert_zero(is_round * (t1[1] + CH[1] + K[1] + d[1] + ce_lo - en[1] - 65536 * ce_hi) + is_init * (d[1] - en[1]))
    ch.assert_zero(is_round * (t1[0] + CH[0] + K[0] + S0[0] + MJ[0] - an[0] - 65536 * ca_lo) + is_init * (x[0] - an[0]))
    ch.assert_zero(is_round * (t1[1] + CH[1] + K[1] + S0[1] + MJ[1] + ca_lo - an[1] - 65536 * ca_hi) + is_init * (x[1] - an[1]))
    ch.assert_zero((is_fin - is_last) * (d[0] - en[0]), "trans")
    ch.assert_zero((is_fin - is_last) * (d[1] - en[1]), "trans")
    # ---- write-back group: state[7 - o] = old + h  (mod 2^32)
    ch.assert_zero(is_fin * (xo[0] + h[0] - x[0] - 65536 * cf[0]))
    ch.assert_zero(is_fin * (xo[1] + h[1] + cf[0] - x[1] - 65536 * cf[1]))
    ch.send("byte", [B_RANGE, 0, mv[0], mv[1]], is_fin)
    ch.send("byte", [B_RANGE, 0, mv[2], mv[3]], is_fin)
    for i in range(4):
        ch.assert_zero((is_real - is_fin) * (mv[i] - mo[i]))                # reads leave memory unchanged
    # ---- the row's memory access
    pos = esum(o * oc[o] for o in range(8))
    rnd = 8 * esum((g - 1) * gr[g] for g in range(1, 9)) + pos          # round index on round rows
    ch.assert_zero(maddr - is_round * (word(wp) + 4 * rnd) - (is_init + is_fin) * (word(hp) + 28 - 4 * pos))
    ts = clk + 2 + is_fin
    ch.receive("mem", [maddr] + mo + [m_sh, m_ts], is_real)
    ch.send("mem", [maddr] + mv + [shard, ts], is_real)
    ch.assert_zero(is_real * (m_same * (m_same - 1)))
    ch.assert_zero(is_real * (m_same * (shard - m_sh)))
    ch.assert_zero(is_real * (m_same * (ts - m_ts - 1) + (1 - m_same) * (shard - m_sh - 1) - m_lo - 65536 * m_hi))
    ch.send("byte", [B_U16, 0, m_lo, 0], is_real)
    ch.send("byte", [B_RANGE, 0, m_hi, 0], is_real)
    ch.quotient_parts = 3
    return ch


# ------------------------------------------------------------------------------------------------------------------
# Field and curve precompiles: one row per call.  Operands are little-endian byte vectors in guest memory (SP1's layout:
# a BLS12-381 Fp element is 12 words, an affine point x || y).  A relation "V = 0 (mod p)" is proven as the INTEGER identity
# V + OFF * p - q * p = 0 over byte limbs (dsl.Chip.assert_poly_zero): q is a witness of range-checked bytes, OFF a constant
# that keeps q non-negative.  Every value written back is canonical (< p): the comparison is witnessed on 3-byte groups.
def limbs_of(v, n):
    assert 0 <= v < 1 << (8 * n)
    return [(v >> (8 * i)) & 0xFF for i in range(n)]


def groups3(vec):
    """a byte vector as values of 3-byte groups (below 2^24: no wrap in the field)"""
    return [esum((x * (1 << (8 * t)) if not isinstance(x, int) else Expr.const(x << (8 * t))) for t, x in enumerate(vec[g:g + 3])) for g in range(0, len(vec), 3)]


def range_bytes(ch, vec, mult):
    for i in range(0, len(vec) - 1, 2):
        ch.send("byte", [B_RANGE, 0, vec[i], vec[i + 1]], mult)
    if len(vec) % 2:
        ch.send("byte", [B_RANGE, 0, vec[-1], 0], mult)


def assert_lt_const(ch, name, vec, modulus, sel):
    """vec (range-checked bytes) < modulus as integers, on rows where sel = 1: one-hot flag of the most significant 3-byte
    group that differs, equality above it, and modulus_g - vec_g - 1 = a 24-bit number there."""
    gv, gm = groups3(vec), groups3(limbs_of(modulus, len(vec)))
    G = len(gv)
    f = ch.cols(name + "_f", G)
    d = ch.cols(name + "_d", 3)
    for x in f:
        ch.assert_zero(x * (x - 1))
    ch.assert_eq(esum(f), sel)
    for g in range(G):
        ch.assert_zero((sel - esum(f[g:])) * (gv[g] - gm[g]))
    ch.assert_zero(esum(f[g] * (gm[g] - gv[g] - 1) for g in range(G)) - (d[0] + 256 * d[1] + 65536 * d[2]))
    ch.send("byte", [B_RANGE, 0, d[0], d[1]], sel)
    ch.send("byte", [B_RANGE, 0, d[2], 0], sel)


def assert_differ(ch, name, va, vb, sel):
    """va != vb as integers (both canonical byte vectors) on rows where sel = 1: sum_g (va_g - vb_g) z_g = 1 has a
    solution exactly when some 3-byte group differs"""
    ga, gb = groups3(va), groups3(vb)
    z = ch.cols(name + "_z", len(ga))
    ch.assert_zero(esum((ga[g] - gb[g]) * z[g] for g in range(len(ga))) - sel)


def mem_words(ch, name, ptr, old, new, shard, ts, mult):
    """len(old) / 4 consecutive words at the word-aligned pointer `ptr` (four bytes): each is consumed with the (shard, clk)
    of its previous access and produced with (shard, ts); old = new for a read"""
    n = len(old) // 4
    his = []
    for k in range(n):
        m_sh, m_ts, m_same, m_lo, m_hi = (ch.col(f"{name}_{t}[{k}]") for t in ("sh", "ts", "same", "lo", "hi"))
        addr = word(ptr) + 4 * k
        ch.receive("mem", [addr] + old[4 * k:4 * k + 4] + [m_sh, m_ts], mult)
        ch.send("mem", [addr] + new[4 * k:4 * k + 4] + [shard, ts], mult)
        ch.assert_zero(mult * (m_same * (m_same - 1)))
        ch.assert_zero(mult * (m_same * (shard - m_sh)))
        ch.assert_zero(mult * (m_same * (ts - m_ts - 1) + (1 - m_same) * (shard - m_sh - 1) - m_lo - 65536 * m_hi))
        ch.send("byte", [B_U16, 0, m_lo, 0], mult)
        his.append(m_hi)
    range_bytes(ch, his, mult)


def code_bytes(sel_codes):
    """the four bytes of the syscall code as affine expressions of the operation selectors"""
    return [esum(s * ((c >> (8 * i)) & 0xFF) for s, c in sel_codes) for i in range(4)]


def build_fp_op():
    """BLS12381_FP_ADD / _SUB / _MUL (a0 = x, a1 = y: 12 little-endian words each): x := x op y mod p.  Operands need not be
    reduced; the result is.  y is read at (shard, clk + 2), x read and replaced at (shard, clk + 3) (so x may be y)."""
    L, Pm = 48, BLS12381_P
    ch = Chip("fp_op")
    shard = ch.pub(PUB_SHARD)
    is_real, is_add, is_sub, is_mul, clk = ch.col("is_real"), ch.col("is_add"), ch.col("is_sub"), ch.col("is_mul"), ch.col("clk")
    xp, yp = ch.cols("xp", 4), ch.cols("yp", 4)
    x, y, r, q = ch.cols("x", L), ch.cols("y", L), ch.cols("r", L), ch.cols("q", L + 1)
    for f in (is_real, is_add, is_sub, is_mul):
        ch.assert_bool(f)
    ch.assert_eq(is_add + is_sub + is_mul, is_real)
    ch.assert_zero(is_real.next() * (1 - is_real), "trans")
    ch.receive("sys", code_bytes([(is_add, SYS_BLS12381_FP_ADD), (is_sub, SYS_BLS12381_FP_SUB), (is_mul, SYS_BLS12381_FP_MUL)]) + xp + yp + [clk, shard], is_real)
    ch.send("byte", [B_ADDR, 0, xp[0], xp[3]], is_real)
    ch.send("byte", [B_ADDR, 0, yp[0], yp[3]], is_real)
    mem_words(ch, "my", yp, y, y, shard, clk + 2, is_real)
    mem_words(ch, "mx", xp, x, r, shard, clk + 3, is_real)
    range_bytes(ch, r, is_real)
    range_bytes(ch, q, is_real)
    assert_lt_const(ch, "rlt", r, Pm, is_real)
    Pl = limbs_of(Pm, L)
    w = ch.assert_poly_zero("rel", [
        (1, is_mul, x, y), (1, is_add, x, None), (1, is_add, y, None),
        (1, is_sub, x, None), (-1, is_sub, y, None), (10, is_sub, Pl, None),       # x - y + 10 p >= 0 for any 384-bit y
        (-1, is_real, r, None)],
        q, Pl, is_real, [{is_real.id: 1, s.id: 1} for s in (is_add, is_sub, is_mul)])
    for t in w:
        ch.send("byte", [B_U16, 0, t, 0], is_real)
    ch.quotient_parts = 1
    ch.logup_parts = 8      # (groups of LogUp batches: the unit of the part-parallel K4 / K5 launches of short tables, stark.cuh)
    return ch


def build_fp2_op():
    """BLS12381_FP2_ADD / _SUB / _MUL (a0 = x, a1 = y: c0 || c1, 24 words each): x := x op y in Fp2 = Fp[u] / (u^2 + 1)."""
    L, Pm = 48, BLS12381_P
    ch = Chip("fp2_op")
    shard = ch.pub(PUB_SHARD)
    is_real, is_add, is_sub, is_mul, clk = ch.col("is_real"), ch.col("is_add"), ch.col("is_sub"), ch.col("is_mul"), ch.col("clk")
    xp, yp = ch.cols("xp", 4), ch.cols("yp", 4)
    x0, x1, y0, y1 = ch.cols("x0", L), ch.cols("x1", L), ch.cols("y0", L), ch.cols("y1", L)
    r0, r1, q0, q1 = ch.cols("r0", L), ch.cols("r1", L), ch.cols("q0", L + 1), ch.cols("q1", L + 1)
    for f in (is_real, is_add, is_sub, is_mul):
        ch.assert_bool(f)
    ch.assert_eq(is_add + is_sub + is_mul, is_real)
    ch.assert_zero(is_real.next() * (1 - is_real), "trans")
    ch.receive("sys", code_bytes([(is_add, SYS_BLS12381_FP2_ADD), (is_sub, SYS_BLS12381_FP2_SUB), (is_mul, SYS_BLS12381_FP2_MUL)]) + xp + yp + [clk, shard], is_real)
    ch.send("byte", [B_ADDR, 0, xp[0], xp[3]], is_real)
    ch.send("byte", [B_ADDR, 0, yp[0], yp[3]], is_real)
    mem_words(ch, "my", yp, y0 + y1, y0 + y1, shard, clk + 2, is_real)
    mem_words(ch, "mx", xp, x0 + x1, r0 + r1, shard, clk + 3, is_real)
    for v in (r0, r1, q0, q1):
        range_bytes(ch, v, is_real)
    assert_lt_const(ch, "r0lt", r0, Pm, is_real)
    assert_lt_const(ch, "r1lt", r1, Pm, is_real)
    Pl = limbs_of(Pm, L)
    OFF = limbs_of(1 << 388, L + 1)          # x1 y1 < 2^768 < 2^388 p
    cases = [{is_real.id: 1, s.id: 1} for s in (is_add, is_sub, is_mul)]
    w0 = ch.assert_poly_zero("rel0", [
        (1, is_mul, x0, y0), (-1, is_mul, x1, y1), (1, is_mul, OFF, Pl),
        (1, is_add, x0, None), (1, is_add, y0, None), (1, is_sub, x0, None), (-1, is_sub, y0, None), (10, is_sub, Pl, None),
        (-1, is_real, r0, None)], q0, Pl, is_real, cases)
    w1 = ch.assert_poly_zero("rel1", [
        (1, is_mul, x0, y1), (1, is_mul, x1, y0),
        (1, is_add, x1, None), (1, is_add, y1, None), (1, is_sub, x1, None), (-1, is_sub, y1, None), (10, is_sub, Pl, None),
        (-1, is_real, r1, None)], q1, Pl, is_real, cases)
    for t in w0 + w1:
        ch.send("byte", [B_U16, 0, t, 0], is_real)
    ch.quotient_parts = 1
    ch.logup_parts = 16      # (groups of LogUp batches: the unit of the part-parallel K4 / K5 launches of short tables, stark.cuh)
    return ch


def build_weierstrass(name, L, Pm, code_add, code_dbl):
    """Affine point addition / doubling on y^2 = x^3 + b over F_p (a = 0: BLS12-381 G1 and secp256k1 alike).
    a0 = p (x || y, L / 4 little-endian words each), a1 = q for ADD, 0 for DOUBLE;  p := p + q  /  p := 2 p.
    Witness lambda, x3, y3 (range-checked bytes) with
        ADD     lambda (x2 - x1) = y2 - y1          DOUBLE   2 lambda y1 = 3 x1^2          (mod p)
                x3 = lambda^2 - x1 - x2                      x3 = lambda^2 - 2 x1
                y3 = lambda (x1 - x3) - y1                   y3 = lambda (x1 - x3) - y1
    All coordinates read are canonical (< p, checked), so are x3 and y3.  ADD requires x1 != x2 (witnessed: with equal
    abscissae lambda would be unconstrained); neither operation knows the point at infinity: the guest handles p = +-q and
    p = 0 itself, as with SP1's precompiles.  q is read at (shard, clk + 2), p read and replaced at (shard, clk + 3)."""
    ch = Chip(name)
    shard = ch.pub(PUB_SHARD)
    is_real, is_add, is_dbl, clk = ch.col("is_real"), ch.col("is_add"), ch.col("is_dbl"), ch.col("clk")
    pp, qp = ch.cols("pp", 4), ch.cols("qp", 4)
    x1, y1, x2, y2 = ch.cols("x1", L), ch.cols("y1", L), ch.cols("x2", L), ch.cols("y2", L)
    lam, x3, y3 = ch.cols("lam", L), ch.cols("x3", L), ch.cols("y3", L)
    q1, q2, q3 = ch.cols("q1", L + 1), ch.cols("q2", L + 1), ch.cols("q3", L + 1)
    for f in (is_real, is_add, is_dbl):
        ch.assert_bool(f)
    ch.assert_eq(is_add + is_dbl, is_real)
    ch.assert_zero(is_real.next() * (1 - is_real), "trans")
    for i in range(4):
        ch.assert_zero(is_dbl * qp[i])                   # DOUBLE: a1 = 0
    ch.receive("sys", code_bytes([(is_add, code_add), (is_dbl, code_dbl)]) + pp + qp + [clk, shard], is_real)
    ch.send("byte", [B_ADDR, 0, pp[0], pp[3]], is_real)
    ch.send("byte", [B_ADDR, 0, qp[0], qp[3]], is_add)
    mem_words(ch, "mq", qp, x2 + y2, x2 + y2, shard, clk + 2, is_add)
    mem_words(ch, "mp", pp, x1 + y1, x3 + y3, shard, clk + 3, is_real)
    for v in (lam, x3, y3, q1, q2, q3):
        range_bytes(ch, v, is_real)
    for nm_, v, s in (("x1lt", x1, is_real), ("y1lt", y1, is_real), ("x2lt", x2, is_add), ("y2lt", y2, is_add), ("x3lt", x3, is_real), ("y3lt", y3, is_real)):
        assert_lt_const(ch, nm_, v, Pm, s)
    assert_differ(ch, "xne", x1, x2, is_add)
    Pl = limbs_of(Pm, L)
    M = 1 << (8 * L)
    cases = [{is_real.id: 1, s.id: 1} for s in (is_add, is_dbl)]
    w1 = ch.assert_poly_zero("rel1", [
        (1, is_add, lam, x2), (-1, is_add, lam, x1), (-1, is_add, y2, None), (1, is_add, y1, None),
        (2, is_dbl, lam, y1), (-3, is_dbl, x1, x1),
        (1, is_real, limbs_of(4 * M, L + 1), Pl)], q1, Pl, is_real, cases)
    w2 = ch.assert_poly_zero("rel2", [
        (1, is_real, lam, lam), (-1, is_real, x1, None), (-1, is_add, x2, None), (-1, is_dbl, x1, None), (-1, is_real, x3, None),
        (4, is_real, Pl, None)], q2, Pl, is_real, cases)
    w3 = ch.assert_poly_zero("rel3", [
        (1, is_real, lam, x1), (-1, is_real, lam, x3), (-1, is_real, y1, None), (-1, is_real, y3, None),
        (1, is_real, limbs_of(2 * M, L + 1), Pl)], q3, Pl, is_real, cases)
    for t in w1 + w2 + w3:
        ch.send("byte", [B_U16, 0, t, 0], is_real)
    ch.quotient_parts = 1
    ch.logup_parts = 16      # (groups of LogUp batches: the unit of the part-parallel K4 / K5 launches of short tables, stark.cuh)
    return ch


def build_bls_g1():
    return build_weierstrass("bls_g1", 48, BLS12381_P, SYS_BLS12381_ADD, SYS_BLS12381_DOUBLE)


def build_secp_k1():
    return build_weierstrass("secp_k1", 32, SECP256K1_P, SYS_SECP256K1_ADD, SYS_SECP256K1_DOUBLE)


def assert_lt_vec(ch, name, vec, bound, sel):
    """vec < bound as integers (both range-checked byte vectors of one length; bound is made of columns), where sel = 1:
    the variable-bound form of assert_lt_const"""
    gv, gb = groups3(vec), groups3(bound)
    G = len(gv)
    f = ch.cols(name + "_f", G)
    d = ch.cols(name + "_d", 3)
    for x in f:
        ch.assert_zero(x * (x - 1))
    ch.assert_eq(esum(f), sel)
    for g in range(G):
        ch.assert_zero((sel - esum(f[g:])) * (gv[g] - gb[g]))
    ch.assert_zero(esum(f[g] * (gb[g] - gv[g] - 1) for g in range(G)) - (d[0] + 256 * d[1] + 65536 * d[2]))
    ch.send("byte", [B_RANGE, 0, d[0], d[1]], sel)
    ch.send("byte", [B_RANGE, 0, d[2], 0], sel)


def build_u256_mul():
    """UINT256_MUL (a0 = x: 8 little-endian words; a1 = y: 8 words followed by the 8 words of the modulus m):
    x := x * y mod m, with m = 0 standing for 2^256 (SP1's convention; what the patched bls12_381 crate's scalar field
    arithmetic calls).  x * y - r - q * M = 0 as one big-integer identity with the VARIABLE modulus M = m + is_zero * 2^256;
    the result is below M."""
    L = 32
    ch = Chip("u256_mul")
    shard = ch.pub(PUB_SHARD)
    is_real, m_zero, clk = ch.col("is_real"), ch.col("m_zero"), ch.col("clk")
    xp, yp = ch.cols("xp", 4), ch.cols("yp", 4)
    x, y, m, r, q = ch.cols("x", L), ch.cols("y", L), ch.cols("m", L), ch.cols("r", L), ch.cols("q", L + 1)
    mz = ch.cols("mz", (L + 2) // 3)
    ch.assert_bool(is_real)
    ch.assert_bool(m_zero)
    ch.assert_zero(m_zero * (1 - is_real))
    ch.assert_zero(is_real.next() * (1 - is_real), "trans")
    ch.receive("sys", code_bytes([(is_real, SYS_UINT256_MUL)]) + xp + yp + [clk, shard], is_real)
    ch.send("byte", [B_ADDR, 0, xp[0], xp[3]], is_real)
    ch.send("byte", [B_ADDR, 0, yp[0], yp[3]], is_real)
    mem_words(ch, "my", yp, y + m, y + m, shard, clk + 2, is_real)
    mem_words(ch, "mx", xp, x, r, shard, clk + 3, is_real)
    range_bytes(ch, r, is_real)
    range_bytes(ch, q, is_real)
    # m_zero = 1 exactly when m = 0:  m_zero * m_i = 0,  sum_g m_g z_g = is_real - m_zero (an inverse of one non-zero group)
    for i in range(L):
        ch.assert_zero(m_zero * m[i])
    gm = groups3(m)
    ch.assert_zero(esum(gm[g] * mz[g] for g in range(len(gm))) - (is_real - m_zero))
    assert_lt_vec(ch, "rlt", r, m, is_real - m_zero)
    M = m + [m_zero]                                      # the modulus as L + 1 limbs: m, or 2^256 when m = 0
    w = ch.assert_poly_zero("rel", [(1, is_real, x, y), (-1, is_real, r, None)], q, M, is_real, [{is_real.id: 1}])
    for t in w:
        ch.send("byte", [B_U16, 0, t, 0], is_real)
    ch.quotient_parts = 1
    ch.logup_parts = 8      # (groups of LogUp batches: the unit of the part-parallel K4 / K5 launches of short tables, stark.cuh)
    return ch


def build_mem_image():
    ch = Chip("mem_image")
    addr, v, real = ch.prep("addr"), ch.preps("v", 4), ch.prep("is_real")
    pad = ch.col("pad")
    ch.assert_zero(pad)
    # the image is consumed once per execution: by the mem_init table, which only the last shard carries
    ch.receive("image", [addr] + v, real * ch.pub(PUB_IS_LAST))
    return ch


def build_mem_init():
    ch = Chip("mem_init")
    ab, v, f, fts, fsh = ch.cols("ab", 4), ch.cols("v", 4), ch.cols("f", 4), ch.col("fts"), ch.col("fsh")
    d = ch.cols("d", 4)
    is_img, is_real = ch.col("is_img"), ch.col("is_real")
    addr = word(ab)
    addr_next = word([x.next() for x in ab])
    ch.assert_bool(is_real)
    ch.assert_bool(is_img)
    ch.assert_zero(is_img * (1 - is_real))
    ch.assert_zero(is_real.next() * (1 - is_real), "trans")
    # strictly increasing addresses OVER THE INTEGERS: the address itself is four range-checked bytes below 0x39000000
    # (guest memory below 0x38000000, then the registers), and so is the gap d = addr' - addr - 1,
    # hence addr + 1 + d < 0x72000000 < p cannot wrap (one initial tuple per address)
    ch.assert_zero(is_real.next() * (addr_next - addr - 1 - word([x.next() for x in d])), "trans")
    ch.send("byte", [B_RANGE, 0, ab[0], ab[1]], is_real)
    ch.send("byte", [B_RANGE, 0, ab[2], ab[3]], is_real)
    ch.send("byte", [B_LTU, 1, ab[3], ADDR_TOP_BYTE + 1], is_real)     # (+ 1: the registers at REG_BASE .. REG_BASE + 31)
    ch.send("byte", [B_RANGE, 0, d[0], d[1]], is_real)
    ch.send("byte", [B_RANGE, 0, d[2], d[3]], is_real)
    ch.send("byte", [B_LTU, 1, d[3], ADDR_TOP_BYTE + 1], is_real)
    # words outside the program image start with a prover-chosen value: that is how HINT_READ delivers the (private)
    # stdin buffers, as in SP1, whose executor files hinted words as "uninitialized memory" values [EXTERNAL]; .bss is
    # part of the image (zero words), stack and heap are written before they are read
    ch.send("byte", [B_RANGE, 0, v[0], v[1]], is_real - is_img)
    ch.send("byte", [B_RANGE, 0, v[2], v[3]], is_real - is_img)
    ch.send("image", [addr] + v, is_img)
    ch.send("mem", [addr] + v + [0, 0], is_real)                  # initial tuple: shard 0, clk 0
    ch.receive("mem", [addr] + f + [fsh, fts], is_real)          # final tuple: last (shard, clk) that touched it
    return ch


def build():
    return Machine("rv32", [build_program(), build_byte(), build_cpu(), build_mem_image(), build_mem_init(), build_shift(), build_muldiv(),
                            build_sha_extend(), build_sha_compress(), build_fp_op(), build_fp2_op(), build_bls_g1(), build_secp_k1(), build_u256_mul()], BUSES)
