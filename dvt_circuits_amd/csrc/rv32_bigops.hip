// Field / curve precompiles of the rv32 machine (SP1's BLS12381_FP_*, BLS12381_FP2_*,
// BLS12381_ADD / _DOUBLE, SECP256K1_ADD / _DOUBLE syscalls, which the reference's guests reach through the patched
// bls12_381 / secp256k1 crates, reference crates/dkg/Cargo.toml:24-25): what a call computes (the guest machine's
// semantics) and the rows of the four chips that prove the calls (tools/airgen/rv32.py: build_fp_op, build_fp2_op,
// build_weierstrass).  What a call computes is host code (the executor); the rows are built on the GPU (one thread per
// call, k0_bigop_cells / _rels / _lookups kernels) and, for the CPU-only debug / test entry points, by the same code on the host.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "bigfield.h"
#include "rv32.h"

// (only the interactions of the five precompile chips are instantiated here, and those are loops)
#include "gen/air_rv32.inc"
#include "gen/rv32_rels.h"

namespace dvt {
namespace rv32 {

#if !defined(__HIP_DEVICE_COMPILE__)

// ------------------------------------------------------------------ the fields
static const uint64_t BLS_P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const uint64_t SECP_P[4] = {0xfffffffefffffc2full, 0xffffffffffffffffull, 0xffffffffffffffffull, 0xffffffffffffffffull};
static const MontField<6> &bls() { static const MontField<6> f(BLS_P); return f; }
static const MontField<4> &secp() { static const MontField<4> f(SECP_P); return f; }

template <int N> static void load(uint64_t *o, const uint32_t *w) { for (int i = 0; i < N; i++) o[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32); }
template <int N> static void store(uint32_t *w, const uint64_t *a) { for (int i = 0; i < N; i++) { w[2 * i] = (uint32_t)a[i]; w[2 * i + 1] = (uint32_t)(a[i] >> 32); } }

bool bigop_info(uint32_t code, BigOpInfo *out) {
    switch (code) {
    case SYS_BLS12381_FP_ADD: case SYS_BLS12381_FP_SUB: case SYS_BLS12381_FP_MUL: *out = {RV32_CHIP_FP_OP, 12, 12}; return true;
    case SYS_BLS12381_FP2_ADD: case SYS_BLS12381_FP2_SUB: case SYS_BLS12381_FP2_MUL: *out = {RV32_CHIP_FP2_OP, 24, 24}; return true;
    case SYS_BLS12381_ADD: *out = {RV32_CHIP_BLS_G1, 24, 24}; return true;
    case SYS_BLS12381_DOUBLE: *out = {RV32_CHIP_BLS_G1, 24, 0}; return true;
    case SYS_SECP256K1_ADD: *out = {RV32_CHIP_SECP_K1, 16, 16}; return true;
    case SYS_SECP256K1_DOUBLE: *out = {RV32_CHIP_SECP_K1, 16, 0}; return true;
    case SYS_UINT256_MUL: *out = {RV32_CHIP_U256_MUL, 8, 16}; return true;
    default: return false;
    }
}

// x op y mod p for arbitrary N-limb x, y (Montgomery conversion reduces them)
template <int N>
static void field_op(const MontField<N> &F, int op, const uint32_t *x, const uint32_t *y, uint32_t *r) {
    uint64_t a[N], b[N], c[N];
    load<N>(a, x); load<N>(b, y);
    F.to_mont(a, a); F.to_mont(b, b);
    if (op == 0) F.add(c, a, b);
    else if (op == 1) F.sub(c, a, b);
    else F.mul(c, a, b);
    F.from_mont(c, c);
    store<N>(r, c);
}

// affine add (q != nullptr) / double on y^2 = x^3 + b (a = 0); coordinates canonical; no point at infinity
template <int N>
static const char *curve_op(const MontField<N> &F, const uint32_t *p, const uint32_t *q, uint32_t *r, uint32_t *lam_out) {
    uint64_t x1[N], y1[N], x2[N], y2[N], num[N], den[N], lam[N], x3[N], y3[N], t[N];
    load<N>(x1, p); load<N>(y1, p + 2 * N);
    if (!F.is_canonical(x1) || !F.is_canonical(y1)) return "curve precompile: coordinate of p not reduced";
    if (q) {
        load<N>(x2, q); load<N>(y2, q + 2 * N);
        if (!F.is_canonical(x2) || !F.is_canonical(y2)) return "curve precompile: coordinate of q not reduced";
        if (F.cmp(x1, x2) == 0) return "ADD precompile with equal abscissae (p = q or p = -q: the guest must handle those)";
    } else if (F.is_zero(y1)) return "DOUBLE precompile of a point with y = 0";
    F.to_mont(x1, x1); F.to_mont(y1, y1);
    if (q) {
        F.to_mont(x2, x2); F.to_mont(y2, y2);
        F.sub(num, y2, y1); F.sub(den, x2, x1);
    } else {
        F.mul(t, x1, x1); F.add(num, t, t); F.add(num, num, t);   // 3 x1^2
        F.add(den, y1, y1);
        memcpy(x2, x1, sizeof x1);
    }
    F.inv(den, den);
    F.mul(lam, num, den);
    F.mul(x3, lam, lam); F.sub(x3, x3, x1); F.sub(x3, x3, x2);
    F.sub(t, x1, x3); F.mul(y3, lam, t); F.sub(y3, y3, y1);
    F.from_mont(lam, lam); F.from_mont(x3, x3); F.from_mont(y3, y3);
    store<N>(r, x3); store<N>(r + 2 * N, y3); store<N>(lam_out, lam);
    return nullptr;
}

// x * y mod m for 256-bit numbers (m = 0: mod 2^256): schoolbook product, then one long division (polyrel.h poly_divmnu)
static void u256_mulmod(const uint32_t *x, const uint32_t *y, const uint32_t *m, uint32_t *r) {
    uint32_t prod[16] = {0};
    for (int i = 0; i < 8; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 8; j++) {
            const uint64_t t = (uint64_t)x[i] * y[j] + prod[i + j] + carry;
            prod[i + j] = (uint32_t)t;
            carry = t >> 32;
        }
        prod[i + 8] = (uint32_t)carry;
    }
    int n = 8;
    while (n > 0 && m[n - 1] == 0) n--;
    if (n == 0) { memcpy(r, prod, 32); return; }
    uint32_t q[16], rem[8] = {0};
    poly_divmnu(q, rem, prod, m, 16, n);
    memcpy(r, rem, 32);
}

const char *bigop_compute(uint32_t code, const uint32_t *a, const uint32_t *b, uint32_t *r, uint32_t *lam) {
    switch (code) {
    case SYS_UINT256_MUL: u256_mulmod(a, b, b + 8, r); return nullptr;
    case SYS_BLS12381_FP_ADD: case SYS_BLS12381_FP_SUB: case SYS_BLS12381_FP_MUL:
        field_op<6>(bls(), (int)(code - SYS_BLS12381_FP_ADD), a, b, r);
        return nullptr;
    case SYS_BLS12381_FP2_ADD: case SYS_BLS12381_FP2_SUB:
        field_op<6>(bls(), (int)(code - SYS_BLS12381_FP2_ADD), a, b, r);
        field_op<6>(bls(), (int)(code - SYS_BLS12381_FP2_ADD), a + 12, b + 12, r + 12);
        return nullptr;
    case SYS_BLS12381_FP2_MUL: {   // (a0 + a1 u)(b0 + b1 u), u^2 = -1
        uint32_t t0[12], t1[12];
        field_op<6>(bls(), 2, a, b, t0); field_op<6>(bls(), 2, a + 12, b + 12, t1);
        field_op<6>(bls(), 1, t0, t1, r);
        field_op<6>(bls(), 2, a, b + 12, t0); field_op<6>(bls(), 2, a + 12, b, t1);
        field_op<6>(bls(), 0, t0, t1, r + 12);
        return nullptr;
    }
    case SYS_BLS12381_ADD: return curve_op<6>(bls(), a, b, r, lam);
    case SYS_BLS12381_DOUBLE: return curve_op<6>(bls(), a, nullptr, r, lam);
    case SYS_SECP256K1_ADD: return curve_op<4>(secp(), a, b, r, lam);
    case SYS_SECP256K1_DOUBLE: return curve_op<4>(secp(), a, nullptr, r, lam);
    default: return "unknown syscall";
    }
}

#endif  // host pass

// ------------------------------------------------------------------ rows (host and device)
namespace {
// A row under construction.  Host: a contiguous scratch row (the chips are up to 1245 columns wide: writing the cells
// straight into the column-major trace would be one cache miss per cell), scattered into the trace when complete.
// Device: the cells of row `row` of the column-major trace itself (lane = row: coalesced).
struct RowRef {
    uint32_t *m;
    DVT_HD uint32_t get(int col) const { return m[col]; }
    DVT_HD void put(int col, uint32_t v) { m[col] = v; }
};
struct DevRow {
    uint32_t *m;
    size_t n, row;
    DVT_HD uint32_t get(int col) const { return m[(size_t)col * n + row]; }
    DVT_HD void put(int col, uint32_t v) { m[(size_t)col * n + row] = v; }
};
template <class Row>
DVT_HD void put_bytes(Row &R, int col0, const uint32_t *w, int nwords) {
    for (int k = 0; k < nwords; k++)
        for (int i = 0; i < 4; i++) R.put(col0 + 4 * k + i, (w[k] >> (8 * i)) & 0xffu);
}
// canonical residues with plain 64-bit arithmetic: what the walker below evaluates the interaction values in (the cells
// are canonical while a row is built; no Montgomery conversion per cell)
struct Cn {
    uint32_t v;
};
DVT_HD Cn operator+(Cn a, Cn b) { uint32_t s = a.v + b.v; return Cn{s >= P ? s - P : s}; }
DVT_HD Cn operator-(Cn a, Cn b) { return Cn{a.v >= b.v ? a.v - b.v : a.v + P - b.v}; }
DVT_HD Cn operator-(Cn a) { return Cn{a.v ? P - a.v : 0}; }
DVT_HD Cn operator*(Cn a, Cn b) { return Cn{(uint32_t)((uint64_t)a.v * b.v % P)}; }
// walks the generated interactions of a chip on one row and counts its byte-table lookups (plain integer counts; on the
// device the table is the shard's, shared with K0 of the cpu rows: atomics)
constexpr uint32_t LOOKUP_SLOTS = 4096;   // direct-mapped (key -> count) cache of one workgroup of the lookups kernel: 32 KiB of LDS
template <class Row>
struct LookupCtx {
    using T = Cn;
    const Row &r;
    const uint32_t *pubs;
    uint32_t *byte_mult;
    uint32_t *lds;     // device: keys [LOOKUP_SLOTS] then counts [LOOKUP_SLOTS] of the workgroup's cache, or nullptr
    DVT_HD static T K(uint32_t monty) { return Cn{Fp::raw(monty).canonical()}; }
    DVT_HD static T KI(uint32_t canonical) { return Cn{canonical}; }
    DVT_HD T main(int c, int rot) const { (void)rot; return Cn{r.get(c)}; }   // (the precompile chips' interactions read the local row only)
    DVT_HD T prep(int, int) const { return Cn{0}; }
    DVT_HD T pub(int k) const { return Cn{pubs[k]}; }
    DVT_HD void interaction(int, int bus, int sign, int, const T &mult, const T *vals, int) {
        if (bus != 2 || sign < 0) return;   // bus 2 = byte (tools/airgen/rv32.py BUSES)
        if (!mult.v) return;
        const uint32_t op = vals[0].v, b = vals[2].v, c = vals[3].v;
        const uint32_t key = (op - 1) * 65536u + (op == B_U16 ? b : (b << 8) | c);
#if defined(__HIP_DEVICE_COMPILE__)
        // Hot table rows (zero bytes, the small carries) would serialise plain global atomics at the memory side, as in K0 of the
        // cpu chip (rv32_trace.hip): hits of the workgroup's cache are LDS atomics, conflicts fall through to a global atomic,
        // the cache is flushed once per workgroup.
        if (lds) {
            const uint32_t slot = (key * 2654435761u) >> 20;  // 12 bits
            const uint32_t old = atomicCAS(&lds[slot], 0xffffffffu, key);
            if (old == 0xffffffffu || old == key) atomicAdd(&lds[LOOKUP_SLOTS + slot], mult.v);
            else atomicAdd(byte_mult + key, mult.v);
        } else {
            atomicAdd(byte_mult + key, mult.v);
        }
#else
        byte_mult[key] += mult.v;
#endif
    }
};

// the (shard, clk) a word carried before the access at (shard, ts): same-shard flag and the two limbs of the gap
template <class Row>
DVT_HD void mem_meta(Row &R, int col_sh0, int k, uint32_t psh, uint32_t pts, uint32_t shard, uint32_t ts) {
    const uint32_t d = psh == shard ? ts - pts - 1 : shard - psh - 1;
    const int c = col_sh0 + 5 * k;   // sh, ts, same, lo, hi of word k are consecutive columns
    R.put(c, psh); R.put(c + 1, pts); R.put(c + 2, psh == shard); R.put(c + 3, d & 0xffff); R.put(c + 4, d >> 16);
}
// v < modulus witnessed on 3-byte groups: one-hot flag of the most significant differing group, modulus_g - v_g - 1 there
// (Mod: uint8_t mod(int i))
template <class Row, class Mod>
DVT_HD bool fill_lt(Row &R, int col_f0, int col_d0, int col_v0, int L, const Mod &mod) {
    const int G = (L + 2) / 3;
    for (int g = G - 1; g >= 0; g--) {
        uint32_t vg = 0, mg = 0;
        for (int t = 0; t < 3 && 3 * g + t < L; t++) { vg |= R.get(col_v0 + 3 * g + t) << (8 * t); mg |= (uint32_t)mod(3 * g + t) << (8 * t); }
        if (vg == mg) continue;
        if (vg > mg) return false;
        const uint32_t d = mg - vg - 1;
        R.put(col_f0 + g, 1);
        R.put(col_d0, d & 0xff); R.put(col_d0 + 1, (d >> 8) & 0xff); R.put(col_d0 + 2, d >> 16);
        return true;
    }
    return false;   // equal to the modulus
}
struct ConstMod {
    const uint8_t *m;
    DVT_HD uint8_t operator()(int i) const { return m[i]; }
};
template <class Row>
struct RowMod {   // a modulus held in the row itself (UINT256_MUL)
    const Row &R;
    int col0;
    DVT_HD uint8_t operator()(int i) const { return (uint8_t)R.get(col0 + i); }
};
template <class Row>
DVT_HD bool fill_differ(Row &R, int col_z0, int col_a0, int col_b0, int L) {
    const int G = (L + 2) / 3;
    for (int g = 0; g < G; g++) {
        uint32_t ag = 0, bg = 0;
        for (int t = 0; t < 3 && 3 * g + t < L; t++) { ag |= R.get(col_a0 + 3 * g + t) << (8 * t); bg |= R.get(col_b0 + 3 * g + t) << (8 * t); }
        if (ag == bg) continue;
        R.put(col_z0 + g, inv(Fp::from_canonical(ag) - Fp::from_canonical(bg)).canonical());
        return true;
    }
    return false;
}

// reasons a row cannot be built (the executor traps on every one of them first: seeing one here is an internal error)
enum RowError { ROW_OK = 0, ROW_NO_WITNESS, ROW_FP_NOT_REDUCED, ROW_FP2_NOT_REDUCED, ROW_U256_NOT_BELOW, ROW_CURVE_BAD, ROW_N_ERRORS };
#if !defined(__HIP_DEVICE_COMPILE__)
const char *const ROW_ERROR_TEXT[ROW_N_ERRORS] = {
    "", "precompile row: an identity has no witness", "fp_op: result not reduced", "fp2_op: result not reduced", "u256_mul: result not below the modulus",
    "curve precompile row: a coordinate is not reduced or the abscissae are equal"};
#endif

// STAGE_ALL: the whole row by one thread (host).  On the device the row is built in three launches: STAGE_CELLS (one thread
// per row: everything but the identities' witnesses), the identities (one wave per row, solve_poly_rel_wave below), then
// STAGE_LOOKUPS (one thread per row and group of LogUp batches: the byte-table lookups of the finished row).
enum RowStage { STAGE_ALL = 0, STAGE_CELLS = 1, STAGE_LOOKUPS = 2 };
// the interactions of group `part` of the chip's LogUp batches (the generator's split, tools/airgen/emit.py split_lparts)
template <class Air, int LP, class Ctx>
DVT_HD void lookups_of_part(Ctx &ctx, int part) {
    if (part == LP) Air::template interactions_part<LP>(ctx);
    else if constexpr (LP + 1 < Air::N_LPARTS) lookups_of_part<Air, LP + 1>(ctx, part);
}
// part < 0: every interaction of the row; otherwise one group (the device walks the groups of a row in parallel)
template <class Air, int STAGE, class Row>
DVT_HD RowError finish_row(Row &R, const PolyRelDesc *rels, int nrels, const uint32_t *pubs, uint32_t *byte_mult, int part, uint32_t *lds) {
    if constexpr (STAGE == STAGE_ALL)
        for (int i = 0; i < nrels; i++)
            if (!solve_poly_rel(rels[i], R)) return ROW_NO_WITNESS;
    if constexpr (STAGE != STAGE_CELLS) {
        LookupCtx<Row> ctx{R, pubs, byte_mult, lds};
        if (part < 0) Air::interactions(ctx);
        else lookups_of_part<Air, 0>(ctx, part);
    }
    return ROW_OK;
}

struct WCols {   // column ids of a short-Weierstrass chip
    int is_real, is_add, is_dbl, clk, pp, qp, x1, y1, x2, y2, lam, x3, y3, mq_sh, mp_sh;
    int x1lt_f, x1lt_d, y1lt_f, y1lt_d, x2lt_f, x2lt_d, y2lt_f, y2lt_d, x3lt_f, x3lt_d, y3lt_f, y3lt_d, xne_z;
};
#define DVT_WCOLS(P) WCols{P##is_real, P##is_add, P##is_dbl, P##clk, P##pp_0, P##qp_0, P##x1_0, P##y1_0, P##x2_0, P##y2_0, P##lam_0, P##x3_0, P##y3_0, \
                           P##mq_sh_0, P##mp_sh_0, P##x1lt_f_0, P##x1lt_d_0, P##y1lt_f_0, P##y1lt_d_0, P##x2lt_f_0, P##x2lt_d_0, P##y2lt_f_0, P##y2lt_d_0, \
                           P##x3lt_f_0, P##x3lt_d_0, P##y3lt_f_0, P##y3lt_d_0, P##xne_z_0}

// every cell of the row of call `e` in chip CHIP (the row starts all-zero), its byte-table lookups counted
template <int CHIP, int STAGE, class Row>
DVT_HD RowError fill_bigop_row(const BigOpEvent &e, uint32_t shard, Row &R, uint32_t *byte_mult, int part = -1, uint32_t *lds = nullptr) {
    const uint32_t pubs[N_PUBLIC] = {0, 0, 0, shard, 0};   // (the chips read PUB_SHARD only)
    const uint32_t ptrs[2] = {e.a_ptr, e.b_ptr};
    if constexpr (CHIP == RV32_CHIP_FP_OP) {
        if constexpr (STAGE != STAGE_LOOKUPS) {
        const int op = (int)(e.code - SYS_BLS12381_FP_ADD);
        R.put(RV32_FP_OP_is_real, 1); R.put(op == 0 ? RV32_FP_OP_is_add : op == 1 ? RV32_FP_OP_is_sub : RV32_FP_OP_is_mul, 1);
        R.put(RV32_FP_OP_clk, e.clk);
        put_bytes(R, RV32_FP_OP_xp_0, &ptrs[0], 1); put_bytes(R, RV32_FP_OP_yp_0, &ptrs[1], 1);
        put_bytes(R, RV32_FP_OP_x_0, e.a, 12); put_bytes(R, RV32_FP_OP_y_0, e.b, 12); put_bytes(R, RV32_FP_OP_r_0, e.r, 12);
        for (int k = 0; k < 12; k++) {
            mem_meta(R, RV32_FP_OP_my_sh_0, k, e.b_sh[k], e.b_ts[k], shard, e.clk + 2);
            mem_meta(R, RV32_FP_OP_mx_sh_0, k, e.a_sh[k], e.a_ts[k], shard, e.clk + 3);
        }
        if (!fill_lt(R, RV32_FP_OP_rlt_f_0, RV32_FP_OP_rlt_d_0, RV32_FP_OP_r_0, 48, ConstMod{rels_rv32::fp_op_0_mod})) return ROW_FP_NOT_REDUCED;
        }
        return finish_row<air_rv32::FpOp, STAGE>(R, rels_rv32::fp_op, rels_rv32::fp_op_n, pubs, byte_mult, part, lds);
    } else if constexpr (CHIP == RV32_CHIP_U256_MUL) {
        if constexpr (STAGE != STAGE_LOOKUPS) {
        R.put(RV32_U256_MUL_is_real, 1); R.put(RV32_U256_MUL_clk, e.clk);
        put_bytes(R, RV32_U256_MUL_xp_0, &ptrs[0], 1); put_bytes(R, RV32_U256_MUL_yp_0, &ptrs[1], 1);
        put_bytes(R, RV32_U256_MUL_x_0, e.a, 8); put_bytes(R, RV32_U256_MUL_y_0, e.b, 8); put_bytes(R, RV32_U256_MUL_m_0, e.b + 8, 8); put_bytes(R, RV32_U256_MUL_r_0, e.r, 8);
        for (int k = 0; k < 16; k++) mem_meta(R, RV32_U256_MUL_my_sh_0, k, e.b_sh[k], e.b_ts[k], shard, e.clk + 2);
        for (int k = 0; k < 8; k++) mem_meta(R, RV32_U256_MUL_mx_sh_0, k, e.a_sh[k], e.a_ts[k], shard, e.clk + 3);
        bool mzero = true;
        for (int i = 0; i < 8; i++) mzero = mzero && e.b[8 + i] == 0;
        R.put(RV32_U256_MUL_m_zero, mzero);
        if (!mzero) {
            // an inverse of one non-zero 3-byte group of m, and r < m
            for (int g = 0; g < 11; g++) {
                uint32_t mg = 0;
                for (int t = 0; t < 3 && 3 * g + t < 32; t++) mg |= R.get(RV32_U256_MUL_m_0 + 3 * g + t) << (8 * t);
                if (mg) { R.put(RV32_U256_MUL_mz_0 + g, inv(Fp::from_canonical(mg)).canonical()); break; }
            }
            if (!fill_lt(R, RV32_U256_MUL_rlt_f_0, RV32_U256_MUL_rlt_d_0, RV32_U256_MUL_r_0, 32, RowMod<Row>{R, RV32_U256_MUL_m_0})) return ROW_U256_NOT_BELOW;
        }
        }
        return finish_row<air_rv32::U256Mul, STAGE>(R, rels_rv32::u256_mul, rels_rv32::u256_mul_n, pubs, byte_mult, part, lds);
    } else if constexpr (CHIP == RV32_CHIP_FP2_OP) {
        if constexpr (STAGE != STAGE_LOOKUPS) {
        const int op = (int)(e.code - SYS_BLS12381_FP2_ADD);
        R.put(RV32_FP2_OP_is_real, 1); R.put(op == 0 ? RV32_FP2_OP_is_add : op == 1 ? RV32_FP2_OP_is_sub : RV32_FP2_OP_is_mul, 1);
        R.put(RV32_FP2_OP_clk, e.clk);
        put_bytes(R, RV32_FP2_OP_xp_0, &ptrs[0], 1); put_bytes(R, RV32_FP2_OP_yp_0, &ptrs[1], 1);
        put_bytes(R, RV32_FP2_OP_x0_0, e.a, 12); put_bytes(R, RV32_FP2_OP_x1_0, e.a + 12, 12);
        put_bytes(R, RV32_FP2_OP_y0_0, e.b, 12); put_bytes(R, RV32_FP2_OP_y1_0, e.b + 12, 12);
        put_bytes(R, RV32_FP2_OP_r0_0, e.r, 12); put_bytes(R, RV32_FP2_OP_r1_0, e.r + 12, 12);
        for (int k = 0; k < 24; k++) {
            mem_meta(R, RV32_FP2_OP_my_sh_0, k, e.b_sh[k], e.b_ts[k], shard, e.clk + 2);
            mem_meta(R, RV32_FP2_OP_mx_sh_0, k, e.a_sh[k], e.a_ts[k], shard, e.clk + 3);
        }
        if (!fill_lt(R, RV32_FP2_OP_r0lt_f_0, RV32_FP2_OP_r0lt_d_0, RV32_FP2_OP_r0_0, 48, ConstMod{rels_rv32::fp2_op_0_mod}) ||
            !fill_lt(R, RV32_FP2_OP_r1lt_f_0, RV32_FP2_OP_r1lt_d_0, RV32_FP2_OP_r1_0, 48, ConstMod{rels_rv32::fp2_op_0_mod})) return ROW_FP2_NOT_REDUCED;
        }
        return finish_row<air_rv32::Fp2Op, STAGE>(R, rels_rv32::fp2_op, rels_rv32::fp2_op_n, pubs, byte_mult, part, lds);
    } else {
        constexpr bool is_bls = CHIP == RV32_CHIP_BLS_G1;
        if constexpr (STAGE != STAGE_LOOKUPS) {
        const WCols C = is_bls ? DVT_WCOLS(RV32_BLS_G1_) : DVT_WCOLS(RV32_SECP_K1_);
        constexpr int L = is_bls ? 48 : 32, W = L / 4;     // bytes / words per coordinate
        const ConstMod mod{is_bls ? rels_rv32::bls_g1_0_mod : rels_rv32::secp_k1_0_mod};
        const bool add = e.code == SYS_BLS12381_ADD || e.code == SYS_SECP256K1_ADD;
        R.put(C.is_real, 1); R.put(add ? C.is_add : C.is_dbl, 1); R.put(C.clk, e.clk);
        put_bytes(R, C.pp, &ptrs[0], 1); put_bytes(R, C.qp, &ptrs[1], 1);
        put_bytes(R, C.x1, e.a, W); put_bytes(R, C.y1, e.a + W, W);
        if (add) { put_bytes(R, C.x2, e.b, W); put_bytes(R, C.y2, e.b + W, W); }
        put_bytes(R, C.lam, e.lam, W); put_bytes(R, C.x3, e.r, W); put_bytes(R, C.y3, e.r + W, W);
        for (int k = 0; k < 2 * W; k++) {
            if (add) mem_meta(R, C.mq_sh, k, e.b_sh[k], e.b_ts[k], shard, e.clk + 2);
            mem_meta(R, C.mp_sh, k, e.a_sh[k], e.a_ts[k], shard, e.clk + 3);
        }
        bool ok = fill_lt(R, C.x1lt_f, C.x1lt_d, C.x1, L, mod) && fill_lt(R, C.y1lt_f, C.y1lt_d, C.y1, L, mod) &&
                  fill_lt(R, C.x3lt_f, C.x3lt_d, C.x3, L, mod) && fill_lt(R, C.y3lt_f, C.y3lt_d, C.y3, L, mod);
        if (add) ok = ok && fill_lt(R, C.x2lt_f, C.x2lt_d, C.x2, L, mod) && fill_lt(R, C.y2lt_f, C.y2lt_d, C.y2, L, mod) && fill_differ(R, C.xne_z, C.x1, C.x2, L);
        if (!ok) return ROW_CURVE_BAD;
        }
        if constexpr (is_bls) return finish_row<air_rv32::BlsG1, STAGE>(R, rels_rv32::bls_g1, rels_rv32::bls_g1_n, pubs, byte_mult, part, lds);
        else return finish_row<air_rv32::SecpK1, STAGE>(R, rels_rv32::secp_k1, rels_rv32::secp_k1_n, pubs, byte_mult, part, lds);
    }
}
}  // namespace

// ------------------------------------------------------------------ device: K0 of the precompile chips
#if defined(__HIPCC__)
// The identities of one row by one wave: coefficient k of V, of low * m^-1 and of q * m belongs to lane k mod 64 (operand
// bytes in LDS); the carry sweeps (base-256 digits of V, of the quotient, the carries W_k) are short serial loops of lane 0
// over LDS.  Same results as solve_poly_rel (polyrel.h), which one thread per row runs out of scratch memory — 50 k
// multiply-adds per G1 row through ~5 KB of private arrays took 25 ms for 13 k rows; this takes well under one.
struct RelShared {
    int32_t a[POLY_MAX_K], b[POLY_MAX_K], s[POLY_MAX_K], w[POLY_MAX_K];
    long long c[POLY_MAX_K];
    uint8_t low[POLY_MAX_K + 8], q[POLY_MAX_K], mod[POLY_MAX_K];
    int ok;
};
__device__ static void load_vec(const PolyVec &v, const DevRow &row, int32_t *o, int lane) {
    for (int i = lane; i < v.len; i += 64) o[i] = v.cols ? (int32_t)row.get(v.cols[i]) : (int32_t)v.cst[i];
}
__device__ static bool solve_poly_rel_wave(const PolyRelDesc &d, DevRow &row, RelShared &S, int lane) {
    if (d.K > POLY_MAX_K) return false;
    for (int k = lane; k < POLY_MAX_K; k += 64) S.c[k] = 0;
    if (lane == 0) S.ok = 1;
    __syncthreads();
    for (int t = 0; t < d.n_terms; t++) {
        const PolyTerm &tm = d.terms[t];
        if (!row.get(tm.sel_col)) continue;     // (the same cell for every lane)
        load_vec(tm.a, row, S.a, lane);
        if (tm.b.len) load_vec(tm.b, row, S.b, lane);
        __syncthreads();
        const int la = tm.a.len, lb = tm.b.len;
        if (lb == 0) {
            for (int k = lane; k < la; k += 64) S.c[k] += (long long)tm.coef * S.a[k];
        } else {
            for (int k = lane; k < la + lb - 1; k += 64) {
                const int i0 = k - lb + 1 > 0 ? k - lb + 1 : 0, i1 = k < la - 1 ? k : la - 1;
                int32_t acc = 0;
                for (int i = i0; i <= i1; i++) acc += S.a[i] * S.b[k - i];
                S.c[k] += (long long)tm.coef * acc;
            }
        }
        __syncthreads();
    }
    for (int j = lane; j < d.nmod; j += 64) S.mod[j] = d.modv ? (uint8_t)row.get(d.modv[j]) : d.mod[j];
    __syncthreads();
    if (!d.modv) {
        if (lane == 0) {
            long long t = 0;
            for (int k = 0; k < d.nq; k++) {
                if (k < d.K) t += S.c[k];
                const uint8_t lo = (uint8_t)(t & 255);
                S.low[k] = lo;
                t = (t - lo) / 256;
            }
        }
        __syncthreads();
        for (int k = lane; k < d.nq; k += 64) {
            int32_t acc = 0;
            for (int i = 0; i <= k; i++) acc += (int32_t)S.low[i] * (int32_t)d.pinv[k - i];
            S.s[k] = acc;
        }
        __syncthreads();
        if (lane == 0) {
            uint64_t carry = 0;
            for (int k = 0; k < d.nq; k++) {
                const uint64_t v = carry + (uint32_t)S.s[k];
                S.q[k] = (uint8_t)(v & 255);
                carry = v >> 8;
            }
        }
    } else if (lane == 0) {
        // the modulus comes from the row and may be even: all K digits of V, then one long division (lane 0; UINT256_MUL only)
        long long t = 0;
        for (int k = 0; k < d.K; k++) {
            t += S.c[k];
            const uint8_t lo = (uint8_t)(t & 255);
            S.low[k] = lo;
            t = (t - lo) / 256;
        }
        int nm = d.nmod;
        while (nm > 0 && S.mod[nm - 1] == 0) nm--;
        bool ok = t == 0 && nm > 0;
        uint32_t u[POLY_MAX_K / 4 + 2] = {0}, v[POLY_MAX_K / 4 + 2] = {0}, qw[POLY_MAX_K / 4 + 2] = {0}, rw[POLY_MAX_K / 4 + 2] = {0};
        if (ok) {
            for (int k = 0; k < d.K; k++) u[k >> 2] |= (uint32_t)S.low[k] << (8 * (k & 3));
            for (int k = 0; k < nm; k++) v[k >> 2] |= (uint32_t)S.mod[k] << (8 * (k & 3));
            const int mu = (d.K + 3) / 4, nv = (nm + 3) / 4;
            if (mu >= nv) poly_divmnu(qw, rw, u, v, mu, nv);
            else for (int k = 0; k < mu; k++) rw[k] = u[k];
            for (int k = 0; k < nv; k++) ok = ok && rw[k] == 0;
            for (int k = 0; k < d.nq; k++) S.q[k] = 0;
            for (int k = 0; k < 4 * (mu >= nv ? mu - nv + 1 : 0); k++) {
                const uint8_t digit = (uint8_t)(qw[k >> 2] >> (8 * (k & 3)));
                if (!digit) continue;
                if (k >= d.nq) { ok = false; break; }
                S.q[k] = digit;
            }
        }
        if (!ok) S.ok = 0;
    }
    __syncthreads();
    if (!S.ok) return false;
    for (int k = lane; k < d.nq; k += 64) row.put(d.q[k], S.q[k]);
    for (int k = lane; k < d.nq + d.nmod - 1 && k < POLY_MAX_K; k += 64) {
        const int i0 = k - d.nmod + 1 > 0 ? k - d.nmod + 1 : 0, i1 = k < d.nq - 1 ? k : d.nq - 1;
        int32_t acc = 0;
        for (int i = i0; i <= i1; i++) acc += (int32_t)S.q[i] * (int32_t)S.mod[k - i];
        S.c[k] -= acc;
    }
    __syncthreads();
    if (lane == 0) {
        long long W = 0;
        bool ok = true;
        for (int k = 0; k + 1 < d.K; k++) {
            const long long v = S.c[k] + W;
            if (v & 255) { ok = false; break; }
            W = v / 256;
            const long long wv = W + d.w_off[k];
            if (wv < 0 || wv >= (d.wb ? 131072 : 65536)) { ok = false; break; }
            S.w[k] = (int32_t)wv;
        }
        if (ok && S.c[d.K - 1] + W != 0) ok = false;
        if (!ok) S.ok = 0;
    }
    __syncthreads();
    if (!S.ok) return false;
    for (int k = lane; k + 1 < d.K; k += 64) {
        row.put(d.w[k], (uint32_t)(S.w[k] & 0xffff));
        if (d.wb) row.put(d.wb[k], (uint32_t)(S.w[k] >> 16));
    }
    __syncthreads();
    return true;
}

template <int CHIP> struct ChipRels;
template <> struct ChipRels<RV32_CHIP_FP_OP> { static constexpr int LPARTS = air_rv32::FpOp::N_LPARTS; static __device__ const PolyRelDesc *get(int *n) { *n = rels_rv32::fp_op_n; return rels_rv32::fp_op; } };
template <> struct ChipRels<RV32_CHIP_FP2_OP> { static constexpr int LPARTS = air_rv32::Fp2Op::N_LPARTS; static __device__ const PolyRelDesc *get(int *n) { *n = rels_rv32::fp2_op_n; return rels_rv32::fp2_op; } };
template <> struct ChipRels<RV32_CHIP_BLS_G1> { static constexpr int LPARTS = air_rv32::BlsG1::N_LPARTS; static __device__ const PolyRelDesc *get(int *n) { *n = rels_rv32::bls_g1_n; return rels_rv32::bls_g1; } };
template <> struct ChipRels<RV32_CHIP_SECP_K1> { static constexpr int LPARTS = air_rv32::SecpK1::N_LPARTS; static __device__ const PolyRelDesc *get(int *n) { *n = rels_rv32::secp_k1_n; return rels_rv32::secp_k1; } };
template <> struct ChipRels<RV32_CHIP_U256_MUL> { static constexpr int LPARTS = air_rv32::U256Mul::N_LPARTS; static __device__ const PolyRelDesc *get(int *n) { *n = rels_rv32::u256_mul_n; return rels_rv32::u256_mul; } };

// one thread per call: everything but the identities' witnesses
template <int CHIP>
__global__ void __launch_bounds__(64) k0_bigop_cells_kernel(const BigOpEvent *ev, uint32_t n_ev, uint32_t shard, uint32_t *main, uint32_t log_n, uint32_t *err) {
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_ev) return;
    DevRow R{main, (size_t)1 << log_n, row};
    const RowError e = fill_bigop_row<CHIP, STAGE_CELLS>(ev[row], shard, R, nullptr);
    if (e != ROW_OK) atomicMax(err, (uint32_t)e);
}
// after the identities: one thread per (call, group of LogUp batches) walks the generated interactions of the finished row
// and counts its byte-table lookups (blockIdx.y = the group: a G1 row has 615 interactions)
template <int CHIP>
__global__ void __launch_bounds__(256) k0_bigop_lookups_kernel(const BigOpEvent *ev, uint32_t n_ev, uint32_t shard, uint32_t *main, uint32_t log_n, uint32_t *byte_mult) {
    __shared__ uint32_t cache[2 * LOOKUP_SLOTS];
    for (uint32_t s = threadIdx.x; s < LOOKUP_SLOTS; s += blockDim.x) { cache[s] = 0xffffffffu; cache[LOOKUP_SLOTS + s] = 0; }
    __syncthreads();
    const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < n_ev) {
        DevRow R{main, (size_t)1 << log_n, row};
        (void)fill_bigop_row<CHIP, STAGE_LOOKUPS>(ev[row], shard, R, byte_mult, (int)blockIdx.y, cache);
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < LOOKUP_SLOTS; s += blockDim.x)
        if (cache[s] != 0xffffffffu && cache[LOOKUP_SLOTS + s]) atomicAdd(byte_mult + cache[s], cache[LOOKUP_SLOTS + s]);
}
// one wave per call: q and the carries of every identity of the chip
template <int CHIP>
__global__ void __launch_bounds__(64) k0_bigop_rels_kernel(uint32_t n_ev, uint32_t *main, uint32_t log_n, uint32_t *err) {
    __shared__ RelShared S;
    const uint32_t row = blockIdx.x;
    if (row >= n_ev) return;
    DevRow R{main, (size_t)1 << log_n, row};
    int nrels = 0;
    const PolyRelDesc *rels = ChipRels<CHIP>::get(&nrels);
    for (int i = 0; i < nrels; i++)
        if (!solve_poly_rel_wave(rels[i], R, S, (int)threadIdx.x)) {
            if (threadIdx.x == 0) atomicMax(err, (uint32_t)ROW_NO_WITNESS);
            return;     // (the whole wave: S.ok is shared)
        }
}

template <int CHIP>
static void launch_chip_rows(hipStream_t st, const BigOpEvent *d_ev, uint32_t n_ev, uint32_t shard, uint32_t *d_main, uint32_t log_n, uint32_t *d_byte_mult, uint32_t *d_err) {
    k0_bigop_cells_kernel<CHIP><<<(n_ev + 63) / 64, 64, 0, st>>>(d_ev, n_ev, shard, d_main, log_n, d_err);
    k0_bigop_rels_kernel<CHIP><<<n_ev, 64, 0, st>>>(n_ev, d_main, log_n, d_err);
    k0_bigop_lookups_kernel<CHIP><<<dim3((n_ev + 255) / 256, ChipRels<CHIP>::LPARTS), 256, 0, st>>>(d_ev, n_ev, shard, d_main, log_n, d_byte_mult);
}
hipError_t launch_k0_bigop_rows(hipStream_t st, int chip, const BigOpEvent *d_ev, uint32_t n_ev, uint32_t shard, uint32_t *d_main, uint32_t log_n,
                                uint32_t *d_byte_mult, uint32_t *d_err) {
    if (!n_ev) return hipSuccess;
    if (((size_t)1 << log_n) < n_ev) return hipErrorInvalidValue;
    switch (chip) {
    case RV32_CHIP_FP_OP: launch_chip_rows<RV32_CHIP_FP_OP>(st, d_ev, n_ev, shard, d_main, log_n, d_byte_mult, d_err); break;
    case RV32_CHIP_FP2_OP: launch_chip_rows<RV32_CHIP_FP2_OP>(st, d_ev, n_ev, shard, d_main, log_n, d_byte_mult, d_err); break;
    case RV32_CHIP_BLS_G1: launch_chip_rows<RV32_CHIP_BLS_G1>(st, d_ev, n_ev, shard, d_main, log_n, d_byte_mult, d_err); break;
    case RV32_CHIP_SECP_K1: launch_chip_rows<RV32_CHIP_SECP_K1>(st, d_ev, n_ev, shard, d_main, log_n, d_byte_mult, d_err); break;
    case RV32_CHIP_U256_MUL: launch_chip_rows<RV32_CHIP_U256_MUL>(st, d_ev, n_ev, shard, d_main, log_n, d_byte_mult, d_err); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#endif

// ------------------------------------------------------------------ host: shapes, and the rows for the CPU-only entry points
#if !defined(__HIP_DEVICE_COMPILE__)
static uint32_t ceil_log2(size_t n) { uint32_t l = 0; while (((size_t)1 << l) < n) l++; return l; }
const char *bigop_row_error_text(uint32_t code) { return code < ROW_N_ERRORS ? ROW_ERROR_TEXT[code] : "precompile row: unknown error"; }

bool build_bigop_traces(const std::vector<BigOpEvent> &big, uint32_t shard, HostTraces *out, uint32_t *byte_mult, std::string *err, BigOpBatches *device_rows) {
    HostTraces &T = *out;
    const auto t_begin = std::chrono::steady_clock::now();
    struct Report {
        std::chrono::steady_clock::time_point t0; size_t n;
        ~Report() { if (n && getenv("DVT_TIME_BIGOPS")) fprintf(stderr, "[bigops] %zu rows in %.3f ms\n", n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
    } report{t_begin, big.size()};
    std::vector<const BigOpEvent *> by_chip[N_CHIPS];
    for (auto &e : big) {
        BigOpInfo bi;
        if (!bigop_info(e.code, &bi)) { if (err) *err = "precompile event with an unknown code"; return false; }
        by_chip[bi.chip].push_back(&e);
    }
    static const int widths[N_CHIPS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, RV32_FP_OP_MAIN_W, RV32_FP2_OP_MAIN_W, RV32_BLS_G1_MAIN_W, RV32_SECP_K1_MAIN_W, RV32_U256_MUL_MAIN_W};
    for (int chip : {RV32_CHIP_FP_OP, RV32_CHIP_FP2_OP, RV32_CHIP_BLS_G1, RV32_CHIP_SECP_K1, RV32_CHIP_U256_MUL}) {
        auto &evs = by_chip[chip];
        T.present[chip] = !evs.empty();
        T.log_n[chip] = 0;
        T.main[chip].clear();
        if (device_rows) device_rows->ev[chip].clear();
        if (evs.empty()) continue;
        const uint32_t lg = ceil_log2(evs.size());
        const size_t n = (size_t)1 << lg;
        T.log_n[chip] = lg;
        if (device_rows) {   // the product path: the GPU builds the rows (launch_k0_bigop_rows) from the calls themselves
            device_rows->ev[chip].reserve(evs.size());
            for (auto *e : evs) device_rows->ev[chip].push_back(*e);
            continue;
        }
        T.main[chip].assign((size_t)widths[chip] * n, 0);
        std::vector<uint32_t> scratch(widths[chip]);
        for (size_t row = 0; row < evs.size(); row++) {
            const BigOpEvent &e = *evs[row];
            std::fill(scratch.begin(), scratch.end(), 0u);
            RowRef R{scratch.data()};
            RowError re = ROW_OK;
            switch (chip) {
            case RV32_CHIP_FP_OP: re = fill_bigop_row<RV32_CHIP_FP_OP, STAGE_ALL>(e, shard, R, byte_mult); break;
            case RV32_CHIP_FP2_OP: re = fill_bigop_row<RV32_CHIP_FP2_OP, STAGE_ALL>(e, shard, R, byte_mult); break;
            case RV32_CHIP_BLS_G1: re = fill_bigop_row<RV32_CHIP_BLS_G1, STAGE_ALL>(e, shard, R, byte_mult); break;
            case RV32_CHIP_SECP_K1: re = fill_bigop_row<RV32_CHIP_SECP_K1, STAGE_ALL>(e, shard, R, byte_mult); break;
            default: re = fill_bigop_row<RV32_CHIP_U256_MUL, STAGE_ALL>(e, shard, R, byte_mult); break;
            }
            if (re != ROW_OK) { if (err) *err = ROW_ERROR_TEXT[re]; return false; }
            uint32_t *dst = T.main[chip].data();
            for (int c = 0; c < widths[chip]; c++)
                if (scratch[c]) dst[(size_t)c * n + row] = scratch[c];
        }
    }
    return true;
}
#endif  // host pass

}  // namespace rv32
}  // namespace dvt
