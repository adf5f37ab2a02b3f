// Generic multi-chip STARK pieces shared by every machine (toy, rv32):
//   - ConstraintFolder<T,...>: folds a chip's AIR constraints and its LogUp
//     constraints with powers of alpha.  T = Fp on the gfx950 quotient domain
//     (K5), T = Fp4 in the host verifier at the out-of-domain point.
//   - PermRowCtx: evaluates a chip's interactions on one trace row and produces
//     the LogUp permutation row (K4).
//   - kernel templates instantiated per generated Air struct (gen/air_*.inc).
//
// LogUp layout of a chip with n interactions in nb = ceil(n/2) batches (pairs): one
// column per batch EXCEPT THE LAST, then the running-sum column phi, all in F_p^4,
// stored flattened as 4 nb base columns.  With d_j = alpha_p + bus_j + sum_k beta^(k+1) v_jk
// a batch {j0,j1} satisfies  v_b * d_j0 * d_j1 = s_j0 m_j0 d_j1 + s_j1 m_j1 d_j0,  where v_b is
// its column, and for the last batch the expression
//     v_last = phi[r+1] - phi[r] - sum_{b < nb-1} perm_b[r] + S/N      (r+1 cyclic: row N-1 -> row 0)
// with S = the chip's cumulative sum and N its height: phi is the exclusive prefix sum of
// (row total - S/N), so the differences around the cycle add up to zero exactly when the row totals
// add up to S.  No boundary constraint and no selector is needed (the constraint has degree 3 on every row),
// and the last batch costs no column.
// Constraint numbering (powers of alpha): the chip's own constraints 0..C-1, then batch b -> C+b.
//
// (stock SP1 keeps this in sp1-stark, absent.)
#pragma once
#include "bb.cuh"
#include "f64dot.cuh"
#include "kernels.h"

namespace dvt {

constexpr int WHEN_ALL = 0, WHEN_FIRST = 1, WHEN_LAST = 2, WHEN_TRANS = 3;

DVT_HD Fp4 to_ext(Fp a) { return Fp4::from_base(a); }
DVT_HD Fp4 to_ext(const Fp4 &a) { return a; }

template <class Air>
struct PermShape {
    static constexpr int NB = (Air::N_INTERACTIONS + 1) / 2;
    static constexpr int EXT_W = Air::N_INTERACTIONS ? NB : 0;  // ext columns: NB - 1 batches + phi
    static constexpr int PHI = NB - 1;                          // index of phi
    static constexpr int BASE_W = 4 * EXT_W;
    static constexpr int N_FOLDED = Air::N_CONSTRAINTS + (Air::N_INTERACTIONS ? NB : 0);
};

// Access must provide: T main(col,rot), T prep(col,rot), T pub(k),
//                      Fp4 perm(extcol,rot)   (rot 0 = local, 1 = next)
template <class Air, class TT, class Access>
struct ConstraintFolder {
    using T = TT;
    using Shape = PermShape<Air>;
    const Access &ax;
    const Fp4 *alpha_pows;  // [N_FOLDED]
    const Fp4 *beta_pows;   // beta^1 .. beta^MAX_ARITY at [0..]
    // T = Fp (the gfx950 quotient kernel) only: the same powers as centred doubles [..][4].  Products of a (wave-uniform)
    // power with a base-field value are then accumulated exactly on the FP64 pipe (f64dot.cuh) instead of as four
    // Montgomery products each; F_p^4 x F_p^4 products stay in Montgomery arithmetic.
    const double *alpha_d = nullptr, *beta_d = nullptr;
    DotAcc4 acc_d;
    Fp4 perm_alpha;
    T sel_first, sel_last, sel_trans;
    Fp4 cum_over_n;   // S / N
    Fp4 acc;
    // pending first half of a LogUp batch
    T pend_m;
    Fp4 pend_d;
    // T = Fp: the LogUp batch constraints (F_p^4 products of per-row values) are evaluated on canonical residues in
    // doubles (Fd4, f64dot.cuh) and accumulated, already multiplied by their alpha power, in acc_c
    Fd4 pend_dd, acc_c = {{0.0, 0.0, 0.0, 0.0}}, perm_alpha_c;
    double pend_md = 0.0;

    DVT_HD ConstraintFolder(const Access &a) : ax(a), acc(Fp4::zero()) {}
    DVT_HD static T K(uint32_t monty) {
        if constexpr (sizeof(T) == sizeof(Fp)) return Fp::raw(monty);
        else return Fp4::from_base(Fp::raw(monty));
    }
    DVT_HD static T KI(uint32_t canonical) {   // a constant computed at run time (looped interactions of the generated code)
        if constexpr (sizeof(T) == sizeof(Fp)) return Fp::from_canonical(canonical);
        else return Fp4::from_base(Fp::from_canonical(canonical));
    }
    DVT_HD T main(int c, int r) const { return ax.main(c, r); }
    DVT_HD T prep(int c, int r) const { return ax.prep(c, r); }
    DVT_HD T pub(int k) const { return ax.pub(k); }

    static constexpr bool BASE = sizeof(T) == sizeof(Fp);
    DVT_HD void fold(int idx, const Fp4 &v) { acc += alpha_pows[idx] * v; }
    DVT_HD void fold_t(int idx, const T &v) {
        if constexpr (BASE) {
            acc_d.add(alpha_d + 4 * idx, v);
            if ((idx & 31) == 31) acc_d.reduce();   // idx is a literal in the generated code: resolved at compile time
        } else {
            acc += alpha_pows[idx] * v;
        }
    }
    // d = alpha_p + bus + sum_k beta^(k+1) v_k
    DVT_HD Fp4 denominator(int bus, const T *vals, int n) const {
        Fp4 d = perm_alpha + Fp::from_canonical((uint32_t)bus);
        if constexpr (BASE) {
            DotAcc4 s;
            for (int k = 0; k < n; k++) {
                s.add(beta_d + 4 * k, vals[k]);
                if ((k & 31) == 31) s.reduce();
            }
            d += s.value();
        } else {
            for (int k = 0; k < n; k++) d += beta_pows[k] * vals[k];
        }
        return d;
    }

    // ---- polynomial identities (tools/airgen/dsl.py assert_poly_zero): the coefficient constraints first .. first + K - 1
    // of one big-integer identity are folded in closed form, alpha^first (C(alpha) + (alpha - 256) W(alpha)).
    // V(alpha) = sum_i alpha^i v_i of a limb vector:
    DVT_HD Fp4 poly(const T *v, int n) const {
        if constexpr (BASE) {
            DotAcc4 s;
            for (int k = 0; k < n; k++) {
                s.add(alpha_d + 4 * k, v[k]);
                if ((k & 31) == 31) s.reduce();
            }
            return s.value();
        } else {
            Fp4 a = Fp4::zero();
            for (int k = 0; k < n; k++) a += alpha_pows[k] * v[k];
            return a;
        }
    }
    DVT_HD Fp4 alpha_minus(uint32_t k) const { return alpha_pows[1] - Fp::from_canonical(k); }
    DVT_HD void fold_poly(int first, const Fp4 &v) { acc += alpha_pows[first] * v; }

    DVT_HD void constraint(int idx, int when, const T &v) {
        if (when == WHEN_ALL) fold_t(idx, v);
        else if (when == WHEN_FIRST) fold_t(idx, v * sel_first);
        else if (when == WHEN_LAST) fold_t(idx, v * sel_last);
        else fold_t(idx, v * sel_trans);
    }

    // (T = Fp) canonical d and signed multiplicity of an interaction, canonical permutation-trace value
    DVT_HD Fd4 denominator_c(int bus, const T *vals, int n) const {
        DotAcc4 s;
        for (int k = 0; k < n; k++) {
            if constexpr (BASE) s.add(beta_d + 4 * k, vals[k]);
            if ((k & 31) == 31) s.reduce();
        }
        Fd4 d;
        s.value_canonical(d.c);
        for (int k = 0; k < 4; k++) d.c[k] += perm_alpha_c.c[k];
        d.c[0] += (double)bus;
        return d;
    }
    // value of batch b: its column, or (last batch) the running-sum difference that stands for it
    DVT_HD Fp4 batch_value(int b) const {
        if (b < Shape::PHI) return ax.perm(b, 0);
        Fp4 v = ax.perm(Shape::PHI, 1) - ax.perm(Shape::PHI, 0) + cum_over_n;
        for (int k = 0; k < Shape::PHI; k++) v -= ax.perm(k, 0);
        return v;
    }
    DVT_HD Fd4 perm_c(int e) const {
        const Fp4 p = batch_value(e);
        Fd4 r;
        for (int k = 0; k < 4; k++) r.c[k] = centred_from_mont(p.c[k].v);
        return r;
    }
    DVT_HD void fold_c(int idx, const Fd4 &v) {   // acc_c += alpha^idx * v
        Fd4 a;
        for (int k = 0; k < 4; k++) a.c[k] = alpha_d[4 * idx + k];
        acc_c = acc_c + a * v;
    }
    DVT_HD void interaction(int j, int bus, int sign, int /*scope*/, const T &mult, const T *vals, int n) {
        if constexpr (BASE) {
            const Fd4 d = denominator_c(bus, vals, n);
            const double m0 = centred_from_mont(mult.v), m = sign > 0 ? m0 : -m0;
            if ((j & 1) == 0) {
                pend_md = m;
                pend_dd = d;
                if (j == Air::N_INTERACTIONS - 1) {  // odd tail: perm * d - m = 0
                    Fd4 v = perm_c(j >> 1) * d;
                    v.c[0] -= m;
                    fold_c(Air::N_CONSTRAINTS + (j >> 1), v);
                }
            } else {
                const Fd4 lhs = (perm_c(j >> 1) * pend_dd) * d;
                const Fd4 rhs = d * pend_md + pend_dd * m;
                fold_c(Air::N_CONSTRAINTS + (j >> 1), lhs - rhs);
            }
            return;
        }
        Fp4 d = denominator(bus, vals, n);
        T m = sign > 0 ? mult : -mult;
        if ((j & 1) == 0) {
            pend_m = m;
            pend_d = d;
            if (j == Air::N_INTERACTIONS - 1) {  // odd tail: perm * d - m = 0
                Fp4 p = batch_value(j >> 1);
                fold(Air::N_CONSTRAINTS + (j >> 1), p * d - to_ext(m));
            }
        } else {
            Fp4 p = batch_value(j >> 1);
            Fp4 lhs = p * pend_d * d;
            Fp4 rhs = d * pend_m + pend_d * m;
            fold(Air::N_CONSTRAINTS + (j >> 1), lhs - rhs);
        }
    }

    // PART < 0: everything (the host verifier).  0 <= PART < Air::N_PARTS: that group of the chip's own constraints;
    // N_PARTS <= PART < N_PARTS + N_LPARTS: that group of LogUp batches.
    // The folded value is the sum over the parts.
    template <int PART = -1>
    DVT_HD Fp4 run() {
        if constexpr (BASE)
            for (int k = 0; k < 4; k++) perm_alpha_c.c[k] = centred_from_mont(perm_alpha.c[k].v);
        if constexpr (PART < 0) {
            Air::constraints(*this);
            Air::interactions(*this);
        } else if constexpr (PART < Air::N_PARTS) {
            Air::template constraints_part<PART>(*this);
        } else {
            Air::template interactions_part<PART - Air::N_PARTS>(*this);
        }
        if constexpr (BASE) {
            acc += acc_d.value();
            Fp4 lc;   // acc_c holds canonical residues: back to Montgomery words
            for (int k = 0; k < 4; k++) lc.c[k] = Fp::raw(p2f::to_mont(acc_c.c[k]));
            acc += lc;
        }
        return acc;
    }
};

#if defined(__HIPCC__)
// ------------------------------------------------------------------ K4: permutation trace
struct PermArgs {
    const uint32_t *main;  // [MAIN_W][N]
    const uint32_t *prep;  // [PREP_W][N]
    const uint32_t *pub;   // device public values (Montgomery)
    uint32_t *perm;        // [4*EXT_W][N] out: the batch columns (the phi columns are filled from `totals` afterwards)
    uint32_t *totals;      // [4][N] out: row totals (all batches)
    const Fp4 *beta_pows;  // device
    const double *beta_d;  // the same powers as centred doubles [..][4] (f64dot.cuh)
    Fp4 perm_alpha;
    uint32_t log_n;
    // scratch [PARTS_MAX][4][N] or nullptr.  With it, a short table runs one (row, group of LogUp batches) pair per thread
    // (perm_rows_parts_kernel): a precompile chip has hundreds of interactions per row and, in most shards, too few rows to
    // fill the GPU with one thread each.
    uint32_t *partial = nullptr;
};
constexpr int PARTS_MAX = 32;              // groups per chip (constraint groups + LogUp groups) the scratch buffers are sized for
constexpr uint32_t PARTS_PARALLEL_LOG = 15;  // tables of at most 2^15 rows use the part-parallel launches

template <class Air>
struct PermRowCtx {
    using T = Fp;
    const PermArgs &a;
    size_t n, row;
    Fp4 batch, total;
    __device__ PermRowCtx(const PermArgs &args, size_t r) : a(args), n((size_t)1 << args.log_n), row(r), batch(Fp4::zero()), total(Fp4::zero()) {}
    __device__ static T K(uint32_t m) { return Fp::raw(m); }
    __device__ static T KI(uint32_t canonical) { return Fp::from_canonical(canonical); }
    __device__ T main(int c, int r) const { return Fp::raw(a.main[(size_t)c * n + ((row + r) & (n - 1))]); }
    __device__ T prep(int c, int r) const { return Fp::raw(a.prep[(size_t)c * n + ((row + r) & (n - 1))]); }
    __device__ T pub(int k) const { return Fp::raw(a.pub[k]); }
    __device__ void store_ext(int extcol, const Fp4 &v) {
#pragma unroll
        for (int k = 0; k < 4; k++) a.perm[((size_t)(4 * extcol + k)) * n + row] = v.c[k].v;
    }
    __device__ void interaction(int j, int bus, int sign, int /*scope*/, const T &mult, const T *vals, int nv) {
        Fp4 term = Fp4::zero();
        if (!mult.is_zero()) {   // most interactions of a row are switched off by their selector
            DotAcc4 s;
            for (int k = 0; k < nv; k++) {
                s.add(a.beta_d + 4 * k, vals[k]);
                if ((k & 31) == 31) s.reduce();
            }
            Fp4 d = a.perm_alpha + Fp::from_canonical((uint32_t)bus) + s.value();
            term = inv(d) * (sign > 0 ? mult : -mult);
        }
        if ((j & 1) == 0) batch = term; else batch += term;
        if ((j & 1) || j == Air::N_INTERACTIONS - 1) {
            if ((j >> 1) < PermShape<Air>::PHI) store_ext(j >> 1, batch);   // the last batch has no column
            total += batch;
        }
    }
};

template <class Air>
__global__ void __launch_bounds__(256) perm_rows_kernel(PermArgs a) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= ((size_t)1 << a.log_n)) return;
    PermRowCtx<Air> ctx(a, row);
    Air::interactions(ctx);
#pragma unroll
    for (int k = 0; k < 4; k++) a.totals[(size_t)k * ctx.n + row] = ctx.total.c[k].v;   // scanned into phi afterwards
}

template <class Air, int LP, class Ctx>
__device__ __forceinline__ void interactions_of_part(Ctx &ctx, int part) {
    if (part == LP) Air::template interactions_part<LP>(ctx);
    else if constexpr (LP + 1 < Air::N_LPARTS) interactions_of_part<Air, LP + 1>(ctx, part);
}
// grid (row blocks, Air::N_LPARTS): the batch columns of group blockIdx.y, and the group's share of the row total
template <class Air>
__global__ void __launch_bounds__(256) perm_rows_parts_kernel(PermArgs a) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= ((size_t)1 << a.log_n)) return;
    PermRowCtx<Air> ctx(a, row);
    interactions_of_part<Air, 0>(ctx, (int)blockIdx.y);
#pragma unroll
    for (int k = 0; k < 4; k++) a.partial[((size_t)blockIdx.y * 4 + k) * ctx.n + row] = ctx.total.c[k].v;
}
// out[w] = sum over the parts of partial[part][w]  (field words)
template <int UNUSED>
__global__ void __launch_bounds__(256) sum_parts_kernel(const uint32_t *partial, uint32_t nparts, size_t words, uint32_t *out) {
    size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    Fp acc = Fp::raw(partial[w]);
    for (uint32_t p = 1; p < nparts; p++) acc = acc + Fp::raw(partial[(size_t)p * words + w]);
    out[w] = acc.v;
}

// ------------------------------------------------------------------ K5: quotient
struct QuotientArgs {
    const uint32_t *main_lde;  // [MAIN_W][2N]
    const uint32_t *prep_lde;  // [PREP_W][2N]
    const uint32_t *perm_lde;  // [4*EXT_W][2N]
    const uint32_t *pub;
    uint32_t *out;             // two chunks: [2][4][N]  (even rows, odd rows)
    const Fp4 *alpha_pows;
    const Fp4 *beta_pows;
    const double *alpha_d, *beta_d;   // centred doubles [..][4]
    Fp4 perm_alpha, cum_over_n;   // cum_over_n = the chip's cumulative sum / N
    Fp zinv_even, zinv_odd;    // 1 / Z_H(x) on even / odd LDE rows
    Fp z_even, z_odd;          // Z_H(x) itself
    Fp w_inv;                  // omega_N^-1
    uint32_t log_n;
    NttTables tabs;
    // per-height selector table [3][2N] (sel_first, sel_last, sel_trans on the LDE rows), built once per prover by
    // selector_table_kernel: every launch of every chip of that height reads three coalesced words per row instead of
    // computing x = g w^i (two table gathers, two products) and inverting (x - 1)(x - w^-1) (about sixty products) —
    // the cpu chip's quotient is nine launches.  nullptr: computed in the kernel (the first proof of a height).
    const uint32_t *sel = nullptr;
    // scratch [PARTS_MAX][2][4][N] or nullptr: with it a short table runs all its groups in ONE launch (quotient_parts_kernel,
    // blockIdx.y = group, partial quotients summed afterwards) instead of one latency-bound launch per group
    uint32_t *partial = nullptr;
};

// the three selectors of one LDE row (what quotient_kernel used to compute inline)
__device__ __forceinline__ void selectors_of_row(const QuotientArgs &a, size_t i, Fp *sel_first, Fp *sel_last, Fp *sel_trans) {
    const uint32_t log_m = a.log_n + 1;
    // x = g * w_{2N}^i
    Fp x = Fp::raw(a.tabs.sh_lo[1]) * (Fp::raw(a.tabs.tw_hi[((uint32_t)i << (24 - log_m)) >> 12]) *
                                        Fp::raw(a.tabs.tw_lo[((uint32_t)i << (24 - log_m)) & 4095]));
    const Fp zh = (i & 1) ? a.z_odd : a.z_even;
    Fp d1 = x - Fp::one(), d2 = x - a.w_inv;
    Fp pinv = inv(d1 * d2);
    *sel_first = zh * (pinv * d2);
    *sel_last = zh * (pinv * d1);
    *sel_trans = d2;
}
template <int UNUSED>   // (a template only so that the header may define it in several translation units)
__global__ void __launch_bounds__(256) selector_table_kernel(QuotientArgs a, uint32_t *out) {
    const size_t m = (size_t)2 << a.log_n;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Fp f, l, t;
    selectors_of_row(a, i, &f, &l, &t);
    out[i] = f.v; out[m + i] = l.v; out[2 * m + i] = t.v;
}

struct QuotAccess {
    const QuotientArgs &a;
    size_t m, i, inext;
    __device__ Fp main(int c, int r) const { return Fp::raw(a.main_lde[(size_t)c * m + (r ? inext : i)]); }
    __device__ Fp prep(int c, int r) const { return Fp::raw(a.prep_lde[(size_t)c * m + (r ? inext : i)]); }
    __device__ Fp pub(int k) const { return Fp::raw(a.pub[k]); }
    __device__ Fp4 perm(int e, int r) const {
        Fp4 v;
        size_t at = r ? inext : i;
#pragma unroll
        for (int k = 0; k < 4; k++) v.c[k] = Fp::raw(a.perm_lde[(size_t)(4 * e + k) * m + at]);
        return v;
    }
};

// One launch per part (ConstraintFolder::run<PART>): part 0 writes the quotient values, the later parts add to them.
template <class Air, int PART>
__global__ void __launch_bounds__(256) quotient_kernel(QuotientArgs a) {
    const size_t m = (size_t)2 << a.log_n;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    QuotAccess ax{a, m, i, (i + 2) & (m - 1)};
    ConstraintFolder<Air, Fp, QuotAccess> f(ax);
    f.alpha_pows = a.alpha_pows;
    f.beta_pows = a.beta_pows;
    f.alpha_d = a.alpha_d;
    f.beta_d = a.beta_d;
    f.perm_alpha = a.perm_alpha;
    f.cum_over_n = a.cum_over_n;
    const bool odd = i & 1;
    const Fp zhinv = odd ? a.zinv_odd : a.zinv_even;
    if (a.sel) {
        f.sel_first = Fp::raw(a.sel[i]); f.sel_last = Fp::raw(a.sel[m + i]); f.sel_trans = Fp::raw(a.sel[2 * m + i]);
    } else {
        selectors_of_row(a, i, &f.sel_first, &f.sel_last, &f.sel_trans);
    }
    Fp4 q = f.template run<PART>() * zhinv;
    const size_t n = m >> 1;
    uint32_t *o = a.out + (odd ? 4 * n : 0) + (i >> 1);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if constexpr (PART == 0) o[(size_t)k * n] = q.c[k].v;
        else o[(size_t)k * n] = (Fp::raw(o[(size_t)k * n]) + q.c[k]).v;
    }
}
template <class Air, int PART, class F>
__device__ __forceinline__ Fp4 run_part(F &f, int part) {
    constexpr int LAST = Air::N_PARTS + (Air::N_INTERACTIONS > 0 ? Air::N_LPARTS : 0) - 1;
    if (part == PART) return f.template run<PART>();
    if constexpr (PART < LAST) return run_part<Air, PART + 1>(f, part);
    return Fp4::zero();
}
template <class Air>
__global__ void __launch_bounds__(256) quotient_parts_kernel(QuotientArgs a) {
    const size_t m = (size_t)2 << a.log_n;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    QuotAccess ax{a, m, i, (i + 2) & (m - 1)};
    ConstraintFolder<Air, Fp, QuotAccess> f(ax);
    f.alpha_pows = a.alpha_pows;
    f.beta_pows = a.beta_pows;
    f.alpha_d = a.alpha_d;
    f.beta_d = a.beta_d;
    f.perm_alpha = a.perm_alpha;
    f.cum_over_n = a.cum_over_n;
    const bool odd = i & 1;
    const Fp zhinv = odd ? a.zinv_odd : a.zinv_even;
    if (a.sel) {
        f.sel_first = Fp::raw(a.sel[i]); f.sel_last = Fp::raw(a.sel[m + i]); f.sel_trans = Fp::raw(a.sel[2 * m + i]);
    } else {
        selectors_of_row(a, i, &f.sel_first, &f.sel_last, &f.sel_trans);
    }
    Fp4 q = run_part<Air, 0>(f, (int)blockIdx.y) * zhinv;
    const size_t n = m >> 1;
    uint32_t *o = a.partial + (size_t)blockIdx.y * 8 * n + (odd ? 4 * n : 0) + (i >> 1);
#pragma unroll
    for (int k = 0; k < 4; k++) o[(size_t)k * n] = q.c[k].v;
}
#endif  // __HIPCC__

}  // namespace dvt
