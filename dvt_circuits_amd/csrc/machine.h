// Runtime description of a machine (a fixed list of chips) as the prover and
// verifier see it: widths, LogUp shape, and per-chip entry points instantiated
// from the generated Air structs (gen/air_<machine>.inc).
#pragma once
#include "stark.cuh"

namespace dvt {

// opened values of one chip at zeta (local) and zeta*omega (next); perm_* hold the
// openings of the flattened base columns (4 per extension column)
struct VerifierAccess {
    const Fp4 *main_l, *main_n, *prep_l, *prep_n, *perm_l, *perm_n;
    const Fp *pubv;
    Fp4 main(int c, int r) const { return r ? main_n[c] : main_l[c]; }
    Fp4 prep(int c, int r) const { return r ? prep_n[c] : prep_l[c]; }
    Fp4 pub(int k) const { return Fp4::from_base(pubv[k]); }
    Fp4 perm(int e, int r) const {
        const Fp4 *p = (r ? perm_n : perm_l) + 4 * e;
        Fp4 acc = p[0];
        for (int k = 1; k < 4; k++) {
            Fp4 basis = Fp4::zero();
            basis.c[k] = Fp::one();
            acc += basis * p[k];
        }
        return acc;
    }
};

struct VerifierPoint {
    const Fp4 *alpha_pows, *beta_pows;
    Fp4 perm_alpha, cum_over_n, sel_first, sel_last, sel_trans;
};

struct ChipDesc {
    const char *name;
    int main_w, prep_w, n_pub, n_constraints, n_interactions, max_arity;
    int perm_ext_w;  // extension columns of the permutation trace (0 if no interactions)
    int n_folded;    // number of alpha powers consumed
    hipError_t (*launch_perm)(hipStream_t, const PermArgs &);
    hipError_t (*launch_quotient)(hipStream_t, const QuotientArgs &);
    Fp4 (*verify_eval)(const VerifierAccess &, const VerifierPoint &);
};

struct MachineDesc {
    const char *name;
    int n_chips;
    const ChipDesc *chips;
};

constexpr int RV32_FIRST_WIDE_CHIP = 9;   // fp_op: the first of the field / curve precompile chips (gen/rv32_cols.h RV32_CHIP_FP_OP)
const MachineDesc *machine_toy();
const MachineDesc *machine_rv32();
const MachineDesc *machine_by_name(const char *name);

#if defined(__HIPCC__)
template <class Air>
hipError_t launch_perm_t(hipStream_t st, const PermArgs &a) {
    if (Air::N_INTERACTIONS == 0) return hipSuccess;
    size_t n = (size_t)1 << a.log_n;
    if constexpr (Air::N_LPARTS > 1 && Air::N_LPARTS <= PARTS_MAX) {
        if (a.partial) {
            perm_rows_parts_kernel<Air><<<dim3((unsigned)((n + 255) / 256), Air::N_LPARTS), 256, 0, st>>>(a);
            sum_parts_kernel<0><<<(unsigned)((4 * n + 255) / 256), 256, 0, st>>>(a.partial, Air::N_LPARTS, 4 * n, a.totals);
            return hipGetLastError();
        }
    }
    perm_rows_kernel<Air><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(a);
    return hipGetLastError();
}
template <class Air, int PART>
void launch_quotient_parts(hipStream_t st, const QuotientArgs &a, unsigned blocks) {
    // parts 0 .. N_PARTS - 1: the chip's own constraints; then N_LPARTS groups of its LogUp constraints (none without interactions)
    constexpr int LAST = Air::N_PARTS + (Air::N_INTERACTIONS > 0 ? Air::N_LPARTS : 0) - 1;
    quotient_kernel<Air, PART><<<blocks, 256, 0, st>>>(a);
    if constexpr (PART < LAST) launch_quotient_parts<Air, PART + 1>(st, a, blocks);
}
template <class Air>
hipError_t launch_quotient_t(hipStream_t st, const QuotientArgs &a) {
    size_t m = (size_t)2 << a.log_n;
    constexpr int NP = Air::N_PARTS + (Air::N_INTERACTIONS > 0 ? Air::N_LPARTS : 0);
    if constexpr (NP > 2 && NP <= PARTS_MAX && Air::MAIN_W >= 128) {   // (the wide chips: the others' groups are few and short)
        if (a.partial && a.log_n <= PARTS_PARALLEL_LOG) {
            quotient_parts_kernel<Air><<<dim3((unsigned)((m + 255) / 256), NP), 256, 0, st>>>(a);
            sum_parts_kernel<0><<<(unsigned)((4 * m + 255) / 256), 256, 0, st>>>(a.partial, NP, 4 * m, a.out);
            return hipGetLastError();
        }
    }
    launch_quotient_parts<Air, 0>(st, a, (unsigned)((m + 255) / 256));
    return hipGetLastError();
}
template <class Air>
Fp4 verify_eval_t(const VerifierAccess &ax, const VerifierPoint &pt) {
    ConstraintFolder<Air, Fp4, VerifierAccess> f(ax);
    f.alpha_pows = pt.alpha_pows;
    f.beta_pows = pt.beta_pows;
    f.perm_alpha = pt.perm_alpha;
    f.cum_over_n = pt.cum_over_n;
    f.sel_first = pt.sel_first;
    f.sel_last = pt.sel_last;
    f.sel_trans = pt.sel_trans;
    return f.run();
}
template <class Air>
constexpr ChipDesc make_chip_desc() {
    return ChipDesc{Air::NAME, Air::MAIN_W, Air::PREP_W, Air::N_PUB, Air::N_CONSTRAINTS, Air::N_INTERACTIONS,
                    Air::MAX_ARITY, PermShape<Air>::EXT_W, PermShape<Air>::N_FOLDED,
                    &launch_perm_t<Air>, &launch_quotient_t<Air>, &verify_eval_t<Air>};
}
#endif

}  // namespace dvt
