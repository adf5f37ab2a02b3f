/* ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see field.h).
 *
 * CPU restatement of the per-chip AIR evaluation.  The per-chip functions are
 * generated (oracle/gen/air_<machine>.c, by tools/airgen) from the same symbolic
 * description as the HIP kernels but through a different backend (plain C,
 * canonical field, `%` arithmetic); the generic code here walks trace rows,
 * applies the first/last/transition selectors and balances the LogUp multiset
 * exactly (a hash map of tuples), which is how the tests validate the traces the
 * product generates (K0) and the permutation/quotient kernels (K4, K5).
 * No reference file exists for these AIRs: the reference delegates them to the
 * absent sp1-core-machine crate. */
#ifndef DVT_AIR_ORACLE_H
#define DVT_AIR_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t bus, sign, scope, arity;
} orc_interaction_info;

typedef struct {
    const char *name;
    uint32_t main_w, prep_w, n_pub, n_constraints, n_interactions, max_arity;
    const uint8_t *when; /* per constraint: 0 all rows, 1 first, 2 last, 3 transition (all but last) */
    const orc_interaction_info *inter;
    void (*constraints)(const uint32_t *main_l, const uint32_t *main_n, const uint32_t *prep_l, const uint32_t *prep_n,
                        const uint32_t *pub, uint32_t *out);
    void (*interactions)(const uint32_t *main_l, const uint32_t *main_n, const uint32_t *prep_l, const uint32_t *prep_n,
                         const uint32_t *pub, uint32_t *mult, uint32_t *vals);
} orc_chip_air;

extern const orc_chip_air orc_machine_toy[];
extern const unsigned orc_machine_toy_nchips;
extern const orc_chip_air orc_machine_rv32[];
extern const unsigned orc_machine_rv32_nchips;
const orc_chip_air *orc_machine(const char *name, unsigned *nchips);

/* Evaluate every constraint on every row (traces column-major, canonical).
 * Returns the number of violated (constraint,row) pairs; the first one found is
 * reported through *bad_constraint / *bad_row. */
size_t orc_check_constraints(const orc_chip_air *chip, const uint32_t *main, const uint32_t *prep, uint32_t log_n,
                             const uint32_t *pub, int *bad_constraint, uint32_t *bad_row);

/* Exact LogUp multiset. */
typedef struct orc_multiset orc_multiset;
orc_multiset *orc_multiset_new(void);
void orc_multiset_free(orc_multiset *ms);
void orc_multiset_add_chip(orc_multiset *ms, const orc_chip_air *chip, const uint32_t *main, const uint32_t *prep,
                           uint32_t log_n, const uint32_t *pub);
/* one explicit tuple (e.g. the verifier-side receives of the public-values bus) */
void orc_multiset_add_tuple(orc_multiset *ms, uint32_t bus, const uint32_t *vals, uint32_t arity, int sign, uint32_t mult);
/* number of tuples whose signed multiplicities do not cancel (mod p); the first
 * one is copied to out[0] = bus, out[1] = arity, out[2] = net multiplicity, out[3..] = values */
size_t orc_multiset_unbalanced(const orc_multiset *ms, uint32_t *out, size_t out_cap);

/* K4 restatement: the LogUp permutation trace of one chip (flattened, column-major
 * [4*(ceil(n_inter/2)+1)][2^log_n], canonical) and its cumulative sum. alpha/beta: 4 words each. */
void orc_perm_trace(const orc_chip_air *chip, const uint32_t *main, const uint32_t *prep, uint32_t log_n,
                    const uint32_t *pub, const uint32_t alpha[4], const uint32_t beta[4], uint32_t *perm_out,
                    uint32_t cumsum_out[4]);

/* K5..K8 restatements (stark_oracle.c).  All arrays canonical; extension elements are 4 words. */
void orc_quotient(const orc_chip_air *chip, const uint32_t *main_lde, const uint32_t *prep_lde, const uint32_t *perm_lde,
                  uint32_t log_n, const uint32_t *pub, const uint32_t perm_alpha[4], const uint32_t beta[4],
                  const uint32_t alpha[4], const uint32_t cumsum[4], uint32_t *out);
void orc_eval_columns(const uint32_t *cols, uint32_t width, uint32_t log_n, uint32_t coset_shift, const uint32_t z[4], uint32_t *out);
void orc_reduced_opening(const uint32_t *const *cols, uint32_t n_two, uint32_t n_all, uint32_t log_m, const uint32_t alpha[4],
                         const uint32_t *open_local, const uint32_t *open_next, const uint32_t zeta[4], const uint32_t zeta_next[4],
                         uint32_t *out);
void orc_fri_fold(const uint32_t *v, uint32_t log_m, const uint32_t beta[4], const uint32_t *ro, uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif
