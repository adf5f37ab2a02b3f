/* ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see air_oracle.h). */
#include "air_oracle.h"
#include "field.h"
#include <stdlib.h>
#include <string.h>

const orc_chip_air *orc_machine(const char *name, unsigned *nchips) {
    if (!strcmp(name, "toy")) { *nchips = orc_machine_toy_nchips; return orc_machine_toy; }
    if (!strcmp(name, "rv32")) { *nchips = orc_machine_rv32_nchips; return orc_machine_rv32; }
    *nchips = 0;
    return NULL;
}

static void gather_row(const uint32_t *m, uint32_t w, size_t n, size_t row, uint32_t *out) {
    for (uint32_t c = 0; c < w; c++) out[c] = m[(size_t)c * n + row];
}

size_t orc_check_constraints(const orc_chip_air *chip, const uint32_t *main, const uint32_t *prep, uint32_t log_n,
                             const uint32_t *pub, int *bad_constraint, uint32_t *bad_row) {
    size_t n = (size_t)1 << log_n, bad = 0;
    uint32_t *ml = malloc(4 * (chip->main_w + 1)), *mn = malloc(4 * (chip->main_w + 1));
    uint32_t *pl = malloc(4 * (chip->prep_w + 1)), *pn = malloc(4 * (chip->prep_w + 1));
    uint32_t *out = malloc(4 * (chip->n_constraints + 1));
    for (size_t r = 0; r < n; r++) {
        size_t rn = (r + 1) & (n - 1);
        gather_row(main, chip->main_w, n, r, ml);
        gather_row(main, chip->main_w, n, rn, mn);
        gather_row(prep, chip->prep_w, n, r, pl);
        gather_row(prep, chip->prep_w, n, rn, pn);
        chip->constraints(ml, mn, pl, pn, pub, out);
        for (uint32_t k = 0; k < chip->n_constraints; k++) {
            int active = chip->when[k] == 0 || (chip->when[k] == 1 && r == 0) || (chip->when[k] == 2 && r == n - 1) ||
                         (chip->when[k] == 3 && r != n - 1);
            if (active && out[k] != 0) {
                if (!bad) { if (bad_constraint) *bad_constraint = (int)k; if (bad_row) *bad_row = (uint32_t)r; }
                bad++;
            }
        }
    }
    free(ml); free(mn); free(pl); free(pn); free(out);
    return bad;
}

/* ---------------------------------------------------------------- exact multiset */
#define MS_MAX_ARITY 48
typedef struct { uint32_t used, bus, arity; uint32_t vals[MS_MAX_ARITY]; uint64_t mult; } ms_entry;
struct orc_multiset { ms_entry *tab; size_t cap, count; };

orc_multiset *orc_multiset_new(void) {
    orc_multiset *ms = calloc(1, sizeof *ms);
    ms->cap = 1 << 16;
    ms->tab = calloc(ms->cap, sizeof(ms_entry));
    return ms;
}
void orc_multiset_free(orc_multiset *ms) { if (ms) { free(ms->tab); free(ms); } }

static uint64_t ms_hash(uint32_t bus, const uint32_t *v, uint32_t n) {
    uint64_t h = 1469598103934665603ull ^ bus;
    for (uint32_t i = 0; i < n; i++) { h ^= v[i]; h *= 1099511628211ull; h ^= h >> 29; }
    return h;
}
static void ms_insert(orc_multiset *ms, uint32_t bus, const uint32_t *v, uint32_t n, uint32_t signed_mult);
static void ms_grow(orc_multiset *ms) {
    ms_entry *old = ms->tab;
    size_t oc = ms->cap;
    ms->cap *= 2;
    ms->tab = calloc(ms->cap, sizeof(ms_entry));
    ms->count = 0;
    for (size_t i = 0; i < oc; i++)
        if (old[i].used) ms_insert(ms, old[i].bus, old[i].vals, old[i].arity, (uint32_t)old[i].mult);
    free(old);
}
static void ms_insert(orc_multiset *ms, uint32_t bus, const uint32_t *v, uint32_t n, uint32_t signed_mult) {
    if (ms->count * 2 > ms->cap) ms_grow(ms);
    size_t i = ms_hash(bus, v, n) & (ms->cap - 1);
    for (;;) {
        ms_entry *e = &ms->tab[i];
        if (!e->used) {
            e->used = 1; e->bus = bus; e->arity = n; memcpy(e->vals, v, 4 * n); e->mult = signed_mult;
            ms->count++;
            return;
        }
        if (e->bus == bus && e->arity == n && !memcmp(e->vals, v, 4 * n)) { e->mult = bb_add((bb_t)e->mult, signed_mult); return; }
        i = (i + 1) & (ms->cap - 1);
    }
}

void orc_multiset_add_chip(orc_multiset *ms, const orc_chip_air *chip, const uint32_t *main, const uint32_t *prep,
                           uint32_t log_n, const uint32_t *pub) {
    if (!chip->n_interactions) return;
    size_t n = (size_t)1 << log_n;
    uint32_t *ml = malloc(4 * (chip->main_w + 1)), *mn = malloc(4 * (chip->main_w + 1));
    uint32_t *pl = malloc(4 * (chip->prep_w + 1)), *pn = malloc(4 * (chip->prep_w + 1));
    uint32_t *mult = malloc(4 * chip->n_interactions), *vals = malloc(4 * (size_t)chip->n_interactions * chip->max_arity);
    for (size_t r = 0; r < n; r++) {
        size_t rn = (r + 1) & (n - 1);
        gather_row(main, chip->main_w, n, r, ml);
        gather_row(main, chip->main_w, n, rn, mn);
        gather_row(prep, chip->prep_w, n, r, pl);
        gather_row(prep, chip->prep_w, n, rn, pn);
        chip->interactions(ml, mn, pl, pn, pub, mult, vals);
        for (uint32_t j = 0; j < chip->n_interactions; j++) {
            if (!mult[j]) continue;
            uint32_t m = chip->inter[j].sign > 0 ? mult[j] : bb_neg(mult[j]);
            ms_insert(ms, (uint32_t)chip->inter[j].bus, vals + (size_t)j * chip->max_arity, (uint32_t)chip->inter[j].arity, m);
        }
    }
    free(ml); free(mn); free(pl); free(pn); free(mult); free(vals);
}

void orc_multiset_add_tuple(orc_multiset *ms, uint32_t bus, const uint32_t *vals, uint32_t arity, int sign, uint32_t mult) {
    ms_insert(ms, bus, vals, arity, sign > 0 ? mult : bb_neg(mult));
}

size_t orc_multiset_unbalanced(const orc_multiset *ms, uint32_t *out, size_t out_cap) {
    size_t bad = 0;
    for (size_t i = 0; i < ms->cap; i++) {
        const ms_entry *e = &ms->tab[i];
        if (!e->used || e->mult == 0) continue;
        if (!bad && out && out_cap >= 3 + e->arity) {
            out[0] = e->bus; out[1] = e->arity; out[2] = (uint32_t)e->mult;
            memcpy(out + 3, e->vals, 4 * e->arity);
        }
        bad++;
    }
    return bad;
}

/* ---------------------------------------------------------------- K4 restatement
 * Layout: nb = ceil(n_interactions / 2) batches; one F_p^4 column per batch except the last,
 * then phi.  phi[r] = sum_{r' < r} total[r'] - r * S / N with total = the sum of ALL batches of a row and S = the
 * chip's cumulative sum: the last batch is the difference phi[r+1] - phi[r] - (the other batches) + S / N. */
void orc_perm_trace(const orc_chip_air *chip, const uint32_t *main, const uint32_t *prep, uint32_t log_n,
                    const uint32_t *pub, const uint32_t alpha[4], const uint32_t beta[4], uint32_t *perm_out,
                    uint32_t cumsum_out[4]) {
    size_t n = (size_t)1 << log_n;
    uint32_t ni = chip->n_interactions, nb = (ni + 1) / 2, phi_col = nb - 1;
    ef_t al, be;
    memcpy(al.c, alpha, 16);
    memcpy(be.c, beta, 16);
    ef_t *bp = malloc(sizeof(ef_t) * (chip->max_arity + 1));
    ef_t *totals = malloc(sizeof(ef_t) * n);
    bp[0] = be;
    for (uint32_t k = 1; k < chip->max_arity; k++) bp[k] = ef_mul(bp[k - 1], be);
#pragma omp parallel
    {
    uint32_t *ml = malloc(4 * (chip->main_w + 1)), *mn = malloc(4 * (chip->main_w + 1));
    uint32_t *pl = malloc(4 * (chip->prep_w + 1)), *pn = malloc(4 * (chip->prep_w + 1));
    uint32_t *mult = malloc(4 * (ni + 1)), *vals = malloc(4 * (size_t)(ni + 1) * chip->max_arity);
#pragma omp for schedule(static)
    for (size_t r = 0; r < n; r++) {
        ef_t total = ef_zero();
        size_t rn = (r + 1) & (n - 1);
        gather_row(main, chip->main_w, n, r, ml);
        gather_row(main, chip->main_w, n, rn, mn);
        gather_row(prep, chip->prep_w, n, r, pl);
        gather_row(prep, chip->prep_w, n, rn, pn);
        chip->interactions(ml, mn, pl, pn, pub, mult, vals);
        for (uint32_t b = 0; b < nb; b++) {
            ef_t acc = ef_zero();
            for (uint32_t j = 2 * b; j < 2 * b + 2 && j < ni; j++) {
                if (!mult[j]) continue;
                ef_t d = ef_add(al, ef_from_base((bb_t)chip->inter[j].bus));
                for (int k = 0; k < chip->inter[j].arity; k++) d = ef_add(d, ef_mul_base(bp[k], vals[(size_t)j * chip->max_arity + k]));
                ef_t t = ef_mul_base(ef_inv(d), mult[j]);
                acc = chip->inter[j].sign > 0 ? ef_add(acc, t) : ef_sub(acc, t);
            }
            if (b < phi_col)
                for (int k = 0; k < 4; k++) perm_out[(size_t)(4 * b + k) * n + r] = acc.c[k];
            total = ef_add(total, acc);
        }
        totals[r] = total;
    }
    free(ml); free(mn); free(pl); free(pn); free(mult); free(vals);
    }
    ef_t sum = ef_zero();
    for (size_t r = 0; r < n; r++) sum = ef_add(sum, totals[r]);
    const ef_t step = ef_mul_base(sum, bb_inv((bb_t)(n % BB_P)));     /* S / N */
    ef_t run = ef_zero();
    for (size_t r = 0; r < n; r++) {
        for (int k = 0; k < 4; k++) perm_out[(size_t)(4 * phi_col + k) * n + r] = run.c[k];
        run = ef_sub(ef_add(run, totals[r]), step);
    }
    memcpy(cumsum_out, sum.c, 16);
    free(bp);
    free(totals);
}
