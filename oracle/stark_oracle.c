/* ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see field.h).
 *
 * CPU restatements of the prover stages K5 (quotient), K6 (out-of-domain
 * openings), K7 (FRI input) and K8 (FRI fold) of SURVEY.md section 8(a), written
 * against the protocol text in DESIGN.md, not against the HIP kernels: canonical
 * field arithmetic, row-at-a-time loops, openings by interpolation + Horner
 * (the GPU uses barycentric weights).  tests/_oracle_prover.py chains them with a
 * Python duplex challenger into a full CPU prover whose proof bytes must equal
 * the GPU's.  In stock SP1 these stages live in sp1-stark / p3-fri / p3-uni-stark
 * (absent; reference call site src/main.rs:463-466). */
#include "air_oracle.h"
#include "dvt_oracle.h"
#include "field.h"
#include <stdlib.h>
#include <string.h>

static ef_t ef_load(const uint32_t *p) { ef_t r; memcpy(r.c, p, 16); return r; }
static void ef_store(uint32_t *p, ef_t v) { memcpy(p, v.c, 16); }
static ef_t ef_sub_base(ef_t a, bb_t b) { a.c[0] = bb_sub(a.c[0], b); return a; }

/* K5.  LDEs are on the coset 31*<w_2N>, natural order; next row = index + 2.
 * out: chunk 0 (even LDE rows) [4][N] then chunk 1 (odd rows) [4][N]. */
void orc_quotient(const orc_chip_air *chip, const uint32_t *main_lde, const uint32_t *prep_lde, const uint32_t *perm_lde,
                  uint32_t log_n, const uint32_t *pub, const uint32_t perm_alpha[4], const uint32_t beta[4],
                  const uint32_t alpha[4], const uint32_t cumsum[4], uint32_t *out) {
    const size_t n = (size_t)1 << log_n, m = 2 * n;
    const uint32_t ni = chip->n_interactions, nb = (ni + 1) / 2, nc = chip->n_constraints;
    const uint32_t nfold = nc + (ni ? nb : 0), phi_col = nb - 1;
    ef_t al = ef_load(alpha), be = ef_load(beta), pa = ef_load(perm_alpha);
    const ef_t cs_over_n = ef_mul_base(ef_load(cumsum), bb_inv((bb_t)(n % BB_P)));
    ef_t *apow = malloc(sizeof(ef_t) * (nfold + 1)), *bpow = malloc(sizeof(ef_t) * (chip->max_arity + 1));
    apow[0] = ef_one();
    for (uint32_t k = 1; k < nfold; k++) apow[k] = ef_mul(apow[k - 1], al);
    bpow[0] = be;
    for (uint32_t k = 1; k < chip->max_arity; k++) bpow[k] = ef_mul(bpow[k - 1], be);
    const bb_t g = BB_GENERATOR, w2 = bb_two_adic_gen(log_n + 1), w_inv = bb_inv(bb_two_adic_gen(log_n));
    const bb_t gn = bb_pow(g, n);
#pragma omp parallel
    {
    uint32_t *ml = malloc(4 * (chip->main_w + 1)), *mn = malloc(4 * (chip->main_w + 1));
    uint32_t *pl = malloc(4 * (chip->prep_w + 1)), *pn = malloc(4 * (chip->prep_w + 1));
    uint32_t *cv = malloc(4 * (nc + 1)), *mult = malloc(4 * (ni + 1)), *vals = malloc(4 * (size_t)(ni + 1) * chip->max_arity);
#pragma omp for schedule(static)
    for (size_t i = 0; i < m; i++) {
        const bb_t x = bb_mul(g, bb_pow(w2, i));
        const size_t inx = (i + 2) & (m - 1);
        for (uint32_t c = 0; c < chip->main_w; c++) { ml[c] = main_lde[(size_t)c * m + i]; mn[c] = main_lde[(size_t)c * m + inx]; }
        for (uint32_t c = 0; c < chip->prep_w; c++) { pl[c] = prep_lde[(size_t)c * m + i]; pn[c] = prep_lde[(size_t)c * m + inx]; }
        const bb_t zh = (i & 1) ? bb_sub(bb_neg(gn), 1) : bb_sub(gn, 1);   /* x^N - 1 */
        const bb_t sel_first = bb_mul(zh, bb_inv(bb_sub(x, 1)));
        const bb_t sel_last = bb_mul(zh, bb_inv(bb_sub(x, w_inv)));
        const bb_t sel_trans = bb_sub(x, w_inv);
        ef_t acc = ef_zero();
        chip->constraints(ml, mn, pl, pn, pub, cv);
        for (uint32_t k = 0; k < nc; k++) {
            bb_t v = cv[k];
            if (chip->when[k] == 1) v = bb_mul(v, sel_first);
            else if (chip->when[k] == 2) v = bb_mul(v, sel_last);
            else if (chip->when[k] == 3) v = bb_mul(v, sel_trans);
            acc = ef_add(acc, ef_mul_base(apow[k], v));
        }
        if (ni) {
            chip->interactions(ml, mn, pl, pn, pub, mult, vals);
            /* value of the last batch: phi' - phi - (the other batches) + S / N  (it has no column) */
            ef_t last;
            for (int k = 0; k < 4; k++)
                last.c[k] = bb_sub(perm_lde[(size_t)(4 * phi_col + k) * m + inx], perm_lde[(size_t)(4 * phi_col + k) * m + i]);
            last = ef_add(last, cs_over_n);
            for (uint32_t b = 0; b < phi_col; b++) {
                ef_t pcol;
                for (int k = 0; k < 4; k++) pcol.c[k] = perm_lde[(size_t)(4 * b + k) * m + i];
                last = ef_sub(last, pcol);
            }
            for (uint32_t b = 0; b < nb; b++) {
                ef_t pcol = last;
                if (b < phi_col)
                    for (int k = 0; k < 4; k++) pcol.c[k] = perm_lde[(size_t)(4 * b + k) * m + i];
                /* v_b * prod(d_j) - sum_j s_j m_j prod_{k != j} d_k */
                ef_t d[2];
                bb_t sm[2];
                uint32_t cnt = 0;
                for (uint32_t j = 2 * b; j < 2 * b + 2 && j < ni; j++, cnt++) {
                    ef_t dj = ef_add(pa, ef_from_base((bb_t)chip->inter[j].bus));
                    for (int k = 0; k < chip->inter[j].arity; k++) dj = ef_add(dj, ef_mul_base(bpow[k], vals[(size_t)j * chip->max_arity + k]));
                    d[cnt] = dj;
                    sm[cnt] = chip->inter[j].sign > 0 ? mult[j] : bb_neg(mult[j]);
                }
                ef_t cons;
                if (cnt == 2) cons = ef_sub(ef_mul(ef_mul(pcol, d[0]), d[1]), ef_add(ef_mul_base(d[1], sm[0]), ef_mul_base(d[0], sm[1])));
                else cons = ef_sub(ef_mul(pcol, d[0]), ef_from_base(sm[0]));
                acc = ef_add(acc, ef_mul(apow[nc + b], cons));
            }
        }
        ef_t q = ef_mul_base(acc, bb_inv(zh));
        uint32_t *o = out + ((i & 1) ? 4 * n : 0) + (i >> 1);
        for (int k = 0; k < 4; k++) o[(size_t)k * n] = q.c[k];
    }
    free(ml); free(mn); free(pl); free(pn); free(cv); free(mult); free(vals);
    }
    free(apow); free(bpow);
}

/* K6.  cols [width][N]: evaluations over coset_shift * <w_N>.  out[c] = p_c(z) in F_p^4.
 * Interpolate (inverse DFT, undo the shift on the coefficients), then Horner in the extension. */
void orc_eval_columns(const uint32_t *cols, uint32_t width, uint32_t log_n, uint32_t coset_shift, const uint32_t z[4], uint32_t *out) {
    const size_t n = (size_t)1 << log_n;
    ef_t zz = ef_load(z);
    const bb_t sinv = bb_inv(coset_shift);
#pragma omp parallel for schedule(dynamic)
    for (uint32_t c = 0; c < width; c++) {
        bb_t *co = malloc(4 * n);
        memcpy(co, cols + (size_t)c * n, 4 * n);
        orc_idft(co, log_n);
        bb_t s = 1;
        for (size_t k = 0; k < n; k++) { co[k] = bb_mul(co[k], s); s = bb_mul(s, sinv); }
        ef_t acc = ef_zero();
        for (size_t k = n; k-- > 0;) acc = ef_add(ef_mul(acc, zz), ef_from_base(co[k]));
        ef_store(out + 4 * c, acc);
        free(co);
    }
}

/* K7.  cols: n_all LDE columns of height 2^log_m on 31*<w_M>, the first n_two opened at zeta and zeta_next.
 * out[i] = sum_c a^c (p_c(x_i) - p_c(zeta)) / (x_i - zeta)
 *        + a^n_all sum_{c<n_two} a^c (p_c(x_i) - p_c(zeta_next)) / (x_i - zeta_next). */
void orc_reduced_opening(const uint32_t *const *cols, uint32_t n_two, uint32_t n_all, uint32_t log_m, const uint32_t alpha[4],
                         const uint32_t *open_local, const uint32_t *open_next, const uint32_t zeta[4], const uint32_t zeta_next[4],
                         uint32_t *out) {
    const size_t m = (size_t)1 << log_m;
    ef_t al = ef_load(alpha), ze = ef_load(zeta), zn = ef_load(zeta_next);
    ef_t *apow = malloc(sizeof(ef_t) * (n_all + 1));
    apow[0] = ef_one();
    for (uint32_t k = 1; k <= n_all; k++) apow[k] = ef_mul(apow[k - 1], al);
    const bb_t w = bb_two_adic_gen(log_m);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; i++) {
        const bb_t x = bb_mul(BB_GENERATOR, bb_pow(w, i));
        ef_t s1 = ef_zero(), s2 = ef_zero();
        for (uint32_t c = 0; c < n_all; c++) {
            bb_t px = cols[c][i];
            s1 = ef_add(s1, ef_mul(apow[c], ef_neg(ef_sub_base(ef_load(open_local + 4 * c), px))));
            if (c < n_two) s2 = ef_add(s2, ef_mul(apow[c], ef_neg(ef_sub_base(ef_load(open_next + 4 * c), px))));
        }
        ef_t r = ef_mul(s1, ef_inv(ef_neg(ef_sub_base(ze, x))));
        if (n_two) r = ef_add(r, ef_mul(apow[n_all], ef_mul(s2, ef_inv(ef_neg(ef_sub_base(zn, x))))));
        ef_store(out + 4 * i, r);
    }
    free(apow);
}

/* K8.  out[i] = (v[i] + v[i+h])/2 + beta (v[i] - v[i+h]) / (2 w_M^i) (+ ro[i]),  h = M/2. */
void orc_fri_fold(const uint32_t *v, uint32_t log_m, const uint32_t beta[4], const uint32_t *ro, uint32_t *out) {
    const size_t half = (size_t)1 << (log_m - 1);
    ef_t be = ef_load(beta);
    const bb_t inv2 = bb_inv(2), winv = bb_inv(bb_two_adic_gen(log_m));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < half; i++) {
        const bb_t xi = bb_pow(winv, i);
        ef_t lo = ef_load(v + 4 * i), hi = ef_load(v + 4 * (i + half));
        ef_t r = ef_add(ef_mul_base(ef_add(lo, hi), inv2), ef_mul(be, ef_mul_base(ef_sub(lo, hi), bb_mul(inv2, xi))));
        if (ro) r = ef_add(r, ef_load(ro + 4 * i));
        ef_store(out + 4 * i, r);
    }
}
