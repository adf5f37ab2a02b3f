/* ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see field.h).
 *
 * Radix-2 DFT over BabyBear and the coset low-degree extension used by every
 * commitment of the prover (in stock SP1 this is
 * p3-dft's coset_lde_batch, source absent).  Textbook bit-reverse + DIT
 * butterflies, one column at a time — deliberately not the 3-pass LDS-tiled
 * four-step decomposition the HIP kernels use. */
#include "dvt_oracle.h"
#include "field.h"
#include <stdlib.h>
#include <string.h>

static void bitrev_permute(bb_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = 0;
        for (unsigned b = 0; b < log_n; b++) j |= ((i >> b) & 1) << (log_n - 1 - b);
        if (j > i) { bb_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}

static void dft_root(bb_t *a, unsigned log_n, bb_t root) {
    size_t n = (size_t)1 << log_n;
    bitrev_permute(a, log_n);
    bb_t *tw = (bb_t *)malloc(sizeof(bb_t) * (n / 2 + 1));
    tw[0] = 1;
    for (size_t i = 1; i < n / 2; i++) tw[i] = bb_mul(tw[i - 1], root);
    for (unsigned s = 0; s < log_n; s++) {
        size_t half = (size_t)1 << s, step = n >> (s + 1);
        for (size_t blk = 0; blk < n; blk += 2 * half)
            for (size_t j = 0; j < half; j++) {
                bb_t u = a[blk + j], v = bb_mul(a[blk + j + half], tw[j * step]);
                a[blk + j] = bb_add(u, v);
                a[blk + j + half] = bb_sub(u, v);
            }
    }
    free(tw);
}

void orc_dft(uint32_t *a, unsigned log_n) { dft_root(a, log_n, bb_two_adic_gen(log_n)); }

void orc_idft(uint32_t *a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    dft_root(a, log_n, bb_inv(bb_two_adic_gen(log_n)));
    bb_t ninv = bb_inv((bb_t)(n % BB_P));
    for (size_t i = 0; i < n; i++) a[i] = bb_mul(a[i], ninv);
}

void orc_coset_lde(const uint32_t *in, uint32_t *out, uint32_t width, unsigned log_n,
                   unsigned added_bits, uint32_t shift) {
    size_t n = (size_t)1 << log_n, m = (size_t)1 << (log_n + added_bits);
#pragma omp parallel for schedule(dynamic)
    for (uint32_t c = 0; c < width; c++) {
        bb_t *o = out + (size_t)c * m;
        memcpy(o, in + (size_t)c * n, n * sizeof(bb_t));
        memset(o + n, 0, (m - n) * sizeof(bb_t));
        orc_idft(o, log_n);
        bb_t s = 1;
        for (size_t i = 0; i < n; i++) { o[i] = bb_mul(o[i], s); s = bb_mul(s, shift); }
        orc_dft(o, log_n + added_bits);
    }
}
