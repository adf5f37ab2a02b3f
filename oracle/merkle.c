/* ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see field.h).
 *
 * Mixed-height Merkle commitment over Poseidon2 (SURVEY.md section 8(a) rows K2/K3;
 * in stock SP1 this is p3-merkle-tree's field MMCS, source absent):
 *   layer 0   : digest[r] = sponge(row r of every tallest matrix, concatenated)
 *   layer k+1 : digest[i] = compress(prev[i], prev[i + L]) with L = |layer k+1|
 *               (natural-order pairing: row r of a matrix of height H and row
 *               r mod L of a matrix of height L share a path, which is exactly the
 *               relation natural-order FRI folding needs); when matrices of
 *               exactly that height exist, digest[i] = compress(digest[i],
 *               sponge(their rows i, concatenated))
 * Heights are powers of two.  Natural row order. */
#include "dvt_oracle.h"
#include <stdlib.h>
#include <string.h>

static size_t gather_rows(const orc_matrix *mats, size_t n, unsigned log_h, size_t row, uint32_t *buf) {
    size_t k = 0;
    for (size_t m = 0; m < n; m++) {
        if (mats[m].log_height != log_h) continue;
        size_t h = (size_t)1 << log_h;
        for (uint32_t c = 0; c < mats[m].width; c++) buf[k++] = mats[m].data[(size_t)c * h + row];
    }
    return k;
}

size_t orc_merkle_digest_words(const orc_matrix *mats, size_t n) {
    unsigned mx = 0;
    for (size_t m = 0; m < n; m++) if (mats[m].log_height > mx) mx = mats[m].log_height;
    return (((size_t)2 << mx) - 1) * 8;
}

void orc_merkle_commit(const orc_matrix *mats, size_t n, uint32_t *digests) {
    unsigned mx = 0;
    size_t total_w = 0;
    for (size_t m = 0; m < n; m++) {
        if (mats[m].log_height > mx) mx = mats[m].log_height;
        total_w += mats[m].width;
    }
    size_t H = (size_t)1 << mx;
#pragma omp parallel
    {
        uint32_t *buf = (uint32_t *)malloc(sizeof(uint32_t) * (total_w + 1));
#pragma omp for schedule(static)
        for (size_t r = 0; r < H; r++) {
            size_t k = gather_rows(mats, n, mx, r, buf);
            orc_hash_slice(buf, k, digests + 8 * r);
        }
        free(buf);
    }
    uint32_t *prev = digests;
    for (unsigned lh = mx; lh-- > 0;) {
        size_t L = (size_t)1 << lh;
        uint32_t *cur = prev + 8 * (L * 2);
        int inject = 0;
        for (size_t m = 0; m < n; m++) if (mats[m].log_height == lh) inject = 1;
#pragma omp parallel if (L >= 256)
        {
            uint32_t *buf = (uint32_t *)malloc(sizeof(uint32_t) * (total_w + 1));
#pragma omp for schedule(static)
            for (size_t i = 0; i < L; i++) {
                orc_compress(prev + 8 * i, prev + 8 * (i + L), cur + 8 * i);
                if (inject) {
                    uint32_t hr[8];
                    size_t k = gather_rows(mats, n, lh, i, buf);
                    orc_hash_slice(buf, k, hr);
                    orc_compress(cur + 8 * i, hr, cur + 8 * i);
                }
            }
            free(buf);
        }
        prev = cur;
    }
}
