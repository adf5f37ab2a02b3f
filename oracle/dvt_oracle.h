/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see field.h for the full notice).
 * PARITY UNPINNED against stock SP1: no golden proof/vkey/digest exists in the
 * reference tree (its tests run `execute` and compare exit codes only,
 * reference script/run.sh:66-89).  What this oracle pins is the HIP product
 * path, bit-exactly, on the same inputs.
 *
 * All matrices are COLUMN-MAJOR: element (row r, column c) of a matrix with
 * `height` rows lives at data[c * height + r].  All orders are natural (no
 * bit-reversed storage anywhere in this project).
 */
#ifndef DVT_ORACLE_H
#define DVT_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- SHA-256 (used only to derive Poseidon2 round constants) ---- */
void orc_sha256(const uint8_t *msg, size_t len, uint8_t out[32]);

/* ---- Poseidon2, width 16, x^7, 8 external + 13 internal rounds ---- */
/* copies the 8*16 external then 13 internal constants (canonical form) */
void orc_poseidon2_constants(uint32_t ext_rc[128], uint32_t int_rc[13], uint32_t diag[16]);
void orc_poseidon2_permute(uint32_t state[16]);
/* padding-free overwrite sponge, rate 8, output 8 */
void orc_hash_slice(const uint32_t *in, size_t n, uint32_t out[8]);
void orc_compress(const uint32_t l[8], const uint32_t r[8], uint32_t out[8]);

/* ---- NTT / LDE over BabyBear ---- */
/* in-place forward DFT of size 2^log_n in natural order: out[k] = sum_j in[j] w^{jk} */
void orc_dft(uint32_t *a, unsigned log_n);
void orc_idft(uint32_t *a, unsigned log_n);
/* columns of `in` ([width][2^log_n]) are evaluations over H = <w_N> in natural order;
 * out ([width][2^(log_n+added_bits)]) = evaluations of the same polynomials on
 * shift * <w_{N*2^added_bits}>, natural order. */
void orc_coset_lde(const uint32_t *in, uint32_t *out, uint32_t width, unsigned log_n,
                   unsigned added_bits, uint32_t shift);

/* ---- mixed-height Merkle commitment (MMCS) ---- */
typedef struct {
    const uint32_t *data; /* column-major [width][height] */
    uint32_t width;
    uint32_t log_height;
} orc_matrix;
/* Commit to `n` matrices.  digests must hold (2*H - 1)*8 words where H is the
 * tallest height: layer 0 (H digests) first, then H/2, ... , 1 (the root last).
 * Matrices of equal height are hashed in argument order.  Pairing is natural-order:
 * parent i of a layer of L nodes = compress(child i, child i + L). */
void orc_merkle_commit(const orc_matrix *mats, size_t n, uint32_t *digests);
/* words needed for `digests` */
size_t orc_merkle_digest_words(const orc_matrix *mats, size_t n);

#ifdef __cplusplus
}
#endif
#endif
