/* ORACLE — TEST INFRASTRUCTURE ONLY: how many OpenMP threads the CPU baseline used. */
#include <omp.h>
int orc_num_threads(void) { return omp_get_max_threads(); }

/* Proof-of-work grind of the duplex challenger (tests/_oracle_prover.py step 7): the smallest witness w such that the
 * permutation of `state` with state[pos] = w has its word 7 divisible by 2^bits.  Blocks of candidates are tried in
 * parallel; the smallest hit of the first block that has one is THE answer (every smaller candidate was tried). */
#include <stdint.h>
#include <string.h>
void orc_poseidon2_permute(uint32_t state[16]);
uint32_t orc_pow_grind(const uint32_t state[16], uint32_t pos, uint32_t bits) {
    const uint32_t mask = (1u << bits) - 1, BLOCK = 4096;
    for (uint32_t base = 0;; base += BLOCK) {
        uint32_t best = 0xFFFFFFFFu;
#pragma omp parallel for reduction(min : best)
        for (uint32_t k = 0; k < BLOCK; k++) {
            uint32_t s[16];
            memcpy(s, state, sizeof s);
            s[pos] = base + k;
            orc_poseidon2_permute(s);
            if ((s[7] & mask) == 0 && base + k < best) best = base + k;
        }
        if (best != 0xFFFFFFFFu) return best;
    }
}
