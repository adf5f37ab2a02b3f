// Internal launch interface between the C-ABI layer (capi.hip) and the gfx950
// kernels.  Not installed; the public surface is include/dvt_prover.h.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "bb.cuh"
#include "poseidon2.cuh"

namespace dvt {

// Device-resident two-level power tables (built once per prover handle).
//   Omega = primitive 2^24-th root:  Omega^e = tw_hi[e >> 12] * tw_lo[e & 4095]
//   g = 31 (coset shift):            g^k     = sh_hi[k >> 12] * sh_lo[k & 4095], k < 2^23
//   lde_tw:    for every transform size 2^l, 1 <= l <= 13, the 2^(l-1) twiddles w_{2^l}^e at offset 2^(l-1) - 1
//   lde_scale: for every (shift_mode, log_n) the per-position coefficient scaling of lde_block (4096 words each,
//              at ((mode * LDE_MAX_LOG + log_n) << 12)): see ntt.hip
constexpr uint32_t LDE_MAX_LOG = 23;
struct NttTables {
    uint32_t *base = nullptr;
    const uint32_t *tw_hi = nullptr, *tw_lo = nullptr, *sh_hi = nullptr, *sh_lo = nullptr;
    const uint32_t *lde_tw = nullptr, *lde_scale = nullptr;      // Montgomery words
    const uint32_t *lde_tw_c = nullptr, *lde_scale_c = nullptr;  // the same as centred canonical residues (two's complement)
};
hipError_t ntt_tables_create(NttTables *t);
void ntt_tables_destroy(NttTables *t);

// ntt.hip
hipError_t launch_coset_lde(hipStream_t st, const NttTables &tabs, uint32_t *d_in, uint32_t *d_scratch, uint32_t *d_out,
                            uint32_t width, uint32_t log_n, uint32_t shift_mode);
hipError_t launch_to_internal(hipStream_t st, uint32_t *d, size_t n);
hipError_t launch_from_internal(hipStream_t st, uint32_t *d, size_t n);

// merkle.hip
// Row sponges of ONE tree in one launch: the leaves (the tallest matrices) and, for every lower level where shorter matrices
// join the tree, the digests of their rows ("injected" at that level by launch_merkle_level / launch_merkle_top).  A short
// and wide table (a precompile chip: 2^14 rows of 1245 columns = 156 sequential permutations per row on a few hundred
// waves) is latency-bound on its own; in the same launch as the tall leaves its chain hides behind them.  Segments run in
// the order given (the launch puts the longest chains first).
constexpr uint32_t MERKLE_TOP_LOG = 6;   // below this the levels share one single-workgroup launch (16 lanes per node)
constexpr uint32_t MERKLE_COOP_LOG = 13; // levels of at most 2^13 nodes run one node per 16 lanes (latency-, not throughput-bound)
constexpr uint32_t MERKLE_MAX_SEGMENTS = 24;
struct MerkleLeafSegments {
    uint32_t n = 0;
    const uint32_t *const *cols[MERKLE_MAX_SEGMENTS];   // device array of `ncols` column base pointers (2^log_h words each)
    uint32_t *out[MERKLE_MAX_SEGMENTS];                 // [2^log_h][8] digests
    uint32_t ncols[MERKLE_MAX_SEGMENTS], log_h[MERKLE_MAX_SEGMENTS];
    uint32_t coop[MERKLE_MAX_SEGMENTS];                 // set by the launch: 16 lanes per row
    uint32_t block0[MERKLE_MAX_SEGMENTS + 1];           // set by the launch: first workgroup of every segment
    void add(const uint32_t *const *c, uint32_t nc, uint32_t lh, uint32_t *o) { cols[n] = c; ncols[n] = nc; log_h[n] = lh; out[n] = o; n++; }
};
hipError_t launch_merkle_leaves(hipStream_t st, MerkleLeafSegments sg);
// d_out[i] = compress(prev[i], prev[i + len]); with d_inject additionally
// d_out[i] = compress(d_out[i], d_inject[i])   (d_inject: the row digests of the matrices that join at this level)
hipError_t launch_merkle_level(hipStream_t st, const uint32_t *d_prev, const uint32_t *d_inject, uint32_t log_len, uint32_t *d_out);
// every level below a layer of 2^log_start <= 2^MERKLE_TOP_LOG nodes, down to the root, in one launch
struct MerkleTopInject {  // per level lh (nodes = 2^lh): row digests injected at that level (or nullptr)
    const uint32_t *digests[MERKLE_TOP_LOG + 1] = {};
};
hipError_t launch_merkle_top(hipStream_t st, uint32_t *d_layer, uint32_t log_start, const MerkleTopInject &inj);
hipError_t launch_poseidon2_permute(hipStream_t st, uint32_t *d_states, size_t n);

// fri.hip
hipError_t launch_prefix_sum_columns(hipStream_t st, uint32_t *d_cols, uint32_t ncols, size_t n, uint32_t *d_scratch);
size_t prefix_sum_scratch_words(uint32_t ncols, size_t n);
// K4 tail: d_totals = inclusive prefix sums S of the row totals ([4][n], one F_p^4 per row); writes the running-sum
// column phi[r] = S[r-1] - r * S[n-1] / n  (phi[0] = 0) of the folded LogUp layout (stark.cuh) into d_phi ([4][n]) and,
// with cum_out != nullptr, the four words of S[n-1] to cum_out (device-accessible memory, e.g. pinned host memory)
hipError_t launch_phi_from_prefix_sums(hipStream_t st, const uint32_t *d_totals, uint32_t *d_phi, uint32_t log_n, uint32_t *cum_out);
hipError_t launch_open_weights(hipStream_t st, const NttTables &tabs, Fp4 z, uint32_t log_n, Fp4 *d_w);
constexpr uint32_t OPEN_MAX_ROW_BLOCKS = 256;
uint32_t open_row_blocks(uint32_t log_n);
hipError_t launch_open_columns(hipStream_t st, const uint32_t *const *d_cols, uint32_t ncols, uint32_t log_n, const Fp4 *d_w,
                               Fp4 *d_partial, Fp4 *d_out);
// d_alpha_pows_f64: [n_all][4] doubles, alpha^c as centred canonical residues (see centred_canonical in bb.cuh)
hipError_t launch_reduced_opening(hipStream_t st, const NttTables &tabs, const uint32_t *const *d_cols, uint32_t n_two,
                                  uint32_t n_all, uint32_t log_m, const double *d_alpha_pows_f64, Fp4 sz_all, Fp4 sz_two, Fp4 zeta,
                                  Fp4 zeta_next, Fp4 alpha_shift, Fp4 *d_out);
hipError_t launch_fri_fold(hipStream_t st, const NttTables &tabs, const Fp4 *d_v, Fp4 *d_out, const Fp4 *d_ro, Fp4 beta,
                           uint32_t log_m, const Fp4 *d_beta = nullptr);
// transcript step of one FRI round on the device: see fri.hip
hipError_t launch_fri_challenge(hipStream_t st, const uint32_t *d_root, uint32_t *d_state, Fp4 *d_beta_out, uint32_t *d_root_out);
hipError_t launch_fri_leaves(hipStream_t st, const Fp4 *d_v, uint32_t log_m, uint32_t *d_digests);
hipError_t launch_gather_rows(hipStream_t st, const uint32_t *const *d_cols, const uint32_t *d_log_h, uint32_t ncols,
                              const uint32_t *d_idx, uint32_t nq, uint32_t *d_out);
hipError_t launch_gather_paths(hipStream_t st, const uint32_t *d_digests, uint32_t log_h, const uint32_t *d_idx, uint32_t nq,
                               uint32_t *d_out);
hipError_t launch_gather_siblings(hipStream_t st, const Fp4 *d_v, uint32_t log_m, const uint32_t *d_idx, uint32_t nq, Fp4 *d_out);
hipError_t launch_pow_grind(hipStream_t st, const uint32_t state[16], uint32_t pos, uint32_t bits, uint32_t base, uint32_t count,
                            uint32_t *d_found);

}  // namespace dvt
