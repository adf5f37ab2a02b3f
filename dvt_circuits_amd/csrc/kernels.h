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
struct NttTables {
    uint32_t *base = nullptr;
    const uint32_t *tw_hi = nullptr, *tw_lo = nullptr, *sh_hi = nullptr, *sh_lo = nullptr;
};
hipError_t ntt_tables_create(NttTables *t);
void ntt_tables_destroy(NttTables *t);

// ntt.hip
hipError_t launch_coset_lde(hipStream_t st, const NttTables &tabs, uint32_t *d_in, uint32_t *d_out, uint32_t width,
                            uint32_t log_n, uint32_t shift_mode);
hipError_t launch_to_internal(hipStream_t st, uint32_t *d, size_t n);
hipError_t launch_from_internal(hipStream_t st, uint32_t *d, size_t n);

// merkle.hip
// d_cols: device array of `ncols` column base pointers (each column has `height` words)
hipError_t launch_merkle_leaves(hipStream_t st, const uint32_t *const *d_cols, uint32_t ncols, uint32_t log_height,
                                uint32_t *d_out);
// d_out[i] = compress(prev[2i], prev[2i+1]); with ncols > 0 additionally
// d_out[i] = compress(d_out[i], sponge(row i of the injected columns))
hipError_t launch_merkle_level(hipStream_t st, const uint32_t *d_prev, const uint32_t *const *d_cols, uint32_t ncols,
                               uint32_t log_len, uint32_t *d_out);
hipError_t launch_poseidon2_permute(hipStream_t st, uint32_t *d_states, size_t n);

}  // namespace dvt
