// K2 / K3 — Poseidon2 leaf hashing and Merkle levels for gfx950
// (SURVEY.md section 8(a) rows K2, K3).
//
// One thread owns one row (leaf) or one parent node; the 16-word Poseidon2 state
// stays in VGPRs and the permutation is fully unrolled.  Because matrices are
// column-major, the 64 lanes of a wave read 64 consecutive rows of the same
// column: every global load is a fully coalesced 256-B request, and the column
// base pointers come from a wave-uniform (scalar) pointer table, which lets one
// kernel hash the concatenation of any number of equal-height matrices.
// Digests are stored array-of-structs ([node][8] words, 32 B).  Pairing is
// natural-order: parent i of a layer of L nodes = compress(child i, child i+L),
// so row r of a tall matrix and row r mod L of a shorter one share a path (the
// relation natural-order FRI folding needs) and both child reads are coalesced.
//
// ALU-bound (about 141 S-boxes = 564 Montgomery products per permutation, one
// permutation per 8 input words), not HBM-bound: see DESIGN.md.
#include "kernels.h"

namespace dvt {

__device__ __forceinline__ void hash_row(const uint32_t *const *cols, uint32_t ncols, size_t row, Fp s[16]) {
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = Fp::zero();
    uint32_t g = 0;
    for (; g + 8 <= ncols; g += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = Fp::raw(cols[g + k][row]);
        p2_permute(s);
    }
    const uint32_t rem = ncols - g;
    if (rem) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((uint32_t)k < rem) s[k] = Fp::raw(cols[g + k][row]);
        p2_permute(s);
    }
}

__device__ __forceinline__ void store_digest(uint32_t *out, size_t node, const Fp s[16]) {
    uint4 *o = reinterpret_cast<uint4 *>(out + node * 8);
    o[0] = make_uint4(s[0].v, s[1].v, s[2].v, s[3].v);
    o[1] = make_uint4(s[4].v, s[5].v, s[6].v, s[7].v);
}

__global__ void __launch_bounds__(256) merkle_leaves_kernel(const uint32_t *const *cols, uint32_t ncols, size_t height,
                                                           uint32_t *out) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= height) return;
    Fp s[16];
    hash_row(cols, ncols, row, s);
    store_digest(out, row, s);
}

__global__ void __launch_bounds__(256) merkle_level_kernel(const uint32_t *prev, const uint32_t *const *cols,
                                                          uint32_t ncols, size_t len, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    // natural-order pairing: children i and i + len (two coalesced 32-B-per-lane streams)
    const uint4 *pl = reinterpret_cast<const uint4 *>(prev + i * 8);
    const uint4 *pr = reinterpret_cast<const uint4 *>(prev + (i + len) * 8);
    uint4 a = pl[0], b = pl[1], c = pr[0], d = pr[1];
    Fp s[16] = {Fp::raw(a.x), Fp::raw(a.y), Fp::raw(a.z), Fp::raw(a.w), Fp::raw(b.x), Fp::raw(b.y),
                Fp::raw(b.z), Fp::raw(b.w), Fp::raw(c.x), Fp::raw(c.y), Fp::raw(c.z), Fp::raw(c.w),
                Fp::raw(d.x), Fp::raw(d.y), Fp::raw(d.z), Fp::raw(d.w)};
    p2_permute(s);
    if (ncols) {
        Fp h[16];
        hash_row(cols, ncols, i, h);
#pragma unroll
        for (int k = 0; k < 8; k++) s[8 + k] = h[k];
        p2_permute(s);
    }
    store_digest(out, i, s);
}

// All levels from a layer of 2^log_start nodes (log_start <= MERKLE_TOP_LOG) down to the root in ONE
// single-workgroup launch: the upper levels are launch-latency bound (a proof has ~25 trees x ~11 of
// them), so they share a launch and synchronise with __syncthreads().  Every node is written once and
// read once afterwards by the same workgroup, so no cross-CU visibility is involved.
__global__ void __launch_bounds__(1024) merkle_top_kernel(uint32_t *layer, uint32_t log_start, MerkleTopInject inj) {
    uint32_t *prev = layer;
    for (uint32_t lh = log_start; lh-- > 0;) {
        const size_t len = (size_t)1 << lh;
        uint32_t *cur = prev + ((size_t)16 << lh);
        for (size_t i = threadIdx.x; i < len; i += blockDim.x) {
            const uint4 *pl = reinterpret_cast<const uint4 *>(prev + i * 8);
            const uint4 *pr = reinterpret_cast<const uint4 *>(prev + (i + len) * 8);
            uint4 a = pl[0], b = pl[1], c = pr[0], d = pr[1];
            Fp s[16] = {Fp::raw(a.x), Fp::raw(a.y), Fp::raw(a.z), Fp::raw(a.w), Fp::raw(b.x), Fp::raw(b.y),
                        Fp::raw(b.z), Fp::raw(b.w), Fp::raw(c.x), Fp::raw(c.y), Fp::raw(c.z), Fp::raw(c.w),
                        Fp::raw(d.x), Fp::raw(d.y), Fp::raw(d.z), Fp::raw(d.w)};
            p2_permute(s);
            if (inj.ncols[lh]) {
                Fp h[16];
                hash_row(inj.cols[lh], inj.ncols[lh], i, h);
#pragma unroll
                for (int k = 0; k < 8; k++) s[8 + k] = h[k];
                p2_permute(s);
            }
            store_digest(cur, i, s);
        }
        __threadfence_block();
        __syncthreads();
        prev = cur;
    }
}

hipError_t launch_merkle_top(hipStream_t st, uint32_t *d_layer, uint32_t log_start, const MerkleTopInject &inj) {
    if (log_start == 0) return hipSuccess;
    if (log_start > MERKLE_TOP_LOG) return hipErrorInvalidValue;
    unsigned threads = log_start >= 11 ? 1024 : (log_start <= 6 ? 64 : 1u << (log_start - 1));
    merkle_top_kernel<<<1, threads, 0, st>>>(d_layer, log_start, inj);
    return hipGetLastError();
}

__global__ void poseidon2_permute_kernel(uint32_t *states, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = Fp::raw(states[i * 16 + k]);
    p2_permute(s);
#pragma unroll
    for (int k = 0; k < 16; k++) states[i * 16 + k] = s[k].v;
}

hipError_t launch_merkle_leaves(hipStream_t st, const uint32_t *const *d_cols, uint32_t ncols, uint32_t log_height,
                                uint32_t *d_out) {
    size_t h = (size_t)1 << log_height;
    unsigned blocks = (unsigned)((h + 255) / 256);
    merkle_leaves_kernel<<<blocks, 256, 0, st>>>(d_cols, ncols, h, d_out);
    return hipGetLastError();
}

hipError_t launch_merkle_level(hipStream_t st, const uint32_t *d_prev, const uint32_t *const *d_cols, uint32_t ncols,
                               uint32_t log_len, uint32_t *d_out) {
    size_t len = (size_t)1 << log_len;
    unsigned blocks = (unsigned)((len + 255) / 256);
    merkle_level_kernel<<<blocks, 256, 0, st>>>(d_prev, d_cols, ncols, len, d_out);
    return hipGetLastError();
}

hipError_t launch_poseidon2_permute(hipStream_t st, uint32_t *d_states, size_t n) {
    if (!n) return hipSuccess;
    unsigned blocks = (unsigned)((n + 255) / 256);
    poseidon2_permute_kernel<<<blocks, 256, 0, st>>>(d_states, n);
    return hipGetLastError();
}

}  // namespace dvt
