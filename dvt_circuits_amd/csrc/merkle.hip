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
// ALU-bound (141 S-boxes = 564 modular products per permutation, one permutation per 8 input words), not
// HBM-bound.  The permutation runs on the FP64 pipe (poseidon2_f64.cuh: exact integers in doubles, 6 full-rate
// operations per product instead of 3 quarter-rate integer multiplies); states stay in doubles between the
// permutations of one sponge and are converted from / to the Montgomery words of HBM at the edges.
#include <utility>

#include "kernels.h"
#include "poseidon2_f64.cuh"

namespace dvt {

// sponge over the row `row` of the concatenated matrices; leaves the state reduced (digest = s[0..8))
__device__ __forceinline__ void hash_row(const uint32_t *const *cols, uint32_t ncols, size_t row, double s[16]) {
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0.0;
    // the words of the next 8 columns are requested before the current block is permuted (a permutation is ~6 k
    // FP64 operations, far longer than an HBM round trip)
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = (uint32_t)k < ncols ? cols[k][row] : 0u;
    // ONE inlined copy of the permutation per kernel (~40 KB of straight-line code): a second copy for the ragged last
    // block would put the loop beyond the 64 KB instruction cache.  The last block absorbs `ncols - g` < 8 words.
    for (uint32_t g = 0; g < ncols; g += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (g + k < ncols) s[k] = p2f::from_mont(w[k]);      // (wave-uniform condition)
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = g + 8 + k < ncols ? cols[g + 8 + k][row] : 0u;
        p2f::permute(s);
    }
}

__device__ __forceinline__ void store_digest(uint32_t *out, size_t node, const double s[16]) {
    uint4 *o = reinterpret_cast<uint4 *>(out + node * 8);
    o[0] = make_uint4(p2f::to_mont(s[0]), p2f::to_mont(s[1]), p2f::to_mont(s[2]), p2f::to_mont(s[3]));
    o[1] = make_uint4(p2f::to_mont(s[4]), p2f::to_mont(s[5]), p2f::to_mont(s[6]), p2f::to_mont(s[7]));
}

// children i and i + len of a layer of 2 len digests -> the 16-word compression input
__device__ __forceinline__ void load_children(const uint32_t *prev, size_t i, size_t len, double s[16]) {
    const uint4 *pl = reinterpret_cast<const uint4 *>(prev + i * 8);
    const uint4 *pr = reinterpret_cast<const uint4 *>(prev + (i + len) * 8);
    uint4 a = pl[0], b = pl[1], c = pr[0], d = pr[1];
    const uint32_t w[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = p2f::from_mont(w[k]);
}

// the same sponge with one row per 16 lanes (lane e holds state element e; rate = lanes 0..7)
__device__ __forceinline__ double coop_hash_row(const uint32_t *const *cols, uint32_t ncols, size_t row, bool live, uint32_t e, const p2f::CoopConsts &k) {
    double h = 0.0;
    for (uint32_t g = 0; g < ncols; g += 8) {
        if (e < 8 && g + e < ncols) h = p2f::from_mont(live ? cols[g + e][row] : 0u);
        h = p2f::coop_permute(h, k);
    }
    return h;
}

// every row sponge of a tree: segment s owns workgroups [block0[s], block0[s + 1])
__global__ void __launch_bounds__(256) merkle_leaves_kernel(MerkleLeafSegments sg) {
    uint32_t s = 0;
    while (s + 1 < sg.n && blockIdx.x >= sg.block0[s + 1]) s++;     // (wave-uniform)
    const size_t t = (size_t)(blockIdx.x - sg.block0[s]) * blockDim.x + threadIdx.x;
    const size_t height = (size_t)1 << sg.log_h[s];
    if (sg.coop[s]) {
        const uint32_t e = threadIdx.x & 15;
        const p2f::CoopConsts k = p2f::coop_consts(e);
        const size_t row = t >> 4;
        const bool live = row < height;
        const double h = coop_hash_row(sg.cols[s], sg.ncols[s], row, live, e, k);
        if (live && e < 8) sg.out[s][row * 8 + e] = p2f::to_mont(h);
        return;
    }
    if (t >= height) return;
    double st[16];
    hash_row(sg.cols[s], sg.ncols[s], t, st);
    store_digest(sg.out[s], t, st);
}

// INJECT = false: the big levels of a tree (no shorter matrix joins there) - one permutation, one inlined copy of it.
template <bool INJECT>
__global__ void __launch_bounds__(256) merkle_level_kernel(const uint32_t *prev, const uint32_t *inject, size_t len, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    // natural-order pairing: children i and i + len (two coalesced 32-B-per-lane streams)
    double s[16];
    load_children(prev, i, len, s);
    // (a loop, not two calls: ONE inlined copy of the permutation, see hash_row)
#pragma unroll 1
    for (int pass = 0; pass < (INJECT ? 2 : 1); pass++) {
        if (pass) {
            const uint4 *pd = reinterpret_cast<const uint4 *>(inject + i * 8);
            const uint4 a = pd[0], b = pd[1];
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int k = 0; k < 8; k++) s[8 + k] = p2f::from_mont(w[k]);
        }
        p2f::permute(s);
    }
    store_digest(out, i, s);
}

// ---- latency-bound levels: one node per 16 lanes (p2f::coop_permute) ---------------------------------------------------
// node i of a level of `len` nodes = compress(child i, child i + len) [then compress(node, injected row digest i)]; lane e of
// the node's row holds state element e.  Blocks are whole rows; rows beyond `len` run along (DPP needs the full row) and
// store nothing.
__device__ __forceinline__ void coop_node(const uint32_t *prev, const uint32_t *inject, size_t i, size_t len,
                                          bool live, uint32_t e, const p2f::CoopConsts &k, uint32_t *out) {
    const size_t src = e < 8 ? i * 8 + e : (i + len) * 8 + (e - 8);
    double s = p2f::from_mont(live ? prev[src] : 0u);
    s = p2f::coop_permute(s, k);
    if (inject) {
        // lanes 8..15 take the digest words 0..7 of the injected rows, lanes 0..7 keep the node
        const double hs = p2f::from_mont(live && e >= 8 ? inject[i * 8 + (e - 8)] : 0u);
        s = e < 8 ? s : hs;
        s = p2f::coop_permute(s, k);
    }
    if (live && e < 8) out[i * 8 + e] = p2f::to_mont(s);
}

__global__ void __launch_bounds__(256) merkle_level_coop_kernel(const uint32_t *prev, const uint32_t *inject, size_t len, uint32_t *out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t e = threadIdx.x & 15;
    const p2f::CoopConsts k = p2f::coop_consts(e);
    coop_node(prev, inject, t >> 4, len, (t >> 4) < len, e, k, out);
}

// All levels from a layer of 2^log_start nodes (log_start <= MERKLE_TOP_LOG) down to the root in ONE
// single-workgroup launch: the upper levels are launch-latency bound (a proof has ~25 trees x ~11 of
// them), so they share a launch and synchronise with __syncthreads().  Every node is written once and
// read once afterwards by the same workgroup, so no cross-CU visibility is involved.
__global__ void __launch_bounds__(1024) merkle_top_kernel(uint32_t *layer, uint32_t log_start, MerkleTopInject inj) {
    const uint32_t e = threadIdx.x & 15, slot = threadIdx.x >> 4, slots = blockDim.x >> 4;
    const p2f::CoopConsts k = p2f::coop_consts(e);
    uint32_t *prev = layer;
    for (uint32_t lh = log_start; lh-- > 0;) {
        const size_t len = (size_t)1 << lh;
        uint32_t *cur = prev + ((size_t)16 << lh);
        for (size_t base = 0; base < len; base += slots)       // uniform trip count: every row takes part in every pass
            coop_node(prev, inj.digests[lh], base + slot, len, base + slot < len, e, k, cur);
        __threadfence_block();
        __syncthreads();
        prev = cur;
    }
}

hipError_t launch_merkle_top(hipStream_t st, uint32_t *d_layer, uint32_t log_start, const MerkleTopInject &inj) {
    if (log_start == 0) return hipSuccess;
    if (log_start > MERKLE_TOP_LOG) return hipErrorInvalidValue;
    // 16 lanes per node; the first level below the start layer has 2^(log_start-1) nodes
    unsigned threads = 16u << (log_start - 1);
    if (threads < 64) threads = 64;
    if (threads > 1024) threads = 1024;
    merkle_top_kernel<<<1, threads, 0, st>>>(d_layer, log_start, inj);
    return hipGetLastError();
}

__global__ void poseidon2_permute_kernel(uint32_t *states, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = p2f::from_mont(states[i * 16 + k]);
    p2f::permute(s);
#pragma unroll
    for (int k = 0; k < 16; k++) states[i * 16 + k] = p2f::to_mont(s[k]);
}

hipError_t launch_merkle_leaves(hipStream_t st, MerkleLeafSegments sg) {
    if (!sg.n || sg.n > MERKLE_MAX_SEGMENTS) return hipErrorInvalidValue;
    // longest sponge first (workgroups are dispatched in order), the many-row segments behind them.  A segment of few rows
    // runs one row per 16 lanes when its chain at one row per thread (about 9 us per permutation on a wave that has a SIMD
    // to itself) would outlast everything else in the launch (about 7 permutations per ns for the whole GPU).
    for (uint32_t i = 1; i < sg.n; i++)
        for (uint32_t j = i; j > 0 && sg.ncols[j] > sg.ncols[j - 1]; j--) {
            std::swap(sg.cols[j], sg.cols[j - 1]); std::swap(sg.out[j], sg.out[j - 1]);
            std::swap(sg.ncols[j], sg.ncols[j - 1]); std::swap(sg.log_h[j], sg.log_h[j - 1]);
        }
    double work_us = 0;
    for (uint32_t i = 0; i < sg.n; i++) work_us += (double)((size_t)1 << sg.log_h[i]) * ((sg.ncols[i] + 7) / 8) / 7000.0;
    uint32_t blocks = 0;
    for (uint32_t i = 0; i < sg.n; i++) {
        const double chain_us = 9.0 * ((sg.ncols[i] + 7) / 8);
        sg.coop[i] = sg.log_h[i] <= MERKLE_COOP_LOG && chain_us > work_us;
        sg.block0[i] = blocks;
        const size_t threads = ((size_t)1 << sg.log_h[i]) * (sg.coop[i] ? 16 : 1);
        blocks += (uint32_t)((threads + 255) / 256);
    }
    sg.block0[sg.n] = blocks;
    merkle_leaves_kernel<<<blocks, 256, 0, st>>>(sg);
    return hipGetLastError();
}

hipError_t launch_merkle_level(hipStream_t st, const uint32_t *d_prev, const uint32_t *d_inject, uint32_t log_len, uint32_t *d_out) {
    size_t len = (size_t)1 << log_len;
    if (log_len <= MERKLE_COOP_LOG) {  // too few nodes to fill the GPU with one thread each: 16 lanes per node
        unsigned blocks = (unsigned)((len * 16 + 255) / 256);
        merkle_level_coop_kernel<<<blocks, 256, 0, st>>>(d_prev, d_inject, len, d_out);
        return hipGetLastError();
    }
    unsigned blocks = (unsigned)((len + 255) / 256);
    if (d_inject) merkle_level_kernel<true><<<blocks, 256, 0, st>>>(d_prev, d_inject, len, d_out);
    else merkle_level_kernel<false><<<blocks, 256, 0, st>>>(d_prev, d_inject, len, d_out);
    return hipGetLastError();
}

hipError_t launch_poseidon2_permute(hipStream_t st, uint32_t *d_states, size_t n) {
    if (!n) return hipSuccess;
    unsigned blocks = (unsigned)((n + 255) / 256);
    poseidon2_permute_kernel<<<blocks, 256, 0, st>>>(d_states, n);
    return hipGetLastError();
}

}  // namespace dvt
