// K6-K9 — out-of-domain openings, FRI input (reduced openings), FRI folding with
// per-layer Merkle leaves, query gathers and the proof-of-work grind, for gfx950
// (SURVEY.md section 8(a) rows K6-K9).  All vectors are in natural order; FRI folds
// the pair (i, i + M/2), so both operands of every fold are coalesced streams and
// a layer's Merkle leaf i is the pair itself.
#include "kernels.h"
#include "poseidon2_f64.cuh"
#include "f64dot.cuh"

namespace dvt {

__device__ __forceinline__ Fp root_pow24f(const NttTables &t, uint32_t e) {
    return Fp::raw(t.tw_hi[e >> 12]) * Fp::raw(t.tw_lo[e & 4095]);
}
__device__ __forceinline__ Fp4 load_ext(const Fp4 *p) {
    uint4 v = *reinterpret_cast<const uint4 *>(p);
    Fp4 r;
    r.c[0] = Fp::raw(v.x); r.c[1] = Fp::raw(v.y); r.c[2] = Fp::raw(v.z); r.c[3] = Fp::raw(v.w);
    return r;
}
__device__ __forceinline__ void store_ext(Fp4 *p, const Fp4 &v) {
    *reinterpret_cast<uint4 *>(p) = make_uint4(v.c[0].v, v.c[1].v, v.c[2].v, v.c[3].v);
}

// ------------------------------------------------------------------ prefix sums (phi column of K4)
// inclusive scan of `ncols` independent columns of n words; 2048 elements per block
constexpr int SCAN_PER_THREAD = 8, SCAN_BLOCK = 256 * SCAN_PER_THREAD;

__global__ void __launch_bounds__(256) scan_blocks_kernel(uint32_t *data, size_t n, uint32_t *block_sums, size_t nblocks) {
    __shared__ uint32_t wsum[256];
    uint32_t *col = data + (size_t)blockIdx.y * n;
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
    Fp v[SCAN_PER_THREAD];
    Fp run = Fp::zero();
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++) {
        v[k] = base + k < n ? Fp::raw(col[base + k]) : Fp::zero();
        run += v[k];
        v[k] = run;
    }
    wsum[threadIdx.x] = run.v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele over the 256 thread totals
        Fp t = Fp::raw(wsum[threadIdx.x]);
        if ((int)threadIdx.x >= off) t += Fp::raw(wsum[threadIdx.x - off]);
        __syncthreads();
        wsum[threadIdx.x] = t.v;
        __syncthreads();
    }
    Fp excl = threadIdx.x ? Fp::raw(wsum[threadIdx.x - 1]) : Fp::zero();
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++)
        if (base + k < n) col[base + k] = (v[k] + excl).v;
    if (threadIdx.x == 255) block_sums[(size_t)blockIdx.y * nblocks + blockIdx.x] = wsum[255];
}
__global__ void scan_sums_kernel(uint32_t *block_sums, size_t nblocks) {  // one thread per column; nblocks is small
    uint32_t *s = block_sums + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * nblocks;
    Fp run = Fp::zero();
    for (size_t i = 0; i < nblocks; i++) { Fp t = Fp::raw(s[i]); s[i] = run.v; run += t; }
}
__global__ void __launch_bounds__(256) scan_add_kernel(uint32_t *data, size_t n, const uint32_t *block_sums, size_t nblocks) {
    uint32_t *col = data + (size_t)blockIdx.y * n;
    Fp off = Fp::raw(block_sums[(size_t)blockIdx.y * nblocks + blockIdx.x]);
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_PER_THREAD;
#pragma unroll
    for (int k = 0; k < SCAN_PER_THREAD; k++)
        if (base + k < n) col[base + k] = (Fp::raw(col[base + k]) + off).v;
}

hipError_t launch_prefix_sum_columns(hipStream_t st, uint32_t *d_cols, uint32_t ncols, size_t n, uint32_t *d_scratch) {
    size_t nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    dim3 grid((unsigned)nblocks, ncols);
    scan_blocks_kernel<<<grid, 256, 0, st>>>(d_cols, n, d_scratch, nblocks);
    if (nblocks > 1) {
        scan_sums_kernel<<<1, ncols, 0, st>>>(d_scratch, nblocks);
        scan_add_kernel<<<grid, 256, 0, st>>>(d_cols, n, d_scratch, nblocks);
    }
    return hipGetLastError();
}
size_t prefix_sum_scratch_words(uint32_t ncols, size_t n) { return ncols * ((n + SCAN_BLOCK - 1) / SCAN_BLOCK); }

__global__ void __launch_bounds__(256) phi_from_prefix_sums_kernel(const uint32_t *tot, uint32_t *phi, size_t n, Fp n_inv, uint32_t *cum_out) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (r == n - 1 && cum_out) {   // the cumulative sum, straight into the caller's (pinned host) buffer: no copy commands
#pragma unroll
        for (int k = 0; k < 4; k++) cum_out[k] = tot[(size_t)k * n + n - 1];
    }
    const Fp rf = Fp::from_canonical((uint32_t)r) * n_inv;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const Fp c = Fp::raw(tot[(size_t)k * n + n - 1]) * rf;     // r * S / n
        const Fp before = r ? Fp::raw(tot[(size_t)k * n + r - 1]) : Fp::zero();
        phi[(size_t)k * n + r] = (before - c).v;
    }
}
hipError_t launch_phi_from_prefix_sums(hipStream_t st, const uint32_t *d_totals, uint32_t *d_phi, uint32_t log_n, uint32_t *cum_out) {
    const size_t n = (size_t)1 << log_n;
    phi_from_prefix_sums_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(d_totals, d_phi, n, inv(Fp::from_canonical((uint32_t)n)), cum_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K6: openings
// w[i] = omega_N^i / (z - omega_N^i)  (centred canonical words, not Montgomery form: only open_columns reads them)
__global__ void __launch_bounds__(256) open_weights_kernel(Fp4 z, uint32_t log_n, Fp4 *w, NttTables tabs) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << log_n)) return;
    Fp x = root_pow24f(tabs, (uint32_t)i << (24 - log_n));
    Fp4 d = z - x;
    // stored as centred canonical residues (two's-complement words in (-p/2, p/2]): the operand format of the FP64
    // dot products of open_columns_kernel, which turns a word into a double with one conversion
    Fp4 r = inv(d) * x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t c = r.c[k].canonical();
        r.c[k] = Fp::raw(c > P / 2 ? c - P : c);
    }
    store_ext(w + i, r);
}
hipError_t launch_open_weights(hipStream_t st, const NttTables &tabs, Fp4 z, uint32_t log_n, Fp4 *d_w) {
    size_t n = (size_t)1 << log_n;
    open_weights_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(z, log_n, d_w, tabs);
    return hipGetLastError();
}

// partial[rb][col][2] : unscaled sums  sum_i col[i]*w[i]  and  sum_i col[i]*w[i-1]
constexpr int OPEN_CT = 4;  // columns per thread
__global__ void __launch_bounds__(256) open_columns_kernel(const uint32_t *const *cols, uint32_t ncols, uint32_t log_n,
                                                          const Fp4 *w, Fp4 *partial) {
    __shared__ uint32_t red[256 * 4];
    const size_t n = (size_t)1 << log_n;
    const uint32_t c0 = blockIdx.y * OPEN_CT;
    DotAcc acc_l[OPEN_CT][4], acc_n[OPEN_CT][4];
#pragma unroll
    for (int c = 0; c < OPEN_CT; c++)
#pragma unroll
        for (int k = 0; k < 4; k++) acc_l[c][k] = acc_n[c][k] = DotAcc{0.0, 0.0};
    // software-pipelined over the rows of this thread: the words of the next row are in flight while the current one
    // is accumulated (a thread walks its rows with a dependent load -> FMA chain otherwise)
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fp4 wl, wn;
    uint32_t v[OPEN_CT];
    auto fetch = [&](size_t row) {
        wl = load_ext(w + row);
        wn = load_ext(w + ((row + n - 1) & (n - 1)));
#pragma unroll
        for (int c = 0; c < OPEN_CT; c++) v[c] = c0 + c < ncols ? cols[c0 + c][row] : 0u;
    };
    if (i < n) fetch(i);
    for (uint32_t it = 0; i < n; i += stride, it++) {
        double dl[4], dn[4], vlo[OPEN_CT], vhi[OPEN_CT];
#pragma unroll
        for (int k = 0; k < 4; k++) { dl[k] = (double)(int32_t)wl.c[k].v; dn[k] = (double)(int32_t)wn.c[k].v; }
#pragma unroll
        for (int c = 0; c < OPEN_CT; c++) { vlo[c] = (double)(v[c] & 0xffffu); vhi[c] = (double)(v[c] >> 16); }
        if (i + stride < n) fetch(i + stride);
#pragma unroll
        for (int c = 0; c < OPEN_CT; c++)
#pragma unroll
            for (int k = 0; k < 4; k++) { acc_l[c][k].add(dl[k], vlo[c], vhi[c]); acc_n[c][k].add(dn[k], vlo[c], vhi[c]); }
        if ((it & 31) == 31) {
#pragma unroll
            for (int c = 0; c < OPEN_CT; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) { acc_l[c][k].reduce(); acc_n[c][k].reduce(); }
        }
    }
    // block reduction, one (column, point) at a time
    for (int c = 0; c < OPEN_CT; c++)
        for (int pt = 0; pt < 2; pt++) {
#pragma unroll
            for (int k = 0; k < 4; k++) red[k * 256 + threadIdx.x] = (pt ? acc_n[c][k] : acc_l[c][k]).value().v;
            __syncthreads();
            for (int off = 128; off > 0; off >>= 1) {
                if ((int)threadIdx.x < off)
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        red[k * 256 + threadIdx.x] = (Fp::raw(red[k * 256 + threadIdx.x]) + Fp::raw(red[k * 256 + threadIdx.x + off])).v;
                __syncthreads();
            }
            if (threadIdx.x == 0 && c0 + c < ncols) {
                Fp4 r;
                for (int k = 0; k < 4; k++) r.c[k] = Fp::raw(red[k * 256]);
                store_ext(partial + ((size_t)blockIdx.x * ncols + c0 + c) * 2 + pt, r);
            }
            __syncthreads();
        }
}
// 16 lanes per value: lane j sums the row blocks j, j + 16, ..., then a 4-step butterfly over the 16 lanes
__global__ void __launch_bounds__(256) open_reduce_kernel(const Fp4 *partial, uint32_t nrb, uint32_t nvals, Fp4 *out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, v = t >> 4, j = t & 15;
    Fp4 acc = Fp4::zero();
    if (v < nvals)
        for (uint32_t rb = j; rb < nrb; rb += 16) acc += load_ext(partial + (size_t)rb * nvals + v);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 4; k++) acc.c[k] = acc.c[k] + Fp::raw(__shfl_xor(acc.c[k].v, off, 16));
    if (v < nvals && j == 0) store_ext(out + v, acc);
}
uint32_t open_row_blocks(uint32_t log_n) {  // <= OPEN_MAX_ROW_BLOCKS (kernels.h): callers size d_partial with it
    size_t n = (size_t)1 << log_n;
    size_t rb = (n + 255) / 256;
    return (uint32_t)(rb > OPEN_MAX_ROW_BLOCKS ? OPEN_MAX_ROW_BLOCKS : rb);
}
// d_out[col][2]; d_partial must hold open_row_blocks(log_n) * ncols * 2 ext elements
hipError_t launch_open_columns(hipStream_t st, const uint32_t *const *d_cols, uint32_t ncols, uint32_t log_n, const Fp4 *d_w,
                               Fp4 *d_partial, Fp4 *d_out) {
    if (!ncols) return hipSuccess;
    uint32_t rb = open_row_blocks(log_n);
    dim3 grid(rb, (ncols + OPEN_CT - 1) / OPEN_CT);
    open_columns_kernel<<<grid, 256, 0, st>>>(d_cols, ncols, log_n, d_w, d_partial);
    open_reduce_kernel<<<(ncols * 2 * 16 + 255) / 256, 256, 0, st>>>(d_partial, rb, ncols * 2, d_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K7: reduced openings (FRI input)
struct ReducedArgs {
    const uint32_t *const *cols;  // LDE columns of this height: two-point columns first
    uint32_t n_two, n_all, log_m;
    const double *alpha_pows;     // [n_all][4]: alpha^c as centred canonical residues (exact doubles)
    Fp4 sz_all, sz_two;  // sum_c alpha^c p_c(zeta) over all columns / p_c(zeta*omega) over two-point columns
    Fp4 zeta, zeta_next, alpha_shift;  // alpha^{n_all}
    Fp4 *out;
    NttTables tabs;
};
__global__ void __launch_bounds__(256) reduced_opening_kernel(ReducedArgs a) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t m = (size_t)1 << a.log_m;
    if (i >= m) return;
    // s = sum_c alpha^c * col_c[i] in F_p^4: four exact FP64 dot products (DotAcc above); alpha^c is wave-uniform
    DotAcc acc[4] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
    Fp4 s_two = Fp4::zero();
    // columns [from, to): eight loads in flight per step (a short-and-wide table — a precompile chip with a few hundred rows and
    // thousands of columns — has too few rows to hide a pointer fetch + a load per column behind other waves)
    auto sweep = [&](uint32_t from, uint32_t to) {
        uint32_t c = from;
        for (; c + 8 <= to; c += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = a.cols[c + u][i];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const double vlo = (double)(v[u] & 0xffffu), vhi = (double)(v[u] >> 16);
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k].add(a.alpha_pows[4 * (c + u) + k], vlo, vhi);
            }
            // (at most 16 terms since the last reduction of this sweep: below the 32-term bound of DotAcc)
            if (((c - from) & 8) != 0)
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k].reduce();
        }
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k].reduce();
        for (; c < to; c++) {
            const uint32_t v = a.cols[c][i];
            const double vlo = (double)(v & 0xffffu), vhi = (double)(v >> 16);
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k].add(a.alpha_pows[4 * c + k], vlo, vhi);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k].reduce();
    };
    sweep(0, a.n_two);
#pragma unroll
    for (int k = 0; k < 4; k++) s_two.c[k] = acc[k].value();
    sweep(a.n_two, a.n_all);
    Fp4 s;
#pragma unroll
    for (int k = 0; k < 4; k++) s.c[k] = acc[k].value();
    if (a.n_two == a.n_all) s_two = s;
    Fp x = Fp::raw(a.tabs.sh_lo[1]) * root_pow24f(a.tabs, (uint32_t)i << (24 - a.log_m));
    Fp4 r = (s - a.sz_all) * inv(Fp4::from_base(x) - a.zeta);
    if (a.n_two) r += a.alpha_shift * ((s_two - a.sz_two) * inv(Fp4::from_base(x) - a.zeta_next));
    store_ext(a.out + i, r);
}
hipError_t launch_reduced_opening(hipStream_t st, const NttTables &tabs, const uint32_t *const *d_cols, uint32_t n_two,
                                  uint32_t n_all, uint32_t log_m, const double *d_alpha_pows_f64, Fp4 sz_all, Fp4 sz_two, Fp4 zeta,
                                  Fp4 zeta_next, Fp4 alpha_shift, Fp4 *d_out) {
    ReducedArgs a{d_cols, n_two, n_all, log_m, d_alpha_pows_f64, sz_all, sz_two, zeta, zeta_next, alpha_shift, d_out, tabs};
    size_t m = (size_t)1 << log_m;
    reduced_opening_kernel<<<(unsigned)((m + 255) / 256), 256, 0, st>>>(a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K8: fold + layer leaves
// out[i] = (v[i] + v[i+h])/2 + beta * (v[i] - v[i+h]) / (2 w_M^i)  (+ ro[i])
// (beta comes from device memory when beta_dev is given: the challenge of a FRI round is derived on the device)
__global__ void __launch_bounds__(256) fri_fold_kernel(const Fp4 *v, Fp4 *out, const Fp4 *ro, Fp4 beta, const Fp4 *beta_dev,
                                                      uint32_t log_m, Fp inv2, NttTables tabs) {
    const size_t half = (size_t)1 << (log_m - 1);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    if (beta_dev) beta = load_ext(beta_dev);
    Fp4 lo = load_ext(v + i), hi = load_ext(v + i + half);
    Fp xinv = i ? root_pow24f(tabs, (uint32_t)(2 * half - i) << (24 - log_m)) : Fp::one();
    Fp4 r = (lo + hi) * inv2 + beta * ((lo - hi) * (inv2 * xinv));
    if (ro) r += load_ext(ro + i);
    store_ext(out + i, r);
}
hipError_t launch_fri_fold(hipStream_t st, const NttTables &tabs, const Fp4 *d_v, Fp4 *d_out, const Fp4 *d_ro, Fp4 beta,
                           uint32_t log_m, const Fp4 *d_beta) {
    size_t half = (size_t)1 << (log_m - 1);
    Fp inv2 = inv(Fp::two());
    fri_fold_kernel<<<(unsigned)((half + 255) / 256), 256, 0, st>>>(d_v, d_out, d_ro, beta, d_beta, log_m, inv2, tabs);
    return hipGetLastError();
}
// One round of the FRI commit phase of the transcript, on the device (challenger.h semantics with an empty input
// buffer: observe(root) fills the rate exactly, one duplexing, sample_ext pops the squeezed rate from the back):
//   state[0..8) <- root;  state <- Poseidon2(state);  beta = (state[7], state[6], state[5], state[4]).
// Removes a device->host->device round trip per FRI round.  The root is also copied to roots_out for the proof.
__global__ void __launch_bounds__(64) fri_challenge_kernel(const uint32_t *root, uint32_t *state, Fp4 *beta_out, uint32_t *root_out) {
    const uint32_t e = threadIdx.x & 15;
    const p2f::CoopConsts k = p2f::coop_consts(e);
    const uint32_t w_in = e < 8 ? root[e] : state[e];
    const double s = p2f::coop_permute(p2f::from_mont(w_in), k);
    const uint32_t w = p2f::to_mont(s);
    if (threadIdx.x < 16) {
        state[e] = w;
        if (e < 8) root_out[e] = w_in;
        if (e >= 4 && e < 8) reinterpret_cast<uint32_t *>(beta_out)[7 - e] = w;
    }
}
hipError_t launch_fri_challenge(hipStream_t st, const uint32_t *d_root, uint32_t *d_state, Fp4 *d_beta_out, uint32_t *d_root_out) {
    fri_challenge_kernel<<<1, 64, 0, st>>>(d_root, d_state, d_beta_out, d_root_out);
    return hipGetLastError();
}
// digest[i] = sponge(v[i] || v[i+half])  (8 words = one permutation)
__global__ void __launch_bounds__(256) fri_leaves_kernel(const Fp4 *v, size_t half, uint32_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    Fp4 lo = load_ext(v + i), hi = load_ext(v + i + half);
    double s[16];
#pragma unroll
    for (int k = 0; k < 4; k++) { s[k] = p2f::from_mont(lo.c[k].v); s[4 + k] = p2f::from_mont(hi.c[k].v); }
#pragma unroll
    for (int k = 8; k < 16; k++) s[k] = 0.0;
    p2f::permute(s);
    uint4 *o = reinterpret_cast<uint4 *>(out + i * 8);
    o[0] = make_uint4(p2f::to_mont(s[0]), p2f::to_mont(s[1]), p2f::to_mont(s[2]), p2f::to_mont(s[3]));
    o[1] = make_uint4(p2f::to_mont(s[4]), p2f::to_mont(s[5]), p2f::to_mont(s[6]), p2f::to_mont(s[7]));
}
hipError_t launch_fri_leaves(hipStream_t st, const Fp4 *d_v, uint32_t log_m, uint32_t *d_digests) {
    size_t half = (size_t)1 << (log_m - 1);
    fri_leaves_kernel<<<(unsigned)((half + 255) / 256), 256, 0, st>>>(d_v, half, d_digests);
    return hipGetLastError();
}

// ------------------------------------------------------------------ K9: gathers + grind
// out[q][c] = cols[c][idx[q] mod 2^log_h[c]]
__global__ void gather_rows_kernel(const uint32_t *const *cols, const uint32_t *log_h, uint32_t ncols, const uint32_t *idx,
                                   uint32_t nq, uint32_t *out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * ncols) return;
    uint32_t q = t / ncols, c = t % ncols;
    out[t] = cols[c][idx[q] & ((1u << log_h[c]) - 1)];
}
hipError_t launch_gather_rows(hipStream_t st, const uint32_t *const *d_cols, const uint32_t *d_log_h, uint32_t ncols,
                              const uint32_t *d_idx, uint32_t nq, uint32_t *d_out) {
    if (!ncols || !nq) return hipSuccess;
    gather_rows_kernel<<<(nq * ncols + 255) / 256, 256, 0, st>>>(d_cols, d_log_h, ncols, d_idx, nq, d_out);
    return hipGetLastError();
}
// out[q][level][8] = sibling digest of leaf idx[q] at each level of a natural-order tree of 2^log_h leaves
__global__ void gather_paths_kernel(const uint32_t *digests, uint32_t log_h, const uint32_t *idx, uint32_t nq, uint32_t *out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nq * log_h * 8) return;
    uint32_t wd = t & 7, lvl = (t >> 3) % log_h, q = (t >> 3) / log_h;
    uint32_t s = log_h - lvl;  // layer of 2^s nodes
    size_t off = ((size_t)2 << log_h) - ((size_t)2 << s);
    uint32_t j = idx[q] & ((1u << s) - 1);
    out[t] = digests[(off + (j ^ (1u << (s - 1)))) * 8 + wd];
}
hipError_t launch_gather_paths(hipStream_t st, const uint32_t *d_digests, uint32_t log_h, const uint32_t *d_idx, uint32_t nq,
                               uint32_t *d_out) {
    if (!log_h || !nq) return hipSuccess;
    gather_paths_kernel<<<(nq * log_h * 8 + 255) / 256, 256, 0, st>>>(d_digests, log_h, d_idx, nq, d_out);
    return hipGetLastError();
}
// out[q] = v[(idx[q] mod M) xor M/2]
__global__ void gather_siblings_kernel(const Fp4 *v, uint32_t log_m, const uint32_t *idx, uint32_t nq, Fp4 *out) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint32_t j = idx[q] & ((1u << log_m) - 1);
    store_ext(out + q, load_ext(v + (j ^ (1u << (log_m - 1)))));
}
hipError_t launch_gather_siblings(hipStream_t st, const Fp4 *d_v, uint32_t log_m, const uint32_t *d_idx, uint32_t nq, Fp4 *d_out) {
    gather_siblings_kernel<<<(nq + 255) / 256, 256, 0, st>>>(d_v, log_m, d_idx, nq, d_out);
    return hipGetLastError();
}

// proof-of-work: smallest w in [base, base+count) with (perm(state | s[pos]=w)[7] & mask) == 0
struct GrindArgs { uint32_t state[16]; uint32_t pos, mask, base, count; };
__global__ void __launch_bounds__(256) pow_grind_kernel(GrindArgs a, uint32_t *found) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.count) return;
    uint32_t w = a.base + t;
    if (w >= P) return;
    double s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = p2f::from_mont(a.state[k]);
#pragma unroll
    for (int k = 0; k < 8; k++)
        if ((uint32_t)k == a.pos) s[k] = p2f::from_canonical(w);
    p2f::permute(s);
    if ((p2f::to_canonical(s[7]) & a.mask) == 0) atomicMin(found, w);
}
hipError_t launch_pow_grind(hipStream_t st, const uint32_t state[16], uint32_t pos, uint32_t bits, uint32_t base, uint32_t count,
                            uint32_t *d_found) {
    GrindArgs a;
    for (int k = 0; k < 16; k++) a.state[k] = state[k];
    a.pos = pos; a.mask = (1u << bits) - 1; a.base = base; a.count = count;
    pow_grind_kernel<<<(count + 255) / 256, 256, 0, st>>>(a, d_found);
    return hipGetLastError();
}

}  // namespace dvt
