// K1 — column-batched BabyBear coset LDE for gfx950.
//
// Data layout: column-major device field arrays; every column is one
// contiguous polynomial of N = 2^n evaluations in natural order.  Output: 2N
// evaluations on shift * <w_2N>, natural order.
//
// Decomposition (four-step on both transforms, fused in the middle), with
// N = N1*N2, N2 = 2^min(n,12), M2 = 2*N2, M = 2N = N1*M2:
//   P1  ntt_strided<inverse>   for every n2: N1-point inverse DFT over the
//                              stride-N2 samples (its twiddle w_N^(-n2*k1) is applied by P2 on load).
//                              LDS tile = N1 x 2^b (2^b consecutive n2 -> 64..128-B
//                              coalesced HBM segments).               [n > 12 only]
//   P2  lde_block              for every k1: N2-point inverse DFT of the contiguous
//                              block (LDS), scale by shift^k / N, zero-pad to M2,
//                              M2-point forward DFT, times w_M^(j2*k1); N2 words in,
//                              M2 words out, both contiguous.
//   P3  ntt_strided<forward>   for every j2: N1-point forward DFT over the stride-M2
//                              samples; in place on the output.       [n > 12 only]
// HBM traffic per column: P1 4N+4N, P2 4N+8N, P3 8N+8N = 36N bytes against the
// compulsory 12N (read N, write 2N); for n <= 12 only P2 runs: exactly 12N.
//
// Inverse transforms are computed as forward DFTs with the output index negated
// (iDFT(x)[k] = DFT(x)[-k]/N), so one forward twiddle table per kernel suffices.
// Local transforms: DIF (natural in, bit-reversed out) / DIT (bit-reversed in,
// natural out) in LDS with an LDS-resident twiddle table; the bit reversals are
// absorbed into LDS addressing, never into HBM addressing.  Butterflies run in
// registers, three stages (radix 8) per LDS round trip: a thread gathers the 8
// words of a group, does 12 butterflies, scatters them back — one barrier per three
// stages instead of per stage.
#include "kernels.h"
#include "poseidon2_f64.cuh"

namespace dvt {

// Arithmetic: the butterflies run on the FP64 pipe (see poseidon2_f64.cuh for the formulation: exact integers in
// doubles, a modular product = 6 full-rate operations, additions unreduced).  Measured on the integer version of these
// kernels: the time went into VALU issue slots (3-instruction modular add / sub, the non-multiply half of the Montgomery
// product), not into the multiplier and not into LDS.  LDS tiles hold 32-bit signed residues in (-p, p); twiddles are
// CANONICAL residues (the data words are Montgomery words: x~ * w mod p is the Montgomery word of x * w), one
// conversion per LDS word on the way in, one reduction + conversion on the way out.
using Lw = int32_t;
__device__ __forceinline__ Lw to_lds(double x) { return (Lw)x; }                    // |x| < 2^31
__device__ __forceinline__ uint32_t lds_to_word(Lw v) { return v < 0 ? (uint32_t)(v + (Lw)P) : (uint32_t)v; }
// canonical residue of a Montgomery-form field element as a centred double (per-thread twiddle seeds)
__device__ __forceinline__ double centred(Fp x) { return p2f::mm((double)x.v, p2f::MONT_RINV); }


__device__ __forceinline__ Fp root_pow24(const NttTables &t, uint32_t e) {  // Omega^e, Omega of order 2^24
    return Fp::raw(t.tw_hi[e >> 12]) * Fp::raw(t.tw_lo[e & 4095]);
}
__device__ __forceinline__ Fp shift_pow(const NttTables &t, uint32_t k) {  // 31^k, k < 2^23
    return Fp::raw(t.sh_hi[k >> 12]) * Fp::raw(t.sh_lo[k & 4095]);
}
__device__ __forceinline__ uint32_t bitrev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// ---- LDS addressing ---------------------------------------------------------------------------
// Every pass below walks the tile with power-of-two strides, which on a 32-bank LDS (ds_read_b32 / ds_write_b32:
// bank = word address mod 32, 32 lanes per group) means 4..32 lanes per bank for the late DIF / early DIT stages
// and for the strided twiddle reads (measured on lde_block: 76 % of all LDS cycles were conflict cycles).
// The arrays are therefore stored XOR-swizzled: logical index i lives at  i ^ ((i >> 5) & 31) ^ ((i >> 10) & 31),
// i.e. index bits 5..9 and 10..14 are folded onto the bank bits.  The map is linear over GF(2), so for index parts
// with disjoint bits swz(a | b) = swz(a) ^ swz(b): one XOR per access once the parts are swizzled.
// A 32-lane group is conflict-free iff the index bits that vary across its lanes map onto the 5 bank bits
// bijectively: consecutive indices do (bits 0..4), a stride 2^k does (bits k..k+4 -> banks k..4, 0..k-1), and the
// radix passes pick WHICH block bits vary inside a lane group accordingly (lane_block below).
__device__ __forceinline__ uint32_t swz(uint32_t i) { return i ^ ((i >> 5) & 31) ^ ((i >> 10) & 31); }

// Work item w of a radix-2^R pass = (low: ell in-group offset bits, blk: which group of 2^R * 2^ell elements).
// Element index bits: [0, ell) offset, [ell, ell+R) position inside the radix group (a per-instruction constant),
// [ell+R, ..) block.  For ell < 5 the lanes of a group must also spread over 5 - ell block bits; bank bit k in
// [ell, 5) is fed (through the swizzle) by index bit k + 5 = block bit k + 5 - ell - R, so the lane-varying block
// bits are [5 - R, 10 - ell - R): the block number is the work-item number with its low bits rotated up by 5 - R.
template <int R>
__device__ __forceinline__ uint32_t lane_block(uint32_t bw, uint32_t ell, uint32_t blk_bits) {
    if (ell >= 5) return bw;
    const uint32_t v = 5 - ell, skip = 5 - R;
    if (blk_bits < v + skip) return bw;  // tiny transform: not enough blocks to spread over, conflicts are harmless there
    return ((bw & ((1u << v) - 1)) << skip) | ((bw >> v) & ((1u << skip) - 1)) | ((bw >> (v + skip)) << (v + skip));
}

// ---- fused radix-2^R passes over an LDS array ------------------------------------------------
// Element (i, c) of a [2^L][2^log_cols] tile has logical index (i << log_cols) | c; the transform runs
// over i for every c.  tw[swz(e << tw_shift)] = w_{2^L}^e.
// DIF stages s .. s+R-1 (stage t pairs distance 2^(L-1-t), twiddle exponent (index mod half) << t).
// Geometry: the runtime arguments (L, s, tw_shift, log_cols) serve every size; with SL >= 0 the template arguments replace
// them, every swizzled offset below folds to a literal and only the block / row part of an address is computed at run
// time (measured: with runtime geometry the address arithmetic of a pass costs as many VALU slots as its butterflies).
template <int R, int SL = -1, int SS = 0, int STW = 0, int SLC = 0, class TW = Lw>
__device__ __forceinline__ void dif_pass(Lw *sm, const TW *tw, uint32_t L_rt, uint32_t s_rt, uint32_t tw_shift_rt, uint32_t log_cols_rt,
                                         uint32_t tid, uint32_t nt) {
    const uint32_t L = SL >= 0 ? (uint32_t)SL : L_rt, s = SL >= 0 ? (uint32_t)SS : s_rt;
    const uint32_t tw_shift = SL >= 0 ? (uint32_t)STW : tw_shift_rt, log_cols = SL >= 0 ? (uint32_t)SLC : log_cols_rt;
    constexpr uint32_t G = 1u << R;
    const uint32_t lh_last = L - s - R, cmask = (1u << log_cols) - 1;
    const uint32_t ell = lh_last + log_cols, blk_bits = s;
    const uint32_t work = (1u << (L - R)) << log_cols;
    uint32_t joff[G];  // swizzled offset of the j-th element of a group
#pragma unroll
    for (uint32_t j = 0; j < G; j++) joff[j] = swz((j << lh_last) << log_cols);
    for (uint32_t w = tid; w < work; w += nt) {
        const uint32_t low = w & ((1u << ell) - 1);
        const uint32_t c = low & cmask, r = low >> log_cols;
        const uint32_t blk = lane_block<R>(w >> ell, ell, blk_bits);
        const uint32_t base = swz((((blk << (L - s)) | r) << log_cols) | c);
        double v[G];
#pragma unroll
        for (uint32_t j = 0; j < G; j++) v[j] = (double)sm[base ^ joff[j]];
#pragma unroll
        for (uint32_t t = 0; t < (uint32_t)R; t++) {
            const uint32_t dist = G >> (t + 1);
            const uint32_t twr = swz((r << (s + t)) << tw_shift);
#pragma unroll
            for (uint32_t j = 0; j < G; j++) {
                if (j & dist) continue;
                const uint32_t twj = swz((((j & (dist - 1)) << lh_last) << (s + t)) << tw_shift);
                const double a = v[j], b = v[j + dist];
                v[j] = a + b;                                            // grows by one bit per stage: < 2^34 after three
                v[j + dist] = p2f::mm(a - b, (double)tw[twr ^ twj]);     // |a - b| < 2^35, |w| < 2^31: reduced again
            }
        }
        // odd positions left the last stage through a product (already reduced); even ones are sums
#pragma unroll
        for (uint32_t j = 0; j < G; j++) sm[base ^ joff[j]] = to_lds((j & 1) ? v[j] : p2f::red(v[j]));
    }
}
// DIT stages s .. s+R-1 (stage t pairs distance 2^t, twiddle exponent (index mod 2^t) << (L-1-t)).
template <int R, int SL = -1, int SS = 0, int STW = 0>
__device__ __forceinline__ void dit_pass(Lw *sm, const Lw *tw, uint32_t L_rt, uint32_t s_rt, uint32_t tw_shift_rt, uint32_t tid, uint32_t nt) {
    const uint32_t L = SL >= 0 ? (uint32_t)SL : L_rt, s = SL >= 0 ? (uint32_t)SS : s_rt;
    const uint32_t tw_shift = SL >= 0 ? (uint32_t)STW : tw_shift_rt;
    constexpr uint32_t G = 1u << R;
    const uint32_t work = 1u << (L - R);
    const uint32_t blk_bits = L - R - s;
    uint32_t joff[G];
#pragma unroll
    for (uint32_t j = 0; j < G; j++) joff[j] = swz(j << s);
    for (uint32_t g = tid; g < work; g += nt) {
        const uint32_t r = g & ((1u << s) - 1);
        const uint32_t blk = lane_block<R>(g >> s, s, blk_bits);
        const uint32_t base = swz((blk << (s + R)) | r);
        double v[G];
#pragma unroll
        for (uint32_t j = 0; j < G; j++) v[j] = (double)sm[base ^ joff[j]];
#pragma unroll
        for (uint32_t t = 0; t < (uint32_t)R; t++) {
            const uint32_t dist = 1u << t;
            const uint32_t twr = swz((r << (L - 1 - s - t)) << tw_shift);
#pragma unroll
            for (uint32_t j = 0; j < G; j++) {
                if (j & dist) continue;
                const uint32_t twj = swz((((j & (dist - 1)) << s) << (L - 1 - s - t)) << tw_shift);
                const double a = v[j], b = p2f::mm(v[j + dist], (double)tw[twr ^ twj]);   // |v| < 2^33
                v[j] = a + b;
                v[j + dist] = a - b;
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < G; j++) sm[base ^ joff[j]] = to_lds(p2f::red(v[j]));
    }
}
template <bool DIF, class TW = Lw>
__device__ __forceinline__ void run_stages(Lw *sm, const TW *tw, uint32_t L, uint32_t first, uint32_t count, uint32_t tw_shift,
                                           uint32_t log_cols, uint32_t tid, uint32_t nt) {
    uint32_t s = first, left = count;
    while (left) {
        if (left >= 3) {
            if constexpr (DIF) dif_pass<3, -1, 0, 0, 0, TW>(sm, tw, L, s, tw_shift, log_cols, tid, nt); else dit_pass<3>(sm, tw, L, s, tw_shift, tid, nt);
            s += 3; left -= 3;
        } else if (left == 2) {
            if constexpr (DIF) dif_pass<2, -1, 0, 0, 0, TW>(sm, tw, L, s, tw_shift, log_cols, tid, nt); else dit_pass<2>(sm, tw, L, s, tw_shift, tid, nt);
            s += 2; left -= 2;
        } else {
            if constexpr (DIF) dif_pass<1, -1, 0, 0, 0, TW>(sm, tw, L, s, tw_shift, log_cols, tid, nt); else dit_pass<1>(sm, tw, L, s, tw_shift, tid, nt);
            s += 1; left -= 1;
        }
        __syncthreads();
    }
}

// the same with compile-time geometry: stages FIRST .. FIRST+COUNT-1 of a 2^L-point transform
template <bool DIF, int L, int FIRST, int COUNT, int TWS, int LC, class TW = Lw>
__device__ __forceinline__ void run_stages_static(Lw *sm, const TW *tw, uint32_t tid, uint32_t nt) {
    if constexpr (COUNT > 0) {
        constexpr int R = COUNT >= 3 ? 3 : COUNT;
        if constexpr (DIF) dif_pass<R, L, FIRST, TWS, LC, TW>(sm, tw, 0, 0, 0, 0, tid, nt);
        else dit_pass<R, L, FIRST, TWS>(sm, tw, 0, 0, 0, tid, nt);
        __syncthreads();
        run_stages_static<DIF, L, FIRST + R, COUNT - R, TWS, LC, TW>(sm, tw, tid, nt);
    }
}

// ---------------------------------------------------------------- P1 / P3
// grid.x = row_stride >> log_cols (tiles along the contiguous axis), grid.y = column
template <bool INVERSE>
__global__ void __launch_bounds__(1024) ntt_strided_kernel(const uint32_t *src, uint32_t *data, size_t col_stride, uint32_t log_rows,
                                                         uint32_t row_stride, uint32_t log_cols, uint32_t log_n, NttTables tabs) {
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t rows = 1u << log_rows, cols = 1u << log_cols, cmask = cols - 1;
    const uint32_t tile_elems = rows << log_cols;
    Lw *sm = reinterpret_cast<Lw *>(lds);
    double *tw = reinterpret_cast<double *>(sm + tile_elems);  // rows/2 entries: w_rows^e (canonical) as doubles: no conversion per use
    uint32_t *col = data + (size_t)blockIdx.y * col_stride;
    const uint32_t *scol = src + (size_t)blockIdx.y * col_stride;
    const uint32_t c0 = blockIdx.x << log_cols;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;

    for (uint32_t e = tid; e < rows / 2; e += nt) tw[swz(e)] = (double)(Lw)tabs.lde_tw_c[rows / 2 - 1 + e];
    for (uint32_t idx = tid; idx < tile_elems; idx += nt) {
        uint32_t r = idx >> log_cols, c = idx & cmask;
        sm[swz(idx)] = (Lw)scol[(size_t)r * row_stride + c0 + c];      // a word in [0, p) is a valid residue in (-p, p)
    }
    __syncthreads();
    // forward DIF over the row index (the two tile shapes of 2^21- and 2^22-row traces with literal geometry)
    if (log_rows == 9 && log_cols == 4) run_stages_static<true, 9, 0, 9, 0, 4, double>(sm, tw, tid, nt);
    else if (log_rows == 10 && log_cols == 4) run_stages_static<true, 10, 0, 10, 0, 4, double>(sm, tw, tid, nt);
    else run_stages<true, double>(sm, tw, log_rows, 0, log_rows, 0, log_cols, tid, nt);
    for (uint32_t idx = tid; idx < tile_elems; idx += nt) {
        uint32_t q = idx >> log_cols, c = idx & cmask;
        uint32_t kf = bitrev(q, log_rows);
        const Lw val = sm[swz(idx)];
        // inverse: DFT with the output index negated; the four-step twiddle w_N^-(n2 k1) that belongs here is applied
        // by lde_block when it loads row k1 (there it is a geometric sequence per thread, here it would be a gather)
        const uint32_t orow = INVERSE ? (rows - kf) & (rows - 1) : kf;
        col[(size_t)orow * row_stride + c0 + c] = lds_to_word(val);
    }
}


// ---------------------------------------------------------------- P1 / P3, four tile columns per thread
// The 2^9-row x 16-column tile of a 2^21-row trace (and of its 2^22-row LDE).  One work item = one radix-8 group of FOUR
// adjacent columns: the 16 columns of a tile row are consecutive n2, so they share every twiddle (a twiddle depends on the
// row only), one address computation and one twiddle fetch serve four butterflies, and LDS moves 16 bytes per lane
// (ds_read_b128 / ds_write_b128: twice the bytes per LDS cycle of the 4-byte forms).  256 threads = the 256 work items
// of a pass.
// LDS image: 16-byte units, unit (r, c4) = (r << 2) | c4 at physical unit u ^ (((u >> 5) & 3) << 2).  A b128 access is
// served in 16-lane groups that must hit 16 distinct units mod 16: the lanes of a group take all 16 values of the low
// four work-item bits; in the passes with 2^lh >= 4 rows per group element those are (c4, two row bits) = unit bits
// 0..3 directly, in the last pass (lh = 0: unit = block << 5 | j << 2 | c4) the XOR brings block bits 0..1 down.
constexpr uint32_t V4_L = 9, V4_ROWS = 1u << V4_L, V4_UNITS = V4_ROWS * 4;
__device__ __forceinline__ uint32_t swzu(uint32_t u) { return u ^ (((u >> 5) & 3u) << 2); }

template <int S>   // DIF stages S, S+1, S+2 of the 2^9-point transform over the rows
__device__ __forceinline__ void dif3_v4(int4 *sm, const double *tw, uint32_t w) {
    constexpr uint32_t lh = V4_L - S - 3;
    const uint32_t c4 = w & 3, rlow = (w >> 2) & ((1u << lh) - 1), blk = w >> (2 + lh);
    const uint32_t r0 = (blk << (V4_L - S)) | rlow;
    double v[8][4];
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
        const int4 x = sm[swzu((((j << lh) | r0) << 2) | c4)];
        v[j][0] = (double)x.x; v[j][1] = (double)x.y; v[j][2] = (double)x.z; v[j][3] = (double)x.w;
    }
#pragma unroll
    for (uint32_t t = 0; t < 3; t++) {
        const uint32_t dist = 4u >> t;
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            if (j & dist) continue;
            // twiddle exponent: (index mod half) << stage, index mod half = (j mod dist) * 2^lh + rlow
            const double wv = tw[((((j & (dist - 1)) << lh) | rlow) << (S + t)) & (V4_ROWS / 2 - 1)];
            const double wp = wv * p2f::PINV;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double a = v[j][k], b = v[j + dist][k];
                v[j][k] = a + b;                                   // < 2^34 after three stages
                v[j + dist][k] = p2f::mm_pre(a - b, wv, wp);       // |a - b| < 2^35: reduced again
            }
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
        int4 x;
        // odd positions left the last stage through a product (already reduced); even ones are sums
        x.x = to_lds((j & 1) ? v[j][0] : p2f::red(v[j][0])); x.y = to_lds((j & 1) ? v[j][1] : p2f::red(v[j][1]));
        x.z = to_lds((j & 1) ? v[j][2] : p2f::red(v[j][2])); x.w = to_lds((j & 1) ? v[j][3] : p2f::red(v[j][3]));
        sm[swzu((((j << lh) | r0) << 2) | c4)] = x;
    }
}

// grid.x = row_stride / 16 (tiles along the contiguous axis), grid.y = column; 256 threads; rows = 2^9
template <bool INVERSE>
__global__ void __launch_bounds__(256) ntt_strided_v4_kernel(const uint32_t *src, uint32_t *data, size_t col_stride, uint32_t row_stride, NttTables tabs) {
    __shared__ int4 sm[V4_UNITS];
    __shared__ double tw[V4_ROWS / 2];
    uint32_t *col = data + (size_t)blockIdx.y * col_stride;
    const uint32_t *scol = src + (size_t)blockIdx.y * col_stride;
    const uint32_t c0 = blockIdx.x << 4, tid = threadIdx.x;
    tw[tid] = (double)(Lw)tabs.lde_tw_c[V4_ROWS / 2 - 1 + tid];       // w_512^e, e < 256 (canonical, centred)
#pragma unroll
    for (uint32_t k = 0; k < V4_UNITS / 256; k++) {
        const uint32_t u = tid + 256 * k, r = u >> 2, c4 = u & 3;
        const uint4 x = *reinterpret_cast<const uint4 *>(scol + (size_t)r * row_stride + c0 + 4 * c4);
        sm[swzu(u)] = make_int4((int)x.x, (int)x.y, (int)x.z, (int)x.w);   // a word in [0, p) is a valid residue in (-p, p)
    }
    __syncthreads();
    dif3_v4<0>(sm, tw, tid);
    __syncthreads();
    dif3_v4<3>(sm, tw, tid);
    __syncthreads();
    dif3_v4<6>(sm, tw, tid);
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < V4_UNITS / 256; k++) {
        const uint32_t u = tid + 256 * k, q = u >> 2, c4 = u & 3;
        const uint32_t kf = bitrev(q, V4_L);
        const uint32_t orow = INVERSE ? (V4_ROWS - kf) & (V4_ROWS - 1) : kf;    // inverse: DFT with the output index negated
        const int4 x = sm[swzu(u)];
        *reinterpret_cast<uint4 *>(col + (size_t)orow * row_stride + c0 + 4 * c4) =
            make_uint4(lds_to_word(x.x), lds_to_word(x.y), lds_to_word(x.z), lds_to_word(x.w));
    }
}

// ---------------------------------------------------------------- P2
// grid.x = N1 (block index k1), grid.y = column.  N2 <= 4096; N2 / 16 <= blockDim.x <= M2.
__global__ void __launch_bounds__(1024) lde_block_kernel(const uint32_t *in, uint32_t *out, uint32_t log_n, uint32_t log_n1,
                                                       uint32_t shift_mode, NttTables tabs) {
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t log_n2 = log_n - log_n1, n2 = 1u << log_n2, m2 = n2 * 2, log_m2 = log_n2 + 1;
    const uint32_t log_m = log_n + 1;
    Lw *sm = reinterpret_cast<Lw *>(lds);
    Lw *tw = sm + m2;  // n2 entries: w_M2^e, e < M2/2 (canonical)
    const uint32_t k1 = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const uint32_t *src = in + ((size_t)blockIdx.y << log_n) + ((size_t)k1 << log_n2);
    uint32_t *dst = out + ((size_t)blockIdx.y << log_m) + ((size_t)k1 << log_m2);

    // twiddles w_M2^e and the position-dependent part of the coefficient scaling come from tables built once per
    // prover (coalesced reads) instead of two table gathers and a product per element
    const uint32_t *W = tabs.lde_tw_c + (n2 - 1);
    const uint32_t *T = tabs.lde_scale_c + ((size_t)(shift_mode * LDE_MAX_LOG + log_n) << 12);
    for (uint32_t e = tid; e < n2; e += nt) tw[swz(e)] = (Lw)W[e];
    if (log_n1 == 0 || k1 == 0) {
        for (uint32_t i = tid; i < n2; i += nt) sm[swz(i)] = (Lw)src[i];
    } else if (tid < n2) {
        // four-step twiddle between P1 and this transform: x[i] *= w_N^-(i k1), i = tid + t nt (i k1 < N):
        // a geometric sequence per thread
        const uint32_t nn = 1u << log_n;
        double cur = tid ? centred(root_pow24(tabs, (nn - tid * k1) << (24 - log_n))) : 1.0;
        const double step = centred(root_pow24(tabs, (nn - nt * k1) << (24 - log_n)));  // nt <= n2 / 8, so 0 < nt * k1 < N
        for (uint32_t i = tid; i < n2; i += nt) {
            sm[swz(i)] = to_lds(p2f::mm((double)src[i], cur));
            cur = p2f::mm(cur, step);
        }
    }
    __syncthreads();
    // forward DIF of size N2 (w_N2^e = tw[2e]); every trace of 2^12 rows or more has N2 = 2^12
    if (log_n2 == 12) run_stages_static<true, 12, 0, 12, 1, 0>(sm, tw, tid, nt);
    else run_stages<true>(sm, tw, log_n2, 0, log_n2, 1, 0, tid, nt);
    // position q holds DFT[bitrev(q)] = N * coeff[k], k = k1 + (k2 << log_n1), k2 = (N2 - bitrev(q)) mod N2.
    // The coefficient must be scaled by s^k / N (s = coset shift, mode 0; 1, mode 1; w_M^-1, mode 2):
    // s^k / N = s^k1 * T[q] with T[q] = s^(k2 << log_n1) / N; the per-block factor s^k1 commutes with the
    // (linear) M2-point transform and is folded into the output twiddle below.
    // Scale, then place at the bit-reversed slot of the zero-padded M2 array with the first DIT stage
    // (pairs (c,0) -> (c,c)) folded in.
    Lw regs[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        uint32_t q = tid + t * nt;
        if (q < n2) regs[t] = to_lds(p2f::mm((double)sm[swz(q)], (double)(Lw)T[q]));
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; t++) {
        uint32_t q = tid + t * nt;
        if (q < n2) {
            uint32_t k2 = (n2 - bitrev(q, log_n2)) & (n2 - 1);
            uint32_t d = 2 * bitrev(k2, log_n2);
            sm[swz(d)] = regs[t];
            sm[swz(d) ^ 1] = regs[t];
        }
    }
    __syncthreads();
    // remaining DIT stages of the M2-point transform
    if (log_n2 == 12) run_stages_static<false, 13, 1, 12, 0, 0>(sm, tw, tid, nt);
    else run_stages<false>(sm, tw, log_m2, 1, log_m2 - 1, 0, 0, tid, nt);
    if (log_n1 == 0) {  // single block: k1 = 0, nothing left to multiply
        for (uint32_t j2 = tid; j2 < m2; j2 += nt) dst[j2] = lds_to_word(sm[swz(j2)]);
        return;
    }
    if (tid >= m2) return;
    // out[j2] = val * s^k1 * w_M^(j2 k1); j2 = tid + t nt: a geometric sequence per thread (j2 k1 < M)
    Fp c0 = root_pow24(tabs, (tid * k1) << (24 - log_m));
    if (shift_mode == 0) c0 = c0 * shift_pow(tabs, k1);
    else if (shift_mode == 2 && k1) c0 = c0 * root_pow24(tabs, ((2u << log_n) - k1) << (24 - log_m));
    double cur = centred(c0);
    const double step = centred(root_pow24(tabs, (nt * k1) << (24 - log_m)));  // nt <= m2, so nt * k1 < M
    for (uint32_t j2 = tid; j2 < m2; j2 += nt) {
        dst[j2] = p2f::fix(p2f::mm((double)sm[swz(j2)], cur));
        cur = p2f::mm(cur, step);
    }
}


// ---------------------------------------------------------------- P2, two trace columns per workgroup
// The same transform as lde_block_kernel (below) for log_n2 = 12 on TWO adjacent trace columns at once: element i of
// both columns lives in one 8-byte LDS slot (int2), so the swizzle, the lane -> work-item maps and every twiddle / scaling
// factor (functions of the position only) are computed once for two butterflies, LDS moves 8 bytes per lane
// (ds_read_b64 / ds_write_b64: twice the bytes per LDS cycle of the 4-byte forms; the bank of an 8-byte slot pair is
// (i mod 32), so the 32-bank analysis of swz() carries over), and the per-thread geometric twiddle sequences of the
// load / store loops are shared.  64 KB tile + 16 KB twiddles: two workgroups (four columns) per CU.
template <int R, int L, int S, int TWS>
__device__ __forceinline__ void dif_pass2(int2 *sm, const Lw *tw, uint32_t tid, uint32_t nt) {
    constexpr uint32_t G = 1u << R, lh = L - S - R, work = 1u << (L - R);
    for (uint32_t w = tid; w < work; w += nt) {
        const uint32_t r = w & ((1u << lh) - 1);
        const uint32_t blk = lane_block<R>(w >> lh, lh, S);
        const uint32_t base = swz((blk << (L - S)) | r);
        double v[G][2];
#pragma unroll
        for (uint32_t j = 0; j < G; j++) {
            const int2 x = sm[base ^ swz(j << lh)];
            v[j][0] = (double)x.x; v[j][1] = (double)x.y;
        }
#pragma unroll
        for (uint32_t t = 0; t < (uint32_t)R; t++) {
            const uint32_t dist = G >> (t + 1);
            const uint32_t twr = swz((r << (S + t)) << TWS);
#pragma unroll
            for (uint32_t j = 0; j < G; j++) {
                if (j & dist) continue;
                const double wv = (double)tw[twr ^ swz((((j & (dist - 1)) << lh) << (S + t)) << TWS)];
                const double wp = wv * p2f::PINV;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const double a = v[j][c], b = v[j + dist][c];
                    v[j][c] = a + b;
                    v[j + dist][c] = p2f::mm_pre(a - b, wv, wp);
                }
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < G; j++)
            sm[base ^ swz(j << lh)] = make_int2(to_lds((j & 1) ? v[j][0] : p2f::red(v[j][0])), to_lds((j & 1) ? v[j][1] : p2f::red(v[j][1])));
    }
}
template <int R, int L, int S, int TWS>
__device__ __forceinline__ void dit_pass2(int2 *sm, const Lw *tw, uint32_t tid, uint32_t nt) {
    constexpr uint32_t G = 1u << R, work = 1u << (L - R), blk_bits = L - R - S;
    for (uint32_t g = tid; g < work; g += nt) {
        const uint32_t r = g & ((1u << S) - 1);
        const uint32_t blk = lane_block<R>(g >> S, S, blk_bits);
        const uint32_t base = swz((blk << (S + R)) | r);
        double v[G][2];
#pragma unroll
        for (uint32_t j = 0; j < G; j++) {
            const int2 x = sm[base ^ swz(j << S)];
            v[j][0] = (double)x.x; v[j][1] = (double)x.y;
        }
#pragma unroll
        for (uint32_t t = 0; t < (uint32_t)R; t++) {
            const uint32_t dist = 1u << t;
            const uint32_t twr = swz((r << (L - 1 - S - t)) << TWS);
#pragma unroll
            for (uint32_t j = 0; j < G; j++) {
                if (j & dist) continue;
                const double wv = (double)tw[twr ^ swz((((j & (dist - 1)) << S) << (L - 1 - S - t)) << TWS)];
                const double wp = wv * p2f::PINV;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const double a = v[j][c], b = p2f::mm_pre(v[j + dist][c], wv, wp);   // |v| < 2^33
                    v[j][c] = a + b;
                    v[j + dist][c] = a - b;
                }
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < G; j++) sm[base ^ swz(j << S)] = make_int2(to_lds(p2f::red(v[j][0])), to_lds(p2f::red(v[j][1])));
    }
}
template <bool DIF, int L, int FIRST, int COUNT, int TWS>
__device__ __forceinline__ void run_stages2(int2 *sm, const Lw *tw, uint32_t tid, uint32_t nt) {
    if constexpr (COUNT > 0) {
        constexpr int R = COUNT >= 3 ? 3 : COUNT;
        if constexpr (DIF) dif_pass2<R, L, FIRST, TWS>(sm, tw, tid, nt);
        else dit_pass2<R, L, FIRST, TWS>(sm, tw, tid, nt);
        __syncthreads();
        run_stages2<DIF, L, FIRST + R, COUNT - R, TWS>(sm, tw, tid, nt);
    }
}

// grid.x = N1, grid.y = column pairs; log_n2 = 12, 512 threads; columns 2 y and 2 y + 1 (both exist)
__global__ void __launch_bounds__(512) lde_block2_kernel(const uint32_t *in, uint32_t *out, uint32_t log_n, uint32_t log_n1,
                                                       uint32_t shift_mode, NttTables tabs) {
    extern __shared__ __align__(16) uint32_t lds[];
    constexpr uint32_t log_n2 = 12, n2 = 1u << log_n2, m2 = n2 * 2, log_m2 = log_n2 + 1;
    const uint32_t log_m = log_n + 1;
    int2 *sm = reinterpret_cast<int2 *>(lds);
    Lw *tw = reinterpret_cast<Lw *>(sm + m2);  // n2 entries: w_M2^e, e < M2/2 (canonical)
    const uint32_t k1 = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const uint32_t *src0 = in + ((size_t)(2 * blockIdx.y) << log_n) + ((size_t)k1 << log_n2), *src1 = src0 + ((size_t)1 << log_n);
    uint32_t *dst0 = out + ((size_t)(2 * blockIdx.y) << log_m) + ((size_t)k1 << log_m2), *dst1 = dst0 + ((size_t)1 << log_m);
    const uint32_t *W = tabs.lde_tw_c + (n2 - 1);
    const uint32_t *T = tabs.lde_scale_c + ((size_t)(shift_mode * LDE_MAX_LOG + log_n) << 12);
    for (uint32_t e = tid; e < n2; e += nt) tw[swz(e)] = (Lw)W[e];
    if (log_n1 == 0 || k1 == 0) {
        for (uint32_t i = tid; i < n2; i += nt) sm[swz(i)] = make_int2((Lw)src0[i], (Lw)src1[i]);
    } else {
        const uint32_t nn = 1u << log_n;
        double cur = tid ? centred(root_pow24(tabs, (nn - tid * k1) << (24 - log_n))) : 1.0;
        const double step = centred(root_pow24(tabs, (nn - nt * k1) << (24 - log_n)));
        for (uint32_t i = tid; i < n2; i += nt) {
            const double cp = cur * p2f::PINV;
            sm[swz(i)] = make_int2(to_lds(p2f::mm_pre((double)src0[i], cur, cp)), to_lds(p2f::mm_pre((double)src1[i], cur, cp)));
            cur = p2f::mm(cur, step);
        }
    }
    __syncthreads();
    run_stages2<true, 12, 0, 12, 1>(sm, tw, tid, nt);
    int2 regs[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const uint32_t q = tid + t * nt;
        const double sc = (double)(Lw)T[q], sp = sc * p2f::PINV;
        const int2 x = sm[swz(q)];
        regs[t] = make_int2(to_lds(p2f::mm_pre((double)x.x, sc, sp)), to_lds(p2f::mm_pre((double)x.y, sc, sp)));
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const uint32_t q = tid + t * nt;
        const uint32_t k2 = (n2 - bitrev(q, log_n2)) & (n2 - 1);
        const uint32_t d = 2 * bitrev(k2, log_n2);
        sm[swz(d)] = regs[t];
        sm[swz(d) ^ 1] = regs[t];
    }
    __syncthreads();
    run_stages2<false, 13, 1, 12, 0>(sm, tw, tid, nt);
    if (log_n1 == 0) {
        for (uint32_t j2 = tid; j2 < m2; j2 += nt) { const int2 x = sm[swz(j2)]; dst0[j2] = lds_to_word(x.x); dst1[j2] = lds_to_word(x.y); }
        return;
    }
    Fp c0 = root_pow24(tabs, (tid * k1) << (24 - log_m));
    if (shift_mode == 0) c0 = c0 * shift_pow(tabs, k1);
    else if (shift_mode == 2 && k1) c0 = c0 * root_pow24(tabs, ((2u << log_n) - k1) << (24 - log_m));
    double cur = centred(c0);
    const double step = centred(root_pow24(tabs, (nt * k1) << (24 - log_m)));
    for (uint32_t j2 = tid; j2 < m2; j2 += nt) {
        const double cp = cur * p2f::PINV;
        const int2 x = sm[swz(j2)];
        dst0[j2] = p2f::fix(p2f::mm_pre((double)x.x, cur, cp));
        dst1[j2] = p2f::fix(p2f::mm_pre((double)x.y, cur, cp));
        cur = p2f::mm(cur, step);
    }
}

// ---------------------------------------------------------------- elementwise representation change
__global__ void to_internal_kernel(uint32_t *d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) d[i] = Fp::from_canonical(d[i]).v;
}
__global__ void from_internal_kernel(uint32_t *d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) d[i] = Fp::raw(d[i]).canonical();
}

// ---------------------------------------------------------------- host side
static std::vector<uint32_t> power_table(Fp base, size_t n) {
    std::vector<uint32_t> t(n);
    Fp x = Fp::one();
    for (size_t i = 0; i < n; i++) { t[i] = x.v; x = x * base; }
    return t;
}

static uint32_t bitrev_host(uint32_t x, uint32_t bits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

hipError_t ntt_tables_create(NttTables *t) {
    Fp omega = two_adic_generator(24);
    Fp g = Fp::from_canonical(COSET_SHIFT);
    std::vector<uint32_t> lo = power_table(omega, 4096), hi = power_table(pow(omega, 4096), 4096);
    std::vector<uint32_t> slo = power_table(g, 4096), shi = power_table(pow(g, 4096), 2048);
    std::vector<uint32_t> all;
    all.insert(all.end(), hi.begin(), hi.end());
    all.insert(all.end(), lo.begin(), lo.end());
    all.insert(all.end(), shi.begin(), shi.end());
    all.insert(all.end(), slo.begin(), slo.end());
    // lde_tw: w_{2^l}^e, e < 2^(l-1), l = 1..13, at offset 2^(l-1) - 1 (8191 words, padded to 8192)
    const size_t off_tw = all.size();
    for (uint32_t l = 1; l <= 13; l++) {
        std::vector<uint32_t> w = power_table(pow(omega, (uint64_t)1 << (24 - l)), (size_t)1 << (l - 1));
        all.insert(all.end(), w.begin(), w.end());
    }
    all.push_back(0);
    // lde_scale[mode][log_n][q], q < N2 = 2^min(log_n, 12): s^(k2 << log_n1) / N with k2 = (N2 - bitrev(q)) mod N2,
    // s = g (mode 0), 1 (mode 1), w_{2N}^-1 (mode 2)
    const size_t off_scale = all.size();
    all.resize(off_scale + ((size_t)3 * LDE_MAX_LOG << 12), 0);
    for (uint32_t mode = 0; mode < 3; mode++)
        for (uint32_t log_n = 0; log_n < LDE_MAX_LOG; log_n++) {
            const uint32_t log_n2 = log_n < 12 ? log_n : 12, log_n1 = log_n - log_n2, n2 = 1u << log_n2;
            const Fp ninv = inv(Fp::from_canonical((uint32_t)(((uint64_t)1 << log_n) % P)));
            Fp s = Fp::one();
            if (mode == 0) s = g;
            else if (mode == 2) s = inv(pow(omega, (uint64_t)1 << (24 - (log_n + 1))));
            std::vector<uint32_t> pw = power_table(pow(s, (uint64_t)1 << log_n1), n2);
            uint32_t *dst = all.data() + off_scale + ((size_t)(mode * LDE_MAX_LOG + log_n) << 12);
            for (uint32_t q = 0; q < n2; q++) dst[q] = (ninv * Fp::raw(pw[(n2 - bitrev_host(q, log_n2)) & (n2 - 1)])).v;
        }
    // canonical (centred, two's complement) copies of lde_tw and lde_scale: operands of the FP64 products
    const size_t off_tw_c = all.size();
    auto centred_word = [](uint32_t mont) -> uint32_t { uint32_t c = Fp::raw(mont).canonical(); return c > P / 2 ? c - P : c; };
    for (size_t i = off_tw; i < off_scale; i++) all.push_back(centred_word(all[i]));
    const size_t off_scale_c = all.size();
    all.resize(off_scale_c + ((size_t)3 * LDE_MAX_LOG << 12));
    for (size_t i = 0; i < ((size_t)3 * LDE_MAX_LOG << 12); i++) all[off_scale_c + i] = centred_word(all[off_scale + i]);
    uint32_t *d = nullptr;
    hipError_t e = hipMalloc(&d, all.size() * 4);
    if (e != hipSuccess) return e;
    e = hipMemcpy(d, all.data(), all.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return e; }
    t->base = d;
    t->tw_hi = d;
    t->tw_lo = d + 4096;
    t->sh_hi = d + 8192;
    t->sh_lo = d + 8192 + 2048;
    t->lde_tw = d + off_tw;
    t->lde_scale = d + off_scale;
    t->lde_tw_c = d + off_tw_c;
    t->lde_scale_c = d + off_scale_c;
    return hipSuccess;
}
void ntt_tables_destroy(NttTables *t) {
    if (t->base) (void)hipFree(t->base);
    t->base = nullptr;
}

hipError_t launch_coset_lde(hipStream_t st, const NttTables &tabs, uint32_t *d_in, uint32_t *d_scratch, uint32_t *d_out,
                            uint32_t width, uint32_t log_n, uint32_t shift_mode) {
    // d_scratch ([width][N]) receives the P1 result so that d_in survives; NULL = run P1 in place on d_in.
    if (!d_scratch) d_scratch = d_in;
    if (width == 0) return hipSuccess;
    if (log_n > 22 || shift_mode > 2) return hipErrorInvalidValue;
    const uint32_t log_n2 = log_n < 12 ? log_n : 12, log_n1 = log_n - log_n2;
    const size_t n = (size_t)1 << log_n;
    // 512 threads: 8 waves share a tile (measured 15 % faster than 256 at n = 2^21; 1024 is slower again)
    const unsigned T_STRIDED = 512;
    const unsigned T_BLOCK = log_n2 >= 8 ? 512 : 64;  // N2 / 16 <= threads <= M2
    // strided passes: tile = 2^log_n1 rows x 2^b columns, 2^13 elements (32 KiB) per workgroup
    // (at least 16 consecutive words = 64-B segments, so 64 KiB tiles at n = 22)
    uint32_t b = 13 - log_n1 < 4 ? 4 : 13 - log_n1;
    if (log_n1) {
        size_t lds = ((size_t)(1u << (log_n1 + b)) + (1u << log_n1)) * 4;   // tile words + rows / 2 twiddles as doubles
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_strided_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_strided_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        dim3 grid((1u << log_n2) >> b, width);
        if (log_n1 == V4_L && b == 4) ntt_strided_v4_kernel<true><<<grid, 256, 0, st>>>(d_in, d_scratch, n, 1u << log_n2, tabs);
        else ntt_strided_kernel<true><<<grid, T_STRIDED, lds, st>>>(d_in, d_scratch, n, log_n1, 1u << log_n2, b, log_n, tabs);
    }
    {
        const uint32_t *p2_in = log_n1 ? d_scratch : d_in;
        uint32_t done = 0;
        if (log_n2 == 12 && width >= 2) {   // column pairs through the two-column kernel
            static bool attr_set = false;
            const size_t lds2 = ((size_t)(2u << log_n2) * 2 + (1u << log_n2)) * 4;   // 64 KB of int2 + 16 KB of twiddles
            if (!attr_set) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lde_block2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                if (e != hipSuccess) return e;
                attr_set = true;
            }
            dim3 grid2(1u << log_n1, width / 2);
            lde_block2_kernel<<<grid2, 512, lds2, st>>>(p2_in, d_out, log_n, log_n1, shift_mode, tabs);
            done = width & ~1u;
        }
        if (done < width) {
            size_t lds = ((size_t)(2u << log_n2) + (1u << log_n2)) * 4;
            dim3 grid(1u << log_n1, width - done);
            lde_block_kernel<<<grid, T_BLOCK, lds, st>>>(p2_in + ((size_t)done << log_n), d_out + ((size_t)done << (log_n + 1)), log_n, log_n1, shift_mode, tabs);
        }
    }
    if (log_n1) {
        size_t lds = ((size_t)(1u << (log_n1 + b)) + (1u << log_n1)) * 4;   // tile words + rows / 2 twiddles as doubles
        dim3 grid((2u << log_n2) >> b, width);
        if (log_n1 == V4_L && b == 4) ntt_strided_v4_kernel<false><<<grid, 256, 0, st>>>(d_out, d_out, 2 * n, 2u << log_n2, tabs);
        else ntt_strided_kernel<false><<<grid, T_STRIDED, lds, st>>>(d_out, d_out, 2 * n, log_n1, 2u << log_n2, b, log_n + 1, tabs);
    }
    return hipGetLastError();
}

hipError_t launch_to_internal(hipStream_t st, uint32_t *d, size_t n) {
    if (!n) return hipSuccess;
    unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    to_internal_kernel<<<blocks, 256, 0, st>>>(d, n);
    return hipGetLastError();
}
hipError_t launch_from_internal(hipStream_t st, uint32_t *d, size_t n) {
    if (!n) return hipSuccess;
    unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    from_internal_kernel<<<blocks, 256, 0, st>>>(d, n);
    return hipGetLastError();
}

}  // namespace dvt
