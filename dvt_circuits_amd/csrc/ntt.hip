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

namespace dvt {

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
template <int R>
__device__ __forceinline__ void dif_pass(Fp *sm, const Fp *tw, uint32_t L, uint32_t s, uint32_t tw_shift, uint32_t log_cols,
                                         uint32_t tid, uint32_t nt) {
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
        Fp v[G];
#pragma unroll
        for (uint32_t j = 0; j < G; j++) v[j] = sm[base ^ joff[j]];
#pragma unroll
        for (uint32_t t = 0; t < (uint32_t)R; t++) {
            const uint32_t dist = G >> (t + 1);
            const uint32_t twr = swz((r << (s + t)) << tw_shift);
#pragma unroll
            for (uint32_t j = 0; j < G; j++) {
                if (j & dist) continue;
                const uint32_t twj = swz((((j & (dist - 1)) << lh_last) << (s + t)) << tw_shift);
                Fp a = v[j], b = v[j + dist];
                v[j] = a + b;
                // a - b + p < 2p < 2^32 needs no reduction before the Montgomery product (2p * p < p * 2^32)
                v[j + dist] = Fp::raw(Fp::reduce64((uint64_t)(a.v + (P - b.v)) * tw[twr ^ twj].v));
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < G; j++) sm[base ^ joff[j]] = v[j];
    }
}
// DIT stages s .. s+R-1 (stage t pairs distance 2^t, twiddle exponent (index mod 2^t) << (L-1-t)).
template <int R>
__device__ __forceinline__ void dit_pass(Fp *sm, const Fp *tw, uint32_t L, uint32_t s, uint32_t tw_shift, uint32_t tid, uint32_t nt) {
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
        Fp v[G];
#pragma unroll
        for (uint32_t j = 0; j < G; j++) v[j] = sm[base ^ joff[j]];
#pragma unroll
        for (uint32_t t = 0; t < (uint32_t)R; t++) {
            const uint32_t dist = 1u << t;
            const uint32_t twr = swz((r << (L - 1 - s - t)) << tw_shift);
#pragma unroll
            for (uint32_t j = 0; j < G; j++) {
                if (j & dist) continue;
                const uint32_t twj = swz((((j & (dist - 1)) << s) << (L - 1 - s - t)) << tw_shift);
                Fp a = v[j], b = v[j + dist] * tw[twr ^ twj];
                v[j] = a + b;
                v[j + dist] = a - b;
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < G; j++) sm[base ^ joff[j]] = v[j];
    }
}
template <bool DIF>
__device__ __forceinline__ void run_stages(Fp *sm, const Fp *tw, uint32_t L, uint32_t first, uint32_t count, uint32_t tw_shift,
                                           uint32_t log_cols, uint32_t tid, uint32_t nt) {
    uint32_t s = first, left = count;
    while (left) {
        if (left >= 3) {
            if (DIF) dif_pass<3>(sm, tw, L, s, tw_shift, log_cols, tid, nt); else dit_pass<3>(sm, tw, L, s, tw_shift, tid, nt);
            s += 3; left -= 3;
        } else if (left == 2) {
            if (DIF) dif_pass<2>(sm, tw, L, s, tw_shift, log_cols, tid, nt); else dit_pass<2>(sm, tw, L, s, tw_shift, tid, nt);
            s += 2; left -= 2;
        } else {
            if (DIF) dif_pass<1>(sm, tw, L, s, tw_shift, log_cols, tid, nt); else dit_pass<1>(sm, tw, L, s, tw_shift, tid, nt);
            s += 1; left -= 1;
        }
        __syncthreads();
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
    Fp *sm = reinterpret_cast<Fp *>(lds);
    Fp *tw = sm + tile_elems;  // rows/2 entries: w_rows^e
    uint32_t *col = data + (size_t)blockIdx.y * col_stride;
    const uint32_t *scol = src + (size_t)blockIdx.y * col_stride;
    const uint32_t c0 = blockIdx.x << log_cols;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;

    for (uint32_t e = tid; e < rows / 2; e += nt) tw[swz(e)] = Fp::raw(tabs.lde_tw[rows / 2 - 1 + e]);
    for (uint32_t idx = tid; idx < tile_elems; idx += nt) {
        uint32_t r = idx >> log_cols, c = idx & cmask;
        sm[swz(idx)] = Fp::raw(scol[(size_t)r * row_stride + c0 + c]);
    }
    __syncthreads();
    run_stages<true>(sm, tw, log_rows, 0, log_rows, 0, log_cols, tid, nt);  // forward DIF over the row index
    for (uint32_t idx = tid; idx < tile_elems; idx += nt) {
        uint32_t q = idx >> log_cols, c = idx & cmask;
        uint32_t kf = bitrev(q, log_rows);
        Fp val = sm[swz(idx)];
        uint32_t orow;
        // inverse: DFT with the output index negated; the four-step twiddle w_N^-(n2 k1) that belongs here is applied
        // by lde_block when it loads row k1 (there it is a geometric sequence per thread, here it would be a gather)
        if (INVERSE) orow = (rows - kf) & (rows - 1);
        else orow = kf;
        col[(size_t)orow * row_stride + c0 + c] = val.v;
    }
}

// ---------------------------------------------------------------- P2
// grid.x = N1 (block index k1), grid.y = column.  N2 <= 4096; N2 / 16 <= blockDim.x <= M2.
__global__ void __launch_bounds__(1024) lde_block_kernel(const uint32_t *in, uint32_t *out, uint32_t log_n, uint32_t log_n1,
                                                       uint32_t shift_mode, NttTables tabs) {
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t log_n2 = log_n - log_n1, n2 = 1u << log_n2, m2 = n2 * 2, log_m2 = log_n2 + 1;
    const uint32_t log_m = log_n + 1;
    Fp *sm = reinterpret_cast<Fp *>(lds);
    Fp *tw = sm + m2;  // n2 entries: w_M2^e, e < M2/2
    const uint32_t k1 = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const uint32_t *src = in + ((size_t)blockIdx.y << log_n) + ((size_t)k1 << log_n2);
    uint32_t *dst = out + ((size_t)blockIdx.y << log_m) + ((size_t)k1 << log_m2);

    // twiddles w_M2^e and the position-dependent part of the coefficient scaling come from tables built once per
    // prover (coalesced reads) instead of two table gathers and a product per element
    const uint32_t *W = tabs.lde_tw + (n2 - 1);
    const uint32_t *T = tabs.lde_scale + ((size_t)(shift_mode * LDE_MAX_LOG + log_n) << 12);
    for (uint32_t e = tid; e < n2; e += nt) tw[swz(e)] = Fp::raw(W[e]);
    if (log_n1 == 0 || k1 == 0) {
        for (uint32_t i = tid; i < n2; i += nt) sm[swz(i)] = Fp::raw(src[i]);
    } else if (tid < n2) {
        // four-step twiddle between P1 and this transform: x[i] *= w_N^-(i k1), i = tid + t nt (i k1 < N)
        const uint32_t nn = 1u << log_n;
        Fp cur = tid ? root_pow24(tabs, (nn - tid * k1) << (24 - log_n)) : Fp::one();
        const Fp step = root_pow24(tabs, (nn - nt * k1) << (24 - log_n));  // nt <= n2 / 8, so 0 < nt * k1 < N
        for (uint32_t i = tid; i < n2; i += nt) {
            sm[swz(i)] = Fp::raw(src[i]) * cur;
            cur = cur * step;
        }
    }
    __syncthreads();
    run_stages<true>(sm, tw, log_n2, 0, log_n2, 1, 0, tid, nt);  // forward DIF of size N2 (w_N2^e = tw[2e])
    // position q holds DFT[bitrev(q)] = N * coeff[k], k = k1 + (k2 << log_n1), k2 = (N2 - bitrev(q)) mod N2.
    // The coefficient must be scaled by s^k / N (s = coset shift, mode 0; 1, mode 1; w_M^-1, mode 2):
    // s^k / N = s^k1 * T[q] with T[q] = s^(k2 << log_n1) / N; the per-block factor s^k1 commutes with the
    // (linear) M2-point transform and is folded into the output twiddle below.
    // Scale, then place at the bit-reversed slot of the zero-padded M2 array with the first DIT stage
    // (pairs (c,0) -> (c,c)) folded in.
    Fp regs[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        uint32_t q = tid + t * nt;
        if (q < n2) regs[t] = sm[swz(q)] * Fp::raw(T[q]);
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
    run_stages<false>(sm, tw, log_m2, 1, log_m2 - 1, 0, 0, tid, nt);  // remaining DIT stages of the M2-point transform
    if (log_n1 == 0) {  // single block: k1 = 0, nothing left to multiply
        for (uint32_t j2 = tid; j2 < m2; j2 += nt) dst[j2] = sm[swz(j2)].v;
        return;
    }
    if (tid >= m2) return;
    // out[j2] = val * s^k1 * w_M^(j2 k1); j2 = tid + t nt: a geometric sequence per thread (j2 k1 < M)
    Fp cur = root_pow24(tabs, (tid * k1) << (24 - log_m));
    if (shift_mode == 0) cur = cur * shift_pow(tabs, k1);
    else if (shift_mode == 2 && k1) cur = cur * root_pow24(tabs, ((2u << log_n) - k1) << (24 - log_m));
    const Fp step = root_pow24(tabs, (nt * k1) << (24 - log_m));  // nt <= m2, so nt * k1 < M
    for (uint32_t j2 = tid; j2 < m2; j2 += nt) {
        dst[j2] = (sm[swz(j2)] * cur).v;
        cur = cur * step;
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
        size_t lds = ((size_t)(1u << (log_n1 + b)) + (1u << (log_n1 - 1))) * 4;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_strided_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_strided_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        dim3 grid((1u << log_n2) >> b, width);
        ntt_strided_kernel<true><<<grid, T_STRIDED, lds, st>>>(d_in, d_scratch, n, log_n1, 1u << log_n2, b, log_n, tabs);
    }
    {
        size_t lds = ((size_t)(2u << log_n2) + (1u << log_n2)) * 4;
        dim3 grid(1u << log_n1, width);
        lde_block_kernel<<<grid, T_BLOCK, lds, st>>>(log_n1 ? d_scratch : d_in, d_out, log_n, log_n1, shift_mode, tabs);
    }
    if (log_n1) {
        size_t lds = ((size_t)(1u << (log_n1 + b)) + (1u << (log_n1 - 1))) * 4;
        dim3 grid((2u << log_n2) >> b, width);
        ntt_strided_kernel<false><<<grid, T_STRIDED, lds, st>>>(d_out, d_out, 2 * n, log_n1, 2u << log_n2, b, log_n + 1, tabs);
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
