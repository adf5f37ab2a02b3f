// K1 — column-batched BabyBear coset LDE for gfx950.
//
// Data layout: column-major device field arrays; every column is one
// contiguous polynomial of N = 2^n evaluations in natural order.  Output: 2N
// evaluations on shift * <w_2N>, natural order.
//
// Decomposition (four-step on both transforms, fused in the middle), with
// N = N1*N2, N2 = 2^min(n,12), M2 = 2*N2, M = 2N = N1*M2:
//   P1  ntt_strided<inverse>   for every n2: N1-point inverse DFT over the
//                              stride-N2 samples, times w_N^(-n2*k1).
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

// ---- fused radix-2^R passes over an LDS array ------------------------------------------------
// Element (i, c) of a [2^L][2^log_cols] tile lives at sm[(i << log_cols) | c]; the transform runs
// over i for every c.  tw[e << tw_shift] = w_{2^L}^e.
// DIF stages s .. s+R-1 (stage t pairs distance 2^(L-1-t), twiddle exponent (index mod half) << t).
template <int R>
__device__ __forceinline__ void dif_pass(Fp *sm, const Fp *tw, uint32_t L, uint32_t s, uint32_t tw_shift, uint32_t log_cols,
                                         uint32_t tid, uint32_t nt) {
    constexpr uint32_t G = 1u << R;
    const uint32_t lh_last = L - s - R, h_last = 1u << lh_last, cmask = (1u << log_cols) - 1;
    const uint32_t work = (1u << (L - R)) << log_cols;
    for (uint32_t w = tid; w < work; w += nt) {
        const uint32_t c = w & cmask, g = w >> log_cols;
        const uint32_t r = g & (h_last - 1), blk = g >> lh_last;
        const uint32_t base = (blk << (L - s)) | r;
        Fp v[G];
#pragma unroll
        for (uint32_t j = 0; j < G; j++) v[j] = sm[((base + (j << lh_last)) << log_cols) | c];
#pragma unroll
        for (uint32_t t = 0; t < (uint32_t)R; t++) {
            const uint32_t dist = G >> (t + 1);
#pragma unroll
            for (uint32_t j = 0; j < G; j++) {
                if (j & dist) continue;
                const uint32_t lo = ((j & (dist - 1)) << lh_last) | r;
                Fp a = v[j], b = v[j + dist];
                v[j] = a + b;
                v[j + dist] = (a - b) * tw[(lo << (s + t)) << tw_shift];
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < G; j++) sm[((base + (j << lh_last)) << log_cols) | c] = v[j];
    }
}
// DIT stages s .. s+R-1 (stage t pairs distance 2^t, twiddle exponent (index mod 2^t) << (L-1-t)).
template <int R>
__device__ __forceinline__ void dit_pass(Fp *sm, const Fp *tw, uint32_t L, uint32_t s, uint32_t tw_shift, uint32_t tid, uint32_t nt) {
    constexpr uint32_t G = 1u << R;
    const uint32_t h0 = 1u << s;
    const uint32_t work = 1u << (L - R);
    for (uint32_t g = tid; g < work; g += nt) {
        const uint32_t r = g & (h0 - 1), blk = g >> s;
        const uint32_t base = (blk << (s + R)) | r;
        Fp v[G];
#pragma unroll
        for (uint32_t j = 0; j < G; j++) v[j] = sm[base + (j << s)];
#pragma unroll
        for (uint32_t t = 0; t < (uint32_t)R; t++) {
            const uint32_t dist = 1u << t;
#pragma unroll
            for (uint32_t j = 0; j < G; j++) {
                if (j & dist) continue;
                const uint32_t lo = ((j & (dist - 1)) << s) | r;
                Fp a = v[j], b = v[j + dist] * tw[(lo << (L - 1 - s - t)) << tw_shift];
                v[j] = a + b;
                v[j + dist] = a - b;
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < G; j++) sm[base + (j << s)] = v[j];
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
__global__ void __launch_bounds__(256) ntt_strided_kernel(const uint32_t *src, uint32_t *data, size_t col_stride, uint32_t log_rows,
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

    for (uint32_t e = tid; e < rows / 2; e += nt) tw[e] = root_pow24(tabs, e << (24 - log_rows));
    for (uint32_t idx = tid; idx < tile_elems; idx += nt) {
        uint32_t r = idx >> log_cols, c = idx & cmask;
        sm[idx] = Fp::raw(scol[(size_t)r * row_stride + c0 + c]);
    }
    __syncthreads();
    run_stages<true>(sm, tw, log_rows, 0, log_rows, 0, log_cols, tid, nt);  // forward DIF over the row index
    for (uint32_t idx = tid; idx < tile_elems; idx += nt) {
        uint32_t q = idx >> log_cols, c = idx & cmask;
        uint32_t kf = bitrev(q, log_rows);
        Fp val = sm[idx];
        uint32_t orow;
        if (INVERSE) {
            orow = (rows - kf) & (rows - 1);
            uint32_t e = (c0 + c) * orow;  // < N
            if (e) val = val * root_pow24(tabs, ((1u << log_n) - e) << (24 - log_n));
        } else {
            orow = kf;
        }
        col[(size_t)orow * row_stride + c0 + c] = val.v;
    }
}

// ---------------------------------------------------------------- P2
// grid.x = N1 (block index k1), grid.y = column.  blockDim.x = 256, N2 <= 4096.
__global__ void __launch_bounds__(256) lde_block_kernel(const uint32_t *in, uint32_t *out, uint32_t log_n, uint32_t log_n1,
                                                       uint32_t shift_mode, uint32_t ninv_m, NttTables tabs) {
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t log_n2 = log_n - log_n1, n2 = 1u << log_n2, m2 = n2 * 2, log_m2 = log_n2 + 1;
    const uint32_t log_m = log_n + 1;
    Fp *sm = reinterpret_cast<Fp *>(lds);
    Fp *tw = sm + m2;  // n2 entries: w_M2^e, e < M2/2
    const uint32_t k1 = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const uint32_t *src = in + ((size_t)blockIdx.y << log_n) + ((size_t)k1 << log_n2);
    uint32_t *dst = out + ((size_t)blockIdx.y << log_m) + ((size_t)k1 << log_m2);

    for (uint32_t e = tid; e < n2; e += nt) tw[e] = root_pow24(tabs, e << (24 - log_m2));
    for (uint32_t i = tid; i < n2; i += nt) sm[i] = Fp::raw(src[i]);
    __syncthreads();
    run_stages<true>(sm, tw, log_n2, 0, log_n2, 1, 0, tid, nt);  // forward DIF of size N2 (w_N2^e = tw[2e])
    // position q holds DFT[bitrev(q)] = N * coeff[(N2 - bitrev(q)) mod N2 (+ block k1)];
    // scale, then place at the bit-reversed slot of the zero-padded M2 array with the
    // first DIT stage (pairs (c,0) -> (c,c)) folded in.
    Fp regs[16];
    const Fp ninv = Fp::raw(ninv_m);
#pragma unroll
    for (int t = 0; t < 16; t++) {
        uint32_t q = tid + t * 256;
        if (q < n2) {
            uint32_t k2 = (n2 - bitrev(q, log_n2)) & (n2 - 1);
            uint32_t k = k1 + (k2 << log_n1);  // coefficient index < N
            Fp sc = ninv;
            if (shift_mode == 0) sc = sc * shift_pow(tabs, k);
            else if (shift_mode == 2 && k) sc = sc * root_pow24(tabs, ((2u << log_n) - k) << (24 - log_m));
            regs[t] = sm[q] * sc;
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; t++) {
        uint32_t q = tid + t * 256;
        if (q < n2) {
            uint32_t k2 = (n2 - bitrev(q, log_n2)) & (n2 - 1);
            uint32_t d = 2 * bitrev(k2, log_n2);
            sm[d] = regs[t];
            sm[d + 1] = regs[t];
        }
    }
    __syncthreads();
    run_stages<false>(sm, tw, log_m2, 1, log_m2 - 1, 0, 0, tid, nt);  // remaining DIT stages of the M2-point transform
    for (uint32_t j2 = tid; j2 < m2; j2 += nt) {
        Fp val = sm[j2];
        if (log_n1) {
            uint32_t e = j2 * k1;  // < M
            if (e) val = val * root_pow24(tabs, e << (24 - log_m));
        }
        dst[j2] = val.v;
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

hipError_t ntt_tables_create(NttTables *t) {
    Fp omega = two_adic_generator(24);
    Fp g = Fp::from_canonical(COSET_SHIFT);
    std::vector<uint32_t> lo = power_table(omega, 4096), hi = power_table(pow(omega, 4096), 4096);
    std::vector<uint32_t> slo = power_table(g, 4096), shi = power_table(pow(g, 4096), 2048);
    uint32_t *d = nullptr;
    size_t words = 4096 * 3 + 2048;
    hipError_t e = hipMalloc(&d, words * 4);
    if (e != hipSuccess) return e;
    std::vector<uint32_t> all;
    all.insert(all.end(), hi.begin(), hi.end());
    all.insert(all.end(), lo.begin(), lo.end());
    all.insert(all.end(), shi.begin(), shi.end());
    all.insert(all.end(), slo.begin(), slo.end());
    e = hipMemcpy(d, all.data(), words * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return e; }
    t->base = d;
    t->tw_hi = d;
    t->tw_lo = d + 4096;
    t->sh_hi = d + 8192;
    t->sh_lo = d + 8192 + 2048;
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
    Fp ninv = inv(Fp::from_canonical((uint32_t)(n % P)));
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
        ntt_strided_kernel<true><<<grid, 256, lds, st>>>(d_in, d_scratch, n, log_n1, 1u << log_n2, b, log_n, tabs);
    }
    {
        size_t lds = ((size_t)(2u << log_n2) + (1u << log_n2)) * 4;
        dim3 grid(1u << log_n1, width);
        lde_block_kernel<<<grid, 256, lds, st>>>(log_n1 ? d_scratch : d_in, d_out, log_n, log_n1, shift_mode, ninv.v, tabs);
    }
    if (log_n1) {
        size_t lds = ((size_t)(1u << (log_n1 + b)) + (1u << (log_n1 - 1))) * 4;
        dim3 grid((2u << log_n2) >> b, width);
        ntt_strided_kernel<false><<<grid, 256, lds, st>>>(d_out, d_out, 2 * n, log_n1, 2u << log_n2, b, log_n + 1, tabs);
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
