// Minimal reproducer for the K0 carry-chain miscompile (hipcc 7.2, gfx950) that csrc/rv32.h pins with an `asm volatile`
// fence.  The MUL family's byte-product loop of fill_cpu_row, isolated: each thread computes the
// eight product bytes / carries of b * c and emits them twice, as the kernel does — through the double conversion of
// DeviceSink::put (the trace cell) and as the integer lookup key of DeviceSink::byte.  Both must equal the host's values.
//   hipcc -O3 --offload-arch=gfx950 k0_carry_repro.hip -o k0_carry_repro            (no fence: miscompiled?)
//   hipcc -O3 --offload-arch=gfx950 -DFENCE k0_carry_repro.hip -o k0_carry_repro_f  (with the fence)
// Prints the number of (thread, k) pairs whose cell / key differ from the host computation.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

struct Sink {
    double *cells;     // [8][n]
    uint32_t *keys;    // [9][n]
    uint32_t *hist;    // [65536] lookup counts, as the kernel's atomics
    size_t n, row;
    __device__ void put(int col, uint32_t v) { cells[(size_t)col * n + row] = (double)v * 3.0; }
    __device__ void byte(int slot, uint32_t key) { keys[(size_t)slot * n + row] = key; atomicAdd(&hist[key & 0xffff], 1u); }
    __device__ void fence(uint32_t &v) {
#ifdef FENCE
        asm volatile("" : "+v"(v));
#else
        (void)v;
#endif
    }
};

template <class S>
__device__ __host__ void mul_rows(uint32_t b, uint32_t c, int xo, S &s) {
    auto B = [](uint32_t w, int i) -> uint32_t { return (w >> (8 * i)) & 0xffu; };
    uint32_t pbyte[8], pcarry[8], acc = 0;
    for (int k = 0; k < 8; k++) {
        uint32_t t = acc;
        for (int i = 0; i < 4; i++) { int j = k - i; if (j >= 0 && j < 4) t += B(b, i) * B(c, j); }
        s.fence(t);
        pbyte[k] = t & 0xff;
        pcarry[k] = t >> 8;
        acc = t >> 8;
    }
    for (int i = 0; i < 4; i++) s.put(i, pbyte[xo + i]);
    for (int k = 0; k < 4; k++) s.put(4 + k, pcarry[k]);
    for (int k = 0; k < 7; k++) s.byte(k, 6u * 65536u + pcarry[k]);
    s.byte(7, 5u * 65536u + ((pbyte[xo] << 8) | pbyte[xo + 1]));
    s.byte(8, 5u * 65536u + ((pbyte[xo + 2] << 8) | pbyte[xo + 3]));
}

__global__ void kern(const uint32_t *b, const uint32_t *c, double *cells, uint32_t *keys, uint32_t *hist, size_t n) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    Sink s{cells, keys, hist, n, r};
    if (r & 1) mul_rows(b[r], c[r], 4, s); else mul_rows(b[r], c[r], 0, s);
}

struct HostSink {
    std::vector<double> cells;
    std::vector<uint32_t> keys;
    size_t n, row;
    void put(int col, uint32_t v) { cells[(size_t)col * n + row] = (double)v * 3.0; }
    void byte(int slot, uint32_t key) { keys[(size_t)slot * n + row] = key; }
    void fence(uint32_t &) {}
};

int main() {
    const size_t n = 1 << 16;
    std::vector<uint32_t> b(n), c(n);
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        b[i] = (uint32_t)x; c[i] = (uint32_t)(x >> 32);
        if (i % 7 == 0) b[i] = 0xffffffffu;
        if (i % 11 == 0) c[i] = 0xffffffffu;
    }
    uint32_t *db, *dc, *dk, *dh;
    double *dcells;
    hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dk, 9 * n * 4); hipMalloc(&dh, 65536 * 4); hipMalloc(&dcells, 8 * n * 8);
    hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(dh, 0, 65536 * 4);
    kern<<<(unsigned)(n / 256), 256>>>(db, dc, dcells, dk, dh, n);
    std::vector<uint32_t> keys(9 * n);
    std::vector<double> cells(8 * n);
    hipMemcpy(keys.data(), dk, 9 * n * 4, hipMemcpyDeviceToHost); hipMemcpy(cells.data(), dcells, 8 * n * 8, hipMemcpyDeviceToHost);
    HostSink h{std::vector<double>(8 * n), std::vector<uint32_t>(9 * n), n, 0};
    for (size_t r = 0; r < n; r++) { h.row = r; mul_rows(b[r], c[r], (r & 1) ? 4 : 0, h); }
    size_t bad_cells = 0, bad_keys = 0, first = (size_t)-1;
    for (size_t i = 0; i < 8 * n; i++) bad_cells += cells[i] != h.cells[i];
    for (size_t i = 0; i < 9 * n; i++) if (keys[i] != h.keys[i]) { bad_keys++; if (first == (size_t)-1) first = i; }
    printf("wrong cells %zu, wrong lookup keys %zu of %zu", bad_cells, bad_keys, 9 * n);
    if (bad_keys) printf(" (first: slot %zu row %zu b=%08x c=%08x device %08x host %08x)", first / n, first % n, b[first % n], c[first % n], keys[first], h.keys[first]);
    printf("\n");
    return 0;
}
