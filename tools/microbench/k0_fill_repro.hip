// Second-stage reproducer for the K0 carry-chain miscompile (see k0_carry_repro.hip, which does NOT reproduce it): the real
// fill_cpu_row template of csrc/rv32.h instantiated with a plain sink (stores instead of LDS-cached atomics), device
// against host, on MUL / MULHU rows.  -DFENCE adds the asm fence the product uses; -DATOMIC makes byte() also do a global
// atomic add, as the product's sink does.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../dvt_circuits_amd/csrc -I../../include k0_fill_repro.hip -o k0_fill_repro
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "rv32.h"

using namespace dvt;
using namespace dvt::rv32;

constexpr int MAXK = 24;
struct Sink {
    uint32_t *cells;   // [MAIN_W][n]
    uint32_t *keys;    // [MAXK][n]
    uint32_t *hist;
    size_t n, row;
    int nk;
    DVT_HD void put(int col, uint32_t v) { cells[(size_t)col * n + row] = v; }
    DVT_HD void byte(int op, uint32_t table_row) {
        const uint32_t key = (uint32_t)op * 65536u + table_row;
        if (nk < MAXK) keys[(size_t)nk++ * n + row] = key;
#if defined(ATOMIC) && defined(__HIP_DEVICE_COMPILE__)
        atomicAdd(&hist[key & 0xffff], 1u);
#endif
    }
    DVT_HD void prog(uint32_t) {}
    DVT_HD void fence(uint32_t &v) {
#if defined(FENCE) && defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(v));
#else
        (void)v;
#endif
    }
};

__global__ void kern(const CycleRec *recs, const Instr *ins, uint32_t *cells, uint32_t *keys, uint32_t *hist, size_t n) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    Sink s{cells, keys, hist, n, r, 0};
    fill_cpu_row(recs[r], ins[recs[r].idx], (uint32_t)r, 1, 0x200800u + 4, s);
}

int main() {
    const size_t n = 1 << 16;
    std::vector<Instr> ins(2);
    for (int k = 0; k < 2; k++) {
        Instr in{};
        in.pc = 0x200800; in.rd = 5; in.rs1 = 6; in.rs2 = 7; in.supported = 1;
        in.flags = (1u << F_RD_EN) | (1u << F_RS1_EN) | (1u << F_RS2_EN) | (1u << (k ? F_MULHU : F_MUL));
        ins[k] = in;
    }
    std::vector<CycleRec> recs(n);
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        CycleRec r{};
        r.idx = i & 1;
        r.b = (uint32_t)x; r.c = (uint32_t)(x >> 32);
        if (i % 7 == 0) r.b = 0xffffffffu;
        if (i % 11 == 0) r.c = 0xffffffffu;
        r.a = (i & 1) ? (uint32_t)(((uint64_t)r.b * r.c) >> 32) : r.b * r.c;
        r.sh_ab = 1 | (1u << 16); r.sh_cm = 1 | (1u << 16);
        recs[i] = r;
    }
    CycleRec *dr; Instr *di; uint32_t *dcells, *dk, *dh;
    hipMalloc(&dr, n * sizeof(CycleRec)); hipMalloc(&di, 2 * sizeof(Instr));
    hipMalloc(&dcells, (size_t)RV32_CPU_MAIN_W * n * 4); hipMalloc(&dk, (size_t)MAXK * n * 4); hipMalloc(&dh, 65536 * 4);
    hipMemcpy(dr, recs.data(), n * sizeof(CycleRec), hipMemcpyHostToDevice); hipMemcpy(di, ins.data(), 2 * sizeof(Instr), hipMemcpyHostToDevice);
    hipMemset(dcells, 0, (size_t)RV32_CPU_MAIN_W * n * 4); hipMemset(dk, 0, (size_t)MAXK * n * 4); hipMemset(dh, 0, 65536 * 4);
    kern<<<(unsigned)(n / 256), 256>>>(dr, di, dcells, dk, dh, n);
    std::vector<uint32_t> cells((size_t)RV32_CPU_MAIN_W * n), keys((size_t)MAXK * n);
    hipMemcpy(cells.data(), dcells, cells.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(keys.data(), dk, keys.size() * 4, hipMemcpyDeviceToHost);
    std::vector<uint32_t> hc(cells.size(), 0), hk(keys.size(), 0);
    for (size_t r = 0; r < n; r++) {
        Sink s{hc.data(), hk.data(), nullptr, n, r, 0};
        fill_cpu_row(recs[r], ins[recs[r].idx], (uint32_t)r, 1, 0x200800u + 4, s);
    }
    size_t bad_cells = 0, bad_keys = 0, fc = (size_t)-1, fk = (size_t)-1;
    for (size_t i = 0; i < cells.size(); i++) if (cells[i] != hc[i]) { bad_cells++; if (fc == (size_t)-1) fc = i; }
    for (size_t i = 0; i < keys.size(); i++) if (keys[i] != hk[i]) { bad_keys++; if (fk == (size_t)-1) fk = i; }
    printf("wrong cells %zu, wrong lookup keys %zu", bad_cells, bad_keys);
    if (bad_cells) printf(" (first cell: col %zu row %zu device %u host %u)", fc / n, fc % n, cells[fc], hc[fc]);
    if (bad_keys) printf(" (first key: slot %zu row %zu b=%08x c=%08x device %08x host %08x)", fk / n, fk % n, recs[fk % n].b, recs[fk % n].c, keys[fk], hk[fk]);
    printf("\n");
    return 0;
}
