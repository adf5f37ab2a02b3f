// K0 — trace generation on the GPU.
//
// The host executor uploads one compact 48-byte record per retired instruction;
// one thread expands one record into the 86 columns of the cpu chip (byte limbs,
// carries, comparator flags, product bytes, ...) and counts its table lookups with
// integer atomics on the byte-table and program-table multiplicity columns.
// Column-major output: the 64 lanes of a wave write 64 consecutive rows of a
// column, so every store is a coalesced 256-B request.  HBM-write bound
// (4 B x 86 columns per cycle against 48 B read).
#include "kernels.h"
#include "rv32.h"

namespace dvt {
namespace rv32 {

struct DeviceSink {
    uint32_t *cpu;
    size_t n, row;
    uint32_t *byte_mult, *prog_mult;
    const uint32_t *prog_row;
    __device__ void put(int col, uint32_t v) { cpu[(size_t)col * n + row] = Fp::from_canonical(v).v; }
    __device__ void byte(int op, uint32_t table_row) {
        // Materialise the row index in a VGPR before it enters the address computation: with the
        // carry chain of the MUL family folded into the atomic's address arithmetic, hipcc 7.2 (gfx950)
        // produced ((t0 + terms1) >> 8) instead of (((t0 >> 8) + terms1) >> 8) for the k = 1 lookup
        // (tests/test_gpu_k0_parity.py caught it; the stored columns were correct).
        asm volatile("" : "+v"(table_row));
        atomicAdd(&byte_mult[(size_t)op * 65536 + table_row], 1u);
    }
    __device__ void fence(uint32_t &v) { asm volatile("" : "+v"(v)); }
    __device__ void prog(uint32_t idx) { atomicAdd(&prog_mult[prog_row[idx]], 1u); }
};

__global__ void __launch_bounds__(256) k0_cpu_rows_kernel(const CycleRec *recs, size_t n_recs, const Instr *instrs, const uint32_t *prog_row,
                                                         uint32_t *cpu, uint32_t log_n, uint32_t *byte_mult, uint32_t *prog_mult) {
    size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_recs) return;
    const CycleRec rec = recs[r];
    const Instr in = instrs[rec.idx];
    DeviceSink s{cpu, (size_t)1 << log_n, r, byte_mult, prog_mult, prog_row};
    fill_cpu_row(rec, in, (uint32_t)r, s);
}

hipError_t launch_k0_cpu_rows(hipStream_t st, const CycleRec *d_recs, size_t n_recs, const Instr *d_instrs, const uint32_t *d_prog_row,
                              uint32_t *d_cpu, uint32_t log_n, uint32_t *d_byte_mult, uint32_t *d_prog_mult) {
    hipError_t e = hipMemsetAsync(d_cpu, 0, ((size_t)RV32_CPU_MAIN_W << log_n) * 4, st);
    if (e != hipSuccess) return e;
    k0_cpu_rows_kernel<<<(unsigned)((n_recs + 255) / 256), 256, 0, st>>>(d_recs, n_recs, d_instrs, d_prog_row, d_cpu, log_n, d_byte_mult,
                                                                         d_prog_mult);
    return hipGetLastError();
}

}  // namespace rv32
}  // namespace dvt
