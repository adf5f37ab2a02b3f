// K0 — trace generation on the GPU.
//
// The host executor uploads one compact 48-byte record per retired instruction;
// one thread expands one record into the 103 columns of the cpu chip (byte limbs,
// carries, comparator flags, product bytes, ...) and counts its table lookups in
// the byte-table and program-table multiplicity columns.
// Column-major output: the 64 lanes of a wave write 64 consecutive rows of a
// column, so every store is a coalesced 256-B request.  HBM-write bound
// (4 B x 103 columns per cycle against 48 B read).
//
// Lookup counting: a few table rows are extremely hot (the high limbs of the
// timestamp differences are almost always (0,0); loop bodies hit the same dozen
// program rows), and plain global atomics on them serialise at the memory side
// (50 ms per 2^21-row shard; without any counting the kernel takes 0.31 ms).  So every workgroup keeps a direct-mapped LDS cache
// of (key -> count): hits are LDS atomics, conflicts fall through to a global
// atomic, and the cache is flushed once per workgroup.  A wave whose active lanes
// all carry the same key adds its lane count with a single LDS atomic.
#include "kernels.h"
#include "poseidon2_f64.cuh"
#include "rv32.h"

namespace dvt {
namespace rv32 {

constexpr uint32_t K0_SLOTS = 8192;            // 64 KiB of LDS per workgroup (4096 slots: 1.58 ms per shard, 8192: 0.59, 16384: 0.61)
constexpr uint32_t K0_ROWS_PER_BLOCK = 4096;   // rows expanded by one workgroup (16 per thread)
constexpr uint32_t K0_EMPTY = 0xffffffffu;
constexpr uint32_t K0_PROG_KEY_BASE = N_BYTE_OPS * 65536;  // program-table keys follow the byte-table keys

struct DeviceSink {
    uint32_t *cpu;
    size_t n, row;
    uint32_t *byte_mult, *prog_mult;
    const uint32_t *prog_row;
    uint32_t *lds_keys, *lds_counts;

    // canonical value -> Montgomery word on the FP64 pipe (6-operation exact product with 2^32 mod p instead of three
    // quarter-rate integer multiplies; a row issues ~60 of these)
    __device__ void put(int col, uint32_t v) { cpu[(size_t)col * n + row] = p2f::to_mont((double)v); }
    // (see rv32.h: keeps hipcc from re-associating the MUL carry chain)
    // Keeps hipcc 7.2 from miscompiling the MUL family's carry chain once its k-loop is unrolled (rv32.h).  Root cause narrowed
    // in round 2 with tools/microbench/k0_fill_repro.hip (the real fill_cpu_row with a plain store-only sink, no LDS / ballots /
    // atomics): wrong at -O2 / -O3 (carry k = 1 is computed from the UNSHIFTED accumulator: (t0 + S1) >> 8 instead of
    // ((t0 >> 8) + S1) >> 8), right at -O1, right with -fno-unroll-loops, unaffected by -amdgpu-sdwa-peephole=0 and
    // -amdgpu-codegenprepare-mul24=0, and the bare loop alone (k0_carry_repro.hip) compiles correctly: a gfx950 code-generation
    // bug after full unrolling, not the lookup-counting fast path below.
    __device__ void fence(uint32_t &v) { asm volatile("" : "+v"(v)); }

    __device__ void global_add(uint32_t key, uint32_t cnt) {
        if (key < K0_PROG_KEY_BASE) atomicAdd(&byte_mult[key], cnt);
        else atomicAdd(&prog_mult[key - K0_PROG_KEY_BASE], cnt);
    }
    __device__ void count(uint32_t key) {
        // wave-uniform fast path: every active lane has the same key -> one add of the lane count
        const uint32_t first = __builtin_amdgcn_readfirstlane(key);
        const unsigned long long active = __ballot(1), same = __ballot(key == first);
        uint32_t cnt = 1;
        if (same == active) {
            if (__builtin_amdgcn_mbcnt_hi((uint32_t)(active >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)active, 0)) != 0) return;
            cnt = (uint32_t)__popcll(active);
        }
        const uint32_t slot = (key * 2654435761u) >> 19;  // 13 bits
        const uint32_t old = atomicCAS(&lds_keys[slot], K0_EMPTY, key);
        if (old == K0_EMPTY || old == key) atomicAdd(&lds_counts[slot], cnt);
        else global_add(key, cnt);
    }
    __device__ void byte(int op, uint32_t table_row) { count((uint32_t)op * 65536u + table_row); }
    __device__ void prog(uint32_t idx) { count(K0_PROG_KEY_BASE + prog_row[idx]); }
};

__global__ void __launch_bounds__(256) k0_cpu_rows_kernel(const CycleRec *recs, size_t n_recs, uint32_t shard, uint32_t shard_next_pc, const Instr *instrs, const uint32_t *prog_row,
                                                         uint32_t *cpu, uint32_t log_n, uint32_t *byte_mult, uint32_t *prog_mult) {
    __shared__ uint32_t keys[K0_SLOTS], counts[K0_SLOTS];
    for (uint32_t s = threadIdx.x; s < K0_SLOTS; s += blockDim.x) { keys[s] = K0_EMPTY; counts[s] = 0; }
    __syncthreads();
    DeviceSink sink{cpu, (size_t)1 << log_n, 0, byte_mult, prog_mult, prog_row, keys, counts};
    const size_t base = (size_t)blockIdx.x * K0_ROWS_PER_BLOCK;
    for (uint32_t k = 0; k < K0_ROWS_PER_BLOCK / 256; k++) {
        size_t r = base + (size_t)k * 256 + threadIdx.x;
        if (r < n_recs) {
            const CycleRec rec = recs[r];
            const Instr in = instrs[rec.idx];
            const uint32_t next_pc = r + 1 < n_recs ? instrs[recs[r + 1].idx].pc : shard_next_pc;
            sink.row = r;
            fill_cpu_row(rec, in, (uint32_t)r, shard, next_pc, sink);
        }   // (padding rows stay all-zero: the launch wrapper cleared the matrix)
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < K0_SLOTS; s += blockDim.x)
        if (keys[s] != K0_EMPTY && counts[s]) sink.global_add(keys[s], counts[s]);
}

// K0 of the shift (WHICH = 0: one AluEvent per row) and mem_init (WHICH = 1: one MemInitRow per row) chips: the same sink as the
// cpu chip's K0 (Montgomery words straight into the column-major trace, lookups through the workgroup's LDS cache).  The
// host used to build these rows and upload them: 168 MB of mem_init table for the largest legal input against 26 MB of rows.
template <int WHICH>
__global__ void __launch_bounds__(256) k0_aux_rows_kernel(const void *events, size_t n_ev, uint32_t *main, uint32_t log_n, uint32_t *byte_mult) {
    __shared__ uint32_t keys[K0_SLOTS], counts[K0_SLOTS];
    for (uint32_t s = threadIdx.x; s < K0_SLOTS; s += blockDim.x) { keys[s] = K0_EMPTY; counts[s] = 0; }
    __syncthreads();
    DeviceSink sink{main, (size_t)1 << log_n, 0, byte_mult, nullptr, nullptr, keys, counts};
    const size_t base = (size_t)blockIdx.x * K0_ROWS_PER_BLOCK;
    for (uint32_t k = 0; k < K0_ROWS_PER_BLOCK / 256; k++) {
        const size_t r = base + (size_t)k * 256 + threadIdx.x;
        if (r < n_ev) {
            sink.row = r;
            if constexpr (WHICH == 0) fill_shift_row(static_cast<const AluEvent *>(events)[r], sink);
            else fill_mem_init_row(static_cast<const MemInitRow *>(events), r, sink);
        }
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < K0_SLOTS; s += blockDim.x)
        if (keys[s] != K0_EMPTY && counts[s]) sink.global_add(keys[s], counts[s]);
}
hipError_t launch_k0_shift_rows(hipStream_t st, const AluEvent *d_ev, size_t n_ev, uint32_t *d_main, uint32_t log_n, uint32_t *d_byte_mult) {
    if (!n_ev) return hipSuccess;
    if (((size_t)1 << log_n) < n_ev) return hipErrorInvalidValue;
    k0_aux_rows_kernel<0><<<(unsigned)((n_ev + K0_ROWS_PER_BLOCK - 1) / K0_ROWS_PER_BLOCK), 256, 0, st>>>(d_ev, n_ev, d_main, log_n, d_byte_mult);
    return hipGetLastError();
}
hipError_t launch_k0_mem_init_rows(hipStream_t st, const MemInitRow *d_rows, size_t n_rows, uint32_t *d_main, uint32_t log_n, uint32_t *d_byte_mult) {
    if (!n_rows) return hipSuccess;
    if (((size_t)1 << log_n) < n_rows) return hipErrorInvalidValue;
    k0_aux_rows_kernel<1><<<(unsigned)((n_rows + K0_ROWS_PER_BLOCK - 1) / K0_ROWS_PER_BLOCK), 256, 0, st>>>(d_rows, n_rows, d_main, log_n, d_byte_mult);
    return hipGetLastError();
}

hipError_t launch_k0_cpu_rows(hipStream_t st, const CycleRec *d_recs, size_t n_recs, uint32_t shard, uint32_t shard_next_pc, const Instr *d_instrs,
                              const uint32_t *d_prog_row, uint32_t *d_cpu, uint32_t log_n, uint32_t *d_byte_mult, uint32_t *d_prog_mult) {
    hipError_t e = hipMemsetAsync(d_cpu, 0, ((size_t)RV32_CPU_MAIN_W << log_n) * 4, st);
    if (e != hipSuccess) return e;
    unsigned blocks = (unsigned)((((size_t)1 << log_n) + K0_ROWS_PER_BLOCK - 1) / K0_ROWS_PER_BLOCK);
    k0_cpu_rows_kernel<<<blocks, 256, 0, st>>>(d_recs, n_recs, shard, shard_next_pc, d_instrs, d_prog_row, d_cpu, log_n, d_byte_mult, d_prog_mult);
    return hipGetLastError();
}

}  // namespace rv32
}  // namespace dvt
