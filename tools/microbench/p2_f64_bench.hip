// Micro-benchmark (not part of the product): Poseidon2 permutations per second on gfx950 with
//   (a) the product's Montgomery / integer-multiply field (poseidon2.cuh), and
//   (b) an FP64-FMA formulation of the same permutation (exact integers carried in doubles, lazily reduced).
// Prints both rates and checks that (b) reproduces (a) bit for bit.  Build: see tools/microbench/Makefile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../dvt_circuits_amd/csrc/poseidon2.cuh"

using namespace dvt;

#include "../../dvt_circuits_amd/csrc/poseidon2_f64.cuh"
namespace f64 {
using namespace dvt::p2f;
struct Consts { int unused; };
template <bool SMALL = false>
__host__ __device__ __forceinline__ void permute(double s[16], const Consts &) { dvt::p2f::permute(s); }
}  // namespace f64

__constant__ f64::Consts d_consts;

__global__ void __launch_bounds__(256) int_kernel(uint32_t *out, int reps) {
    Fp s[16];
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < 16; i++) s[i] = Fp::from_canonical((tid * 16u + i) % P);
    for (int r = 0; r < reps; r++) p2_permute(s);
    for (int i = 0; i < 16; i++) out[(size_t)tid * 16 + i] = s[i].canonical();
}
__global__ void __launch_bounds__(256) f64_kernel(uint32_t *out, int reps) {
    double s[16];
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < 16; i++) s[i] = (double)((tid * 16u + i) % P);
    for (int r = 0; r < reps; r++) {
        f64::permute(s, d_consts);
        for (int i = 0; i < 16; i++) s[i] = f64::red(s[i]);
    }
    for (int i = 0; i < 16; i++) out[(size_t)tid * 16 + i] = f64::to_canonical(s[i]);
}

static f64::Consts make_consts() { return f64::Consts(); }

int main(int argc, char **argv) {
    f64::Consts k = make_consts();
    // host check first (also what runs in the CPU-only container)
    {
        int bad = 0;
        for (uint32_t t = 0; t < 2000; t++) {
            Fp a[16]; double b[16];
            for (int i = 0; i < 16; i++) { uint32_t v = (t * 2654435761u + i * 40503u) % P; if (t == 0) v = i ? P - i : 0; a[i] = Fp::from_canonical(v); b[i] = (double)v; }
            p2_permute(a); f64::permute(b, k);
            for (int i = 0; i < 16; i++) bad += a[i].canonical() != f64::to_canonical(b[i]);
        }
        printf("host check: %d mismatches\n", bad);
        if (bad) return 1;
    }
    if (argc > 1 && !strcmp(argv[1], "--host-only")) return 0;
    hipMemcpyToSymbol(HIP_SYMBOL(d_consts), &k, sizeof(k));
    const int blocks = 256 * 16, threads = 256, reps = 64;
    size_t n = (size_t)blocks * threads;
    uint32_t *d_a, *d_b;
    hipMalloc(&d_a, n * 64); hipMalloc(&d_b, n * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms_int = 0, ms_f64 = 0;
    for (int it = 0; it < 3; it++) {
        hipEventRecord(e0); int_kernel<<<blocks, threads>>>(d_a, reps); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_int, e0, e1);
        hipEventRecord(e0); f64_kernel<<<blocks, threads>>>(d_b, reps); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_f64, e0, e1);
    }
    std::vector<uint32_t> a(n * 16), b(n * 16);
    hipMemcpy(a.data(), d_a, n * 64, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d_b, n * 64, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < n * 16; i++) bad += a[i] != b[i];
    double perms = (double)n * reps;
    printf("int  : %.3f ms  %.2f Gperm/s\nf64  : %.3f ms  %.2f Gperm/s\nmismatches: %zu of %zu\n", ms_int, perms / ms_int / 1e6, ms_f64, perms / ms_f64 / 1e6, bad, n * 16);
    return bad != 0;
}
