// Instantiates the generic STARK kernels for the RV32IM core machine.
#include "machine.h"
// The generated AIR code is instantiated twice: for the gfx950 kernels (T = Fp, device pass, fully optimised) and for the
// host verifier (T = Fp4).  The host instantiation is thousands of F_p^4 operations of straight-line code per chip that
// run once per proof; optimising it at -O3 (inlining every operator) took 18 of this file's 22 minutes of compile time.
// It is compiled unoptimised instead (each F_p^4 operator stays a call).
#if !defined(__HIP_DEVICE_COMPILE__)
#pragma clang optimize off
#endif
#include "gen/air_rv32.inc"
#if !defined(__HIP_DEVICE_COMPILE__)
#pragma clang optimize on
#endif

// The chips are instantiated in two translation units compiled side by side (this file: the cpu machine proper; the five
// field / curve precompile chips with their part-parallel kernels in machine_rv32_wide.hip): one unit took seven minutes.
namespace dvt {
ChipDesc rv32_wide_chip_desc(int chip);   // machine_rv32_wide.hip
namespace {
template <int I, class A>
ChipDesc chip_desc_here() {
    if constexpr (I < RV32_FIRST_WIDE_CHIP) return make_chip_desc<A>();
    else return rv32_wide_chip_desc(I);
}
}  // namespace
#define DVT_X(i, A) chip_desc_here<i, A>(),
static const ChipDesc rv32_chips[] = {DVT_AIR_RV32_CHIPS(DVT_X)};
#undef DVT_X
static const MachineDesc rv32_machine = {"rv32", air_rv32::N_CHIPS, rv32_chips};
const MachineDesc *machine_rv32() { return &rv32_machine; }
}  // namespace dvt
