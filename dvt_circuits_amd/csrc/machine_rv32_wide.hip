// The field / curve precompile chips of the RV32IM core machine (fp_op, fp2_op, bls_g1, secp_k1, u256_mul): their STARK kernels,
// instantiated apart from machine_rv32.hip so that the two halves compile side by side.
#include "machine.h"
#if !defined(__HIP_DEVICE_COMPILE__)
#pragma clang optimize off   // (the host verifier's F_p^4 instantiation: see machine_rv32.hip)
#endif
#include "gen/air_rv32.inc"
#if !defined(__HIP_DEVICE_COMPILE__)
#pragma clang optimize on
#endif

#include "gen/rv32_cols.h"
static_assert(dvt::RV32_FIRST_WIDE_CHIP == RV32_CHIP_FP_OP, "machine.h RV32_FIRST_WIDE_CHIP must be the fp_op chip");

namespace dvt {
namespace {
template <int I, class A>
bool pick(int chip, ChipDesc *out) {
    if constexpr (I >= RV32_FIRST_WIDE_CHIP) {
        if (chip == I) { *out = make_chip_desc<A>(); return true; }
    }
    return false;
}
}  // namespace
ChipDesc rv32_wide_chip_desc(int chip) {
    ChipDesc d{};
#define DVT_X(i, A) if (pick<i, A>(chip, &d)) return d;
    DVT_AIR_RV32_CHIPS(DVT_X)
#undef DVT_X
    return d;
}
}  // namespace dvt
