// Instantiates the generic STARK kernels for the RV32IM core machine.
#include "machine.h"
#include "gen/air_rv32.inc"

namespace dvt {
#define DVT_X(i, A) make_chip_desc<A>(),
static const ChipDesc rv32_chips[] = {DVT_AIR_RV32_CHIPS(DVT_X)};
#undef DVT_X
static const MachineDesc rv32_machine = {"rv32", air_rv32::N_CHIPS, rv32_chips};
const MachineDesc *machine_rv32() { return &rv32_machine; }
}  // namespace dvt
