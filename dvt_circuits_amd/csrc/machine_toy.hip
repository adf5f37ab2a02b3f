// Instantiates the generic STARK kernels for the toy machine (engine unit tests).
#include "machine.h"
#include "gen/air_toy.inc"

namespace dvt {
#define DVT_X(i, A) make_chip_desc<A>(),
static const ChipDesc toy_chips[] = {DVT_AIR_TOY_CHIPS(DVT_X)};
#undef DVT_X
static const MachineDesc toy_machine = {"toy", air_toy::N_CHIPS, toy_chips};
const MachineDesc *machine_toy() { return &toy_machine; }
}  // namespace dvt
