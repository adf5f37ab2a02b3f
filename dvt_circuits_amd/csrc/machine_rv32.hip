// Instantiates the generic STARK kernels for the RV32IM core machine.
#include "machine.h"

namespace dvt {
const MachineDesc *machine_rv32() { return nullptr; }  // chips land with tools/airgen/rv32.py
}  // namespace dvt
