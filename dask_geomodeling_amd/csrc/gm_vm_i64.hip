// VM kernels of register class i64 (see gm_vm.h)
#include "gm_vm.h"

namespace gmvm {
GM_VM_DEFINE_LAUNCHER(long long)
}  // namespace gmvm
