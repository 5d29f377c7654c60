// Explicit instantiations of the 8-phase NT GEMM kernel, part B of 3 (see gemm8p_nt.h: the list EZ_8P_INSTANCES_B).
#include "gemm8p_nt.h"

namespace ezclip {
namespace nt8p {
EZ_8P_INSTANCES_B(EZ_8P_DEFINE)
}  // namespace nt8p
}  // namespace ezclip
