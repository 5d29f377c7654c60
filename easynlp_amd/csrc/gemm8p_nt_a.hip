// Explicit instantiations of the 8-phase NT GEMM kernel, part A of 3 (see gemm8p_nt.h: the list EZ_8P_INSTANCES_A).
#include "gemm8p_nt.h"

namespace ezclip {
namespace nt8p {
EZ_8P_INSTANCES_A(EZ_8P_DEFINE)
}  // namespace nt8p
}  // namespace ezclip
