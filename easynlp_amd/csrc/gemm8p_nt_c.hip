// Explicit instantiations of the 8-phase NT GEMM kernel, part C of 3 (see gemm8p_nt.h: the list EZ_8P_INSTANCES_C).
#include "gemm8p_nt.h"

namespace ezclip {
namespace nt8p {
EZ_8P_INSTANCES_C(EZ_8P_DEFINE)
}  // namespace nt8p
}  // namespace ezclip
