// chain kernel variants with chunk length 7 (<= 3584 padded positions per cell)
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_l7(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_v<CHAIN_NT, 7>(a, mode, stream); }
}  // namespace icnv
