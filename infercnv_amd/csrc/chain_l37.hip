// chain kernel variants with chunk length 37 (<= 18944 padded positions per cell)
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_l37(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_v<512, 37>(a, mode, stream); }
}  // namespace icnv
