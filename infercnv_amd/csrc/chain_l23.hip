// chain kernel variants with chunk length 23 (<= 11776 padded positions per cell)
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_l23(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_v<CHAIN_NT, 23>(a, mode, stream); }
}  // namespace icnv
