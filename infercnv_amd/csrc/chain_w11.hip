// chain kernel variants with 1024 threads and chunk length 11 (<= 11264 padded positions per cell)
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_w11(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_v<1024, 11>(a, mode, stream); }
}  // namespace icnv
