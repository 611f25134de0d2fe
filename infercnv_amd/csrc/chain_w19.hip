// chain kernel variants with 1024 threads and chunk length 19 (<= 19456 padded positions per cell)
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_w19(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_v<1024, 19>(a, mode, stream); }
}  // namespace icnv
