// chain kernel variants with 512 threads (2 wavefronts per SIMD, 256 VGPRs) and chunk length 23
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_l23(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_v<512, 23>(a, mode, stream); }
}  // namespace icnv
