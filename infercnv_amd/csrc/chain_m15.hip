// chain kernel variants with 768 threads (3 wavefronts per SIMD, 168 VGPRs) and chunk length 15
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_m15(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_m<768, 15, 2>(a, mode, stream); }
}  // namespace icnv
