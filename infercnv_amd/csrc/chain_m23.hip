// chain kernel variants with 768 threads (3 wavefronts per SIMD, 168 VGPRs) and chunk length 23
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_m23(const ChainArgs &a, int mode, hipStream_t stream) {
    // even gene counts up to 768 * 11 * 2: eleven gene-pair slots instead of the twelve the chunk length would give
    if (a.G <= 768 * 11 * 2) return launch_chain_m<768, 23, 2, 11, 0>(a, mode, stream);
    return launch_chain_m<768, 23, 2>(a, mode, stream);
}
}  // namespace icnv
