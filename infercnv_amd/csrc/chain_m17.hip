// chain kernel variants with 768 threads (3 wavefronts per SIMD, 168 VGPRs), chunk length 17 and -- for even gene counts --
// 8 gene-pair slots per thread (at most 768 * 8 * 2 genes): gene sets between the 10 000-gene geometries and 768 x 23.
// The chunk-layout phases cost a thread its chunk length, the S-layout phases its slots: a cell of 11 000 genes on the
// 768 x 23 geometry (12 slots) did 40 % of both for nothing (profiles/r05_sweep.json).
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_m17(const ChainArgs &a, int mode, hipStream_t stream) {
    return launch_chain_m<768, 17, 2, 8, 0>(a, mode, stream);   // (even and odd gene counts: the pair layout)
}
}  // namespace icnv
