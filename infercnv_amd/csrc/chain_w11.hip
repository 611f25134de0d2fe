// chain kernel variants with 1024 threads (4 wavefronts per SIMD, 128 VGPRs), chunk length 11 and five gene-pair
// slots per thread: at most 11 264 padded positions and 10 240 (even) genes per cell -- the 10 000-gene workload with
// the default window.  Sixteen wavefronts walk a cell through its barrier-separated phases faster than twelve.
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_w11(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_m<1024, 11, 2, 5, 0>(a, mode, stream); }
}  // namespace icnv
