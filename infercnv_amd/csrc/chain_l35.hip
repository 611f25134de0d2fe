// chain kernel variants with 512 threads and chunk length 35 (<= 17920 padded positions per cell: the largest
// cell vector that still leaves room for the median's histogram and candidate buffers in the 160 KiB LDS)
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_l35(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_m<512, 35, 2>(a, mode, stream); }
}  // namespace icnv
