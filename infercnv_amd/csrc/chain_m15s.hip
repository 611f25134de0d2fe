// chain kernel variants with 768 threads, chunk length 15 and SEVEN gene-pair slots per thread: gene sets of at most
// 768 * 7 * 2 = 10 752 (even) genes -- the S-layout phases (steps 8, 9, median, 12, 14, 22, loads and stores) run over
// the genes, not over the padded positions of the chunk layout, and an eighth slot would only repeat the last pair
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_m15s(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_m<768, 15, 2, 7, 0>(a, mode, stream); }
}  // namespace icnv
