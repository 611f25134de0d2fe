// chain_m15s.hip's geometry with the half window fixed at compile time: 50 = window_length 101, the default of
// infercnv::run() (R/inferCNV_ops.R:251) -- the window initialisation of the smoothing unrolls completely.
// Only the passes that smooth exist in this form.
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_m15t(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_m<768, 15, 2, 7, 50>(a, mode, stream); }
}  // namespace icnv
