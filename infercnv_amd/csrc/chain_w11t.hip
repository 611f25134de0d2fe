// chain_w11.hip's geometry with the half window fixed at compile time: 50 = window_length 101, the default of
// infercnv::run() (R/inferCNV_ops.R:251).  Only the passes that smooth exist in this form.
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_w11t(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_m<1024, 11, 2, 5, 50>(a, mode, stream); }
}  // namespace icnv

#ifdef ICNV_CHAIN_PROFILE
extern "C" int icnv_debug_chain_profile(unsigned long long *out32, int reset) {
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(icnv::g_chain_prof), 32 * sizeof(unsigned long long)) != hipSuccess) return 2;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(icnv::g_chain_prof), z, sizeof(z)) != hipSuccess) return 2;
    }
    return 0;
}
#endif
