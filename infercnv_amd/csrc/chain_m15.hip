// chain kernel variants with 768 threads (3 wavefronts per SIMD, 168 VGPRs) and chunk length 15
#include "chain_kernel.inc"

namespace icnv {
int launch_chain_m15(const ChainArgs &a, int mode, hipStream_t stream) { return launch_chain_v<768, 15>(a, mode, stream); }
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
