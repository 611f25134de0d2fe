// The two-cells-per-CU chain kernels (variants/chain2.hip, round 3) are a measured NEGATIVE result -- 3.0-3.4 ms against the
// product kernel's 2.3 ms (docs/KERNEL_LOG.md, profiles/r03_chain2_*) -- and are not part of libicnv_hip.so.  The product
// library links these two stubs; `make -C infercnv_amd/csrc chain2-variant` builds ../libicnv_hip_chain2.so with the real
// translation unit in their place (load it with ICNV_LIB=..., switch the kernels on with ICNV_CHAIN2=1: the chain2 tests
// do: `build()` builds the variant, the CPU plan tests load it directly, the GPU tests run themselves in a process that
// loads it).
#include "icnv_internal.h"

namespace icnv {

bool chain2_build_plan(const int32_t *, int32_t, int32_t, int32_t, std::vector<uint32_t> &plan, std::vector<double> &dict) {
    plan.clear();
    dict.clear();
    return false;   // no plan: launch_chain never asks for these kernels
}

int launch_chain2(const ChainArgs &, int, hipStream_t) { return -1000; }   // "not built for this pass": the caller goes on

}  // namespace icnv
