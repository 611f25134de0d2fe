// The 1024 x 11 chain geometry on a STRIDED VIEW of the matrix (a group of whole chromosomes of every cell): steps 8, 9, 10 only --
// pass 1 of the two-pass chain for gene sets beyond the fused kernel's LDS-resident limit (api.hip: large_smooth_center; round 6:
// replaces the three-pass chain's (2T + 1)-tap pass, 14 x the 10 000-gene per-cell cost, R/inferCNV_ops.R:2406-2532).
#include <cmath>

#include "chain_kernel.inc"

namespace icnv {
bool chain_view_fits(int64_t G_view, int32_t n_chr_view, int32_t T) {
    const int64_t pad = T >= 1 ? ((T + 3) & ~1) : 0;
    return G_view >= 4 && G_view <= 1024 * 5 * 2 && G_view + (int64_t)(n_chr_view + 1) * pad <= 1024 * 11 && n_chr_view <= 510 && !(T >= 1 && T / 11 > 64);
}
int launch_chain_strided(const ChainArgs &a0, hipStream_t stream) {
    ChainArgs a = a0;
    if (a.ld < a.G || !(a.mask & ICNV_ST_SMOOTH) || a.T < 1 || (a.mask & ~(uint32_t)(ICNV_ST_SUBTRACT_REF_1 | ICNV_ST_MAX_THRESH | ICNV_ST_SMOOTH)))
        ICNV_FAIL(ICNV_ERR_ARG, "strided chain launch: steps 8 / 9 / 10 with step 10, leading dimension >= genes of the view");
    if (!chain_view_fits(a.G, a.n_chr, a.T)) ICNV_FAIL(ICNV_ERR_ARG, "strided chain launch: the view does not fit the 1024 x 11 geometry");
    if (a.mask & ICNV_ST_MAX_THRESH) {
        if (a.max_thresh != a.max_thresh) ICNV_FAIL(ICNV_ERR_ARG, "apply_max_threshold_bounds: the threshold is NaN");
        if (std::isinf(a.max_thresh)) a.mask &= ~(uint32_t)ICNV_ST_MAX_THRESH;
    }
    if (!(a.inv_pos && a.inv_codes && a.inv_dict)) ICNV_FAIL(ICNV_ERR_ARG, "smoothing launch without its normalisation table");
    a.pad = (a.T + 3) & ~1;
    // the compile-time form for run()'s case (steps 8 + 9 + 10, window_length 101, a coded normalisation table); the generic kernel otherwise
    if (a.mask == 0x07u && a.T == 50 && a.inv_coded) return launch_chain_t<1024, 11, 2, MODE_APPLY, 0x07, 5, 50, true>(a, stream);
    return launch_chain_t<1024, 11, 2, MODE_APPLY, -1, 5, 0, true>(a, stream);
}
}  // namespace icnv
