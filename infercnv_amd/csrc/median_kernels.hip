// 2-D median denoise (apply_median_filtering / .median_filter,
// R/noise_reduction.R:43-113) for gfx950.
//
// A workgroup produces a 32-gene x 8-cell patch of one (tile, chromosome)
// block: the (32+2h) x (8+2h) input patch (h = half_window+1, so the effective
// window is (window_size+2)^2, clamped at the block's edges) is gathered
// through the tile's cell-index vector into LDS; every thread then selects the
// median of its clamped window by a value-bounded quickselect (exact order
// statistics; even counts average the two middle values like stats::median).
#include "icnv_internal.h"

namespace icnv {

namespace {

constexpr int MF_TG = 32;  // genes per patch
constexpr int MF_TC = 8;   // cells per patch
constexpr int MF_MAXH = 8; // supports window_size <= 15

__global__ void __launch_bounds__(MF_TG *MF_TC) median_filter_kernel(
    const double *__restrict__ in, double *__restrict__ out, int G, const int32_t *__restrict__ chr_start,
    const int32_t *__restrict__ tile_idx, const int32_t *__restrict__ tile_off, int n_tiles,
    const int32_t *__restrict__ blk_off /* n_tiles+1 prefix of cell-patches per tile */, int h, int bz_base) {
    extern __shared__ __attribute__((aligned(16))) double patch[];  // [(MF_TC+2h)][(MF_TG+2h)]
    const int chr = blockIdx.y;
    const int cs = chr_start[chr], xdim = chr_start[chr + 1] - cs;
    const int g0 = blockIdx.x * MF_TG;
    if (g0 >= xdim) return;
    // tile of this block (binary search in the per-tile patch prefix)
    int lo = 0, hi = n_tiles - 1;
    const int bz = bz_base + blockIdx.z;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blk_off[mid] <= bz) lo = mid; else hi = mid - 1;
    }
    const int tile = lo;
    const int32_t *idx = tile_idx + tile_off[tile];
    const int ydim = tile_off[tile + 1] - tile_off[tile];
    const int c0 = (bz - blk_off[tile]) * MF_TC;

    const int PW = MF_TG + 2 * h;  // patch width (genes)
    const int PH = MF_TC + 2 * h;  // patch height (cells)
    for (int e = threadIdx.x; e < PW * PH; e += blockDim.x) {
        const int py = e / PW, px = e - py * PW;
        const int gx = g0 - h + px, cy = c0 - h + py;
        double v = 0.0;
        if (gx >= 0 && gx < xdim && cy >= 0 && cy < ydim) v = in[(int64_t)idx[cy] * G + cs + gx];
        patch[e] = v;
    }
    __syncthreads();

    const int tx = threadIdx.x % MF_TG, ty = threadIdx.x / MF_TG;
    const int gx = g0 + tx, cy = c0 + ty;
    if (gx >= xdim || cy >= ydim) return;
    // clamped window (R/noise_reduction.R:101-106), in patch coordinates
    const int xa = (gx - h < 0 ? 0 : gx - h) - (g0 - h), xb = (gx + h > xdim - 1 ? xdim - 1 : gx + h) - (g0 - h);
    const int ya = (cy - h < 0 ? 0 : cy - h) - (c0 - h), yb = (cy + h > ydim - 1 ? ydim - 1 : cy + h) - (c0 - h);
    const int m = (xb - xa + 1) * (yb - ya + 1);
    const int r_hi = m >> 1, r_lo = (m & 1) ? r_hi : r_hi - 1;
    // Exact selection of rank r_lo by value-bounded quickselect: every pass counts the window's
    // elements below / equal to the pivot and picks the next pivot on both sides, so a pass is the
    // only per-iteration cost (about 2 ln m passes instead of m for plain rank counting).
    double blo = -__builtin_inf(), bhi = __builtin_inf();     // the answer lies strictly between
    double pivot = patch[((ya + yb) >> 1) * PW + ((xa + xb) >> 1)];
    double v_lo = pivot, v_hi = pivot;
    for (int iter = 0; iter <= m; ++iter) {
        int c_lt = 0, c_eq = 0;
        double next_lo = blo, next_hi = bhi;      // candidate pivots inside (lo, pivot) and (pivot, hi)
        double above = __builtin_inf();         // smallest element greater than the pivot
        for (int yy = ya; yy <= yb; ++yy)
            for (int xx = xa; xx <= xb; ++xx) {
                const double o = patch[yy * PW + xx];
                if (o < pivot) { ++c_lt; if (o > blo) next_lo = o; }
                else if (o == pivot) ++c_eq;
                else { above = fmin(above, o); if (o < bhi) next_hi = o; }
            }
        if (r_lo < c_lt) { bhi = pivot; pivot = next_lo; }
        else if (r_lo < c_lt + c_eq) {
            v_lo = pivot;
            v_hi = (r_hi < c_lt + c_eq) ? pivot : above;   // upper middle: same value or the next one up
            break;
        } else { blo = pivot; pivot = next_hi; }
    }
    out[(int64_t)idx[cy] * G + cs + gx] = (m & 1) ? v_lo : (v_lo + v_hi) * 0.5;
}

}  // namespace

int launch_median_filter(const double *in, double *out, int32_t G, int64_t C, const int32_t *chr_start_dev,
                         int32_t n_chr, const int32_t *tile_idx_dev, const int32_t *tile_off_dev, int32_t n_tiles,
                         const int32_t *blk_off_dev, const int32_t *chr_start_host, int32_t total_cell_patches,
                         int32_t window_size, hipStream_t stream) {
    (void)C;
    if (n_tiles <= 0 || n_chr <= 0 || total_cell_patches <= 0) return ICNV_OK;
    const int h = (window_size - 1) / 2 + 1;
    if (h > MF_MAXH) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "median filter supports window_size <= 15");
    int maxlen = 0;
    for (int k = 0; k < n_chr; ++k) {
        const int n = chr_start_host[k + 1] - chr_start_host[k];
        if (n > maxlen) maxlen = n;
    }
    if (n_chr > 65535) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "median filter: more than 65535 chromosomes");
    const size_t lds = (size_t)(MF_TG + 2 * h) * (MF_TC + 2 * h) * sizeof(double);
    KernelTimer kt("median_filter", stream);
    for (int base = 0; base < total_cell_patches; base += 32768) {
        const int nz = (total_cell_patches - base) < 32768 ? (total_cell_patches - base) : 32768;
        hipLaunchKernelGGL(median_filter_kernel, dim3((maxlen + MF_TG - 1) / MF_TG, n_chr, nz), dim3(MF_TG * MF_TC), lds,
                           stream, in, out, G, chr_start_dev, tile_idx_dev, tile_off_dev, n_tiles, blk_off_dev, h, base);
    }
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

}  // namespace icnv
