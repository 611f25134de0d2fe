// 2-D median denoise (apply_median_filtering / .median_filter,
// R/noise_reduction.R:43-113) for gfx950.
//
// A workgroup produces a 32-gene x 8-cell patch of one (tile, chromosome)
// (interior outputs of the default window_size 7 take the sorted-column network path, see below)
// block: the (32+2h) x (8+2h) input patch (h = half_window+1, so the effective
// window is (window_size+2)^2, clamped at the block's edges) is gathered
// through the tile's cell-index vector into LDS; every thread then selects the
// median of its clamped window by a value-bounded quickselect (exact order
// statistics; even counts average the two middle values like stats::median).
#include "icnv_internal.h"
#include "median9x9_net.h"

namespace icnv {

namespace {

constexpr int MF_TG = 32;  // genes per patch
constexpr int MF_TC = 8;   // cells per patch
constexpr int MF_MAXH = 8; // supports window_size <= 15

template <bool FAST9>
__global__ void __launch_bounds__(MF_TG *MF_TC) median_filter_kernel(
    const double *__restrict__ in, double *__restrict__ out, int G, const int32_t *__restrict__ chr_start,
    const int32_t *__restrict__ tile_idx, const int32_t *__restrict__ tile_off, int n_tiles,
    const int32_t *__restrict__ blk_off /* n_tiles+1 prefix of cell-patches per tile */, int h, int bz_base,
    int n_chr) {
    extern __shared__ __attribute__((aligned(16))) double patch[];  // [(MF_TC+2h)][(MF_TG+2h)]
    // blockIdx.x enumerates the 32-gene blocks of all chromosomes back to back (no empty blocks):
    // chromosome k owns ceil(n_k / 32) consecutive blocks
    int chr = 0, gblk_base = 0;
    for (int k = 0; k < n_chr; ++k) {
        const int nb = (chr_start[k + 1] - chr_start[k] + MF_TG - 1) / MF_TG;
        if ((int)blockIdx.x < gblk_base + nb) { chr = k; break; }
        gblk_base += nb;
    }
    const int cs = chr_start[chr], xdim = chr_start[chr + 1] - cs;
    const int g0 = ((int)blockIdx.x - gblk_base) * MF_TG;
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
    if constexpr (FAST9) {
        // window_size 7 -> 9 x 9 windows.  Stage A: every (output gene, patch cell) pair gets its nine
        // values along the genes sorted once (25 compare-exchanges) and shared through LDS by the nine
        // outputs whose window contains it.  Stage B (interior outputs): the exact median of nine sorted
        // columns by pruned odd-even merge networks (median9x9_net.h, 686 min/max, no branches).
        double *sortedc = patch + PW * PH;                       // [PH][MF_TG][9]
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int c = ty + half * MF_TC;
            double v[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) v[k] = patch[c * PW + tx + k];
            ICNV_SORT9(v);
#pragma unroll
            for (int k = 0; k < 9; ++k) sortedc[(c * MF_TG + tx) * 9 + k] = v[k];
        }
        __syncthreads();
        if (gx >= xdim || cy >= ydim) return;
        if (gx - 4 >= 0 && gx + 4 <= xdim - 1 && cy - 4 >= 0 && cy + 4 <= ydim - 1) {
            double a[81];
#pragma unroll
            for (int c = 0; c < 9; ++c)
#pragma unroll
                for (int k = 0; k < 9; ++k) a[9 * c + k] = sortedc[((ty + c) * MF_TG + tx) * 9 + k];
            out[(int64_t)idx[cy] * G + cs + gx] = median81_sorted_columns(a);
            return;
        }
        // Border output: its clamped window (R/noise_reduction.R:101-106) holds m < 81 values.  The
        // missing positions are padded with n_lo x -inf and the rest +inf so that the wanted order
        // statistics of the real values sit at ranks 40 (and 41 for an even m) of the padded 81; the
        // output sorts its own nine columns and runs the same branch-free network.
        {
            const int xa = (gx - 4 < 0 ? 0 : gx - 4) - (g0 - 4), xb = (gx + 4 > xdim - 1 ? xdim - 1 : gx + 4) - (g0 - 4);
            const int ya = (cy - 4 < 0 ? 0 : cy - 4) - (c0 - 4), yb = (cy + 4 > ydim - 1 ? ydim - 1 : cy + 4) - (c0 - 4);
            const int m = (xb - xa + 1) * (yb - ya + 1);
            const int n_lo = (m & 1) ? (81 - m) / 2 : 41 - m / 2;
            int npad = 0;
            double a[81];
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const int yy = ty + c;
                double v[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int xx = tx + k;
                    double val = patch[yy * PW + xx];
                    if (!(yy >= ya && yy <= yb && xx >= xa && xx <= xb)) {
                        val = (npad < n_lo) ? -__builtin_inf() : __builtin_inf();
                        ++npad;
                    }
                    v[k] = val;
                }
                ICNV_SORT9(v);
#pragma unroll
                for (int k = 0; k < 9; ++k) a[9 * c + k] = v[k];
            }
            double r40, r41;
            median81_pair_sorted_columns(a, r40, r41);
            out[(int64_t)idx[cy] * G + cs + gx] = (m & 1) ? r40 : (r40 + r41) * 0.5;
            return;
        }
    }
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
    int gene_blocks = 0;
    for (int k = 0; k < n_chr; ++k) gene_blocks += (chr_start_host[k + 1] - chr_start_host[k] + MF_TG - 1) / MF_TG;
    if (gene_blocks <= 0) return ICNV_OK;
    if (n_chr > 65535) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "median filter: more than 65535 chromosomes");
    const bool fast9 = (h == 4);
    const size_t lds = ((size_t)(MF_TG + 2 * h) * (MF_TC + 2 * h) + (fast9 ? (size_t)(MF_TC + 2 * h) * MF_TG * 9 : 0)) *
                       sizeof(double);
    KernelTimer kt("median_filter", stream);
    for (int base = 0; base < total_cell_patches; base += 32768) {
        const int nz = (total_cell_patches - base) < 32768 ? (total_cell_patches - base) : 32768;
        const dim3 grid(gene_blocks, 1, nz);
        if (fast9)
            hipLaunchKernelGGL(median_filter_kernel<true>, grid, dim3(MF_TG * MF_TC), lds, stream, in, out, G, chr_start_dev,
                               tile_idx_dev, tile_off_dev, n_tiles, blk_off_dev, h, base, n_chr);
        else
            hipLaunchKernelGGL(median_filter_kernel<false>, grid, dim3(MF_TG * MF_TC), lds, stream, in, out, G,
                               chr_start_dev, tile_idx_dev, tile_off_dev, n_tiles, blk_off_dev, h, base, n_chr);
    }
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

}  // namespace icnv
