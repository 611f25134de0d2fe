// 2-D median denoise (apply_median_filtering / .median_filter,
// R/noise_reduction.R:43-113) for gfx950.
//
// A workgroup produces a patch of 32 genes x 8 (16 for the default window) cells of one (tile, chromosome)
// block: the (32+2h) x (cells+2h) input patch (h = half_window+1, so the effective
// window is (window_size+2)^2, clamped at the block's edges) is gathered
// through the tile's cell-index vector into LDS.
//   window_size 7 (9 x 9 windows, the default): median_filter9_kernel -- sorted columns shared through LDS,
//     two outputs per thread that share eight of their nine columns, branch-free min/max networks;
//   other window sizes: median_filter_kernel -- every thread selects the median of its clamped window by a
//     value-bounded quickselect (exact order statistics; even counts average the two middle values like
//     stats::median).
#include "icnv_internal.h"

// v_min_f64 / v_max_f64 without the compiler's canonicalisation of operands that come straight from memory (one
// extra v_max_f64 x, x per loaded value: +13 % on the 9 x 9 networks); the instructions themselves return the
// non-NaN operand like fmin / fmax
__device__ static inline double icnv_mf_min(double x, double y) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ static inline double icnv_mf_max(double x, double y) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
#ifdef ICNV_MF_RAW_MINMAX   // measured: the asm statements cost the register allocator one occupancy step (3.15 vs 2.14 ms)
#define ICNV_FMIN(a, b) icnv_mf_min(a, b)
#define ICNV_FMAX(a, b) icnv_mf_max(a, b)
#endif
#include "median9x9_net.h"

namespace icnv {

namespace {

constexpr int MF_TG = 32;  // genes per patch
constexpr int MF_TC = 8;   // cells per patch
constexpr int MF_MAXH = 8; // supports window_size <= 15
constexpr int MF9_TC = 16; // cells per patch of the 9 x 9 kernel: two per thread

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


// Majority shortcut (exact, data-dependent).  A window in which one value occupies more than half of the positions has that
// value as its median, whatever the rest is.  That is the common case on the matrix this filter is made for:
// apply_median_filtering runs on the DENOISED matrix, where clear_noise_via_ref_mean_sd (R/inferCNV_ops.R:2302-2346) has set
// every entry inside the noise band -- typically 80-90 % of them -- to one and the same value mu.  The kernels count, per
// output, the window positions equal to a candidate value (row masks of the patch through ballots / LDS atomics, one
// popcount per window row); an output whose count exceeds half its window takes the candidate, a wavefront whose outputs all
// do skips its networks, a patch whose outputs all do skips the column sorts as well.  Any candidate is correct (the test is
// exact); a good one is the last value that won, re-seeded from the patch itself when it fails.
//
// One patch of the 9 x 9 interior kernel: stage A (column sorts into LDS), then the outputs of thread (tx, ty).  The
// patch holds interior outputs only: genes [g0, gend) with gend <= xdim - 4, cells [c0, cend) with c0 >= 4, cend <= ydim - 4.
// `majA` / `majB`: this thread's outputs are decided by the majority value `vmaj`; `wave_skip`: so are all outputs of its wavefront.
__device__ inline void median9_patch(int cs, int g0, int gend, int c0, int cend, const int32_t *rows /* LDS: cell of patch row r */,
                                     int tx, int ty, const double *patch, double *sortedc, double *__restrict__ out, int G,
                                     bool majA, bool majB, bool wave_skip, double vmaj) {
    constexpr int h = 4;
    constexpr int PW = MF_TG + 2 * h;
    constexpr int PH = MF9_TC + 2 * h;
#pragma unroll
    for (int q = 0; q < PH / MF_TC; ++q) {
        const int c = ty + q * MF_TC;
        double v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = patch[c * PW + tx + k];
        ICNV_SORT9(v);
#pragma unroll
        for (int k = 0; k < 9; ++k) sortedc[(c * MF_TG + tx) * 9 + k] = v[k];
    }
    __syncthreads();
    const int gx = g0 + tx;
    const int r0 = 2 * ty;                       // patch row of the first column of output A's window
    const int cyA = c0 + r0, cyB = cyA + 1;
    if (wave_skip) {                             // (wave-uniform) every output of this wavefront is its window's majority value
        if (gx < gend && cyA < cend) out[(int64_t)rows[r0 + 4] * G + cs + gx] = vmaj;
        if (gx < gend && cyB < cend) out[(int64_t)rows[r0 + 5] * G + cs + gx] = vmaj;
        return;
    }
    if (gx >= gend) return;
    if (cyA >= cend) return;
    if (cyB < cend) {
        double w[10];
        {
            double s[72];
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int k = 0; k < 9; ++k) s[9 * c + k] = sortedc[((r0 + 1 + c) * MF_TG + tx) * 9 + k];
            median72_window(s, w);
        }
        double p[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) p[k] = sortedc[(r0 * MF_TG + tx) * 9 + k];
        const double mA = median_window_finish(w, p);
        out[(int64_t)rows[r0 + 4] * G + cs + gx] = majA ? vmaj : mA;
#pragma unroll
        for (int k = 0; k < 9; ++k) p[k] = sortedc[((r0 + 9) * MF_TG + tx) * 9 + k];
        const double mB = median_window_finish(w, p);
        out[(int64_t)rows[r0 + 5] * G + cs + gx] = majB ? vmaj : mB;
        return;
    }
    // the last interior cell of a tile with an odd number of them: the single-output network over its nine columns
    double a[81];
#pragma unroll
    for (int c = 0; c < 9; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) a[9 * c + k] = sortedc[((r0 + c) * MF_TG + tx) * 9 + k];
    out[(int64_t)rows[r0 + 4] * G + cs + gx] = median81_sorted_columns(a);
}

// The value held by at least two of three probes (else the first): a cheap guess at a patch's most common value.
__device__ inline double majority_of_three(double a, double b, double c) { return (b == c) ? b : a; }

// Border outputs of the 9 x 9 filter (clamped windows, m < 81 values; R/noise_reduction.R:101-106), one per thread,
// no interior outputs in the workgroup: the missing positions are padded with n_lo x -inf and +inf so that the wanted
// order statistics of the real values sit at ranks 40 (and 41 for an even m: stats::median averages the two middle
// values) of the padded 81; the thread sorts its own nine columns and runs the two-rank variant of the single-output
// network.  Branch-free like the interior path; kept apart from it because a wavefront that mixes the two pays for both
// (border outputs are 3.4 % of a 500-cell x 450-gene block but cost 3 x an interior output).
//   mode 0: the border GENES of one chromosome (first and last four; all genes of a chromosome shorter than nine) for 32
//           consecutive cells of a tile, corners included;  b0 = first cell
//   mode 1: the border CELLS of one tile (first and last four; all cells of a tile smaller than nine) for 32 consecutive
//           interior genes [4, xdim - 4) of a chromosome;  b0 = first gene
__global__ void __launch_bounds__(256, 2) median_filter9_edge_kernel(const double *__restrict__ in, double *__restrict__ out, int G,
                                                                     const int32_t *__restrict__ tile_idx,
                                                                     const int4 *__restrict__ item_desc) {
    __shared__ double ep[40 * 17];   // mode 0 rows are padded to 17 doubles: a lane's cell is its row, 16 would put all lanes on one bank
    const int4 d0 = item_desc[2 * blockIdx.x], d1 = item_desc[2 * blockIdx.x + 1];
    const int mode = d0.x, cs = d0.y, xdim = d0.z, ydim = d1.x, b0 = d1.y;
    const int32_t *idx = tile_idx + d0.w;
    const int t = threadIdx.x;
    int g, cy;          // this thread's output (chromosome-relative gene, tile-relative cell)
    bool active;
    if (mode == 0) {    // LDS: 40 cells (b0 - 4 ..) x 16 genes (0..7 | xdim-8 .. xdim-1)
        for (int e = t; e < 640; e += 256) {
            const int row = e >> 4, col = e & 15;
            const int c = b0 - 4 + row, gx = col < 8 ? col : xdim - 16 + col;
            ep[row * 17 + col] = (c >= 0 && c < ydim && gx >= 0 && gx < xdim) ? in[(int64_t)idx[c] * G + cs + gx] : 0.0;
        }
        const int og = t >> 5;
        cy = b0 + (t & 31);
        g = (xdim >= 9 && og >= 4) ? xdim - 8 + og : og;
        active = cy < ydim && (xdim >= 9 || og < xdim);
    } else {            // LDS: 16 cells (0..7 | ydim-8 .. ydim-1) x 40 genes (b0 - 4 ..)
        for (int e = t; e < 640; e += 256) {
            const int row = e / 40, col = e - row * 40;
            const int c = row < 8 ? row : ydim - 16 + row, gx = b0 - 4 + col;
            ep[e] = (c >= 0 && c < ydim && gx >= 0 && gx < xdim) ? in[(int64_t)idx[c] * G + cs + gx] : 0.0;
        }
        const int oc = t >> 5;
        g = b0 + (t & 31);
        cy = (ydim >= 9 && oc >= 4) ? ydim - 8 + oc : oc;
        active = g < xdim - 4 && (ydim >= 9 || oc < ydim);
    }
    __syncthreads();
    if (!active) return;
    const int xa = g - 4 < 0 ? 0 : g - 4, xb = g + 4 > xdim - 1 ? xdim - 1 : g + 4;
    const int ya = cy - 4 < 0 ? 0 : cy - 4, yb = cy + 4 > ydim - 1 ? ydim - 1 : cy + 4;
    const int m = (xb - xa + 1) * (yb - ya + 1);
    const int n_lo = (m & 1) ? (81 - m) / 2 : 41 - m / 2;
    // LDS position of (gene xx, cell yy) of this thread's window: left / right (upper / lower) group of its mode
    int base, sx, sy;
    if (mode == 0) {
        sx = 1; sy = 17;
        base = ((g < 4 || xdim < 9) ? 0 : 16 - xdim) - (b0 - 4) * 17;
    } else {
        sx = 1; sy = 40;
        base = ((cy < 4 || ydim < 9) ? 0 : (16 - ydim) * 40) - (b0 - 4);
    }
    {   // majority shortcut (see above): a value on more than half of the window's m positions is its median (for an even m both
        // middle values are that value); a wavefront whose outputs are all decided this way skips its networks
        const double vc = mode == 0 ? majority_of_three(ep[8 * 17 + 2], ep[20 * 17 + 10], ep[30 * 17 + 5])
                                    : majority_of_three(ep[2 * 40 + 10], ep[5 * 40 + 20], ep[12 * 40 + 30]);
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const int yy = cy - 4 + c;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int xx = g - 4 + k;
                const bool in_win = yy >= ya && yy <= yb && xx >= xa && xx <= xb;
                cnt += (in_win && ep[in_win ? base + yy * sy + xx * sx : 0] == vc) ? 1 : 0;
            }
        }
        const bool maj = 2 * cnt > m;
        if (__ballot(!maj) == 0ull) {
            out[(int64_t)idx[cy] * G + cs + g] = vc;
            return;
        }
    }
    int npad = 0;
    double a[81];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int yy = cy - 4 + c;
        double v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int xx = g - 4 + k;
            const bool in_win = yy >= ya && yy <= yb && xx >= xa && xx <= xb;
            double val = in_win ? ep[base + yy * sy + xx * sx] : 0.0;
            if (!in_win) {
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
    out[(int64_t)idx[cy] * G + cs + g] = (m & 1) ? r40 : (r40 + r41) * 0.5;
}

// window_size 7 -> 9 x 9 windows.
//   Stage A: every (output gene, patch cell) pair gets its nine values along the genes sorted once (25
//     compare-exchanges) and shared through LDS by the nine outputs whose window contains it.
//   Stage B: a thread owns the outputs of two neighbouring cells.  Their windows share eight of the nine sorted
//     columns: positions 31..40 of the merged 72 shared values -- the only ones that can be the median of 72 + 9 --
//     come from one pruned odd-even merge network per PAIR (668 min/max), and each output finishes with its own
//     column (18 min/max): 352 min/max per output instead of the 686 of a network per output (median9x9_net.h,
//     generated and verified by gen_median_net.py; no branches, no data-dependent loops).
//   This kernel produces the INTERIOR outputs only (gene >= 4 from either end of the chromosome, cell >= 4 from either
//     end of the tile); median_filter9_edge_kernel above produces the rest.
__global__ void __launch_bounds__(MF_TG *MF_TC, 2) median_filter9_kernel(
    const double *__restrict__ in, double *__restrict__ out, int G, const int32_t *__restrict__ tile_idx,
    const int4 *__restrict__ gene_block_desc /* {chromosome's first gene, its length, block's first gene, end of its interior genes} */,
    const int4 *__restrict__ cell_patch_desc /* {offset of the tile's cell list, tile length, patch's first cell, end of its interior cells} */,
    int gene_blocks, int64_t n_patches) {
    constexpr int h = 4;
    constexpr int PW = MF_TG + 2 * h;    // patch width (genes)
    constexpr int PH = MF9_TC + 2 * h;   // patch height (cells)
    constexpr int NT = MF_TG * MF_TC;
    constexpr int EPT = (PW * PH + NT - 1) / NT;   // patch elements per thread
    extern __shared__ __attribute__((aligned(16))) double patch2[];  // 2 x [PH][PW], then the sorted columns [PH][MF_TG][9]
    double *sortedc = patch2 + 2 * PW * PH;
    // Persistent workgroups walk the patches (gene block fastest).  Two things keep the memory latency off the
    // critical path: a patch is described by two 16-byte records built on the host (one scalar load each instead of
    // a chromosome scan and a binary search: ~10 dependent loads), requested TWO patches ahead; and the NEXT patch's
    // values are requested into registers before the current patch's networks run and parked in LDS afterwards, so the
    // gather (indirect rows through the tile's cell index) hides behind ~1 000 min/max instead of standing between
    // barriers.
    struct Where { int cs, xdim, g0, gend, ydim, c0, cend; const int32_t *idx; };
    auto where = [&](const int4 gd, const int4 cd) {
        Where w;
        w.cs = gd.x; w.xdim = gd.y; w.g0 = gd.z; w.gend = gd.w;
        w.idx = tile_idx + cd.x; w.ydim = cd.y; w.c0 = cd.z; w.cend = cd.w;
        return w;
    };
    // cell index of every patch row, double-buffered in LDS: loaded by PH threads two patches ahead, so neither the
    // gather nor the output stores wait for an index load
    // The patch is double-buffered and the row table triple-buffered: a wavefront that runs ahead into the next patch
    // parks its values and writes the row table of the patch after it while the slowest wavefront still reads the
    // current ones, so a patch costs two barriers (patch parked | columns sorted) instead of three.
    int32_t *rowbuf = reinterpret_cast<int32_t *>(sortedc + PH * MF_TG * 9);   // [3][PH]
    // majority shortcut: per patch buffer, one 64-bit mask per patch row (bit g: the value at gene g of the row equals the
    // candidate) and a word of flags (bit 0: some output needs its network, bit 1: some output took the candidate)
    unsigned long long *rowmask = reinterpret_cast<unsigned long long *>(rowbuf + 3 * PH + (PH & 1));   // [2][PH], 8-byte aligned
    unsigned int *mflag = reinterpret_cast<unsigned int *>(rowmask + 2 * PH);                            // [2]
    auto load_rows = [&](const Where &w) -> int32_t {
        const int cy = w.c0 - h + (int)threadIdx.x;
        return ((int)threadIdx.x < PH && cy >= 0 && cy < w.ydim) ? w.idx[cy] : 0;
    };
    double stage[EPT];
    auto gather = [&](const Where &w, const int32_t *rows) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = (int)threadIdx.x + q * NT;
            const int py = e / PW, px = e - py * PW;
            const int gx = w.g0 - h + px, cy = w.c0 - h + py;
            double v = 0.0;
            if (e < PW * PH && gx >= 0 && gx < w.xdim && cy >= 0 && cy < w.ydim) v = in[(int64_t)rows[py] * G + w.cs + gx];
            stage[q] = v;
        }
    };
    const int tx = threadIdx.x % MF_TG, ty = threadIdx.x / MF_TG;
    const int64_t step = gridDim.x;
    int64_t pid = blockIdx.x;
    if (pid >= n_patches) return;
    auto desc = [&](int64_t p, int4 &gd, int4 &cd) {
        gd = gene_block_desc[p % gene_blocks];
        cd = cell_patch_desc[p / gene_blocks];
    };
    int4 gd, cd;
    desc(pid, gd, cd);
    Where cur = where(gd, cd);
    if ((int)threadIdx.x < PH) rowbuf[threadIdx.x] = load_rows(cur);
    if ((int)threadIdx.x < 2 * PH) rowmask[threadIdx.x] = 0ull;
    if ((int)threadIdx.x < 2) mflag[threadIdx.x] = 0u;
    __syncthreads();
    gather(cur, rowbuf);
    // first candidate: an element from the middle of the first patch (wave-uniform address); re-seeded below when it fails
    double vguess = in[(int64_t)rowbuf[PH / 2] * G + cur.cs + cur.g0];
    const int lane = threadIdx.x & 63;
    Where nxt = cur;                 // patch pid + step
    int32_t nxt_row = 0;             // its row table entry of this thread
    int4 gd2 = gd, cd2 = cd;         // descriptors of patch pid + 2 step
    if (pid + step < n_patches) {
        desc(pid + step, gd, cd);
        nxt = where(gd, cd);
        nxt_row = load_rows(nxt);
    }
    if (pid + 2 * step < n_patches) desc(pid + 2 * step, gd2, cd2);
    int it = 0;   // patch counter of this workgroup: patch buffer it & 1, row table it % 3
    for (; pid < n_patches; pid += step, ++it) {
        const bool more = pid + step < n_patches;
        double *patch = patch2 + (it & 1) * (PW * PH);
        const int rb = it % 3, rb_next = (it + 1) % 3;
        unsigned long long *rmask = rowmask + (it & 1) * PH;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = (int)threadIdx.x + q * NT;
            if (e < PW * PH) patch[e] = stage[q];
            // which of the wavefront's 64 consecutive patch elements equal the candidate: one ballot, then at most three
            // lanes add the bits of the (up to three) patch rows those elements lie in to the rows' masks
            const unsigned long long m = __ballot(stage[q] == vguess);
            const int e0 = ((int)threadIdx.x & ~63) + q * NT;
            const int r = e0 / PW + lane;
            if (lane < 3 && r < PH) {
                const int lo = r * PW - e0;                  // element (relative to e0) that sits at gene 0 of row r
                unsigned long long bits = lo >= 64 ? 0ull : (lo >= 0 ? (m >> lo) : (m << (-lo)));
                bits &= (1ull << PW) - 1ull;
                if (bits) atomicOr(&rmask[r], bits);
            }
        }
        if (more && (int)threadIdx.x < PH) rowbuf[rb_next * PH + threadIdx.x] = nxt_row;
        __syncthreads();   // also: every wavefront is done with the previous patch's sorted columns
        // the other buffer's masks and flags (the previous patch's: everybody is past them) are cleared for the patch after next
        if ((int)threadIdx.x < PH) rowmask[((it + 1) & 1) * PH + threadIdx.x] = 0ull;
        if (threadIdx.x == 0) mflag[(it + 1) & 1] = 0u;
        const Where w = cur;
        if (more) {
            cur = nxt;
            gather(cur, rowbuf + rb_next * PH);
            if (pid + 2 * step < n_patches) {
                nxt = where(gd2, cd2);
                nxt_row = load_rows(nxt);
                if (pid + 3 * step < n_patches) desc(pid + 3 * step, gd2, cd2);
            }
        }
        // majority shortcut: positions equal to the candidate in the windows of this thread's two outputs
        bool majA, majB;
        {
            const int r0 = 2 * ty;
            int hsum = 0, h0 = 0, h9 = 0;
#pragma unroll
            for (int c = 0; c < 10; ++c) {
                const unsigned int hc = __builtin_popcount((unsigned int)(rmask[r0 + c] >> tx) & 0x1FFu);
                if (c == 0) h0 = (int)hc;
                if (c == 9) h9 = (int)hc; else hsum += (int)hc;
            }
            majA = hsum >= 41;
            majB = hsum - h0 + h9 >= 41;
        }
        const bool actA = w.g0 + tx < w.gend && w.c0 + 2 * ty < w.cend, actB = actA && w.c0 + 2 * ty + 1 < w.cend;
        const bool wave_net = __ballot((actA && !majA) || (actB && !majB)) != 0ull;       // some output of this wavefront needs its network
        const bool wave_maj = __ballot((actA && majA) || (actB && majB)) != 0ull;
        if (lane == 0 && (wave_net || wave_maj)) atomicOr(&mflag[it & 1], (wave_net ? 1u : 0u) | (wave_maj ? 2u : 0u));
        __syncthreads();
        const unsigned int fl = mflag[it & 1];
        if (fl & 1u) {
            median9_patch(w.cs, w.g0, w.gend, w.c0, w.cend, rowbuf + rb * PH, tx, ty, patch, sortedc, out, G, majA, majB, !wave_net, vguess);
            // a candidate that decided nothing in a whole patch is replaced by the value two of three probes of this patch agree on
            if (!(fl & 2u)) vguess = majority_of_three(patch[6 * PW + 10], patch[12 * PW + 20], patch[18 * PW + 30]);
        } else {
            // every output of the patch is its window's majority value: no column sorts, no networks
            const int32_t *rows = rowbuf + rb * PH;
            if (actA) out[(int64_t)rows[2 * ty + 4] * G + w.cs + w.g0 + tx] = vguess;
            if (actB) out[(int64_t)rows[2 * ty + 5] * G + w.cs + w.g0 + tx] = vguess;
        }
    }
}

}  // namespace

int launch_median_filter(const double *in, double *out, int32_t G, int64_t C, const int32_t *chr_start_dev,
                         int32_t n_chr, const int32_t *tile_idx_dev, const int32_t *tile_off_dev, int32_t n_tiles,
                         const int32_t *blk_off_dev, const int32_t *chr_start_host, int32_t total_cell_patches,
                         int32_t window_size, const Median9Plan &plan9, hipStream_t stream) {
    (void)C;
    static_assert(MF_TG == MEDIAN_GENES_PER_PATCH && MF9_TC == MEDIAN9_CELLS_PER_PATCH && MF_TC == MEDIAN_CELLS_PER_PATCH, "host tables");
    if (n_tiles <= 0 || n_chr <= 0) return ICNV_OK;
    const int h = (window_size - 1) / 2 + 1;
    if (h > MF_MAXH) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "median filter supports window_size <= 15");
    KernelTimer kt("median_filter", stream);
    if (median_is_9x9(window_size)) {
        const int64_t n_patches = (int64_t)plan9.n_gene_blocks * plan9.n_cell_patches;
        if (n_patches > 0) {
            const size_t lds = ((size_t)2 * (MF_TG + 8) * (MF9_TC + 8) + (size_t)(MF9_TC + 8) * MF_TG * 9) * sizeof(double) +
                               (3 * (MF9_TC + 8) + ((MF9_TC + 8) & 1)) * sizeof(int32_t) +
                               2 * (MF9_TC + 8) * sizeof(unsigned long long) + 2 * sizeof(unsigned int);   // + row masks and flags of the majority shortcut
            static DeviceOnce once;
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(median_filter9_kernel), 80 * 1024, once)) return rc;
            int64_t grid = (int64_t)num_cus() * 2;   // two resident workgroups per CU (70 KB of LDS, 256 registers)
            if (grid > n_patches) grid = n_patches;
            hipLaunchKernelGGL(median_filter9_kernel, dim3((unsigned)grid), dim3(MF_TG * MF_TC), lds, stream, in, out, G,
                               tile_idx_dev, reinterpret_cast<const int4 *>(plan9.gene_block_desc),
                               reinterpret_cast<const int4 *>(plan9.cell_patch_desc), plan9.n_gene_blocks, n_patches);
        }
        if (plan9.n_edge_items > 0)
            hipLaunchKernelGGL(median_filter9_edge_kernel, dim3((unsigned)plan9.n_edge_items), dim3(256), 0, stream, in, out, G,
                               tile_idx_dev, reinterpret_cast<const int4 *>(plan9.edge_desc));
    } else {
        if (total_cell_patches <= 0) return ICNV_OK;
        int gene_blocks = 0;
        for (int k = 0; k < n_chr; ++k) gene_blocks += (chr_start_host[k + 1] - chr_start_host[k] + MF_TG - 1) / MF_TG;
        if (gene_blocks <= 0) return ICNV_OK;
        const size_t lds = (size_t)(MF_TG + 2 * h) * (MF_TC + 2 * h) * sizeof(double);
        for (int base = 0; base < total_cell_patches; base += 32768) {
            const int nz = (total_cell_patches - base) < 32768 ? (total_cell_patches - base) : 32768;
            const dim3 grid(gene_blocks, 1, nz);
            hipLaunchKernelGGL(median_filter_kernel, grid, dim3(MF_TG * MF_TC), lds, stream, in, out, G,
                               chr_start_dev, tile_idx_dev, tile_off_dev, n_tiles, blk_off_dev, h, base, n_chr);
        }
    }
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

}  // namespace icnv
