// 2-D median denoise (apply_median_filtering / .median_filter,
// R/noise_reduction.R:43-113) for gfx950.  The effective window is (window_size+2)^2, clamped at the edges of the
// (cell tile, chromosome) block; rows of a block are gathered through the tile's cell-index vector.
//   window_size 7 (9 x 9 windows, the default), launches of one call (round 6):
//     median9_probe_count / _finish   a sample: the dominant value, up to three repeated values, the value range (on the device)
//     median9_sweep_kernel            classification: a wavefront per 56-gene block sweeps down the cells, writes the outputs whose
//                                     median is the dominant value, marks tiles for the dense pass, queues the rest
//                                     (median9_classify_kernel, round 5's tiled form, when the probe is switched off)
//     median9_border_kernel           without a dominant value: the blocks' border outputs straight from the geometry
//     median9_units_kernel + median9_strip_kernel   the dense pass: one gene column per lane sliding down the cells, shared merges in
//                                     registers on 32-bit compound keys; median_filter9_kernel (fp64, rounds 2-5) behind a gate
//     median9_sparse_kernel           single outputs (queues, slow lists)
//   other window sizes: median_filter_kernel -- every thread selects the median of its clamped window by a
//     value-bounded quickselect (exact order statistics; even counts average the two middle values like
//     stats::median).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
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
#include "median9_strip_net.h"

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


// ---------------------------------------------------------------------------------------------------------------------
// 9 x 9 windows (window_size 7, the default): round 5's three launches (kernel 1 = median9_classify_kernel, kernel 2 = median_filter9_kernel,
// kernel 3 = median9_sparse_kernel); round 6 put the sweep in front of kernel 1 and the strip kernel in front of kernel 2 (further down).
//
// Majority shortcut (exact, data-dependent).  A window in which one value occupies more than half of the positions has that
// value as its median, whatever the rest is (for an even number of positions -- clamped border windows -- both middle values
// are that value).  That is the common case on the matrix this filter is made for: apply_median_filtering runs on the
// DENOISED matrix, where clear_noise_via_ref_mean_sd (R/inferCNV_ops.R:2302-2346) has set every entry inside the noise band
// -- typically 80-90 % of them -- to one and the same value mu.
//
//   1. median9_classify_kernel  streams the matrix in tiles of 56 genes x 32 cells of one (tile, chromosome) block (borders
//      included; 2 x 2 tiles of the dense pass each), counts per output the window positions BELOW and ABOVE a candidate value (row masks of the tile built from
//      ballots, two popcounts per window row) and writes the candidate where fewer than half of the window's positions lie on
//      either side of it -- then it IS the median: not only where it fills more than half of the window, also where the other
//      values balance around it, which is what noise around mu does.  What is
//      left undecided goes three ways: a tile with many undecided interior outputs is put on the DENSE list (its interior
//      outputs are all recomputed by kernel 2); every other undecided output -- the few of a neutral region, and every
//      undecided border output -- is queued as one record; a tile whose records do not fit the workgroup's queue segment is
//      put on the SLOW list as a whole.  No LDS patch, ~40 registers: eight workgroups per CU keep enough loads in flight to
//      run at memory speed.  Any candidate is correct (the test is exact); a good one is the last value that won, re-seeded
//      from the tile when it decides nothing; a workgroup that meets no majority for eight tiles in a row stops loading for
//      the next 120 (data without a dominant value: everything is marked for the dense pass or queued unseen).
//   2. median_filter9_kernel    the dense pass over the tiles of the dense list: sorted columns shared through LDS, two
//      outputs per thread that share eight of their nine columns, branch-free min/max networks (interior outputs only).
//   3. median9_sparse_kernel    one queued output per lane (interior or border: clamped windows are padded with -inf / +inf so
//      that the wanted order statistics sit at ranks 40 / 41 of 81), nine column sorts and the two-rank network straight
//      from global memory (L2); also every output of the slow list's tiles.
// All lists are per-workgroup segments of kernel 1's grid: no atomics, deterministic content.

// One tile of the dense pass: stage A (column sorts into LDS), then the outputs of thread (tx, ty).  The patch covers genes
// [g0 - 4, g0 + 36) and cells [c0 - 4, c0 + 20) of the block; outputs are the INTERIOR ones of the tile: genes
// [max(g0, 4), gend) with gend <= xdim - 4, cells [max(c0, 4), cend) with cend <= ydim - 4.
__device__ inline void median9_patch(int cs, int g0, int gend, int c0, int cend, const int32_t *rows /* LDS: cell of patch row r */,
                                     int tx, int ty, const double *patch, double *sortedc, double *__restrict__ out, int G) {
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
    if (gx < 4 || gx >= gend) return;
    const int r0 = 2 * ty;                       // patch row of the first column of output A's window
    const int cyA = c0 + r0, cyB = cyA + 1;
    const bool okA = cyA >= 4 && cyA < cend, okB = cyB >= 4 && cyB < cend;
    if (!okA && !okB) return;
    if (okA && okB) {
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
        out[(int64_t)rows[r0 + 4] * G + cs + gx] = median_window_finish(w, p);
#pragma unroll
        for (int k = 0; k < 9; ++k) p[k] = sortedc[((r0 + 9) * MF_TG + tx) * 9 + k];
        out[(int64_t)rows[r0 + 5] * G + cs + gx] = median_window_finish(w, p);
        return;
    }
    // one interior output only (the first or the last interior cell of the tile): the single-output network over its nine columns
    const int rs = okA ? r0 : r0 + 1;
    double a[81];
#pragma unroll
    for (int c = 0; c < 9; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) a[9 * c + k] = sortedc[((rs + c) * MF_TG + tx) * 9 + k];
    out[(int64_t)rows[rs + 4] * G + cs + gx] = median81_sorted_columns(a);
}

// One output straight from global memory: position p in the tile's cell list (the window's cells are tile_idx[p - up .. p + dn]),
// absolute gene a (the window's genes a - lf .. a + rt); up / dn / lf / rt <= 4 are what the block leaves of the 9 x 9 window
// (R/noise_reduction.R:101-106).  The missing positions are padded with n_lo x -inf and +inf so that the wanted order
// statistics of the m real values sit at ranks 40 (and 41 for an even m: stats::median averages the two middle values).
__device__ __forceinline__ void median9_general_output(const double *__restrict__ in, double *__restrict__ out, int G,
                                                       const int32_t *__restrict__ tile_idx, int p, int a, unsigned int clamp) {
    const int up = (int)(clamp & 15u), dn = (int)((clamp >> 4) & 15u), lf = (int)((clamp >> 8) & 15u), rt = (int)((clamp >> 12) & 15u);
    const int m = (up + dn + 1) * (lf + rt + 1);
    const int n_lo = (m & 1) ? (81 - m) / 2 : 41 - m / 2;
    int npad = 0;
    double arr[81];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const bool row_ok = c - 4 >= -up && c - 4 <= dn;
        const double *src = in + (int64_t)tile_idx[row_ok ? p - 4 + c : p] * G + (a - 4);
        double v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const bool ok = row_ok && k - 4 >= -lf && k - 4 <= rt;
            double val = ok ? src[k] : 0.0;
            if (!ok) {
                val = (npad < n_lo) ? -__builtin_inf() : __builtin_inf();
                ++npad;
            }
            v[k] = val;
        }
        ICNV_SORT9(v);
#pragma unroll
        for (int k = 0; k < 9; ++k) arr[9 * c + k] = v[k];
    }
    double r40, r41;
    median81_pair_sorted_columns(arr, r40, r41);
    out[(int64_t)tile_idx[p] * G + a] = (m & 1) ? r40 : (r40 + r41) * 0.5;
}

constexpr int MF9_SPARSE_T = 128;   // more undecided interior outputs than this in a tile (of up to 512): the tile goes to the dense list
constexpr int MF9_NT = MF_TG * MF_TC;

struct StripParams;         // what the probe kernel found (defined with the strip kernel)
__device__ inline bool median9_has_dominant_value(const StripParams *P);
__device__ inline double median9_dominant_value(const StripParams *P);

struct Median9Lists {       // per-workgroup segments of kernel 1's grid
    uint4 *queue;           // [n_seg][qcap] {p, a, clamp, 0}
    int32_t *qcount;        // [n_seg]
    int32_t *slow;          // [n_seg][lcap] tiles of kernel 1
    int32_t *scount;        // [n_seg]
    uint8_t *dflag;         // [tiles of the dense pass] 1: the dense pass computes this tile (zeroed before kernel 1)
    int qcap, lcap, n_seg;
};

__device__ inline unsigned int median9_clamp_bits(int gx, int xdim, int cy, int ydim) {
    const int up = cy < 4 ? cy : 4, dn = ydim - 1 - cy < 4 ? ydim - 1 - cy : 4;
    const int lf = gx < 4 ? gx : 4, rt = xdim - 1 - gx < 4 ? xdim - 1 - gx : 4;
    return (unsigned int)up | ((unsigned int)dn << 4) | ((unsigned int)lf << 8) | ((unsigned int)rt << 12);
}

// Kernel 1.  A workgroup (four wavefronts) takes a tile of K1G = 56 genes x K1C = 32 cells of one (cell tile, chromosome)
// block.  With the four-gene halo on either side a tile row is exactly 64 genes wide: ONE wavefront reads a whole row with
// one coalesced load per lane, and its two ballots (below / above the candidate) ARE the row's masks -- no atomics, no
// shifting.  Wavefront w reads rows 10 w .. 10 w + 9 of the tile's 40 and decides the outputs of cells 8 w .. 8 w + 7 (eight
// per lane, one gene column: the counts of the window rows are computed once per row and slide down the column).
// A tile covers 2 x 2 tiles of the dense pass (32 genes x 16 cells each): each of the four is classified on its own.
constexpr int K1G = 56, K1C = 32, K1ROWS = K1C + 8, K1RPW = K1ROWS / 4, K1OPW = K1C / 4, K1RUN = MEDIAN9_K1_RUN;

__global__ void __launch_bounds__(256, 4) median9_classify_kernel(
    const double *__restrict__ in, double *__restrict__ out, int G, const int32_t *__restrict__ tile_idx,
    const int4 *__restrict__ gene1_desc /* {chromosome's first gene, its length, tile's first gene, index of the CHROMOSOME's first dense-pass gene block} */,
    const int4 *__restrict__ cell1_desc /* {offset of the cell tile's list, its length, tile's first cell, index of its first dense-pass cell block} */,
    int gene_blocks1, int64_t n_tiles, int gene_blocks2, Median9Lists L, int dev_mode /* developer switch: 1 no test, 2 no queue for interior outputs */,
    const StripParams *__restrict__ probe_result /* nullable */, int64_t n_flags /* entries of L.dflag; 0: no border kernel behind this launch */) {
    constexpr int NW = 4;
    __shared__ unsigned long long lessmask[2][K1ROWS], grtmask[2][K1ROWS];   // per tile row: bit l = the value at gene g0 - 4 + l lies below / above the candidate
    __shared__ unsigned int wcnt[2][NW];
    __shared__ double probe[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 *queue = L.queue + (int64_t)blockIdx.x * L.qcap;
    int32_t *slist = L.slow + (int64_t)blockIdx.x * L.lcap;
    int qn = 0, sn = 0;                  // entries of this workgroup's segments (the same numbers in every thread)
    // Order of the tiles (round 6): a workgroup walks RUNS of K1RUN tiles down the cells of one gene block (run r = blockIdx + k gridDim: gene
    // block r mod gene_blocks1, cell blocks K1RUN (r / gene_blocks1) ...), so that the eight halo rows it shares with the tile above are
    // lines IT asked for one tile earlier -- L2 hits instead of a second trip to HBM (1.25 x the rows otherwise).  The host pads the
    // cell blocks to a multiple of K1RUN with empty ones.  A tile's number stays cell block x gene_blocks1 + gene block.
    const int64_t n_runs = (int64_t)gene_blocks1 * ((n_tiles / gene_blocks1) / K1RUN);
    auto tile_at = [&](int s) -> int64_t {      // the s-th tile of this workgroup, -1: none
        const int64_t r = (int64_t)blockIdx.x + (int64_t)(s / K1RUN) * gridDim.x;
        if (r >= n_runs) return -1;
        return ((r / gene_blocks1) * K1RUN + (s % K1RUN)) * gene_blocks1 + r % gene_blocks1;
    };
    struct Where { int cs, xdim, g0, kb2, ydim, c0, idx_off, kc2; };
    auto where = [&](int64_t p) {
        const int4 gd = gene1_desc[p % gene_blocks1], cd = cell1_desc[p / gene_blocks1];
        Where w;
        w.cs = gd.x; w.xdim = gd.y; w.g0 = gd.z; w.kb2 = gd.w; w.idx_off = cd.x; w.ydim = cd.y; w.c0 = cd.z; w.kc2 = cd.w;
        return w;
    };
    __shared__ int32_t rowtab[2][K1ROWS];   // cell (matrix column) of every tile row, -1 outside the block
    // Three dependent round trips stand between a tile's number and its values (descriptors -> cell indices of its rows ->
    // values): each is requested one tile earlier than the next, so that a tile waits for none of them -- descriptors three
    // tiles ahead, row indices two, values one.
    auto load_ridx = [&](const Where &w) -> int32_t {     // lane j < 10: the cell of row 10 wave + j
        const int cy = w.c0 - 4 + wave * K1RPW + lane;
        return (lane < K1RPW && cy >= 0 && cy < w.ydim) ? tile_idx[w.idx_off + cy] : -1;
    };
    double stage[K1RPW];                 // this lane's gene in the wavefront's ten rows
    auto gather = [&](const Where &w, int32_t ridx) {   // (out-of-block positions: their bits are masked out below, whatever is read here)
        const int gx = w.g0 - 4 + lane;
        const bool gok = gx >= 0 && gx < w.xdim;
#pragma unroll
        for (int j = 0; j < K1RPW; ++j) {
            const int32_t row = __builtin_amdgcn_readlane(ridx, j);      // (wave-uniform)
            double v = 0.0;
            if (row >= 0 && gok) v = in[(int64_t)row * G + w.cs + gx];
            stage[j] = v;
        }
    };
    double vguess = 0.0;
    // (round 6) a matrix in which the probe found no dominant value is not read at all: every tile goes to the dense pass, every border output to the queue
    int miss = 0, cold = ((dev_mode & 1) || (probe_result && !median9_has_dominant_value(probe_result))) ? 0x7fffffff : 0;
    bool loaded = false;                 // the current tile's values are in `stage`
    if (probe_result && !median9_has_dominant_value(probe_result) && n_flags > 0) {
        // (round 6) no dominant value: nothing to classify.  Every tile of the dense pass is marked here and now (bytes of 1, this workgroup's share), the border
        // outputs are computed by median9_border_kernel straight from the blocks' geometry: no queue, no walk over the tiles (0.9 -> 0.05 ms per 50 000 cells)
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n_flags; i += (int64_t)gridDim.x * 1024)
            *reinterpret_cast<uint32_t *>(L.dflag + i) = 0x01010101u;      // (the flag array is padded to 16 bytes)
        if (threadIdx.x == 0) { L.qcount[blockIdx.x] = 0; L.scount[blockIdx.x] = 0; }
        return;
    }
    if (tile_at(0) < 0) {
        if (threadIdx.x == 0) { L.qcount[blockIdx.x] = 0; L.scount[blockIdx.x] = 0; }
        return;
    }
    Where D0 = where(tile_at(0)), D1 = D0, D2 = D0;
    if (tile_at(1) >= 0) D1 = where(tile_at(1));
    if (tile_at(2) >= 0) D2 = where(tile_at(2));
    int32_t R0 = load_ridx(D0), R1 = load_ridx(D1);
    // first candidate: an element of the first tile (wave-uniform address); re-seeded below when it decides nothing
    vguess = in[(int64_t)tile_idx[D0.idx_off + (D0.c0 + 8 < D0.ydim ? D0.c0 + 8 : D0.c0)] * G + D0.cs + D0.g0];
    // (round 6) with the probe's dominant value the candidate is that value for good: a run of tiles down an altered region decides nothing
    // for many tiles in a row, which must neither re-seed the candidate nor stop the loads
    const bool fixed_candidate = probe_result && median9_has_dominant_value(probe_result);
    if (fixed_candidate) vguess = median9_dominant_value(probe_result);
    if (cold == 0) { gather(D0, R0); loaded = true; }
    for (int it = 0;; ++it) {
        const int64_t pid = tile_at(it);
        if (pid < 0) break;
        const bool more1 = tile_at(it + 1) >= 0, more2 = tile_at(it + 2) >= 0;
        const Where w = D0;
        // (workgroup-uniform; a NaN candidate -- a probe can pick one up from the data -- decides nothing: tested on the bits)
        const bool test = loaded && !(((unsigned long long)__double_as_longlong(vguess) & 0x7fffffffffffffffull) > 0x7ff0000000000000ull);
        unsigned long long *lmask = lessmask[it & 1], *gmask = grtmask[it & 1];
        const int gx = w.g0 - 4 + lane;
        if (test) {
            const bool gok = gx >= 0 && gx < w.xdim;
#pragma unroll
            for (int j = 0; j < K1RPW; ++j) {
                // (this file is compiled with -fno-honor-nans for its min / max networks: NaN is tested on the bits; a NaN counts on
                // both sides of the candidate: it decides nothing)
                const int r = wave * K1RPW + j, cy = w.c0 - 4 + r;
                const bool ok = gok && cy >= 0 && cy < w.ydim;
                const bool is_nan = ((unsigned long long)__double_as_longlong(stage[j]) & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
                const unsigned long long ml = __ballot(ok && (stage[j] < vguess || is_nan)), mg = __ballot(ok && (stage[j] > vguess || is_nan));
                if (lane == 0) { lmask[r] = ml; gmask[r] = mg; }
            }
        }
        if (loaded && wave == 1) {
            // a probe of the tile's own values for the re-seed: the value two of three elements agree on
            const double a0 = __shfl(stage[5], 10), a1 = __shfl(stage[5], 30), a2 = __shfl(stage[6], 50);
            if (lane == 0) probe[it & 1] = (a1 == a2) ? a1 : a0;
        }
        if (lane < K1RPW) rowtab[it & 1][wave * K1RPW + lane] = R0;
        __syncthreads();
        // the next tile's values, the row indices of the tile after it and the descriptors of the one after that are requested
        // before this tile's decisions are taken
        const bool had_values = loaded;
        loaded = false;
        if (more1) {
            if (cold > 0) --cold;
            else { gather(D1, R1); loaded = true; }
        }
        const int32_t R2 = more2 ? load_ridx(D2) : -1;
        const Where D3 = tile_at(it + 3) >= 0 ? where(tile_at(it + 3)) : D2;
        // this lane's gene column: outputs of cells c0 + 8 wave + i, i = 0 .. 7
        const bool g_act = lane >= 4 && lane < 4 + K1G && gx < w.xdim;
        const bool g_int = g_act && gx >= 4 && gx < w.xdim - 4;
        const int hg = lane >= 4 + MF_TG ? 1 : 0, hc = wave >> 1;      // which of the 2 x 2 dense-pass tiles
        const int nx = (gx + 4 < w.xdim - 1 ? gx + 4 : w.xdim - 1) - (gx - 4 > 0 ? gx - 4 : 0) + 1;
        unsigned int maj = 0, act = 0, inter = 0;     // bit i: output i decided / exists / is an interior output
#pragma unroll
        for (int i = 0; i < K1OPW; ++i) {
            const int cy = w.c0 + wave * K1OPW + i;
            if (g_act && cy < w.ydim) act |= 1u << i;
            if (g_int && cy >= 4 && cy < w.ydim - 4) inter |= 1u << i;
        }
        if (test) {
            // the candidate is the median of a window of m positions iff fewer than m / 2 of them lie below it and fewer than m / 2
            // above it (odd m: at most (m - 1) / 2 on either side; even m: both middle values are the candidate)
            const int sh = lane >= 4 ? lane - 4 : 0;
            int hl[K1OPW + 8], hgt[K1OPW + 8];
#pragma unroll
            for (int k = 0; k < K1OPW + 8; ++k) {
                hl[k] = __builtin_popcount((unsigned int)(lmask[wave * K1OPW + k] >> sh) & 0x1FFu);
                hgt[k] = __builtin_popcount((unsigned int)(gmask[wave * K1OPW + k] >> sh) & 0x1FFu);
            }
            int sl = 0, sg = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) { sl += hl[k]; sg += hgt[k]; }
#pragma unroll
            for (int i = 0; i < K1OPW; ++i) {
                const int cy = w.c0 + wave * K1OPW + i;
                const int ny = (cy + 4 < w.ydim - 1 ? cy + 4 : w.ydim - 1) - (cy - 4 > 0 ? cy - 4 : 0) + 1;
                if (2 * sl < nx * ny && 2 * sg < nx * ny) maj |= 1u << i;
                if (i + 1 < K1OPW) { sl += hl[i + 9] - hl[i]; sg += hgt[i + 9] - hgt[i]; }
            }
            maj &= act;
        }
        const unsigned int und = act & ~maj;
        {
            // undecided interior outputs of this wavefront in the left / right dense-pass tile, undecided border outputs
            const int ni = __builtin_popcount(und & inter), nb = __builtin_popcount(und & ~inter);
            int n0 = hg == 0 ? ni : 0, n1 = hg == 1 ? ni : 0, nbw = nb;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                n0 += __shfl_xor(n0, o, 64);
                n1 += __shfl_xor(n1, o, 64);
                nbw += __shfl_xor(nbw, o, 64);
            }
            const bool wave_maj = __ballot(maj != 0u) != 0ull;
            if (lane == 0) wcnt[it & 1][wave] = (unsigned int)n0 | ((unsigned int)n1 << 10) | ((unsigned int)nbw << 20) | (wave_maj ? 0x80000000u : 0u);
        }
        __syncthreads();
        int tot[2][2] = {{0, 0}, {0, 0}};      // undecided interior outputs of the dense-pass tile [cell half][gene half]
        bool any_maj = false;
#pragma unroll
        for (int v = 0; v < NW; ++v) {
            const unsigned int c = wcnt[it & 1][v];
            tot[v >> 1][0] += (int)(c & 0x3FFu);
            tot[v >> 1][1] += (int)((c >> 10) & 0x3FFu);
            any_maj = any_maj || (c & 0x80000000u);
        }
        bool dense[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) dense[a][b] = tot[a][b] > MF9_SPARSE_T || ((dev_mode & 2) && tot[a][b] > 0);
        int n_push = 0, base = 0;
#pragma unroll
        for (int v = 0; v < NW; ++v) {
            const unsigned int c = wcnt[it & 1][v];
            const int pv = (int)((c >> 20) & 0x3FFu) + (dense[v >> 1][0] ? 0 : (int)(c & 0x3FFu)) + (dense[v >> 1][1] ? 0 : (int)((c >> 10) & 0x3FFu));
            if (v < wave) base += pv;
            n_push += pv;
        }
        if (qn + n_push > L.qcap) {
            // the records do not fit this workgroup's segment: the whole tile (interior and border) is left to kernel 3
            if (threadIdx.x == 0) slist[sn] = (int32_t)pid;
            ++sn;
        } else {
            const bool my_dense = dense[hc][hg];
            const unsigned int keep = my_dense ? ~inter : ~0u;     // a dense tile's interior outputs are all rewritten by kernel 2
            const unsigned int wr = maj & keep, push = und & keep;
            const int a = w.cs + gx;
            const int p0 = w.idx_off + w.c0 + wave * K1OPW;
#pragma unroll
            for (int i = 0; i < K1OPW; ++i)
                if (wr & (1u << i)) out[(int64_t)rowtab[it & 1][wave * K1OPW + 4 + i] * G + a] = vguess;
            if (n_push > 0) {
                // (round 6) the wavefront's records row by row, the genes of a row behind each other: neighbouring lanes of kernel 3 then gather
                // neighbouring genes of the same cells -- whole lines instead of one 72-byte piece per lane and row (0.54 -> 0.3x ms)
                int at = qn + base;
                const unsigned long long lower = (1ull << lane) - 1ull;
#pragma unroll
                for (int i = 0; i < K1OPW; ++i) {
                    const unsigned long long row = __ballot((push >> i) & 1u);
                    if ((push >> i) & 1u) {
                        const int cy = w.c0 + wave * K1OPW + i;
                        queue[at + __builtin_popcountll(row & lower)] = make_uint4((unsigned int)(p0 + i), (unsigned int)a, median9_clamp_bits(gx, w.xdim, cy, w.ydim), 0u);
                    }
                    at += __builtin_popcountll(row);
                }
                qn += n_push;
            }
            // a dense quarter (cell half ch, genes [g0 + 32 gh, g0 + 32 gh + 32 or 24)) marks the dense-pass tiles it overlaps: cell
            // block kc2 + ch (the cell halves ARE the dense pass's cell blocks) x the one or two 32-gene blocks of the chromosome
            // (numbered from kb2) its genes fall into.  Plain byte stores of 1: several quarters may mark the same tile.
            if (threadIdx.x < 8) {
                const int ch = (threadIdx.x >> 2) & 1, gh = (threadIdx.x >> 1) & 1, which = threadIdx.x & 1;
                const int lo = w.g0 + MF_TG * gh, hi = (lo + (gh ? K1G - MF_TG : MF_TG) < w.xdim ? lo + (gh ? K1G - MF_TG : MF_TG) : w.xdim) - 1;
                if (dense[ch][gh] && lo <= hi) {
                    const int b = which ? hi / MF_TG : lo / MF_TG;
                    L.dflag[(int64_t)(w.kc2 + ch) * gene_blocks2 + (w.kb2 + b)] = 1;
                }
            }
        }
        if (had_values && w.ydim > 0 && !fixed_candidate) {       // (an empty tile -- the padding of the runs -- says nothing about the candidate)
            if (any_maj) {
                miss = 0;
            } else {
                // the candidate decided nothing in a whole tile.  One such tile means little (the dominant value is the same all over
                // the matrix, a workgroup's tiles hop between neutral and altered regions): the candidate is kept; after three in a
                // row it is replaced by the tile's own probe, after eight the workgroup stops loading for the next 120 tiles (data without a
                // dominant value)
                ++miss;
                if (miss >= 3) vguess = probe[it & 1];
                if (miss >= 8) { miss = 0; cold = 120; }
            }
        }
        D0 = D1; D1 = D2; D2 = D3;
        R0 = R1; R1 = R2;
    }
    if (threadIdx.x == 0) {
        L.qcount[blockIdx.x] = qn;
        L.scount[blockIdx.x] = sn;
    }
}

// Kernel 2: the dense pass over the tiles kernel 1 marked (interior outputs only).  Workgroup b owns the tiles b, b + grid,
// b + 2 grid, ...; it reads the marks of its next 64 tiles with one load per lane and walks the set bits of the ballot.
__global__ void __launch_bounds__(MF_TG *MF_TC, 2) median_filter9_kernel(
    const double *__restrict__ in, double *__restrict__ out, int G, const int32_t *__restrict__ tile_idx,
    const int4 *__restrict__ gene_block_desc, const int4 *__restrict__ cell_patch_desc, int gene_blocks,
    const uint8_t *__restrict__ dflag, int64_t n_tiles2, const int32_t *__restrict__ gate, int gate_cap) {
    // (round 6) behind the strip kernel this launch is a fallback: it runs only if the strip kernel's queue overflowed
    if (gate && *gate <= gate_cap) return;
    constexpr int h = 4;
    constexpr int PW = MF_TG + 2 * h;    // patch width (genes)
    constexpr int PH = MF9_TC + 2 * h;   // patch height (cells)
    constexpr int NT = MF_TG * MF_TC;
    constexpr int EPT = (PW * PH + NT - 1) / NT;   // patch elements per thread
    extern __shared__ __attribute__((aligned(16))) double patch2[];  // 2 x [PH][PW], then the sorted columns [PH][MF_TG][9]
    double *sortedc = patch2 + 2 * PW * PH;
    // Persistent workgroups walk the dense list.  Two things keep the memory latency off the critical path: a tile is
    // described by two 16-byte records built on the host, requested TWO tiles ahead; and the NEXT tile's values are
    // requested into registers before the current tile's networks run and parked in LDS afterwards, so the gather
    // (indirect rows through the tile's cell index) hides behind ~1 000 min/max instead of standing between barriers.
    struct Where { int cs, xdim, g0, gend, ydim, c0, cend; const int32_t *idx; };
    auto where = [&](const int4 gd, const int4 cd) {
        Where w;
        w.cs = gd.x; w.xdim = gd.y; w.g0 = gd.z;
        w.gend = gd.w;                                    // end of this block's outputs (host: its share of kernel 1's tile, inside the interior)
        w.idx = tile_idx + cd.x; w.ydim = cd.y; w.c0 = cd.z;
        w.cend = cd.z + MF9_TC < cd.y - 4 ? cd.z + MF9_TC : cd.y - 4;
        return w;
    };
    // The patch is double-buffered and the row table triple-buffered: a wavefront that runs ahead into the next tile
    // parks its values and writes the row table of the tile after it while the slowest wavefront still reads the
    // current ones, so a tile costs two barriers (patch parked | columns sorted) instead of three.
    int32_t *rowbuf = reinterpret_cast<int32_t *>(sortedc + PH * MF_TG * 9);   // [3][PH]
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
    // the marked tiles of this workgroup, in order: a 64-bit window of marks over the tiles wbase + i step
    int64_t wbase = blockIdx.x;
    unsigned long long wmask = 0ull;
    bool wvalid = false;
    auto next_tile = [&]() -> int64_t {      // (the same value in every thread) -1: no more
        for (;;) {
            if (!wvalid) {
                if (wbase >= n_tiles2) return -1;
                const int64_t t = wbase + (int64_t)(threadIdx.x & 63) * step;
                wmask = __ballot(t < n_tiles2 && dflag[t] != 0);
                wvalid = true;
            }
            if (wmask) {
                const int i = __builtin_ctzll(wmask);
                wmask &= wmask - 1ull;
                return wbase + (int64_t)i * step;
            }
            wvalid = false;
            wbase += 64 * step;
        }
    };
    auto desc = [&](int64_t p, int4 &gd, int4 &cd) {
        gd = gene_block_desc[p % gene_blocks];
        cd = cell_patch_desc[p / gene_blocks];
    };
    int64_t t0 = next_tile();
    if (t0 < 0) return;
    int64_t t1 = next_tile(), t2 = t1 >= 0 ? next_tile() : -1;
    int4 gd, cd;
    desc(t0, gd, cd);
    Where cur = where(gd, cd);
    if ((int)threadIdx.x < PH) rowbuf[threadIdx.x] = load_rows(cur);
    __syncthreads();
    gather(cur, rowbuf);
    Where nxt = cur;                 // the next marked tile
    int32_t nxt_row = 0;             // its row table entry of this thread
    int4 gd2 = gd, cd2 = cd;         // descriptors of the marked tile after it
    if (t1 >= 0) {
        desc(t1, gd, cd);
        nxt = where(gd, cd);
        nxt_row = load_rows(nxt);
    }
    if (t2 >= 0) desc(t2, gd2, cd2);
    int it = 0;   // tile counter of this workgroup: patch buffer it & 1, row table it % 3
    for (; t0 >= 0; ++it) {
        const bool more = t1 >= 0;
        double *patch = patch2 + (it & 1) * (PW * PH);
        const int rb = it % 3, rb_next = (it + 1) % 3;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = (int)threadIdx.x + q * NT;
            if (e < PW * PH) patch[e] = stage[q];
        }
        if (more && (int)threadIdx.x < PH) rowbuf[rb_next * PH + threadIdx.x] = nxt_row;
        __syncthreads();   // also: every wavefront is done with the previous tile's sorted columns
        const Where w = cur;
        t0 = t1; t1 = t2;
        if (more) {
            cur = nxt;
            gather(cur, rowbuf + rb_next * PH);
            t2 = t1 >= 0 ? next_tile() : -1;
            if (t1 >= 0) {
                nxt = where(gd2, cd2);
                nxt_row = load_rows(nxt);
                if (t2 >= 0) desc(t2, gd2, cd2);
            }
        }
        median9_patch(w.cs, w.g0, w.gend, w.c0, w.cend, rowbuf + rb * PH, tx, ty, patch, sortedc, out, G);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Kernel 2s (round 6): the STRIP form of the dense pass.  One wavefront slides down the cells of a (cell tile, chromosome)
// block with one gene column per lane (64 output genes, the 4-gene halo on either side through a 72-wide row in LDS) and keeps
// everything that neighbouring windows share in REGISTERS: every matrix row is sorted once (the lane's nine genes), rows are
// merged in pairs once (used by four pair-windows), pairs in quads once (used by two), and a pair of outputs takes positions
// 30..41 of its eight shared rows from two quads and finishes with its own ninth row -- 254 min/max per output instead of 427,
// no barrier, no sorted columns in LDS.  Registers hold this only because the values are 32 bits wide: every element is a
// COMPOUND KEY  (code << 8) | id:
//   code  a monotone (non-decreasing) 24-bit code of the value: 2 q + 1 with q = the value's bucket of 2^23 - 1 linear buckets
//         over the range the probe kernel sampled (saturating: anything outside, +-Inf and NaN land in the end buckets); the
//         dominant value of the matrix, if there is one, has a code of its own (2 q_d + 1, its bucket's other values 2 q_d and
//         2 q_d + 2), so that ties at that value -- 70 % of a denoised matrix -- are never ambiguous;
//   id    (ring slot of the row, gene column mod 16): unique among the elements of a window; it makes the keys of a window
//         distinct and says where the median's 64-bit value lies (rows are kept as doubles in a 16-row ring in LDS).
// Exactness: the element at rank 40 of the window's sorted keys has 40 elements in front of it and 40 behind it.  If no other
// element of the window carries its code, the 40 in front have smaller codes, hence strictly smaller values, the 40 behind
// strictly greater ones: its value IS the median.  Equal codes are neighbours in the sorted order, so the test is
// code(39) != code(40) != code(41) (that is why the networks deliver three ranks); the dominant value's own code passes by
// construction (every element with that code has the same value); the end buckets never pass.  An output that fails the test
// is queued for kernel 3 (a record like kernel 1's, one atomic each: ~1e-5 of the outputs on continuous data).  Data with many
// repeated values that are not the dominant one would flood that queue: when it overflows, the fp64 dense pass
// (median_filter9_kernel) runs over the same tiles behind this kernel and rewrites them -- a gated launch that otherwise
// returns at once.  Only the choice of path depends on the data, never the result.
constexpr int MS_SEG = 8;                         // dense-pass cell blocks (16 cells) per segment: a run is at most 128 output rows
constexpr int MS_W = 72;                          // ring row: 64 genes + 2 x 4 halo
constexpr int MS_RING = 12;                       // rows kept as doubles (a window and the rows in flight: 10)
#ifndef MS_WAVES_PER_SIMD
#define MS_WAVES_PER_SIMD 2
#endif
constexpr uint32_t MS_QMAX = (1u << 23) - 2u;     // last bucket; codes 0 .. 2 QMAX + 2 < 2^24

constexpr int MS_NSP = 3;
struct StripParams {        // written by median9_probe_kernel
    double scale, lo_scaled;   // bucket q = sat_u32(fma(x, scale, lo_scaled))
    double sp[MS_NSP];         // values that repeat (>= 1 % of the sample each, most frequent first): each gets a code of its own; sp[0] is the dominant value if has_dom
    uint32_t has_dom, n_sp;
};

__device__ inline bool median9_has_dominant_value(const StripParams *P) { return P->has_dom != 0u; }
__device__ inline double median9_dominant_value(const StripParams *P) { return P->sp[0]; }

__device__ __forceinline__ uint32_t ms_bucket(double x, double scale, double lo_scaled) {
    const double t = __builtin_fma(x, scale, lo_scaled);
    uint32_t q;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(q) : "v"(t));    // saturating; NaN -> 0
    return q < MS_QMAX ? q : MS_QMAX;
}

// The probe (two small launches): 4 096 pseudo-random elements of the tiles' cells.
//  (1) which values REPEAT: the first 512 samples are candidates; workgroup b of the first launch counts every sample against its
//      sixteen candidates.  The second launch takes the three most frequent repeated values (a value that makes up 0.7 % of the
//      matrix is among 512 candidates with probability 0.97; a miss costs time, never exactness) -- each gets a code of its own in
//      the strip kernel; one that makes up a quarter of the sample is the DOMINANT value.  Without a dominant value the
//      classification pass decides nothing and is told not to read the matrix at all (every tile goes to the dense pass, every border
//      output to the queue).  More repeated values than three (discrete data): the strip kernel is told to leave the tiles to the
//      fp64 dense pass (strip_ok = 0).
//  (2) the range of the finite samples, for the linear buckets.
constexpr int MP_SAMPLES = 4096, MP_CAND = 512, MP_WG = 32;     // candidates per workgroup: 16
__device__ __forceinline__ double median9_probe_sample(const double *__restrict__ in, int G, const int32_t *__restrict__ tile_idx, int n_list, int i) {
    // (two 32-bit hashes scaled into range by a high multiply: no division; 64 samples share a cell -- 64 x 64 elements: a random element
    // of a 4 GB matrix is a page-table walk, 4 096 of them took 20-50 us)
    uint32_t h1 = (uint32_t)(i >> 6) * 0x9E3779B9u + 0x7F4A7C15u, h2 = (uint32_t)i * 0x85EBCA6Bu + 0xC2B2AE35u;
    h1 ^= h1 >> 15; h1 *= 0x2C1B3C6Du; h1 ^= h1 >> 12; h2 ^= h2 >> 13; h2 *= 0x297A2D39u; h2 ^= h2 >> 15;
    const int e = (int)__umulhi(h1, (uint32_t)n_list), g = (int)__umulhi(h2, (uint32_t)G);
    return in[(int64_t)tile_idx[e] * G + g];
}
struct ProbeScratch { int32_t count[MP_CAND]; double mn[MP_WG], mx[MP_WG]; };

__global__ void __launch_bounds__(256) median9_probe_count_kernel(const double *__restrict__ in, int G, const int32_t *__restrict__ tile_idx,
                                                                   int n_list, ProbeScratch *__restrict__ S) {
    constexpr int CPW = MP_CAND / MP_WG, SPT = MP_SAMPLES / 256;
    __shared__ unsigned long long cand[CPW];
    __shared__ int cnt[CPW];
    __shared__ double smin[4], smax[4];
    const int tid = threadIdx.x;
    unsigned long long v[SPT];
    double mn = __builtin_inf(), mx = -__builtin_inf();
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const double x = median9_probe_sample(in, G, tile_idx, n_list, tid + 256 * k);
        v[k] = (unsigned long long)__double_as_longlong(x);
        const bool finite = (v[k] & 0x7ff0000000000000ull) != 0x7ff0000000000000ull;
        if (finite) { mn = x < mn ? x : mn; mx = x > mx ? x : mx; }
    }
    if (tid < CPW) {
        cand[tid] = (unsigned long long)__double_as_longlong(median9_probe_sample(in, G, tile_idx, n_list, (int)blockIdx.x * CPW + tid));
        cnt[tid] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        const unsigned long long cj = cand[j];
        int n = 0;
#pragma unroll
        for (int k = 0; k < SPT; ++k) n += v[k] == cj ? 1 : 0;
        if (n) atomicAdd(&cnt[j], n);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx;
    }
    if ((tid & 63) == 0) { smin[tid >> 6] = mn; smax[tid >> 6] = mx; }
    __syncthreads();
    if (tid < CPW) S->count[blockIdx.x * CPW + tid] = cnt[tid];
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) { mn = smin[w] < mn ? smin[w] : mn; mx = smax[w] > mx ? smax[w] : mx; }
        S->mn[blockIdx.x] = mn; S->mx[blockIdx.x] = mx;      // (every workgroup saw all samples: the same numbers)
    }
}

__global__ void __launch_bounds__(MP_CAND) median9_probe_finish_kernel(const double *__restrict__ in, int G, const int32_t *__restrict__ tile_idx,
                                                                        int n_list, const ProbeScratch *__restrict__ S, StripParams *__restrict__ P) {
    __shared__ unsigned long long key[MP_CAND];      // block arg-max of (count << 32 | candidate index)
    __shared__ unsigned long long picked[MS_NSP];
    __shared__ int picked_cnt[MS_NSP], others;
    const int tid = threadIdx.x;
    const unsigned long long mine = (unsigned long long)__double_as_longlong(median9_probe_sample(in, G, tile_idx, n_list, tid));
    const int my_cnt = S->count[tid];
    const bool is_nan = (mine & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
    if (tid == 0) others = 0;
    int n_sp = 0;
    for (int k = 0; k < MS_NSP; ++k) {
        bool taken = is_nan || my_cnt < 3;       // (a value seen three times in 4 096 samples repeats: continuous data does not)
        for (int i = 0; i < n_sp; ++i) taken = taken || picked[i] == mine;
        key[tid] = taken ? 0ull : (((unsigned long long)my_cnt << 32) | (unsigned long long)(MP_CAND - tid));
        __syncthreads();
        for (int o = MP_CAND / 2; o > 0; o >>= 1) {
            if (tid < o) key[tid] = key[tid] > key[tid + o] ? key[tid] : key[tid + o];
            __syncthreads();
        }
        const unsigned long long best = key[0];
        __syncthreads();
        if (best == 0ull) break;
        if (tid == MP_CAND - (int)(best & 0xffffffffull)) { picked[k] = mine; picked_cnt[k] = my_cnt; }
        ++n_sp;
        __syncthreads();
    }
    // candidates that repeat and are none of the picked values: samples of the matrix's share of OTHER repeated values
    {
        bool other = !is_nan && my_cnt >= 3;
        for (int i = 0; i < n_sp; ++i) other = other && picked[i] != mine;
        if (other) atomicAdd(&others, 1);
    }
    __syncthreads();
    if (tid == 0) {
        double mn = S->mn[0], mx = S->mx[0];
        double lo = mn, hi = mx;
        if (!(lo <= hi)) { lo = 0.0; hi = 1.0; }                  // no finite sample
        double range = hi - lo;
        if (!(range > 0.0) || range > 1e300) range = 1.0;          // constant sample / overflow: any positive scale is correct
        const double scale = (double)MS_QMAX / (range * 1.0625);   // 1/32 of the range spare on either side
        StripParams p;
        p.scale = scale;
        p.lo_scaled = -(lo - range * 0.03125) * scale;
        for (int k = 0; k < MS_NSP; ++k) p.sp[k] = k < n_sp ? __longlong_as_double((long long)picked[k]) : 0.0;
        p.n_sp = (uint32_t)n_sp | (others * 50 > MP_CAND ? 0x100u : 0u);      // bit 8: more than 2 % of the matrix are further repeated values -- not the strip kernel's data
        p.has_dom = (n_sp > 0 && picked_cnt[0] >= MP_SAMPLES / 4) ? 1u : 0u;
        *P = p;
    }
}

struct StripArgs {
    const double *in;
    double *out;
    int G;
    const int32_t *tile_idx;
    const int4 *strip_desc;     // {chromosome's first gene, its length, strip's first gene (in the chromosome), dflag gene block of that gene}
    const int4 *seg_desc;       // {offset of the cell tile's list, its length, segment's first cell, dflag cell block of that cell}
    int n_strips;
    int64_t n_units;            // n_strips x segments
    const uint8_t *dflag;
    int gene_blocks2;
    const StripParams *P;
    uint4 *fq;                  // outputs this kernel could not certify: {position in the tile list, absolute gene, clamp bits, 0}
    int32_t *fq_count;
    int fq_cap;
    int32_t *unit_counter;      // zeroed before the launch
    const int32_t *unit_counts; // {full units, partial units} of the list (median9_units_kernel)
    const int2 *unit_list;      // [n_units]: full units from the front, partial ones from the back
};

// The units of the strip kernel that hold a marked tile, as a list (one thread per unit; a launch of a few microseconds): full units -- eight cell
// blocks inside the tile, every one with a marked tile -- are appended at the front, the others at the back.
__global__ void __launch_bounds__(256) median9_units_kernel(const int4 *__restrict__ strip_desc, const int4 *__restrict__ seg_desc, int n_strips, int64_t n_units,
                                                            const uint8_t *__restrict__ dflag, int gene_blocks2, const StripParams *__restrict__ P,
                                                            int32_t *__restrict__ counts /* {front, back}, zeroed */, int2 *__restrict__ list) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_units || (P->n_sp & 0x100u)) return;
    const int4 sd = strip_desc[u % n_strips], cd = seg_desc[u / n_strips];
    const int xdim = sd.y, g0 = sd.z, kb = sd.w, ydim = cd.y, c0 = cd.z, kc = cd.w;
    uint32_t fm = 0, pm = 0;
    for (int j = 0; j < MS_SEG; ++j)
        for (int b = 0; b < 2; ++b)
            if (c0 + MEDIAN9_CELLS_PER_PATCH * j < ydim && g0 + MF_TG * b < xdim && dflag[(int64_t)(kc + j) * gene_blocks2 + kb + b]) { fm |= 1u << (2 * j + b); pm |= 1u << j; }
    if (!fm) return;
    const bool full = pm == (1u << MS_SEG) - 1u;
    if (full) list[atomicAdd(&counts[0], 1)] = make_int2((int)u, (int)fm);
    else list[n_units - 1 - atomicAdd(&counts[1], 1)] = make_int2((int)u, (int)fm);
}

__global__ void __launch_bounds__(256, MS_WAVES_PER_SIMD) median9_strip_kernel(const StripArgs A) {
    __shared__ double ring_all[4][MS_RING][MS_W];
    __shared__ uint32_t keys_all[4][2][MS_W + 8];
    __shared__ int32_t rcol_all[4][16];
    __shared__ uint32_t park_all[4][5][9][64];     // the sorted second rows of the last five steps (each is an output's ninth row four steps later)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t (*park)[9][64] = park_all[wave];
    double (*ring)[MS_W] = ring_all[wave];
    const double *ring_flat = &ring_all[wave][0][0];
    uint32_t (*keys)[MS_W + 8] = keys_all[wave];
    int32_t *rcol = rcol_all[wave];
    const double scale = A.P->scale, lo_scaled = A.P->lo_scaled;
    // the repeated values' buckets (0xffffffff: none -- also a second value in a bucket that is taken) and their own codes
    // (scalars, not arrays: an array the compiler indexes at run time would live in scratch memory)
    const uint32_t n_sp = A.P->n_sp & 0xffu;
    if (A.P->n_sp & 0x100u) {       // discrete data (the probe): the gated fp64 dense pass takes the tiles
        if (threadIdx.x == 0 && blockIdx.x == 0) *A.fq_count = 0x7fffffff;
        return;
    }
    const double sv0 = A.P->sp[0], sv1 = A.P->sp[1], sv2 = A.P->sp[2];
    const uint32_t sq0 = n_sp > 0u ? ms_bucket(sv0, scale, lo_scaled) : 0xffffffffu;
    uint32_t sq1 = n_sp > 1u ? ms_bucket(sv1, scale, lo_scaled) : 0xffffffffu;
    uint32_t sq2 = n_sp > 2u ? ms_bucket(sv2, scale, lo_scaled) : 0xffffffffu;
    if (sq1 == sq0) sq1 = 0xffffffffu;
    if (sq2 == sq0 || sq2 == sq1) sq2 = 0xffffffffu;
    const uint32_t sc0 = 2u * sq0 + 1u, sc1 = 2u * sq1 + 1u, sc2 = 2u * sq2 + 1u;      // (none: 0xffffffff, never a 24-bit code)
    // (branch-free per lane; the number of repeated values is wave-uniform: data without any pays for none)
    auto make_key = [&](double x, uint32_t id) -> uint32_t {
        const uint32_t q = ms_bucket(x, scale, lo_scaled);
        uint32_t code = 2u * q + 1u;
        if (n_sp > 0u) code += (q == sq0) ? (uint32_t)((int)(x > sv0) - (int)(x < sv0)) : 0u;
        if (n_sp > 1u) code += (q == sq1) ? (uint32_t)((int)(x > sv1) - (int)(x < sv1)) : 0u;
        if (n_sp > 2u) code += (q == sq2) ? (uint32_t)((int)(x > sv2) - (int)(x < sv2)) : 0u;
        return (code << 8) | id;
    };
    // Units are handed out by a counter (the marked tiles cluster in a few chromosomes: dealt round-robin, 5 +- 2 of a wavefront's 32 units
    // carried work and the launch waited for the unluckiest wavefront) from the LIST median9_units_kernel made of the units that hold a marked tile
    // (five of six units of the denoised matrix hold none: finding that out cost a wavefront three dependent loads per unit) -- the full units
    // first, the partial ones (tile ends, partly marked) last, so that the launch ends on its smallest pieces.  The next entry's number is
    // requested before this one is processed.
    const int n_full = A.unit_counts[0], n_part = A.unit_counts[1];
    auto next_unit = [&]() -> int {
        int v = 0;
        if (lane == 0) v = atomicAdd(A.unit_counter, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    int i_next = next_unit();
    for (;;) {
        const int i_cur = i_next;
        if (i_cur >= n_full + n_part) break;
        i_next = next_unit();
        const int2 entry = A.unit_list[i_cur < n_full ? (int64_t)i_cur : A.n_units - 1 - (int64_t)(i_cur - n_full)];      // {unit, its marks: bit 2 j + b}
        const int64_t u = entry.x;
        const uint32_t fm = (uint32_t)entry.y;
        const int4 sd = A.strip_desc[u % A.n_strips], cd = A.seg_desc[u / A.n_strips];
        const int cs = sd.x, xdim = sd.y, g0 = sd.z;
        const int idx_off = cd.x, ydim = cd.y, c0 = cd.z;
        uint32_t pm = 0;     // bit j: cell block j has a marked tile
#pragma unroll
        for (int j = 0; j < MS_SEG; ++j) pm |= ((fm >> (2 * j)) & 3u) ? (1u << j) : 0u;
        const int go = g0 + lane;                              // this lane's output gene (in the chromosome)
        const bool lane_ok = go >= 4 && go < xdim - 4;         // interior outputs only (kernel 1 / 3 own the borders)
        const int gl = g0 - 4 + lane;                          // the gene this lane loads per row: column `lane`
        // (a row's address is a wave-uniform base -- the scalar unit's -- plus these 32-bit lane offsets)
        const int off_l = cs + (gl < 0 ? 0 : (gl < xdim ? gl : xdim - 1));
        const int ge2 = g0 + 60 + (lane & 7);                  // (lanes 0 .. 7 and 8 .. 15: the genes of columns 64 .. 71)
        const int off_e = cs + (ge2 < xdim ? ge2 : xdim - 1);
        const int off_o = cs + go;
        const uint32_t id_l = (uint32_t)lane & 15u;
        while (pm) {
            const int j0 = __builtin_ctz(pm);
            const int run = __builtin_ctz(~(pm >> j0));
            pm &= ~(((1u << run) - 1u) << j0);
            const int c_lo = c0 + MEDIAN9_CELLS_PER_PATCH * j0;
            const int c_hi = c0 + MEDIAN9_CELLS_PER_PATCH * (j0 + run) < ydim ? c0 + MEDIAN9_CELLS_PER_PATCH * (j0 + run) : ydim;
            const int o_lo = c_lo > 4 ? c_lo : 4, o_hi = c_hi < ydim - 4 ? c_hi : ydim - 4;     // output rows [o_lo, o_hi)
            if (o_lo >= o_hi) continue;
            const int r_first = o_lo - 4;                       // stream row j is the tile's cell r_first + j
            const int n_stream = o_hi - o_lo + 8;               // <= 136
            const int j_last = n_stream - 5;                    // last centre
            int32_t rc0, rc1, rc2;                              // matrix column of stream row lane, 64 + lane, 128 + lane
            {
                const int a0 = r_first + lane, a1 = a0 + 64, a2 = a0 + 128;
                rc0 = A.tile_idx[idx_off + (a0 < ydim ? a0 : ydim - 1)];
                rc1 = A.tile_idx[idx_off + (a1 < ydim ? a1 : ydim - 1)];
                rc2 = A.tile_idx[idx_off + (a2 < ydim ? a2 : ydim - 1)];
            }
            // (by-value captures: a closure of references turns the selection below into a selection of addresses in scratch memory)
            auto row_col = [rc0, rc1, rc2, n_stream](int j) -> int32_t {              // (wave-uniform j)
                const int jj = j < n_stream ? j : n_stream - 1;
                const int32_t a = __builtin_amdgcn_readlane(rc0, jj & 63), b = __builtin_amdgcn_readlane(rc1, jj & 63), c = __builtin_amdgcn_readlane(rc2, jj & 63);
                return jj < 64 ? a : (jj < 128 ? b : c);
            };
            // state carried from step to step (32-bit keys): the newest pair, two quads, the window, four second rows
            uint32_t M1h[18], Q2[2][36], W[12];
#pragma unroll
            for (int i = 0; i < 18; ++i) M1h[i] = 0u;
#pragma unroll
            for (int i = 0; i < 36; ++i) { Q2[0][i] = 0u; Q2[1][i] = 0u; }
#pragma unroll
            for (int i = 0; i < 12; ++i) W[i] = 0u;
            // rows of the first step
            int32_t col_a = row_col(0), col_b = row_col(1);
            const double *row_a = A.in + (int64_t)col_a * A.G, *row_b = A.in + (int64_t)col_b * A.G;
            double xa = row_a[off_l], xb = row_b[off_l];
            double xe = 0.0;
            if (lane < 16) xe = (lane < 8 ? row_a : row_b)[off_e];
            int ring_slot = 0, park_slot = 0;
            auto emit = [&](const uint32_t (&r)[3], int jc) {
                if (jc < 4 || jc > j_last) return;              // (wave-uniform)
                const int cy = r_first + jc;
                const uint32_t bits = (fm >> (2 * ((cy - c0) >> 4))) & 3u;      // (wave-uniform: the marks of this row's two tiles)
                const bool mine = lane_ok && ((bits >> (lane >> 5)) & 1u);
                const uint32_t code = r[1] >> 8;
                const bool own_code = code == sc0 || code == sc1 || code == sc2;      // a repeated value's code: every element that carries it has that value
                const bool amb = (((r[0] >> 8) == code || (r[2] >> 8) == code) && !own_code) || code < 3u || code >= 2u * MS_QMAX;
                // the median's 64-bit value: ring slot and column of its id (the column is the one of lane .. lane + 8 with these low four bits)
                const uint32_t col = (uint32_t)lane + ((r[1] - (uint32_t)lane) & 15u);
                const double v = ring_flat[((r[1] >> 4) & 15u) * MS_W + col];
                double *orow = A.out + (int64_t)rcol[jc & 15] * A.G;            // (wave-uniform)
                if (mine && !amb) __builtin_nontemporal_store(v, orow + off_o);
                if (__ballot(mine && amb)) {
                    if (mine && amb) {
                        // (one atomic per wavefront: the lanes' records lie behind each other)
                        const unsigned long long m = __ballot(1);
                        const int first = __builtin_ctzll(m);
                        int at = 0;
                        if (lane == first) at = atomicAdd(A.fq_count, __builtin_popcountll(m));
                        at = __builtin_amdgcn_readlane(at, first) + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                        if (at < A.fq_cap) A.fq[at] = make_uint4((unsigned int)(idx_off + cy), (unsigned int)(cs + go), 0x4444u, 0u);
                    }
                }
            };
            auto step = [&](auto PHC, int p) {
                constexpr int PH = decltype(PHC)::value;
                const int ja = 2 * p, jb = 2 * p + 1;
                const int sa = ring_slot, sb = ring_slot + 1;      // (stream row j lives in ring slot j mod 12)
                ring_slot = ring_slot + 2 == MS_RING ? 0 : ring_slot + 2;
                ring[sa][lane] = xa;
                ring[sb][lane] = xb;
                keys[0][lane] = make_key(xa, ((uint32_t)sa << 4) | id_l);
                keys[1][lane] = make_key(xb, ((uint32_t)sb << 4) | id_l);
                if (lane < 16) {                    // columns 64 .. 71: lanes 0 .. 7 hold row a's, lanes 8 .. 15 row b's
                    const int se = lane < 8 ? sa : sb;
                    ring[se][64 + (lane & 7)] = xe;
                    keys[lane >> 3][64 + (lane & 7)] = make_key(xe, ((uint32_t)se << 4) | (uint32_t)(lane & 7));
                }
                if (lane == 0) { rcol[ja & 15] = col_a; rcol[jb & 15] = col_b; }
                // the next step's rows are requested now and land behind this step's networks
                col_a = row_col(ja + 2);
                col_b = row_col(jb + 2);
                row_a = A.in + (int64_t)col_a * A.G;
                row_b = A.in + (int64_t)col_b * A.G;
                xa = row_a[off_l];
                xb = row_b[off_l];
                if (lane < 16) xe = (lane < 8 ? row_a : row_b)[off_e];
                __builtin_amdgcn_wave_barrier();
                uint32_t Ta[9], Tn[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) { Ta[k] = keys[0][lane + k]; Tn[k] = keys[1][lane + k]; }
                __builtin_amdgcn_wave_barrier();
                MS_SORT9(Ta);
                MS_SORT9(Tn);
                uint32_t r[3];
                ms_finish(W, Ta, r);                 // the previous step's second output: centre 2 p - 4, its ninth row is this step's first
                emit(r, 2 * p - 4);
                uint32_t M1n[18], Qn[36];
                __builtin_amdgcn_sched_barrier(0);
                ms_merge9(Ta, Tn, M1n);
#pragma unroll
                for (int i = 0; i < 9; ++i) park[park_slot][i][lane] = Tn[i];
                park_slot = park_slot == 4 ? 0 : park_slot + 1;       // (now the slot of step p - 4)
                __builtin_amdgcn_sched_barrier(0);
                ms_merge18(M1h, M1n, Qn);            // rows 2 p - 2 .. 2 p + 1
#pragma unroll
                for (int i = 0; i < 18; ++i) M1h[i] = M1n[i];
                __builtin_amdgcn_sched_barrier(0);
                ms_window(Q2[PH & 1], Qn, W);        // rows 2 p - 6 .. 2 p + 1
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 36; ++i) Q2[PH & 1][i] = Qn[i];
                uint32_t To[9];                       // the sorted row 2 p - 7 (second row of step p - 4): ninth row of this step's first output
#pragma unroll
                for (int i = 0; i < 9; ++i) To[i] = park[park_slot][i][lane];
                ms_finish(W, To, r);                 // first output: centre 2 p - 3
                emit(r, 2 * p - 3);
            };
            for (int p = 0;;) {
                if (2 * p - 4 > j_last) break;
                step(std::integral_constant<int, 0>{}, p); ++p;
                if (2 * p - 4 > j_last) break;
                step(std::integral_constant<int, 1>{}, p); ++p;
                if (2 * p - 4 > j_last) break;
                step(std::integral_constant<int, 2>{}, p); ++p;
                if (2 * p - 4 > j_last) break;
                step(std::integral_constant<int, 3>{}, p); ++p;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Kernel 1s (round 6): the classification pass as a SWEEP down the cells -- used whenever the probe ran (its dominant value is the
// candidate; without one this launch marks every tile and returns, like kernel 1).  One wavefront, no workgroup: lane l holds gene
// g0 - 4 + l of a 56-gene block (the same blocks as kernel 1's tiles) and walks down a segment of 128 cells.  Every matrix row is read
// ONCE (kernel 1 reads 40 rows for 32: its tiles overlap by the halo), its two ballots are the row's masks, a lane's 9-bit popcounts of
// the last nine rows live in registers and the window sums slide: no LDS, no barrier, ~45 vector instructions per row.  Decided
// outputs are written at once; the undecided ones of a block of 16 cells are kept as bit masks until the block is complete -- then
// its two 32-gene halves are counted (more than 128 undecided interior outputs: the dense pass takes the tile), the rest is queued
// row by row in the wavefront's own segment (a block that does not fit puts its tile of kernel 1 on the slow list).  Same lists, same
// meaning as kernel 1's: kernels 2 and 3 do not know which of the two ran.
struct SweepArgs {
    const double *in;
    double *out;
    int G;
    const int32_t *tile_idx;
    const int4 *gene_desc;      // per gene block of the sweep, two records: {chromosome's first gene, its length, block's first gene, index of the chromosome's first dense-pass gene block},
                                //                                           {index of the chromosome's first gene block of kernel 1 (the slow list names ITS tiles), 0, 0, 0}
    const int4 *seg_desc;       // {offset of the cell tile's list, its length, segment's first cell, dense-pass cell block of that cell}
    int gene_blocks, gene_blocks1, gene_blocks2;      // of the sweep | of kernel 1 | of the dense pass
    int64_t n_units;            // gene_blocks x segments
    Median9Lists L;             // segments per WAVEFRONT of this grid
    int dev_mode;
    const StripParams *P;
    int64_t n_flags;
};
constexpr int SW_ROWS = MS_SEG * MF9_TC;      // cells per segment (the strip kernel's segments)

// NH = 64-gene halves of a row: 1 -> 56 output genes per wavefront (kernel 1's gene blocks), 2 -> 120 (a 1 024-byte row straddles 9 lines for 120
// outputs where a 512-byte row straddles 5 for 56: 16 % fewer bytes read).  Lane l loads gene g0 - 4 + 64 h + l of half h and owns the outputs
// g0 + 64 k + l, k < NH (the last slot: l < 56); the 128-bit masks of a row are two ballots per side.
template <int NH>
__global__ void __launch_bounds__(256) median9_sweep_kernel(const SweepArgs A) {
    constexpr int NOUT = 64 * NH - 8;         // output genes per block
    const int lane = threadIdx.x & 63;
    const int sgm = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);      // this wavefront's segment of the lists
    if (!median9_has_dominant_value(A.P)) {
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < A.n_flags; i += (int64_t)gridDim.x * 1024)
            *reinterpret_cast<uint32_t *>(A.L.dflag + i) = 0x01010101u;
        if (lane == 0) { A.L.qcount[sgm] = 0; A.L.scount[sgm] = 0; }
        return;
    }
    const double vg = median9_dominant_value(A.P);
    uint4 *queue = A.L.queue + (int64_t)sgm * A.L.qcap;
    int32_t *slist = A.L.slow + (int64_t)sgm * A.L.lcap;
    int qn = 0, sn = 0;
    const unsigned long long lower = (1ull << lane) - 1ull;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t u = sgm; u < A.n_units; u += n_waves) {
        const int gb = (int)(u % A.gene_blocks);
        const int4 gd = A.gene_desc[2 * gb], gd2 = A.gene_desc[2 * gb + 1], sd = A.seg_desc[u / A.gene_blocks];
        const int cs = gd.x, xdim = gd.y, g0 = gd.z, kb2 = gd.w, kb1 = gd2.x;
        const int idx_off = sd.x, ydim = sd.y, c0 = sd.z, kc = sd.w;
        const int n_out = (c0 + SW_ROWS < ydim ? c0 + SW_ROWS : ydim) - c0;       // outputs: cells c0 .. c0 + n_out - 1; stream row j is cell c0 - 4 + j
        const int n_groups = (n_out + 8 + 8) / 9;                                 // groups of nine stream rows
        bool gok[NH], g_act[NH], g_int[NH];
        int nx[NH], off[NH], a_abs[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int gl = g0 - 4 + 64 * h + lane;          // the gene this lane LOADS in half h
            gok[h] = gl >= 0 && gl < xdim;
            off[h] = cs + (gl < 0 ? 0 : (gl < xdim ? gl : xdim - 1));
            const int go = g0 + 64 * h + lane;              // the gene of this lane's OUTPUT slot h
            g_act[h] = 64 * h + lane < NOUT && go < xdim;
            g_int[h] = g_act[h] && go >= 4 && go < xdim - 4;
            nx[h] = (go + 4 < xdim - 1 ? go + 4 : xdim - 1) - (go - 4 > 0 ? go - 4 : 0) + 1;
            a_abs[h] = cs + go;
        }
        auto load_rc = [&](int g) -> int32_t {        // lane i < 9: the matrix column of stream row 9 g + i, -1 outside the tile
            const int cy = c0 - 4 + 9 * g + lane;
            return (lane < 9 && g < n_groups && cy >= 0 && cy < ydim) ? A.tile_idx[idx_off + cy] : -1;
        };
        double cur[NH][9], nxt[NH][9];
        auto gather = [&](double (&v)[NH][9], int32_t rc) {
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int32_t row = __builtin_amdgcn_readlane(rc, i);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    double x = 0.0;
                    if (row >= 0) x = (A.in + (int64_t)row * A.G)[off[h]];
                    v[h][i] = x;
                }
            }
        };
        int32_t rc_prev = -1, rc_cur = load_rc(0), rc_nxt = load_rc(1);
        gather(cur, rc_cur);
        int hl[NH][9], hg[NH][9], sl[NH], sg[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            sl[h] = 0; sg[h] = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i) { hl[h][i] = 0; hg[h][i] = 0; }
        }
        // this lane's outputs of the current and the next block of 16 cells, undecided / interior (bit (c - c0) mod 32: a block is finished outside the
        // unrolled rows, up to nine rows after its last one), and the undecided interior outputs of the 32-gene pieces of the block (2 NH pieces: lanes
        // 0 .. 31 | 32 .. 63 of every slot) per block parity
        uint32_t und32[NH], inter32[NH];
        int np[2][2 * NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) { und32[h] = 0; inter32[h] = 0; np[0][2 * h] = np[0][2 * h + 1] = np[1][2 * h] = np[1][2 * h + 1] = 0; }
        int next_blk = 0;
        auto finish_block = [&](int kblk) {
            const int par = kblk & 1;
            bool dense[2 * NH];
            uint32_t keep16[NH];
            int n_push = 0;
            bool any_dense = false;
#pragma unroll
            for (int t = 0; t < 2 * NH; ++t) {
                const int n = par ? np[1][t] : np[0][t];
                dense[t] = n > MF9_SPARSE_T || ((A.dev_mode & 2) && n > 0);
                any_dense = any_dense || dense[t];
            }
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const uint32_t und16 = (und32[h] >> (16 * par)) & 0xFFFFu, inter16 = (inter32[h] >> (16 * par)) & 0xFFFFu;
                const bool my_dense = lane >= 32 ? dense[2 * h + 1] : dense[2 * h];
                keep16[h] = und16 & ~(my_dense ? inter16 : 0u);      // a dense tile's interior outputs are all rewritten by the dense pass
#pragma unroll
                for (int i = 0; i < MF9_TC; ++i) n_push += __builtin_popcountll(__ballot((keep16[h] >> i) & 1u));
            }
            if (n_push > 0) {
                if (qn + n_push > A.L.qcap) {
                    // the records do not fit this wavefront's segment: the tiles of kernel 1 that hold the block are left to kernel 3 as a whole
                    const int t_lo = g0 / K1G, t_hi = ((g0 + NOUT < xdim ? g0 + NOUT : xdim) - 1) / K1G;
                    for (int t = t_lo; t <= t_hi; ++t) {
                        if (lane == 0 && sn < A.L.lcap) slist[sn] = (int32_t)((int64_t)((kc + kblk) >> 1) * A.gene_blocks1 + kb1 + t);
                        ++sn;
                    }
                } else {
                    int at = qn;
#pragma unroll
                    for (int i = 0; i < MF9_TC; ++i) {
                        const int cy = c0 + MF9_TC * kblk + i;
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
                            const unsigned long long row = __ballot((keep16[h] >> i) & 1u);
                            if ((keep16[h] >> i) & 1u)
                                queue[at + __builtin_popcountll(row & lower)] = make_uint4((unsigned int)(idx_off + cy), (unsigned int)a_abs[h], median9_clamp_bits(a_abs[h] - cs, xdim, cy, ydim), 0u);
                            at += __builtin_popcountll(row);
                        }
                    }
                    qn += n_push;
                }
            }
            if (any_dense && lane < 4 * NH) {
                // piece t (32 genes from g0 + 32 t, the last one 24) marks the one or two dense-pass tiles its genes fall into
                const int t = lane >> 1, which = lane & 1;
                const int lo = g0 + MF_TG * t, wd = t == 2 * NH - 1 ? MF_TG - 8 : MF_TG, hi = (lo + wd < xdim ? lo + wd : xdim) - 1;
                bool d = false;
#pragma unroll
                for (int q = 0; q < 2 * NH; ++q) d = d || (q == t && dense[q]);
                if (d && lo <= hi) A.L.dflag[(int64_t)(kc + kblk) * A.gene_blocks2 + (kb2 + (which ? hi / MF_TG : lo / MF_TG))] = 1;
            }
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                und32[h] &= ~(0xFFFFu << (16 * par));
                inter32[h] &= ~(0xFFFFu << (16 * par));
                if (par) { np[1][2 * h] = 0; np[1][2 * h + 1] = 0; } else { np[0][2 * h] = 0; np[0][2 * h + 1] = 0; }
            }
        };
        for (int g = 0; g < n_groups; ++g) {
            const int32_t rc_nn = load_rc(g + 2);
            if (g + 1 < n_groups) gather(nxt, rc_nxt);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int j = 9 * g + i, cyr = c0 - 4 + j;
                const bool row_ok = cyr >= 0 && cyr < ydim && !(A.dev_mode & 1);
                unsigned long long ml[NH], mg[NH];
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const double x = cur[h][i];
                    const bool is_nan = ((unsigned long long)__double_as_longlong(x) & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
                    ml[h] = __ballot(gok[h] && row_ok && (x < vg || is_nan));
                    mg[h] = __ballot(gok[h] && row_ok && (x > vg || is_nan));
                }
                const int c = cyr - 4;                       // the output whose window this row completes
                const bool out_row = j >= 8 && c < c0 + n_out;      // (wave-uniform)
                const int ny = (c + 4 < ydim - 1 ? c + 4 : ydim - 1) - (c - 4 > 0 ? c - 4 : 0) + 1;
                const int bi = (c - c0) & 31;
                int32_t ccol = 0;
                if (out_row) ccol = i >= 4 ? __builtin_amdgcn_readlane(rc_cur, i >= 4 ? i - 4 : 0) : __builtin_amdgcn_readlane(rc_prev, i < 4 ? i + 5 : 0);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    // the nine mask bits from position 64 h + lane on: this lane's window in the row
                    unsigned long long wl = ml[h] >> lane, wg = mg[h] >> lane;
                    if (h + 1 < NH && lane > 55) { wl |= ml[h + 1 < NH ? h + 1 : h] << (64 - lane); wg |= mg[h + 1 < NH ? h + 1 : h] << (64 - lane); }
                    const int nl = __builtin_popcount((unsigned int)wl & 0x1FFu), ng = __builtin_popcount((unsigned int)wg & 0x1FFu);
                    sl[h] += nl - hl[h][i]; hl[h][i] = nl;
                    sg[h] += ng - hg[h][i]; hg[h][i] = ng;
                    if (out_row) {
                        const bool dec = !(A.dev_mode & 1) && 2 * sl[h] < nx[h] * ny && 2 * sg[h] < nx[h] * ny;
                        if (g_act[h] && dec) __builtin_nontemporal_store(vg, (A.out + (int64_t)ccol * A.G) + a_abs[h]);      // (written once, read by nobody in this call)
                        const bool und = g_act[h] && !dec, inter = g_int[h] && c >= 4 && c < ydim - 4;
                        und32[h] |= (und ? 1u : 0u) << bi;
                        inter32[h] |= (inter ? 1u : 0u) << bi;
                        const unsigned long long ub = __ballot(und && inter);
                        const int u0 = __builtin_popcount((unsigned int)ub), u1 = __builtin_popcount((unsigned int)(ub >> 32));
                        if (bi & 16) { np[1][2 * h] += u0; np[1][2 * h + 1] += u1; } else { np[0][2 * h] += u0; np[0][2 * h + 1] += u1; }
                    }
                }
            }
            // blocks whose last output lies behind us (one per group at most, two at the end of the unit)
            {
                const int done = 9 * g < n_out - 1 ? 9 * g : n_out - 1;      // outputs c0 .. c0 + done are complete
                while (next_blk * MF9_TC <= done && (next_blk * MF9_TC + MF9_TC - 1 <= done || done == n_out - 1)) finish_block(next_blk++);
            }
            rc_prev = rc_cur; rc_cur = rc_nxt; rc_nxt = rc_nn;
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int i = 0; i < 9; ++i) cur[h][i] = nxt[h][i];
        }
    }
    if (lane == 0) { A.L.qcount[sgm] = qn; A.L.scount[sgm] = sn < A.L.lcap ? sn : A.L.lcap; }
}

// Kernel 1b (round 6): the border outputs of EVERY (cell tile, chromosome) block, straight from the geometry -- the partner of kernel 1's early
// exit when the probe finds no dominant value (gated: with a dominant value kernel 1 has queued the undecided border outputs and this launch
// returns at once).  A cell within four rows of its tile's edge has all of its genes on the border (items of 256 genes: tile x 8 rows x
// chunks), any other cell the first and last four genes of every chromosome (one item per cell); persistent workgroups stride over both lists.
__global__ void __launch_bounds__(256, 2) median9_border_kernel(const double *__restrict__ in, double *__restrict__ out, int G, const int32_t *__restrict__ tile_idx,
                                                                 const int32_t *__restrict__ tile_off, int n_tiles, const int32_t *__restrict__ chr_start, int n_chr,
                                                                 const StripParams *__restrict__ probe_result) {
    if (median9_has_dominant_value(probe_result)) return;
    const int n_chunk = (G + 255) / 256;
    const int64_t n_a = (int64_t)n_tiles * 8 * n_chunk;
    for (int64_t i = blockIdx.x; i < n_a; i += gridDim.x) {
        const int t = (int)(i / (8 * n_chunk)), r = (int)(i / n_chunk) % 8, a = (int)(i % n_chunk) * 256 + (int)threadIdx.x;
        const int idx_off = tile_off[t], ydim = tile_off[t + 1] - idx_off;
        const int cy = r < 4 ? r : ydim - 8 + r;         // rows 0 .. 3 and ydim - 4 .. ydim - 1, none twice
        if (!(r < 4 ? cy < ydim : cy >= 4) || a >= G) continue;
        int k = 0, kh = n_chr - 1;                       // the gene's chromosome
        while (k < kh) {
            const int mid = (k + kh + 1) >> 1;
            if (chr_start[mid] <= a) k = mid; else kh = mid - 1;
        }
        const int cs = chr_start[k], xdim = chr_start[k + 1] - cs;
        median9_general_output(in, out, G, tile_idx, idx_off + cy, a, median9_clamp_bits(a - cs, xdim, cy, ydim));
    }
    const int n_list = tile_off[n_tiles];
    for (int e = blockIdx.x; e < n_list; e += gridDim.x) {
        int lo = 0, hi = n_tiles - 1;                    // the cell's tile: the last one that starts at or before e
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tile_off[mid] <= e) lo = mid; else hi = mid - 1;
        }
        const int idx_off = tile_off[lo], ydim = tile_off[lo + 1] - idx_off, cy = e - idx_off;
        if (cy < 4 || cy >= ydim - 4) continue;          // (a border row: done above)
        for (int i = threadIdx.x; i < 8 * n_chr; i += 256) {
            const int k = i >> 3, j = i & 7;
            const int cs = chr_start[k], xdim = chr_start[k + 1] - cs;
            const int gx = j < 4 ? j : xdim - 8 + j;     // genes 0 .. 3 and xdim - 4 .. xdim - 1, none twice
            if (j < 4 ? gx < xdim : gx >= 4)
                median9_general_output(in, out, G, tile_idx, e, cs + gx, median9_clamp_bits(gx, xdim, cy, ydim));
        }
    }
}

// Kernel 3: the queued outputs (one segment per workgroup of kernel 1), one per lane; then every output of the slow list's tiles.
__global__ void __launch_bounds__(256, 2) median9_sparse_kernel(const double *__restrict__ in, double *__restrict__ out, int G,
                                                                 const int32_t *__restrict__ tile_idx, const int4 *__restrict__ gene1_desc,
                                                                 const int4 *__restrict__ cell1_desc, int gene_blocks1, Median9Lists L,
                                                                 const uint4 *__restrict__ fq, const int32_t *__restrict__ fq_count, int fq_cap) {
    if (fq) {   // the outputs the strip kernel could not certify (beyond the capacity: the gated dense pass has rewritten every tile)
        const int nq = *fq_count == 0x7fffffff ? 0 : (*fq_count < fq_cap ? *fq_count : fq_cap);      // (0x7fffffff: the strip kernel did not run)
        for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < nq; i += (int)gridDim.x * 256) {
            const uint4 e = fq[i];
            median9_general_output(in, out, G, tile_idx, (int)e.x, (int)e.y, e.z);
        }
    }
    // (block b starts with segment b and walks on: the segments are about equally long)
    for (int sgm = blockIdx.x; sgm < L.n_seg; sgm += gridDim.x) {
        const int n = L.qcount[sgm];
        const uint4 *q = L.queue + (int64_t)sgm * L.qcap;
        for (int i = (int)threadIdx.x; i < n; i += 256) {
            const uint4 e = q[i];
            median9_general_output(in, out, G, tile_idx, (int)e.x, (int)e.y, e.z);
        }
        const int ns = L.scount[sgm];
        for (int j = 0; j < ns; ++j) {
            // a whole tile of kernel 1 (56 genes x 32 cells): 1 792 outputs, seven per thread
            const int64_t p = L.slow[(int64_t)sgm * L.lcap + j];
            const int4 gd = gene1_desc[p % gene_blocks1], cd = cell1_desc[p / gene_blocks1];
#pragma unroll 1
            for (int o = (int)threadIdx.x; o < K1G * K1C; o += 256) {
                const int gx = gd.z + o % K1G, cy = cd.z + o / K1G;
                if (gx < gd.y && cy < cd.y)
                    median9_general_output(in, out, G, tile_idx, cd.x + cy, gd.x + gx, median9_clamp_bits(gx, gd.y, cy, cd.y));
            }
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
        const int64_t n_tiles9 = (int64_t)plan9.n_gene_blocks1 * plan9.n_cell_patches1;       // kernel 1's tiles
        const int64_t n_tiles2 = (int64_t)plan9.n_gene_blocks * plan9.n_cell_patches;         // the dense pass's (four per tile of kernel 1)
        if (n_tiles9 > 0) {
            static_assert(K1G == MEDIAN9_K1_GENES && K1C == MEDIAN9_K1_CELLS && K1G + 8 == 64 && K1C == 2 * MF9_TC && K1G <= 2 * MF_TG, "host tables");
            if (n_tiles2 > 0x7fffffff) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "median filter: more than 2^31 tiles in one call");
            static const int dev_mode = std::getenv("ICNV_MF9_MODE") ? std::atoi(std::getenv("ICNV_MF9_MODE")) : 0;   // developer switch
            static const int strip_mode = std::getenv("ICNV_MF9_STRIP") ? std::atoi(std::getenv("ICNV_MF9_STRIP")) : 1;   // developer switch: 0 = the fp64 dense pass of rounds 2-5
            static const int probe_mode = std::getenv("ICNV_MF9_PROBE") ? std::atoi(std::getenv("ICNV_MF9_PROBE")) : 1;   // developer switch: 0 = kernel 1 looks for a dominant value by itself, as in round 5
            static const int sweep_mode = std::getenv("ICNV_MF9_SWEEP") ? std::atoi(std::getenv("ICNV_MF9_SWEEP")) : 1;   // developer switch: 0 = kernel 1 (tiles of 56 x 32, four wavefronts each) also when the probe ran, 1 = the sweep on 56-gene blocks, 2 = on 120-gene blocks (16 % fewer bytes read, 209 registers: measured slower, 4.58 vs 4.06 ms)
            const bool strip = strip_mode != 0 && plan9.n_strips > 0 && plan9.n_segs > 0;
            const bool probed = plan9.n_list > 0 && (strip || probe_mode != 0);
            static const int border_mode = std::getenv("ICNV_MF9_BORDER") ? std::atoi(std::getenv("ICNV_MF9_BORDER")) : 1;   // developer switch: 0 = without a dominant value kernel 1 still walks the tiles and queues the border outputs
            const bool sweep = probed && probe_mode != 0 && sweep_mode != 0 && border_mode != 0 && plan9.n_segs > 0 && plan9.n_sweep_blocks1 > 0 && plan9.n_sweep_blocks2 > 0;
            // kernel 1: four workgroups per CU (<= 128 registers; ten 8-byte loads in flight per lane), persistent; their list segments
            const int64_t n_runs1 = n_tiles9 / K1RUN;      // (the host pads kernel 1's cell blocks to a multiple of K1RUN)
            int64_t grid1 = (int64_t)num_cus() * 4;
            if (grid1 > n_runs1) grid1 = n_runs1;
            Median9Lists L;
            L.n_seg = (int)grid1;
            L.lcap = (int)((n_runs1 / grid1 + 1) * K1RUN + 2);
            // kernel 1s (the sweep): one list segment per WAVEFRONT, units (56 or 120 genes x 128 cells) dealt round-robin -- a wavefront's slow list can hold every block it may meet
            const int sweep_nh = sweep_mode == 2 ? 2 : 1;
            const int sweep_blocks = sweep_nh == 1 ? plan9.n_sweep_blocks1 : plan9.n_sweep_blocks2;
            const int4 *sweep_desc = reinterpret_cast<const int4 *>(sweep_nh == 1 ? plan9.sweep_desc1 : plan9.sweep_desc2);
            const int64_t n_units_s = (int64_t)sweep_blocks * plan9.n_segs;
            static const int sweep_wg = std::getenv("ICNV_MF9_SWEEP_WG") ? std::atoi(std::getenv("ICNV_MF9_SWEEP_WG")) : 0;   // developer switch: workgroups per CU of the sweep
            int64_t grid_s = (int64_t)num_cus() * (sweep_wg > 0 ? sweep_wg : 6);
            if (grid_s * 4 > n_units_s) grid_s = (n_units_s + 3) / 4;
            if (sweep) {
                L.n_seg = (int)(grid_s * 4);
                L.lcap = (int)((n_units_s / L.n_seg + 1) * MS_SEG * 4 + 2);      // (a block names up to four of kernel 1's tiles)
                grid1 = L.n_seg;      // (the sizes below are per segment)
            }
            // queue of single outputs: sized for 5 % of a workgroup's outputs, at least two tiles' worth (a neutral region leaves
            // next to nothing undecided, the border outputs of undecided regions are ~3 % of those; a tile that does not fit goes to
            // the slow list as a whole, so the size is a performance knob, not a limit)
            const int64_t outs_per_wg = n_tiles9 * (K1G * K1C) / grid1 + 1;
            L.qcap = (int)std::min<int64_t>(std::max<int64_t>(outs_per_wg * 5 / 100, 2 * K1G * K1C), 1 << 24);
            // ... and never more than 1 GiB over all segments (ADVICE, round 5: 5 % of the outputs at 16 bytes is ~10 % of the matrix bytes --
            // 8 GB for 10 000 x 1 000 000 on a GPU that holds 270 GB of matrices)
            L.qcap = (int)std::min<int64_t>(L.qcap, std::max<int64_t>(((int64_t)1 << 30) / ((int64_t)grid1 * (int64_t)sizeof(uint4)), 2 * K1G * K1C));
            if (const char *e = std::getenv("ICNV_MF9_QCAP")) L.qcap = std::max(1, std::atoi(e));   // developer / test switch: a tiny queue sends tiles to the slow list
            const size_t b_list = ((size_t)grid1 * L.lcap * sizeof(int32_t) + 15) & ~(size_t)15;
            const size_t b_cnt = ((size_t)grid1 * sizeof(int32_t) + 15) & ~(size_t)15;
            const size_t b_flag = ((size_t)n_tiles2 + 15) & ~(size_t)15;
            // (round 6) the strip form of the dense pass: the probe's result, the counter and the queue of its uncertified outputs
            int fq_cap = (int)std::min<int64_t>(std::max<int64_t>(n_tiles2 * (MF_TG * MF9_TC) / 64, 1 << 16), 1 << 22);
            if (const char *e = std::getenv("ICNV_MF9_FQCAP")) fq_cap = std::max(0, std::atoi(e));   // developer / test switch: 0 sends every uncertified output's tile to the gated fp64 pass
            const size_t b_ulist = strip ? (((size_t)plan9.n_strips * plan9.n_segs * sizeof(int2) + 15) & ~(size_t)15) : 0;      // the strip kernel's list of units
            const size_t b_probe = 64 + ((sizeof(ProbeScratch) + 63) & ~(size_t)63), b_fq = (strip ? (size_t)fq_cap * sizeof(uint4) : 0) + b_ulist;
            size_t b_queue = 0;
            for (;;) {   // a pool that cannot give the queue gets a shorter one: more tiles take the slow list, nothing fails
                b_queue = (size_t)grid1 * L.qcap * sizeof(uint4);
                const int rc = plan9.queue->alloc(b_queue + b_list + 2 * b_cnt + b_flag + b_probe + b_fq);
                if (!rc) break;
                if (L.qcap <= 64) return rc;
                L.qcap = std::max(64, L.qcap / 4);
            }
            char *base = plan9.queue->as<char>();
            L.queue = reinterpret_cast<uint4 *>(base); base += b_queue;
            L.slow = reinterpret_cast<int32_t *>(base); base += b_list;
            L.qcount = reinterpret_cast<int32_t *>(base); base += b_cnt;
            L.scount = reinterpret_cast<int32_t *>(base); base += b_cnt;
            L.dflag = reinterpret_cast<uint8_t *>(base); base += b_flag;
            StripParams *probe = reinterpret_cast<StripParams *>(base);
            int32_t *fq_count = reinterpret_cast<int32_t *>(base + 48);
            ProbeScratch *pscratch = reinterpret_cast<ProbeScratch *>(base + 64); base += b_probe;
            uint4 *fq = strip ? reinterpret_cast<uint4 *>(base) : nullptr;
            static_assert(sizeof(StripParams) <= 48 && MS_NSP == 3, "workspace layout");
            ICNV_HIP(hipMemsetAsync(L.dflag, 0, b_flag + b_probe, stream));       // the marks and the strip kernel's counter
            const int4 *gd = reinterpret_cast<const int4 *>(plan9.gene_block_desc), *cd = reinterpret_cast<const int4 *>(plan9.cell_patch_desc);
            const int4 *g1 = reinterpret_cast<const int4 *>(plan9.gene1_desc), *c1 = reinterpret_cast<const int4 *>(plan9.cell1_desc);
            if (probed) {
                hipLaunchKernelGGL(median9_probe_count_kernel, dim3(MP_WG), dim3(256), 0, stream, in, G, tile_idx_dev, plan9.n_list, pscratch);
                hipLaunchKernelGGL(median9_probe_finish_kernel, dim3(1), dim3(MP_CAND), 0, stream, in, G, tile_idx_dev, plan9.n_list,
                                   (const ProbeScratch *)pscratch, probe);
            }
            if (sweep) {
                SweepArgs S;
                S.in = in; S.out = out; S.G = G; S.tile_idx = tile_idx_dev;
                S.gene_desc = sweep_desc; S.seg_desc = reinterpret_cast<const int4 *>(plan9.seg_desc);
                S.gene_blocks = sweep_blocks; S.gene_blocks1 = plan9.n_gene_blocks1; S.gene_blocks2 = plan9.n_gene_blocks;
                S.n_units = n_units_s; S.L = L; S.dev_mode = dev_mode; S.P = probe;
                S.n_flags = (int64_t)b_flag;
                if (sweep_nh == 1) hipLaunchKernelGGL(median9_sweep_kernel<1>, dim3((unsigned)grid_s), dim3(256), 0, stream, S);
                else hipLaunchKernelGGL(median9_sweep_kernel<2>, dim3((unsigned)grid_s), dim3(256), 0, stream, S);
            } else {
                hipLaunchKernelGGL(median9_classify_kernel, dim3((unsigned)grid1), dim3(256), 0, stream, in, out, G, tile_idx_dev, g1, c1,
                                   plan9.n_gene_blocks1, n_tiles9, plan9.n_gene_blocks, L, dev_mode,
                                   (probed && probe_mode != 0) ? (const StripParams *)probe : (const StripParams *)nullptr,
                                   (probed && probe_mode != 0 && border_mode != 0) ? (int64_t)b_flag : (int64_t)0);
            }
            if (probed && probe_mode != 0 && border_mode != 0)
                hipLaunchKernelGGL(median9_border_kernel, dim3((unsigned)std::min<int64_t>((int64_t)num_cus() * 8, (int64_t)plan9.n_list)), dim3(256), 0, stream, in, out, G, tile_idx_dev, tile_off_dev, n_tiles,
                                   chr_start_dev, n_chr, (const StripParams *)probe);
            const size_t lds = ((size_t)2 * (MF_TG + 8) * (MF9_TC + 8) + (size_t)(MF9_TC + 8) * MF_TG * 9) * sizeof(double) +
                               3 * (MF9_TC + 8) * sizeof(int32_t);
            static DeviceOnce once;
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(median_filter9_kernel), 80 * 1024, once)) return rc;
            int64_t grid2 = (int64_t)num_cus() * 2;   // two resident workgroups per CU (70 KB of LDS, 256 registers)
            if (grid2 > n_tiles2) grid2 = n_tiles2;
            if (strip) {
                StripArgs A;
                A.in = in; A.out = out; A.G = G; A.tile_idx = tile_idx_dev;
                A.strip_desc = reinterpret_cast<const int4 *>(plan9.strip_desc);
                A.seg_desc = reinterpret_cast<const int4 *>(plan9.seg_desc);
                A.n_strips = plan9.n_strips;
                A.n_units = (int64_t)plan9.n_strips * plan9.n_segs;
                if (A.n_units > 0x7fffff00) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "median filter: more than 2^31 strip units in one call");      // (a 32-bit counter hands them out)
                A.dflag = L.dflag; A.gene_blocks2 = plan9.n_gene_blocks;
                A.P = probe; A.fq = fq; A.fq_count = fq_count; A.fq_cap = fq_cap;
                A.unit_counter = fq_count + 1;
                A.unit_counts = fq_count + 2;
                int2 *ulist = reinterpret_cast<int2 *>(reinterpret_cast<char *>(fq) + (size_t)fq_cap * sizeof(uint4));
                A.unit_list = ulist;
                hipLaunchKernelGGL(median9_units_kernel, dim3((unsigned)((A.n_units + 255) / 256)), dim3(256), 0, stream, A.strip_desc, A.seg_desc, A.n_strips, A.n_units,
                                   (const uint8_t *)L.dflag, plan9.n_gene_blocks, (const StripParams *)probe, fq_count + 2, ulist);
                int64_t grid2s = (int64_t)num_cus() * MS_WAVES_PER_SIMD;       // resident workgroups of four independent wavefronts
                if (grid2s * 4 > A.n_units) grid2s = (A.n_units + 3) / 4;
                hipLaunchKernelGGL(median9_strip_kernel, dim3((unsigned)grid2s), dim3(256), 0, stream, A);
                // ... and the fp64 dense pass behind it, gated: it returns at once unless the strip kernel's queue overflowed
                hipLaunchKernelGGL(median_filter9_kernel, dim3((unsigned)grid2), dim3(MF_TG * MF_TC), lds, stream, in, out, G, tile_idx_dev, gd, cd,
                                   plan9.n_gene_blocks, L.dflag, n_tiles2, (const int32_t *)fq_count, fq_cap);
            } else {
                hipLaunchKernelGGL(median_filter9_kernel, dim3((unsigned)grid2), dim3(MF_TG * MF_TC), lds, stream, in, out, G, tile_idx_dev, gd, cd,
                                   plan9.n_gene_blocks, L.dflag, n_tiles2, (const int32_t *)nullptr, 0);
            }
            int64_t grid3 = std::min<int64_t>((int64_t)num_cus() * 4, grid1);
            hipLaunchKernelGGL(median9_sparse_kernel, dim3((unsigned)grid3), dim3(256), 0, stream, in, out, G, tile_idx_dev, g1, c1,
                               plan9.n_gene_blocks1, L, (const uint4 *)fq, (const int32_t *)fq_count, fq_cap);
            if (std::getenv("ICNV_MF9_DEBUG")) {   // developer switch: how the tiles were split (synchronises)
                std::vector<int32_t> cnt(2 * (b_cnt / sizeof(int32_t)));
                std::vector<uint8_t> fl((size_t)n_tiles2);
                (void)hipStreamSynchronize(stream);
                (void)hipMemcpy(cnt.data(), L.qcount, 2 * b_cnt, hipMemcpyDeviceToHost);
                (void)hipMemcpy(fl.data(), L.dflag, (size_t)n_tiles2, hipMemcpyDeviceToHost);
                int64_t q = 0, d = 0, sl = 0;
                const size_t stride = b_cnt / sizeof(int32_t);
                for (int64_t i = 0; i < grid1; ++i) { q += cnt[i]; sl += cnt[stride + i]; }
                for (uint8_t f : fl) d += f;
                int32_t nfq = 0;
                StripParams hp;
                (void)hipMemcpy(&nfq, fq_count, sizeof(nfq), hipMemcpyDeviceToHost);
                (void)hipMemcpy(&hp, probe, sizeof(hp), hipMemcpyDeviceToHost);
                fprintf(stderr, "[median9] %lld tiles (56 x 32): %lld of %lld dense-pass tiles (32 x 16) marked, %lld slow, %lld queued outputs (qcap %d per segment, %lld segments); "
                        "strip %d: %d uncertified outputs (capacity %d), probe: dominant value %s (%.17g), scale %.6g\n",
                        (long long)n_tiles9, (long long)d, (long long)n_tiles2, (long long)sl, (long long)q, L.qcap, (long long)grid1,
                        (int)strip, (int)nfq, fq_cap, hp.has_dom ? "yes" : "no", hp.sp[0], hp.scale);
                fprintf(stderr, "[median9] probe: %u repeated values (%s): %.17g %.17g %.17g\n", hp.n_sp & 0xffu, (hp.n_sp & 0x100u) ? "and many more: fp64 dense pass" : "strip kernel", hp.sp[0], hp.sp[1], hp.sp[2]);
            }
        }
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
