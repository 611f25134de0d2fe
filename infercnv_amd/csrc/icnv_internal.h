// Internal declarations shared by the HIP translation units of libicnv_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <stdint.h>

#include <string>

#include "../../include/icnv.h"

namespace icnv {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define ICNV_HIP(call)                                                      \
    do {                                                                    \
        hipError_t _e = (call);                                             \
        if (_e != hipSuccess) return ::icnv::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define ICNV_FAIL(code, msg)        \
    do {                            \
        ::icnv::set_error(msg);     \
        return (code);              \
    } while (0)

// Device workspace pool (api.hip): grow-only, per device (and per pool partition of the calling thread).
int pool_alloc(void **p, size_t bytes);
void pool_free(void *p);
void set_pool_part(int part);
double pool_malloc_ms(bool reset);   // wall time the pool spent in hipMalloc since the last reset
struct DevBuf {  // RAII over the pool
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p) { o.p = nullptr; }
    DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { pool_free(p); p = o.p; o.p = nullptr; } return *this; }
    ~DevBuf() { pool_free(p); }
    int alloc(size_t bytes) { pool_free(p); p = nullptr; return pool_alloc(&p, bytes); }
    void release() { pool_free(p); p = nullptr; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Host-buffer path (host_path.hip): a device copy of a host matrix, either recognised as resident (icnv_residency) or
// uploaded on `s`; and the registration of a freshly downloaded result.
struct MatrixLease {
    const double *dev = nullptr;
    DevBuf own;               // the upload when the matrix is not kept resident
    void *entry = nullptr;    // the resident entry pinned by this lease
    MatrixLease() = default;
    MatrixLease(const MatrixLease &) = delete;
    ~MatrixLease();
};
int acquire_input(const double *host, int64_t n_doubles, hipStream_t s, MatrixLease &lease);
void publish_output(const double *host, int64_t n_doubles, DevBuf &&buf);
int viterbi_groups_host_one(const double *expr, uint8_t *states, int64_t G, int64_t C, const int32_t *chr_start, int32_t n_chr,
                            const int32_t *grp_idx, const int32_t *grp_off, int32_t n_grp, int32_t K, const double *mean,
                            const double *sd_shared_per_grp, const double *logPi, const double *logDelta);
int median_filter_host_one(const double *expr_in, double *expr_out, int64_t G, int64_t C, const int32_t *chr_start, int32_t n_chr,
                           const int32_t *tile_idx, const int32_t *tile_off, int32_t n_tiles, int32_t window_size);
// api.hip: the fused apply pass over the columns [c0, c1) of the chain's matrix (the reference statistics are in place);
// -1000 when this chain needs the whole-matrix call (three-pass chain, noise_logistic, HMM input without a denoise stage)
int chain_apply_columns(icnv_chain_t *ch, const double *expr_in, double *expr_out, double *pre_denoise, int64_t c0, int64_t c1,
                        hipStream_t s);
bool residency_release_idle();   // host_path.hip: hand the idle resident matrices of this pool domain back to the pool
void viterbi_release_contexts();  // api.hip: device tables / pinned words / events of the per-device Viterbi state

// Optional per-kernel timing with hipEvents recorded on the launch stream.
struct KernelTimer {
    KernelTimer(const char *name, hipStream_t s);
    ~KernelTimer();
    const char *name;
    hipStream_t stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int dev = 0;          // device the events belong to (the event pool is per device)
    bool on = false;
};

int num_cus();
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set it once for every device the
// calling thread's launches go to (`done` = one bit per device ordinal, owned by the launch site)
struct DeviceOnce { unsigned long long done = 0; };
int ensure_dynamic_lds(const void *kernel, int bytes, DeviceOnce &once);

// ---- smoothing chain ------------------------------------------------------
enum ChainMode { MODE_APPLY = 0, MODE_GENE_SUMS = 1, MODE_CELL_STATS = 2 };

struct ChainArgs {
    const double *in;       // G x C_total column-major
    double *out;            // MODE_APPLY
    double *pre_out;        // MODE_APPLY, nullable: matrix before step 22
    int32_t G;
    const int32_t *cells;   // nullable: columns to process (else 0..n_cells-1)
    int32_t n_cells;
    const int32_t *chr_start;  // device, n_chr+1
    int32_t n_chr;
    int32_t T;              // half window, (W-1)/2; 0 = no smoothing
    int32_t pad;            // zeros between chromosomes in LDS (set by launch_chain)
    uint32_t mask;          // ICNV_ST_* stages applied in this pass
    int32_t use_bounds;
    int32_t inv_log;        // MODE_GENE_SUMS with no stage before it: accumulate 2^x - 1 (subtract_ref_expr_from_obs(inv_log = TRUE))
    double max_thresh;
    const double *b1;       // [2*G] lo | hi  (step 8)
    const double *b2;       // [2*G] lo | hi  (step 12)
    const double *denoise;  // [2] mu, s
    const double *inv_pos;  // smoothing: 1/denominator per padded LDS position (chain_build_inv_table)
    const uint32_t *inv_codes;  // one byte per position: index into inv_dict; [(LMAX+3)/4][NT]
    int32_t inv_coded;          // 1: every distinct value has a code (else the generic kernels read inv_pos)
    const double *inv_dict;     // [256] distinct 1/denominator values, entry 0 = 0.0 (padding)
    double *partial;        // MODE_GENE_SUMS: [gridDim.x * G]
    double *cache_out;      // MODE_GENE_SUMS, nullable: the stages' output of list position i goes to cache_out[i * G ..]
    int32_t in_by_pos;      // 1: `in` holds one column per list position (a cache written through cache_out)
    double *cell_stats;     // MODE_CELL_STATS: [n_cells * 2] {sum, sd}
    int32_t out_by_pos;     // 1: `out` receives one column per list position (the reference-cell cache written by round B)
    int32_t pre_ld;         // doubles between the columns of pre_out (G unless the caller asked for padded columns)
    int32_t ld;             // 0, or (strided view, launch_chain_strided): doubles between the columns of in / out / pre_out, and between b1 / b2's lower and upper vectors
};

int launch_chain(const ChainArgs &a, int mode, hipStream_t stream);
// Steps 8 / 9 / 10 (a.mask: any of them, with step 10) of a VIEW: the genes [g0, g0 + a.G) of every column -- a group of whole
// chromosomes that fits the 1024 x 11 geometry (chain_view_fits) -- a.in / a.out point at gene g0 of column 0, a.b1 at gene g0 of the
// lower vector, columns a.ld doubles apart (chain_w11s.hip).  Pass 1 of the chain for gene sets beyond the LDS-resident limit.
int launch_chain_strided(const ChainArgs &a, hipStream_t stream);
bool chain_view_fits(int64_t G_view, int32_t n_chr_view, int32_t T);
// chain_na.hip: the list positions of `a` whose input column holds a NaN, recomputed with the reference's NA semantics
// (same arguments as the apply launch it follows; flags_ws: a.n_cells bytes of workspace)
// (all_flag, nullable: a device word that flags every cell when non-zero -- a NaN mean in the no-bounds mode)
int launch_chain_na_fixup(const ChainArgs &a, int32_t max_chr_len, uint8_t *flags_ws, const int32_t *all_flag, hipStream_t stream);
int launch_nan_flags(const ChainArgs &a, uint8_t *flags_ws, const int32_t *all_flag, hipStream_t stream);          // its two halves, for a chain
int launch_chain_na_cells(const ChainArgs &a, int32_t max_chr_len, const uint8_t *flags, hipStream_t stream);       // that runs in place
int launch_gather_columns(const double *x, int32_t G, const int32_t *ids_dev, int32_t n, double *stash, hipStream_t stream);
int chain_max_genes();
bool chain_fused_fits(int64_t G, int32_t n_chr, int32_t T);   // does the LDS-resident fused kernel take this geometry?

// three-pass chain for gene sets beyond the fused kernel's limit (chain_large.hip)
struct LargeChainArgs {
    const double *in;          // rows of G doubles
    double *out;               // S: stage output; M: in place on `out`; E: final matrix (may alias `in`)
    double *pre;               // E, nullable: matrix before step 22
    int32_t G;
    const int32_t *in_rows;    // nullable (identity): row of `in` read by launch row r
    const int32_t *out_rows;   // nullable (identity): row of `out` / `pre` written by launch row r
    int32_t n_rows;
    const int32_t *chr_start;  // device, n_chr + 1
    int32_t n_chr;
    int32_t T;
    uint32_t mask;             // the ICNV_ST_* bits this pass applies
    double max_thresh;
    const double *b1, *b2;     // [2*G] lo | hi
    const double *denoise;     // [2] mu, s
    double *cell_stats;        // E, nullable: [n_rows * 2] {sum, sd} instead of writing the matrix
};
size_t chain_large_lds_bytes(int32_t max_chr_len, int32_t T);
int launch_chain_large_smooth(const LargeChainArgs &a, int32_t max_chr_len, hipStream_t stream);
int launch_chain_large_center(const LargeChainArgs &a, hipStream_t stream);
int launch_chain_large_finish(const LargeChainArgs &a, hipStream_t stream);
bool chain_large_center_finish_covers(int32_t G);
int launch_chain_large_center_finish(const LargeChainArgs &a, hipStream_t stream);   // steps 11, 12, 14, 22 in ONE pass over the smoothed rows
int launch_chain_large_group_sums(const double *x, int32_t G, const int32_t *idx_dev, const int32_t *off_dev, int32_t n_grp,
                                  double *sums_counts, hipStream_t stream);
// Host: per-position normalisation table of the smoothing stage for this geometry, in the kernel's
// [(LMAX+1)/2][NT] double2 layout (R/inferCNV_ops.R:2410-2440: pyramid weights renormalised at chromosome edges).
int chain_build_inv_table(const int32_t *chr_start, int32_t n_chr, int32_t G, int32_t T, std::vector<double> &tab,
                          std::vector<uint32_t> &codes, std::vector<double> &dict, bool &coded, bool view_1024x11 = false);
int launch_reduce_partials(const double *partial, int nblk, int32_t G, double *out, double count,
                           double *count_out, hipStream_t stream);
// raw per-gene sums of all reference groups in one launch (+ the reduction over its splits): sums_counts = [G*n_grp | n_grp]
int launch_group_gene_sums(const double *x, int32_t G, const int32_t *cells_dev, const int32_t *off_dev, int32_t n_grp,
                           double *partial, int32_t partial_rows, double *sums_counts, hipStream_t stream);
// (nan_flag, nullable: set to 1 when a stored bound is NaN -- NA-aware chains in the no-bounds mode look at it)
int launch_bounds_from_sums(const double *sums_counts, int32_t G, int32_t n_grp, int32_t use_bounds, int32_t inv_log,
                            double *bounds, int32_t *nan_flag, hipStream_t stream);
bool cache_cell_stats_covers(int32_t G);   // even G: the streaming kernel reads gene pairs
int launch_cache_cell_stats(const double *cache, int32_t G, int32_t n_cells, uint32_t mask /* steps 12 / 14 still to run */, const double *b2,
                            double *cell_stats, hipStream_t stream);
int launch_reduce_cell_stats(const double *cell_stats, int32_t n_cells, int32_t G, double *out4,
                             hipStream_t stream);
int launch_denoise_from_stats(const double *stats4, double sd_amplifier, double noise_filter,
                              double *mu_s, hipStream_t stream);
// ---- gene filters / block statistics (stats_kernels.hip) ----
int gene_stats_nsplit(int32_t G, int64_t C);
int launch_gene_stats(const double *x, int32_t G, int64_t C, int nsplit, double *part_sum, int32_t *part_nnz, double *sums,
                      int32_t *nnz, hipStream_t stream);
int launch_select_genes(const double *in, int32_t G_in, int64_t C, const int32_t *keep_dev, int32_t G_out, double *out,
                        hipStream_t stream);
int launch_block_cell_reduce(int pass, const double *x, int32_t G, const int32_t *gene_idx_dev, int32_t n_genes,
                             const int32_t *cell_idx_dev, int32_t n_cells, double mean, double *out, hipStream_t stream);
int launch_scale_genes(const double *in, double *out, int32_t G, int64_t C, int nsplit, double *part, double *mean_sd, hipStream_t stream);
int launch_clamp_bounds(const double *in, double *out, int64_t n, double lo, double hi, hipStream_t stream);
int launch_logistic_denoise(double *x, int64_t n, const double *mu_s_dev, hipStream_t stream);
int launch_gather_values(const double *x, const int64_t *offsets_dev, int64_t n, double *out, hipStream_t stream);
int launch_col_sums(const double *x, int32_t G, int64_t C, double *out, hipStream_t stream);
int launch_normalize_log2(const double *in, double *out, int32_t G, int64_t C, const double *col_sums, double factor,
                          int do_norm, int do_log, hipStream_t stream);
int launch_minmax_cells(const double *x, int32_t G, int64_t C, double *out2_dev, hipStream_t stream);

// ---- HMM ------------------------------------------------------------------
struct HmmParams {
    int32_t K;
    double mean[8];
    double logPi[64];   // [j + K*k]
    double logDelta[8];
};

int launch_viterbi(const double *x, uint8_t *states, int32_t G, int64_t n_seq_cols, const int32_t *chr_start_dev,
                   const int32_t *chr_order_dev, int32_t n_chr, int32_t max_chr_len, const HmmParams &p,
                   const double *sd_per_col_dev, double sd_shared, uint32_t *bp_scratch, int32_t *n_underflow,
                   const int32_t *gate_dev, int32_t gate_limit, hipStream_t stream,   // gate_dev: run only if *gate_dev > gate_limit
                   int64_t ld_x = 0, int64_t ld_st = 0);   // elements between the columns of x / states (0: G, contiguous columns)
size_t viterbi_scratch_bytes(int32_t G, int64_t n_cols);

// certified fast path (viterbi_fast.hip): table-driven scores + decision-margin test, flagged sequences redone exactly
struct EmisTable;
struct FastViterbiArgs {
    const double *x;
    uint8_t *states;
    int32_t G;
    int64_t ncols;
    const int32_t *chr_start;   // device
    const int32_t *chr_order;   // device, longest chromosome first
    int32_t n_chr;
    const double *table;        // device image (viterbi_fast_table_image)
    int32_t n_int, n_grid;      // records, grid intervals (emission_table.h)
    double mean[8];
    double logDelta[8];
    double a, b;                // log off-diagonal / diagonal transition probability
    double x_lo, x_hi;          // table domain
    double inv_w;               // interval of x = (int)((x - x_lo) * inv_w)
    double eps;                 // eps_tab + eps_spec
    double b0, s_step;          // |value| bound of the recurrence: B = b0 + (n + 1) s_step
    uint16_t *bp;               // [G][ncols] words, then [(G >> 4) + 3 n_chr][ncols] block summaries (viterbi_fast_scratch_bytes)
    int32_t *task_counter;      // zeroed before the launch
    int32_t *flag_count;        // zeroed before the launch
    int32_t *flag_list;         // [2 * n_chr * ncols] (chromosome, column) pairs
    const int32_t *gate_count;  // null, or: the flag count of a first attempt on this batch -- the kernel runs only if it
    int32_t gate_limit;         // exceeds gate_limit, and hands it on as its own count otherwise
    // Round 6: the columns of x / states may lie further apart than G elements (icnv_viterbi_cells_ld_dev): with a leading dimension
    // that is a multiple of 16 every column starts on a cache line and every state column on a 16-byte word, whatever G is -- the
    // staged kernel's chunks are whole lines again and the traceback's blocks aligned stores (a gene count that is not a multiple of
    // 16 cost +27 % before).  The host-buffer entry points upload into such a layout by themselves.
    int64_t ld_x, ld_st;
};
size_t viterbi_fast_scratch_bytes(int32_t G, int32_t n_chr, int64_t n_cols);
size_t viterbi_fast_lds_bytes(int K, int n_int, int n_grid, bool staged);
int viterbi_fast_max_intervals(int K, bool staged);   // staged: the variant that streams the observations through LDS (shorter table)
void viterbi_fast_table_image(const EmisTable &t, std::vector<double> &img);
int launch_viterbi_fast(const FastViterbiArgs &a, int K, bool staged, hipStream_t stream);
int viterbi_redo_slots();
size_t viterbi_redo_scratch_bytes(int32_t max_chr_len);
int launch_viterbi_redo(const double *x, uint8_t *states, int32_t G, const int32_t *chr_start_dev, const HmmParams &p,
                        const double *sd_per_col_dev, double sd_shared, const int32_t *flag_count_dev,
                        const int32_t *flag_list_dev, int32_t max_count /* lists longer than this are left alone */,
                        uint32_t *bp_redo, int32_t *n_underflow, int32_t max_chr_len, const char *timer_name,
                        hipStream_t stream, int64_t ld_x = 0, int64_t ld_st = 0,
                        const double *params_dev = nullptr /* [mean_0 .. mean_{K-1}, sd] on the device instead of p.mean / sd_shared */);
int group_means_nsplit(int32_t G, int32_t n_grp);
int launch_group_means_ws(const double *x, int32_t G, const int32_t *grp_idx_dev, const int32_t *grp_off_dev,
                          int32_t n_grp, int nsplit, double *part, double *out, hipStream_t stream,
                          const uint8_t *ref_flag_dev = nullptr, double *mom = nullptr,   // (+ the shifted moments of the flagged cells' values, two doubles per workgroup;
                          const uint8_t *grp_kind_dev = nullptr);                         //  grp_kind[q]: 0 no flagged cell in group q, 1 only flagged cells, 2 mixed / unknown)
int64_t group_means_moment_blocks(int32_t G, int32_t n_grp, int nsplit);
int launch_reduce_moments(const double *mom, int64_t n_blocks, double n_values, double *out3, hipStream_t stream);
int launch_i3_params(const double *m3, double z, double delta_abs, double *params6, hipStream_t stream);
int launch_broadcast_states(const uint8_t *grp_states, int32_t G, int64_t C, const int32_t *cell_to_grp_dev,
                            uint8_t *states, hipStream_t stream);
int launch_state_consensus(const uint8_t *states, int32_t G, const int32_t *grp_idx_dev, const int32_t *grp_off_dev,
                           int32_t n_grp, uint8_t *out, hipStream_t stream);
int launch_broadcast_states_keep(const uint8_t *grp_states, int32_t G, int64_t C, const int32_t *cell_to_grp_dev,
                                 uint8_t *states, hipStream_t stream);
int launch_states_to_proxy(const uint8_t *states, double *out, int64_t n, int32_t K, hipStream_t stream);

// ---- cell-cell distances (distance_kernels.hip) ----
int launch_cell_distances(const double *x, int32_t G, const int32_t *idx_dev, int32_t n, const double *mean_dev,
                          double *diag_dev, double *out, hipStream_t stream);

// ---- median filter --------------------------------------------------------
// The default window (9 x 9) runs on tile descriptors built by the host (api.hip): kernel 1's tiles (56 genes x 32 cells of a
// (cell tile, chromosome) block, borders included) and the 2 x 2 dense-pass tiles each of them covers; other windows use blk_off.
struct Median9Plan {
    const int32_t *gene_block_desc = nullptr;   // dense pass, 4 ints per gene block: {chromosome's first gene, its length, block's first gene, end of its interior outputs}
    const int32_t *cell_patch_desc = nullptr;   // dense pass, 4 ints per cell block: {offset of the tile's cells, tile length, block's first cell, 0}
    const int32_t *gene1_desc = nullptr;        // kernel 1, 4 ints: {chromosome's first gene, its length, tile's first gene, index of its first dense-pass gene block}
    const int32_t *cell1_desc = nullptr;        // kernel 1, 4 ints: {offset of the tile's cells, tile length, tile's first cell, index of its first dense-pass cell block}
    int32_t n_gene_blocks = 0, n_cell_patches = 0, n_gene_blocks1 = 0, n_cell_patches1 = 0;
    // the strip form of the dense pass (round 6): 64-gene strips of the chromosomes x segments of eight 16-cell blocks of the cell tiles
    const int32_t *strip_desc = nullptr;        // 4 ints: {chromosome's first gene, its length, strip's first gene, index of that gene's dense-pass gene block}
    const int32_t *seg_desc = nullptr;          // 4 ints: {offset of the tile's cells, tile length, segment's first cell, index of that cell's dense-pass cell block}
    int32_t n_strips = 0, n_segs = 0;
    int32_t n_list = 0;                         // entries of the tiles' cell lists (the probe samples them)
    // the sweep form of the classification pass (round 6): gene blocks of 56 / 120 genes, two records of 4 ints each:
    // {chromosome's first gene, its length, block's first gene, index of the chromosome's first dense-pass gene block}, {index of the chromosome's first gene block of kernel 1, 0, 0, 0}
    const int32_t *sweep_desc1 = nullptr, *sweep_desc2 = nullptr;
    int32_t n_sweep_blocks1 = 0, n_sweep_blocks2 = 0;
    DevBuf *queue = nullptr;                    // workspace of the three-kernel scheme's lists (allocated by the launch, owned by the caller)
};
int launch_median_filter(const double *in, double *out, int32_t G, int64_t C, const int32_t *chr_start_dev,
                         int32_t n_chr, const int32_t *tile_idx_dev, const int32_t *tile_off_dev, int32_t n_tiles,
                         const int32_t *blk_off_dev, const int32_t *chr_start_host, int32_t total_cell_patches,
                         int32_t window_size, const Median9Plan &plan9, hipStream_t stream);
constexpr int MEDIAN_GENES_PER_PATCH = 32;
constexpr int MEDIAN9_CELLS_PER_PATCH = 16;
constexpr int MEDIAN9_K1_RUN = 1;            // kernel 1 walks runs of this many tiles down the cells: the host pads its cell blocks to a multiple
constexpr int MEDIAN9_K1_GENES = 56, MEDIAN9_K1_CELLS = 32;   // kernel 1's tile: with the halo one tile row is the 64 lanes of a wavefront
inline bool median_is_9x9(int32_t window_size) { return (window_size - 1) / 2 + 1 == 4; }
constexpr int MEDIAN_CELLS_PER_PATCH = 8;   // generic kernel
constexpr int MEDIAN9_STRIP_GENES = 64, MEDIAN9_SEG_BLOCKS = 8;   // strip kernel: output genes per wavefront, 16-cell blocks per segment

}  // namespace icnv
