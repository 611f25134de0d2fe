// The reference's NA semantics of the smoothing chain, for the cells that hold a NaN (R's NA_real_ is a NaN payload).
//
// infercnv::run() cannot produce one -- its chain input is log2(x + 1) of counts -- and the fused kernels do not look for
// one.  A caller whose matrix may hold NAs sets ICNV_ST_NA_AWARE in the stage mask (the R glue does when anyNA(expr.data)):
// the fused pass runs as always, then the cells that contain a NaN are found (one pass over the input) and recomputed here,
// stage by stage, the way R treats an NA:
//   step  8 / 12  use_bounds: `which(x > hi)` / `which(x < lo)` never select an NA, so the NA comes out as 0 -- and so does
//                 every value of a gene whose reference mean is NA (R/inferCNV_ops.R:1757-1768); without bounds x - mean keeps it
//   step  9       `x[x > thr] <- thr` leaves an NA alone (:2974-2975)
//   step 10       .smooth_helper strips the NAs of a chromosome, smooths the shortened sequence -- the genes either side of a
//                 gap are neighbours -- and puts the NAs back (:2487-2489, 2529)
//   step 11       median(x, na.rm = TRUE) / mean(x, na.rm = TRUE) over the values present; the NA stays (:2098, 2104)
//   step 14       2^NA = NA;  step 22: `which(x > lo & x < hi)` never selects an NA (:2335)
// A slow path on purpose: one workgroup per flagged cell, the cell's working column in global memory (the output column
// itself), one chromosome at a time compacted into LDS, direct (2T + 1)-tap pyramid sums with the reference's renormalisation
// at the ends, the median by an eight-pass radix select over order-preserving keys.
#include <algorithm>

#include "icnv_internal.h"
#include "icnv_exp2_coef.h"

namespace icnv {

namespace {

constexpr int NA_NT = 256;

__device__ inline bool na_isnan(double x) { return x != x; }

__device__ inline double na_block_sum(double v, double *red) {   // fixed order: lane butterflies, then the wavefronts in order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < NA_NT / 64; ++w) r += red[w];
    __syncthreads();
    return r;
}

// cells that hold at least one NaN: flags[c] = 1
// (`all_flag`, nullable: a device word that flags EVERY cell when it is non-zero -- a step-8 / 12 mean that is NaN in the
// no-bounds mode, where x - NA = NA for that gene in every cell, R/inferCNV_ops.R:1770-1776)
__global__ void __launch_bounds__(256) nan_cells_flag_kernel(const double *__restrict__ x, int G, const int32_t *__restrict__ cells,
                                                            int in_by_pos, int64_t n_cells, uint8_t *__restrict__ flags,
                                                            const int32_t *__restrict__ all_flag) {
    const bool all = all_flag && *all_flag != 0;
    for (int64_t i = blockIdx.x; i < n_cells; i += gridDim.x) {
        const int64_t col = (cells && !in_by_pos) ? cells[i] : i;
        const double *p = x + col * (int64_t)G;
        bool any = all;
        if (!all)
            for (int g = threadIdx.x; g < G; g += 256) any |= na_isnan(p[g]);
        if (__builtin_amdgcn_ballot_w64(any) != 0ull && (threadIdx.x & 63) == 0) flags[i] = 1;
    }
}

// stash[i * G ..] = x[ids[i] * G ..]: the input columns of the flagged cells, kept when the chain runs in place
__global__ void __launch_bounds__(256) gather_columns_kernel(const double *__restrict__ x, int G, const int32_t *__restrict__ ids, int n,
                                                            double *__restrict__ stash) {
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const double *p = x + (int64_t)ids[i] * G;
        double *d = stash + (int64_t)i * G;
        for (int g = threadIdx.x; g < G; g += 256) d[g] = p[g];
    }
}

__device__ inline unsigned long long na_key(double x) {   // order-preserving key of a non-NaN double
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double na_unkey(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// the value of rank r (0-based, ascending) among the non-NaN values of col[0 .. G): radix select, eight bits a pass
__device__ double na_select(const double *col, int G, int r, unsigned int *hist /* LDS [256] */, unsigned long long *sh /* LDS [2] */) {
    unsigned long long prefix = 0ull;
    int rank = r;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        for (int i = threadIdx.x; i < 256; i += NA_NT) hist[i] = 0u;
        __syncthreads();
        for (int g = threadIdx.x; g < G; g += NA_NT) {
            const double v = col[g];
            if (na_isnan(v)) continue;
            const unsigned long long k = na_key(v);
            if (pass == 0 || (k >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(unsigned int)(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int acc = 0, b = 0;
            for (; b < 256; ++b) {
                const int c = (int)hist[b];
                if (rank < acc + c) break;
                acc += c;
            }
            sh[0] = prefix | ((unsigned long long)b << shift);
            sh[1] = (unsigned long long)(rank - acc);
        }
        __syncthreads();
        prefix = sh[0];
        rank = (int)sh[1];
        __syncthreads();
    }
    return na_unkey(prefix);
}

__device__ inline double na_exp2(double x) {
    if (!(__builtin_fabs(x) < 1022.0)) return exp2(x);   // also NaN, +-Inf
    const double n = __builtin_rint(x);
    const double f = x - n;
    double p = ICNV_EXP2_C11;
    p = __builtin_fma(p, f, ICNV_EXP2_C10);
    p = __builtin_fma(p, f, ICNV_EXP2_C9);
    p = __builtin_fma(p, f, ICNV_EXP2_C8);
    p = __builtin_fma(p, f, ICNV_EXP2_C7);
    p = __builtin_fma(p, f, ICNV_EXP2_C6);
    p = __builtin_fma(p, f, ICNV_EXP2_C5);
    p = __builtin_fma(p, f, ICNV_EXP2_C4);
    p = __builtin_fma(p, f, ICNV_EXP2_C3);
    p = __builtin_fma(p, f, ICNV_EXP2_C2);
    p = __builtin_fma(p, f, ICNV_EXP2_C1);
    p = __builtin_fma(p, f, 1.0);
    return __builtin_ldexp(p, (int)n);
}

// .subtract_expr for one value (R/inferCNV_ops.R:1757-1776): comparisons with an NA (value or bound) select nothing
__device__ inline double na_subtract(double x, double lo, double hi, int use_bounds) {
    if (!use_bounds) return x - lo;            // lo == hi == mean of the group means
    if (x > hi) return x - hi;
    if (x < lo) return x - lo;
    return 0.0;
}

struct NaArgs {
    const double *in;          // the stages' input
    double *out;               // final matrix (its column is the working column)
    double *pre;               // nullable: the matrix before step 22
    int32_t G;
    const int32_t *cells;      // nullable: the columns of this launch's list positions
    int32_t in_by_pos, out_by_pos;
    int64_t n_cells;
    const uint8_t *flags;      // per list position
    const int32_t *chr_start;  // device
    int32_t n_chr, T;
    uint32_t mask;
    int32_t use_bounds;
    double max_thresh;
    const double *b1, *b2, *denoise;
};

__global__ void __launch_bounds__(NA_NT) chain_na_cells_kernel(const NaArgs a) {
    extern __shared__ __attribute__((aligned(16))) double cmp[];   // the chromosome's values present, compacted [max chromosome length]
    __shared__ double red[NA_NT / 64];
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long sh[2];
    __shared__ int wcount[NA_NT / 64 + 1];
    const int t = threadIdx.x, G = a.G;
    const uint32_t mask = a.mask;
    for (int64_t it = blockIdx.x; it < a.n_cells; it += gridDim.x) {
        if (!a.flags[it]) continue;            // (workgroup-uniform)
        const int64_t cin = (a.cells && !a.in_by_pos) ? a.cells[it] : it;
        const int64_t cout = (a.cells && !a.out_by_pos) ? a.cells[it] : it;
        const double *src = a.in + cin * (int64_t)G;
        double *col = a.out + cout * (int64_t)G;
        // ---- steps 8, 9 (elementwise) into the working column
        for (int g = t; g < G; g += NA_NT) {
            double x = src[g];
            if (mask & ICNV_ST_SUBTRACT_REF_1) x = na_subtract(x, a.b1[g], a.b1[G + g], a.use_bounds);
            if (mask & ICNV_ST_MAX_THRESH) {
                if (x > a.max_thresh) x = a.max_thresh;
                else if (x < -a.max_thresh) x = -a.max_thresh;
            }
            col[g] = x;
        }
        __syncthreads();
        // ---- step 10: per chromosome, over the values present
        if ((mask & ICNV_ST_SMOOTH) && a.T >= 1) {
            const int T = a.T;
            for (int k = 0; k < a.n_chr; ++k) {
                const int s0 = a.chr_start[k], n = a.chr_start[k + 1] - s0;
                if (n < 2) continue;           // R/inferCNV_ops.R:2417: a chromosome of one gene is left alone
                // stable compaction: position of gene s0 + i among the values present (chunks of NA_NT genes, wavefront ballots)
                int base = 0;
                for (int c0 = 0; c0 < n; c0 += NA_NT) {
                    const int i = c0 + t;
                    const double v = i < n ? col[s0 + i] : 0.0;
                    const bool ok = i < n && !na_isnan(v);
                    const unsigned long long b = __builtin_amdgcn_ballot_w64(ok);
                    if ((t & 63) == 0) wcount[t >> 6] = __builtin_popcountll(b);
                    __syncthreads();
                    int off = base;
                    for (int w = 0; w < (t >> 6); ++w) off += wcount[w];
                    int tot = 0;
                    for (int w = 0; w < NA_NT / 64; ++w) tot += wcount[w];
                    if (ok) cmp[off + __builtin_popcountll(b & ((1ull << (t & 63)) - 1ull))] = v;
                    base += tot;
                    __syncthreads();
                }
                const int m = base;            // values present on this chromosome
                // every present gene: the pyramid over its neighbours among the values present, renormalised at the ends
                int ci_base = 0;
                for (int c0 = 0; c0 < n; c0 += NA_NT) {
                    const int i = c0 + t;
                    const double v = i < n ? col[s0 + i] : 0.0;
                    const bool ok = i < n && !na_isnan(v);
                    const unsigned long long b = __builtin_amdgcn_ballot_w64(ok);
                    if ((t & 63) == 0) wcount[t >> 6] = __builtin_popcountll(b);
                    __syncthreads();
                    int off = ci_base;
                    for (int w = 0; w < (t >> 6); ++w) off += wcount[w];
                    int tot = 0;
                    for (int w = 0; w < NA_NT / 64; ++w) tot += wcount[w];
                    double res = v;
                    if (ok && m >= 1) {
                        const int ci = off + __builtin_popcountll(b & ((1ull << (t & 63)) - 1ull));
                        const int lo = ci - T > 0 ? ci - T : 0, hi = ci + T < m - 1 ? ci + T : m - 1;
                        double num = 0.0, den = 0.0;
                        for (int j = lo; j <= hi; ++j) {
                            const double wgt = (double)(T + 1 - (j > ci ? j - ci : ci - j));
                            num += wgt * cmp[j];
                            den += wgt;
                        }
                        res = num / den;
                    }
                    __syncthreads();           // (everybody has read wcount)
                    if (ok) col[s0 + i] = res;
                    ci_base += tot;
                }
                __syncthreads();
            }
        }
        // ---- step 11: centre on the values present
        double centre = 0.0;
        if (mask & ICNV_ST_CENTER) {
            int cnt = 0;
            double s = 0.0;
            for (int g = t; g < G; g += NA_NT) {
                const double v = col[g];
                if (!na_isnan(v)) { ++cnt; s += v; }
            }
            const int n_ok = (int)(na_block_sum((double)cnt, red) + 0.5);
            if (n_ok == 0) centre = __builtin_nan("");                      // median / mean of nothing: NA
            else if (mask & ICNV_ST_CENTER_MEAN) centre = na_block_sum(s, red) / (double)n_ok;
            else if (n_ok & 1) centre = na_select(col, G, n_ok / 2, hist, sh);
            else centre = (na_select(col, G, n_ok / 2 - 1, hist, sh) + na_select(col, G, n_ok / 2, hist, sh)) * 0.5;
        }
        // ---- steps 11 (subtract), 12, 14, 22 and the stores
        double mu = 0.0, lo_d = 0.0, hi_d = 0.0;
        if (mask & ICNV_ST_DENOISE) { mu = a.denoise[0]; lo_d = mu - a.denoise[1]; hi_d = mu + a.denoise[1]; }
        double *pre = a.pre ? a.pre + cout * (int64_t)G : nullptr;
        for (int g = t; g < G; g += NA_NT) {
            double x = col[g];
            if (mask & ICNV_ST_CENTER) x -= centre;
            if (mask & ICNV_ST_SUBTRACT_REF_2) x = na_subtract(x, a.b2[g], a.b2[G + g], a.use_bounds);
            if (mask & ICNV_ST_INVERT_LOG2) x = na_exp2(x);
            if (pre) pre[g] = x;
            if ((mask & ICNV_ST_DENOISE) && x > lo_d && x < hi_d) x = mu;
            col[g] = x;
        }
        __syncthreads();
    }
}

}  // namespace

// `flags_ws`: n_cells bytes of workspace.  Flags the list positions whose input column holds a NaN.
int launch_nan_flags(const ChainArgs &a, uint8_t *flags_ws, const int32_t *all_flag, hipStream_t stream) {
    if (a.n_cells <= 0) return ICNV_OK;
    ICNV_HIP(hipMemsetAsync(flags_ws, 0, (size_t)a.n_cells, stream));
    const int grid1 = (int)std::min<int64_t>(a.n_cells, (int64_t)num_cus() * 16);
    hipLaunchKernelGGL(nan_cells_flag_kernel, dim3(grid1), dim3(256), 0, stream, a.in, a.G, a.cells, a.in_by_pos, (int64_t)a.n_cells, flags_ws,
                       all_flag);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

// the flagged list positions of `a` recomputed from a.in with the reference's NA semantics
int launch_chain_na_cells(const ChainArgs &a, int32_t max_chr_len, const uint8_t *flags, hipStream_t stream) {
    if (a.n_cells <= 0) return ICNV_OK;
    const size_t lds = (size_t)std::max(max_chr_len, 1) * sizeof(double);
    if (lds > 150 * 1024) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "NA-aware chain: a chromosome does not fit the LDS");
    NaArgs n;
    n.in = a.in; n.out = a.out; n.pre = a.pre_out; n.G = a.G; n.cells = a.cells; n.in_by_pos = a.in_by_pos; n.out_by_pos = a.out_by_pos;
    n.n_cells = a.n_cells; n.flags = flags; n.chr_start = a.chr_start; n.n_chr = a.n_chr; n.T = (a.mask & ICNV_ST_SMOOTH) ? a.T : 0;
    n.mask = a.mask; n.use_bounds = a.use_bounds; n.max_thresh = a.max_thresh; n.b1 = a.b1; n.b2 = a.b2; n.denoise = a.denoise;
    static DeviceOnce once;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(chain_na_cells_kernel), 150 * 1024, once)) return rc;
    const int grid2 = (int)std::min<int64_t>(a.n_cells, (int64_t)num_cus() * 2);
    KernelTimer kt("chain_na_cells", stream);
    hipLaunchKernelGGL(chain_na_cells_kernel, dim3(grid2), dim3(NA_NT), lds, stream, n);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

// Flags the list positions whose input column holds a NaN and recomputes them (a.in must still hold the input: not for a
// chain that runs in place -- api.hip stashes the flagged columns first, see chain_apply_masked).
int launch_chain_na_fixup(const ChainArgs &a, int32_t max_chr_len, uint8_t *flags_ws, const int32_t *all_flag, hipStream_t stream) {
    if (a.n_cells <= 0) return ICNV_OK;
    if (int rc = launch_nan_flags(a, flags_ws, all_flag, stream)) return rc;
    return launch_chain_na_cells(a, max_chr_len, flags_ws, stream);
}

int launch_gather_columns(const double *x, int32_t G, const int32_t *ids_dev, int32_t n, double *stash, hipStream_t stream) {
    if (n <= 0) return ICNV_OK;
    hipLaunchKernelGGL(gather_columns_kernel, dim3((unsigned)std::min(n, num_cus() * 8)), dim3(256), 0, stream, x, (int)G, ids_dev, (int)n, stash);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

}  // namespace icnv
