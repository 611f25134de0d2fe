// Ingest: steps 2, 3, 4 of infercnv::run() from the raw COUNT matrix.
//
//   step 2  require_above_min_mean_expr_cutoff (rowMeans(counts) < cutoff -> gene removed), require_above_min_cells_ref
//           (sum(x > 0 & !is.na(x)) >= min_cells_per_gene -> gene kept)          R/inferCNV_ops.R:2128-2163, 2182-2213
//   step 3  normalize_counts_by_seq_depth: x / colSums(x) * median(colSums(x))     R/inferCNV_ops.R:3064-3111
//   step 4  log2xplus1: log2(x + 1)                                                R/inferCNV_ops.R:2756-2769
//
// The reference does this on a double (or dgCMatrix) matrix in R.  Here the counts cross PCIe ONCE, as what they are --
// int32, dense (4 bytes per entry instead of 8) or CSC (12 bytes per NONZERO: scRNA-seq counts are > 90 % zeros) -- and
// the f64 matrix of the kept genes is produced on the device, where step 8 wants it.  Three streaming passes over the
// integer data (gene statistics -> [filter decision] -> column sums of the kept genes -> [median] -> normalise + log2),
// split so that a cell-sharded caller can all-reduce the gene statistics and all-gather the column sums between them.
// Integer sums are exact (int64), so every split / order gives the same statistics; the per-element arithmetic is the
// reference's, in its order: divide, multiply, add, log2 -- bit for bit what icnv_gene_stats + icnv_select_genes +
// icnv_normalize_log2 give on the f64 copy of the counts (tests/test_gpu_entrypoints.py).
#include <algorithm>
#include <cmath>
#include <vector>

#include "icnv_internal.h"

namespace icnv {

namespace {

constexpr int IG_TILE = 256;

// dense: part_sum[sp*G + g] = sum over the sp-th slice of cells of x[g, c] (int64, exact); part_nnz counts x > 0
__global__ void ingest_gene_stats_dense_kernel(const int32_t *__restrict__ x, int G, int64_t C, int nsplit,
                                               long long *__restrict__ part_sum, int32_t *__restrict__ part_nnz, int32_t *__restrict__ neg_flag) {
    const int g = blockIdx.x * IG_TILE + threadIdx.x;
    const int sp = blockIdx.y;
    if (g >= G) return;
    const int64_t per = (C + nsplit - 1) / nsplit;
    const int64_t lo = sp * per;
    int64_t hi = lo + per;
    if (hi > C) hi = C;
    long long s = 0;
    int32_t n = 0;
    for (int64_t c = lo; c < hi; ++c) {
        const int32_t v = __builtin_nontemporal_load(x + c * (int64_t)G + g);
        s += v;
        n += (v > 0) ? 1 : 0;
        if (v < 0) atomicOr(neg_flag, 1);   // not a count (R's NA_integer_ is INT_MIN): the caller is told, nothing is computed from it
    }
    part_sum[(int64_t)sp * G + g] = s;
    part_nnz[(int64_t)sp * G + g] = n;
}
// stats[g] = sum, stats[G + g] = number of cells with x > 0, both as doubles (one buffer, one all-reduce)
__global__ void ingest_gene_stats_finish_kernel(const long long *__restrict__ part_sum, const int32_t *__restrict__ part_nnz, int G,
                                                int nsplit, double *__restrict__ stats) {
    const int g = blockIdx.x * IG_TILE + threadIdx.x;
    if (g >= G) return;
    long long s = 0;
    long long n = 0;
    for (int sp = 0; sp < nsplit; ++sp) {
        s += part_sum[(int64_t)sp * G + g];
        n += part_nnz ? part_nnz[(int64_t)sp * G + g] : 0;
    }
    stats[g] = (double)s;
    stats[G + g] = (double)n;
}
// CSC: one thread per stored entry; integer atomics are exact and commute -> deterministic
__global__ void ingest_gene_stats_csc_kernel(const int32_t *__restrict__ rowidx, const int32_t *__restrict__ vals, int64_t nnz,
                                             unsigned long long *__restrict__ sums, unsigned long long *__restrict__ cnt,
                                             int32_t *__restrict__ neg_flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t v = vals[i];
        const int32_t r = rowidx[i];
        if (v < 0) atomicOr(neg_flag, 1);
        atomicAdd(&sums[r], (unsigned long long)(long long)v);
        if (v > 0) atomicAdd(&cnt[r], 1ull);
    }
}
__global__ void ingest_stats_from_i64_kernel(const unsigned long long *__restrict__ sums, const unsigned long long *__restrict__ cnt, int G,
                                             double *__restrict__ stats) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    stats[g] = (double)(long long)sums[g];
    stats[G + g] = (double)(long long)cnt[g];
}

// colSums over the kept genes: one workgroup per cell (dense) / one wavefront per column (CSC); int64, exact
__global__ void __launch_bounds__(256) ingest_col_sums_dense_kernel(const int32_t *__restrict__ x, int G, int64_t C, const uint8_t *__restrict__ keep,
                                                                     double *__restrict__ out) {
    __shared__ long long red[4];
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const int32_t *col = x + c * (int64_t)G;
        long long s = 0;
        for (int g = threadIdx.x; g < G; g += 256) s += keep[g] ? (long long)col[g] : 0ll;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) out[c] = (double)((red[0] + red[1]) + (red[2] + red[3]));
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) ingest_col_sums_csc_kernel(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
                                                                   const int32_t *__restrict__ vals, int64_t C, const uint8_t *__restrict__ keep,
                                                                   double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); c < C; c += (int64_t)gridDim.x * 4) {
        long long s = 0;
        for (int64_t i = colptr[c] + lane; i < colptr[c + 1]; i += 64) s += keep[rowidx[i]] ? (long long)vals[i] : 0ll;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) out[c] = (double)s;
    }
}

// .normalize_data_matrix_by_seq_depth + log2xplus1 on one count: operation order as in R (divide, multiply, add, log2)
__device__ inline double ingest_value(int32_t v, double cs, double factor, int do_norm, int do_log) {
    double y = (double)v;
    if (do_norm) y = y / cs * factor;
    if (do_log) y = log2(y + 1.0);
    return y;
}
__global__ void __launch_bounds__(256) ingest_apply_dense_kernel(const int32_t *__restrict__ x, int G, int64_t C, const int32_t *__restrict__ keep_idx,
                                                                  int G_out, const double *__restrict__ col_sums, double factor, int do_norm,
                                                                  int do_log, double *__restrict__ out) {
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const int32_t *src = x + c * (int64_t)G;
        double *dst = out + c * (int64_t)G_out;
        const double cs = do_norm ? col_sums[c] : 1.0;
        for (int j = threadIdx.x; j < G_out; j += 256) dst[j] = ingest_value(src[keep_idx[j]], cs, factor, do_norm, do_log);
    }
}
// CSC: the output was zero-filled (a zero count stays 0 through every step: 0 / cs * f = 0, log2(0 + 1) = 0); the stored
// entries of the kept genes are scattered to their new rows.  A cell whose column sum over the kept genes is 0 is the
// exception: R computes 0 / 0 * f = NaN for every gene of it (so do the dense path and icnv_normalize_log2) -- its column
// is filled with that very expression instead.
__global__ void __launch_bounds__(256) ingest_apply_csc_kernel(const int64_t *__restrict__ colptr, const int32_t *__restrict__ rowidx,
                                                                const int32_t *__restrict__ vals, int64_t C, const int32_t *__restrict__ new_row,
                                                                int G_out, const double *__restrict__ col_sums, double factor, int do_norm,
                                                                int do_log, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); c < C; c += (int64_t)gridDim.x * 4) {
        const double cs = do_norm ? col_sums[c] : 1.0;
        double *dst = out + c * (int64_t)G_out;
        if (do_norm && cs == 0.0) {   // wave-uniform
            const double z = ingest_value(0, cs, factor, do_norm, do_log);
            for (int j = lane; j < G_out; j += 64) dst[j] = z;
        }
        for (int64_t i = colptr[c] + lane; i < colptr[c + 1]; i += 64) {
            const int32_t j = new_row[rowidx[i]];
            if (j >= 0) dst[j] = ingest_value(vals[i], cs, factor, do_norm, do_log);
        }
    }
}

int check_counts(const icnv_counts *cnt, int64_t G, int64_t C) {
    if (!cnt || G < 1 || C < 0 || G > 0x7fffffff) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    const bool dense = cnt->dense != nullptr, csc = cnt->colptr != nullptr;
    if (dense == csc) ICNV_FAIL(ICNV_ERR_ARG, "icnv_counts: give either the dense int32 matrix or the CSC arrays");
    if (csc && cnt->nnz > 0 && (!cnt->rowidx || !cnt->vals)) ICNV_FAIL(ICNV_ERR_ARG, "icnv_counts: CSC row indices / values missing");
    if (csc && cnt->nnz < 0) ICNV_FAIL(ICNV_ERR_ARG, "icnv_counts: negative nnz");
    return ICNV_OK;
}

double host_median_of(std::vector<double> v) {   // stats::median
    if (v.empty()) return NAN;
    const size_t n = v.size(), h = n / 2;
    std::nth_element(v.begin(), v.begin() + (std::ptrdiff_t)h, v.end());
    const double hi = v[h];
    if (n & 1) return hi;
    const double lo = *std::max_element(v.begin(), v.begin() + (std::ptrdiff_t)h);
    return (lo + hi) / 2.0;
}

}  // namespace

}  // namespace icnv

using namespace icnv;

extern "C" {

int icnv_ingest_gene_stats_dev(const icnv_counts *cnt, int64_t G, int64_t C, double *stats2G_dev, void *stream) {
    int rc = check_counts(cnt, G, C);
    if (rc) return rc;
    if (!stats2G_dev) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    if (C == 0 || (cnt->colptr && cnt->nnz == 0)) {
        ICNV_HIP(hipMemsetAsync(stats2G_dev, 0, (size_t)2 * G * sizeof(double), s));
        return ICNV_OK;
    }
    DevBuf dneg;
    if ((rc = dneg.alloc(sizeof(int32_t)))) return rc;
    ICNV_HIP(hipMemsetAsync(dneg.p, 0, sizeof(int32_t), s));
    int32_t neg = 0;
    {
    KernelTimer kt("ingest_gene_stats", s);
    if (cnt->dense) {
        const int tiles = (int)((G + IG_TILE - 1) / IG_TILE);
        int64_t ns = (4096 + tiles - 1) / tiles;
        ns = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ns, C), 1024));
        DevBuf ps, pn;
        if ((rc = ps.alloc((size_t)ns * G * sizeof(long long))) || (rc = pn.alloc((size_t)ns * G * sizeof(int32_t)))) return rc;
        hipLaunchKernelGGL(ingest_gene_stats_dense_kernel, dim3(tiles, (unsigned)ns), dim3(IG_TILE), 0, s, cnt->dense, (int)G, C, (int)ns,
                           ps.as<long long>(), pn.as<int32_t>(), dneg.as<int32_t>());
        hipLaunchKernelGGL(ingest_gene_stats_finish_kernel, dim3(tiles), dim3(IG_TILE), 0, s, ps.as<long long>(), pn.as<int32_t>(), (int)G,
                           (int)ns, stats2G_dev);
        ICNV_HIP(hipGetLastError());
        ICNV_HIP(hipMemcpyAsync(&neg, dneg.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        ICNV_HIP(hipStreamSynchronize(s));   // the partial buffers go back to the pool
    } else {
        DevBuf acc;
        if ((rc = acc.alloc((size_t)2 * G * sizeof(unsigned long long)))) return rc;
        ICNV_HIP(hipMemsetAsync(acc.p, 0, (size_t)2 * G * sizeof(unsigned long long), s));
        const int grid = (int)std::min<int64_t>((cnt->nnz + 255) / 256, (int64_t)num_cus() * 16);
        hipLaunchKernelGGL(ingest_gene_stats_csc_kernel, dim3(grid), dim3(256), 0, s, cnt->rowidx, cnt->vals, cnt->nnz,
                           acc.as<unsigned long long>(), acc.as<unsigned long long>() + G, dneg.as<int32_t>());
        hipLaunchKernelGGL(ingest_stats_from_i64_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, acc.as<unsigned long long>(),
                           acc.as<unsigned long long>() + G, (int)G, stats2G_dev);
        ICNV_HIP(hipGetLastError());
        ICNV_HIP(hipMemcpyAsync(&neg, dneg.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        ICNV_HIP(hipStreamSynchronize(s));
    }
    }
    if (neg) ICNV_FAIL(ICNV_ERR_ARG, "icnv_counts: negative value in the count matrix (an NA_integer_ from R?): counts must be >= 0");
    return ICNV_OK;
}

int icnv_ingest_col_sums_dev(const icnv_counts *cnt, int64_t G, int64_t C, const uint8_t *keep_mask_dev, double *col_sums_dev, void *stream) {
    int rc = check_counts(cnt, G, C);
    if (rc) return rc;
    if (!keep_mask_dev || (C > 0 && !col_sums_dev)) ICNV_FAIL(ICNV_ERR_ARG, "null argument");
    if (C == 0) return ICNV_OK;
    hipStream_t s = (hipStream_t)stream;
    KernelTimer kt("ingest_col_sums", s);
    if (cnt->dense)
        hipLaunchKernelGGL(ingest_col_sums_dense_kernel, dim3((unsigned)std::min<int64_t>(C, 8192)), dim3(256), 0, s, cnt->dense, (int)G, C,
                           keep_mask_dev, col_sums_dev);
    else
        hipLaunchKernelGGL(ingest_col_sums_csc_kernel, dim3((unsigned)std::min<int64_t>((C + 3) / 4, 8192)), dim3(256), 0, s, cnt->colptr,
                           cnt->rowidx, cnt->vals, C, keep_mask_dev, col_sums_dev);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int icnv_ingest_apply_dev(const icnv_counts *cnt, int64_t G, int64_t C, const int32_t *keep_idx_dev, int64_t G_out,
                          const double *col_sums_dev, double factor, int32_t do_normalize, int32_t do_log2, double *expr_out, void *stream) {
    int rc = check_counts(cnt, G, C);
    if (rc) return rc;
    if (G_out < 0 || G_out > G || (G_out > 0 && !keep_idx_dev) || (G_out > 0 && C > 0 && !expr_out) || (do_normalize && C > 0 && !col_sums_dev))
        ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    if (C == 0 || G_out == 0) return ICNV_OK;
    hipStream_t s = (hipStream_t)stream;
    KernelTimer kt("ingest_apply", s);
    if (cnt->dense) {
        hipLaunchKernelGGL(ingest_apply_dense_kernel, dim3((unsigned)std::min<int64_t>(C, 8192)), dim3(256), 0, s, cnt->dense, (int)G, C,
                           keep_idx_dev, (int)G_out, col_sums_dev, factor, do_normalize, do_log2, expr_out);
        ICNV_HIP(hipGetLastError());
        return ICNV_OK;
    }
    // CSC: old row -> new row (-1: dropped), built on the device from the keep list
    DevBuf newrow;
    if ((rc = newrow.alloc((size_t)G * sizeof(int32_t)))) return rc;
    std::vector<int32_t> keep((size_t)G_out), nr((size_t)G, -1);
    ICNV_HIP(hipMemcpyAsync(keep.data(), keep_idx_dev, (size_t)G_out * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ICNV_HIP(hipStreamSynchronize(s));
    for (int64_t j = 0; j < G_out; ++j) {
        if (keep[(size_t)j] < 0 || keep[(size_t)j] >= G) ICNV_FAIL(ICNV_ERR_ARG, "gene index out of range");
        nr[(size_t)keep[(size_t)j]] = (int32_t)j;
    }
    ICNV_HIP(hipMemcpyAsync(newrow.p, nr.data(), (size_t)G * sizeof(int32_t), hipMemcpyHostToDevice, s));
    ICNV_HIP(hipMemsetAsync(expr_out, 0, (size_t)G_out * (size_t)C * sizeof(double), s));
    hipLaunchKernelGGL(ingest_apply_csc_kernel, dim3((unsigned)std::min<int64_t>((C + 3) / 4, 8192)), dim3(256), 0, s, cnt->colptr, cnt->rowidx,
                       cnt->vals, C, newrow.as<int32_t>(), (int)G_out, col_sums_dev, factor, do_normalize, do_log2, expr_out);
    ICNV_HIP(hipGetLastError());
    ICNV_HIP(hipStreamSynchronize(s));   // nr / newrow stay alive until the launch has read them
    return ICNV_OK;
}

// the filter decision of step 2 from the (all-reduced) gene statistics: host arithmetic, identical on every rank
int icnv_ingest_select(const double *stats2G_host, int64_t G, int64_t C_total, double min_mean_expr_cutoff, int32_t min_cells_per_gene,
                       int32_t *keep_idx, int64_t *G_out) {
    if (!stats2G_host || !keep_idx || !G_out || G < 1 || C_total < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    int64_t n = 0;
    for (int64_t g = 0; g < G; ++g) {
        bool keep = true;
        if (!std::isnan(min_mean_expr_cutoff)) keep = !(stats2G_host[g] / (double)C_total < min_mean_expr_cutoff);   // rowMeans < cutoff -> removed (:2157)
        if (keep && min_cells_per_gene > 0) keep = stats2G_host[G + g] >= (double)min_cells_per_gene;                 // :2184
        if (keep) keep_idx[n++] = (int32_t)g;
    }
    *G_out = n;
    if (n == 0) ICNV_FAIL(ICNV_ERR_ARG, "All genes removed! Must revisit your data..., cannot continue here.");   // stop(998), :2194-2198
    return ICNV_OK;
}

int icnv_ingest_counts_dev(const icnv_counts *cnt, int64_t G, int64_t C, double min_mean_expr_cutoff, int32_t min_cells_per_gene,
                           double normalize_factor, int32_t *keep_idx_host, int64_t *G_out, double *expr_out_dev, double *factor_used,
                           void *stream) {
    int rc = check_counts(cnt, G, C);
    if (rc) return rc;
    if (!keep_idx_host || !G_out || C < 1 || !expr_out_dev) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    DevBuf dstats, dkeep, dmask, dcs;
    if ((rc = dstats.alloc((size_t)2 * G * sizeof(double))) || (rc = dcs.alloc((size_t)C * sizeof(double)))) return rc;
    if ((rc = icnv_ingest_gene_stats_dev(cnt, G, C, dstats.as<double>(), stream))) return rc;
    std::vector<double> stats((size_t)2 * G);
    ICNV_HIP(hipMemcpyAsync(stats.data(), dstats.p, stats.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    ICNV_HIP(hipStreamSynchronize(s));
    if ((rc = icnv_ingest_select(stats.data(), G, C, min_mean_expr_cutoff, min_cells_per_gene, keep_idx_host, G_out))) return rc;
    std::vector<uint8_t> mask((size_t)G, 0);
    for (int64_t j = 0; j < *G_out; ++j) mask[(size_t)keep_idx_host[j]] = 1;
    if ((rc = dmask.alloc((size_t)G)) || (rc = dkeep.alloc((size_t)std::max<int64_t>(*G_out, 1) * sizeof(int32_t)))) return rc;
    ICNV_HIP(hipMemcpyAsync(dmask.p, mask.data(), (size_t)G, hipMemcpyHostToDevice, s));
    ICNV_HIP(hipMemcpyAsync(dkeep.p, keep_idx_host, (size_t)*G_out * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if ((rc = icnv_ingest_col_sums_dev(cnt, G, C, dmask.as<uint8_t>(), dcs.as<double>(), stream))) return rc;
    double factor = normalize_factor;
    if (std::isnan(factor)) {   // median(colSums), R/inferCNV_ops.R:3096
        std::vector<double> cs((size_t)C);
        ICNV_HIP(hipMemcpyAsync(cs.data(), dcs.p, (size_t)C * sizeof(double), hipMemcpyDeviceToHost, s));
        ICNV_HIP(hipStreamSynchronize(s));
        factor = host_median_of(std::move(cs));
    }
    if (std::isnan(factor)) ICNV_FAIL(ICNV_ERR_ARG, "normalize factor not estimated");   // :3105
    if (factor_used) *factor_used = factor;
    if ((rc = icnv_ingest_apply_dev(cnt, G, C, dkeep.as<int32_t>(), *G_out, dcs.as<double>(), factor, 1, 1, expr_out_dev, stream))) return rc;
    ICNV_HIP(hipStreamSynchronize(s));   // mask / keep list / column sums go back to the pool
    return ICNV_OK;
}

int icnv_ingest_counts(const icnv_counts *cnt, int64_t G, int64_t C, double min_mean_expr_cutoff, int32_t min_cells_per_gene,
                       double normalize_factor, int32_t *keep_idx, int64_t *G_out, double *expr_out, double *factor_used,
                       int64_t *h2d_bytes) {
    int rc = check_counts(cnt, G, C);
    if (rc) return rc;
    if (!keep_idx || !G_out || !expr_out || C < 1) ICNV_FAIL(ICNV_ERR_ARG, "bad argument");
    icnv_counts d = *cnt;
    DevBuf a, b, c, dout;
    int64_t up = 0;
    if (cnt->dense) {
        const size_t bytes = (size_t)G * (size_t)C * sizeof(int32_t);
        if ((rc = a.alloc(bytes))) return rc;
        ICNV_HIP(hipMemcpy(a.p, cnt->dense, bytes, hipMemcpyHostToDevice));
        d.dense = a.as<int32_t>();
        up = (int64_t)bytes;
    } else {
        if (cnt->colptr[0] != 0 || cnt->colptr[C] != cnt->nnz) ICNV_FAIL(ICNV_ERR_ARG, "icnv_counts: colptr does not span nnz");
        for (int64_t cc = 0; cc < C; ++cc)   // (the kernels walk [colptr[c], colptr[c+1]): every entry inside [0, nnz], non-decreasing)
            if (cnt->colptr[cc] > cnt->colptr[cc + 1] || cnt->colptr[cc] < 0) ICNV_FAIL(ICNV_ERR_ARG, "icnv_counts: colptr is not non-decreasing");
        for (int64_t i = 0; i < cnt->nnz; ++i)
            if (cnt->rowidx[i] < 0 || cnt->rowidx[i] >= G) ICNV_FAIL(ICNV_ERR_ARG, "icnv_counts: gene index out of range");
        const size_t nz = (size_t)std::max<int64_t>(cnt->nnz, 1);
        if ((rc = a.alloc((size_t)(C + 1) * sizeof(int64_t))) || (rc = b.alloc(nz * sizeof(int32_t))) || (rc = c.alloc(nz * sizeof(int32_t)))) return rc;
        ICNV_HIP(hipMemcpy(a.p, cnt->colptr, (size_t)(C + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        if (cnt->nnz) {
            ICNV_HIP(hipMemcpy(b.p, cnt->rowidx, (size_t)cnt->nnz * sizeof(int32_t), hipMemcpyHostToDevice));
            ICNV_HIP(hipMemcpy(c.p, cnt->vals, (size_t)cnt->nnz * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        d.colptr = a.as<int64_t>(); d.rowidx = b.as<int32_t>(); d.vals = c.as<int32_t>();
        up = (int64_t)((C + 1) * sizeof(int64_t)) + cnt->nnz * 8;
    }
    if (h2d_bytes) *h2d_bytes = up;
    if ((rc = dout.alloc((size_t)G * (size_t)C * sizeof(double)))) return rc;
    if ((rc = icnv_ingest_counts_dev(&d, G, C, min_mean_expr_cutoff, min_cells_per_gene, normalize_factor, keep_idx, G_out, dout.as<double>(),
                                     factor_used, nullptr)))
        return rc;
    const int64_t n_out = *G_out * C;
    ICNV_HIP(hipMemcpy(expr_out, dout.p, (size_t)n_out * sizeof(double), hipMemcpyDeviceToHost));
    publish_output(expr_out, n_out, std::move(dout));   // step 8 finds the matrix on the device (icnv_residency)
    return ICNV_OK;
}

}  // extern "C"
