// Gene filters of the ingest (steps 2: require_above_min_mean_expr_cutoff / require_above_min_cells_ref,
// R/inferCNV_ops.R:2128-2213), row selection (remove_genes) and the mean / sd of a gene x cell block
// (get_spike_dists, R/inferCNV_HMM.R:15-99) for gfx950.  All HBM-bound streaming work.
#include <algorithm>

#include "icnv_internal.h"

namespace icnv {

namespace {

constexpr int GS_TILE = 256;   // genes per block (coalesced along the gene axis of the cell-major matrix)

// part_sum[sp*G + g] = sum over the sp-th slice of cells of x[g, c];  part_nnz likewise counts x > 0 (NaN is not > 0,
// which is R's `x > 0 & !is.na(x)`).  Fixed slice order => deterministic.
__global__ void gene_stats_partial_kernel(const double *__restrict__ x, int G, int64_t C, int nsplit,
                                          double *__restrict__ part_sum, int32_t *__restrict__ part_nnz) {
    const int g = blockIdx.x * GS_TILE + threadIdx.x;
    const int sp = blockIdx.y;
    if (g >= G) return;
    const int64_t per = (C + nsplit - 1) / nsplit;
    const int64_t lo = sp * per;
    int64_t hi = lo + per;
    if (hi > C) hi = C;
    double s = 0.0;
    int32_t n = 0;
    for (int64_t c = lo; c < hi; ++c) {
        const double v = __builtin_nontemporal_load(x + c * (int64_t)G + g);
        s += v;
        n += (v > 0.0) ? 1 : 0;
    }
    part_sum[(int64_t)sp * G + g] = s;
    part_nnz[(int64_t)sp * G + g] = n;
}
__global__ void gene_stats_finish_kernel(const double *__restrict__ part_sum, const int32_t *__restrict__ part_nnz, int G,
                                         int nsplit, double *__restrict__ sums, int32_t *__restrict__ nnz) {
    const int g = blockIdx.x * GS_TILE + threadIdx.x;
    if (g >= G) return;
    double s = 0.0;
    int32_t n = 0;
    for (int sp = 0; sp < nsplit; ++sp) {
        s += part_sum[(int64_t)sp * G + g];
        n += part_nnz[(int64_t)sp * G + g];
    }
    sums[g] = s;
    nnz[g] = n;
}

// scale_infercnv_expr (R/inferCNV_ops.R:3174-3185): t(scale(t(x))) -- every gene centred on its mean over the cells and divided
// by sqrt(sum(centred^2) / max(1, C - 1)) (scale.default).  Three kernels over the cell-major matrix, lanes along the genes:
// per-slice sums -> means; per-slice sums of squared deviations -> scales; the elementwise apply.
__global__ void gene_moment_partial_kernel(const double *__restrict__ x, int G, int64_t C, int nsplit, const double *__restrict__ mean,
                                           double *__restrict__ part) {
    const int g = blockIdx.x * GS_TILE + threadIdx.x;
    const int sp = blockIdx.y;
    if (g >= G) return;
    const int64_t per = (C + nsplit - 1) / nsplit;
    const int64_t lo = sp * per;
    int64_t hi = lo + per;
    if (hi > C) hi = C;
    const double m = mean ? mean[g] : 0.0;
    double s = 0.0;
    for (int64_t c = lo; c < hi; ++c) {
        const double v = x[c * (int64_t)G + g];
        if (mean) { const double d = v - m; s += d * d; }
        else s += v;
    }
    part[(int64_t)sp * G + g] = s;
}
// pass 0: out[g] = sum / C;  pass 1: out[g] = sqrt(sum / max(1, C - 1))
__global__ void gene_moment_finish_kernel(const double *__restrict__ part, int G, int nsplit, int64_t C, int pass, double *__restrict__ out) {
    const int g = blockIdx.x * GS_TILE + threadIdx.x;
    if (g >= G) return;
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += part[(int64_t)sp * G + g];
    out[g] = pass == 0 ? s / (double)C : sqrt(s / (double)(C > 1 ? C - 1 : 1));
}
__global__ void __launch_bounds__(256) scale_genes_kernel(const double *__restrict__ in, double *__restrict__ out, int G, int64_t C,
                                                          const double *__restrict__ mean, const double *__restrict__ sdv) {
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double *src = in + c * (int64_t)G;
        double *dst = out + c * (int64_t)G;
        for (int g = threadIdx.x; g < G; g += 256) dst[g] = (src[g] - mean[g]) / sdv[g];
    }
}

// out[j, c] = in[keep[j], c]  (remove_genes: rows dropped, cell-major layout => a per-cell gather)
__global__ void select_genes_kernel(const double *__restrict__ in, int G_in, int64_t C, const int32_t *__restrict__ keep,
                                    int G_out, double *__restrict__ out) {
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double *src = in + c * (int64_t)G_in;
        double *dst = out + c * (int64_t)G_out;
        for (int j = threadIdx.x; j < G_out; j += blockDim.x) dst[j] = src[keep[j]];
    }
}

// per listed cell: sum (PASS 0) or sum of squared deviations from `mean` (PASS 1) over the listed genes
template <int PASS>
__global__ void block_cell_reduce_kernel(const double *__restrict__ x, int G, const int32_t *__restrict__ gene_idx,
                                         int n_genes, const int32_t *__restrict__ cell_idx, double mean,
                                         double *__restrict__ out) {
    __shared__ double red[256 / 64];
    const double *src = x + (int64_t)cell_idx[blockIdx.x] * G;
    auto term = [&](double v) -> double {
        if (PASS == 0) return v;
        const double d = v - mean;
        return d * d;
    };
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;   // four independent chains, combined in a fixed order
    int j = threadIdx.x;
    if (!gene_idx) {
        // all genes of the cell: 16-byte requests (round 5: 8-byte requests read the reference cells at 3.5 TB/s; a column that
        // starts at an 8-byte boundary -- odd G -- is read unaligned), the same four chains over gene PAIRS
        typedef double dbl2_t __attribute__((ext_vector_type(2)));
        const int np = n_genes >> 1;
        auto pair = [&](int p) -> dbl2_t { dbl2_t v; __builtin_memcpy(&v, src + 2 * (int64_t)p, 16); return v; };
        int p = threadIdx.x;
        for (; p + 1792 < np; p += 2048) {   // eight requests in flight per thread (a block is short-lived: latency, not bandwidth)
            const dbl2_t v0 = pair(p), v1 = pair(p + 256), v2 = pair(p + 512), v3 = pair(p + 768);
            const dbl2_t v4 = pair(p + 1024), v5 = pair(p + 1280), v6 = pair(p + 1536), v7 = pair(p + 1792);
            s0 += term(v0.x); s1 += term(v1.x); s2 += term(v2.x); s3 += term(v3.x);
            s0 += term(v0.y); s1 += term(v1.y); s2 += term(v2.y); s3 += term(v3.y);
            s0 += term(v4.x); s1 += term(v5.x); s2 += term(v6.x); s3 += term(v7.x);
            s0 += term(v4.y); s1 += term(v5.y); s2 += term(v6.y); s3 += term(v7.y);
        }
        for (; p + 768 < np; p += 1024) {
            const dbl2_t v0 = pair(p), v1 = pair(p + 256), v2 = pair(p + 512), v3 = pair(p + 768);
            s0 += term(v0.x); s1 += term(v1.x); s2 += term(v2.x); s3 += term(v3.x);
            s0 += term(v0.y); s1 += term(v1.y); s2 += term(v2.y); s3 += term(v3.y);
        }
        for (; p < np; p += 256) {
            const dbl2_t v0 = pair(p);
            s0 += term(v0.x);
            s0 += term(v0.y);
        }
        if ((n_genes & 1) && threadIdx.x == 0) s1 += term(src[n_genes - 1]);
        j = n_genes;   // (nothing left for the loops below)
    }
    for (; j + 768 < n_genes; j += 1024) {
        const double v0 = src[gene_idx ? gene_idx[j] : j], v1 = src[gene_idx ? gene_idx[j + 256] : j + 256];
        const double v2 = src[gene_idx ? gene_idx[j + 512] : j + 512], v3 = src[gene_idx ? gene_idx[j + 768] : j + 768];
        s0 += term(v0); s1 += term(v1); s2 += term(v2); s3 += term(v3);
    }
    for (; j < n_genes; j += 256) s0 += term(src[gene_idx ? gene_idx[j] : j]);
    double s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// step 16, .remove_outliers_norm (R/inferCNV_ops.R:2049-2051): data[data < lower] <- lower; data[data > upper] <- upper
__global__ void clamp_bounds_kernel(const double *__restrict__ in, double *__restrict__ out, int64_t n, double lo, double hi) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v = in[i];
        if (v < lo) v = lo;
        if (v > hi) v = hi;      // (R applies the second assignment to the result of the first; a NaN fails both tests)
        out[i] = v;
    }
}

// step 22 with noise_logistic = TRUE (.apply_logistic_val_adj, R/inferCNV_heatmap.R:2791-2810), in place: den = {m, s}
__global__ void logistic_denoise_kernel(double *__restrict__ x, int64_t n, const double *__restrict__ den) {
    const double m = den[0], s = den[1];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = x[i];
        const double val = fabs(v - m);
        const double p = 1.0 / (1.0 + exp(-20.0 * (val - s)));   // .logistic(val, delta_midpt, slope = 20)
        double r = v;
        if (v > m) r = m + p * val;
        else if (v < m) r = m - p * val;
        x[i] = r;
    }
}
// out[i] = x[offsets[i]]: the resampling of the hidden spike-in's residuals (R/inferCNV_HMM.R:164) on the resident matrix
__global__ void gather_values_kernel(const double *__restrict__ x, const int64_t *__restrict__ offsets, int64_t n, double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = x[offsets[i]];
}

}  // namespace

int launch_clamp_bounds(const double *in, double *out, int64_t n, double lo, double hi, hipStream_t stream) {
    if (n <= 0) return ICNV_OK;
    KernelTimer kt("remove_outliers", stream);
    const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)num_cus() * 16);
    hipLaunchKernelGGL(clamp_bounds_kernel, dim3(grid), dim3(256), 0, stream, in, out, n, lo, hi);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_logistic_denoise(double *x, int64_t n, const double *mu_s_dev, hipStream_t stream) {
    if (n <= 0) return ICNV_OK;
    KernelTimer kt("logistic_denoise", stream);
    const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)num_cus() * 16);
    hipLaunchKernelGGL(logistic_denoise_kernel, dim3(grid), dim3(256), 0, stream, x, n, mu_s_dev);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_gather_values(const double *x, const int64_t *offsets_dev, int64_t n, double *out, hipStream_t stream) {
    if (n <= 0) return ICNV_OK;
    KernelTimer kt("gather_values", stream);
    hipLaunchKernelGGL(gather_values_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, stream, x, offsets_dev, n, out);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int gene_stats_nsplit(int32_t G, int64_t C) {
    const int tiles = (G + GS_TILE - 1) / GS_TILE;
    int64_t ns = (4096 + tiles - 1) / tiles;   // enough blocks to fill 256 CUs several times over
    if (ns > C) ns = C;
    if (ns < 1) ns = 1;
    if (ns > 1024) ns = 1024;
    return (int)ns;
}

int launch_gene_stats(const double *x, int32_t G, int64_t C, int nsplit, double *part_sum, int32_t *part_nnz, double *sums,
                      int32_t *nnz, hipStream_t stream) {
    if (C <= 0) return ICNV_OK;
    KernelTimer kt("gene_stats", stream);
    const int tiles = (G + GS_TILE - 1) / GS_TILE;
    hipLaunchKernelGGL(gene_stats_partial_kernel, dim3(tiles, nsplit), dim3(GS_TILE), 0, stream, x, G, C, nsplit, part_sum,
                       part_nnz);
    hipLaunchKernelGGL(gene_stats_finish_kernel, dim3(tiles), dim3(GS_TILE), 0, stream, part_sum, part_nnz, G, nsplit, sums,
                       nnz);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

// mean_sd: [G means | G scales] (device); part: nsplit * G doubles of workspace
int launch_scale_genes(const double *in, double *out, int32_t G, int64_t C, int nsplit, double *part, double *mean_sd, hipStream_t stream) {
    if (C <= 0) return ICNV_OK;
    KernelTimer kt("scale_genes", stream);
    const int tiles = (G + GS_TILE - 1) / GS_TILE;
    hipLaunchKernelGGL(gene_moment_partial_kernel, dim3(tiles, nsplit), dim3(GS_TILE), 0, stream, in, G, C, nsplit, (const double *)nullptr, part);
    hipLaunchKernelGGL(gene_moment_finish_kernel, dim3(tiles), dim3(GS_TILE), 0, stream, part, G, nsplit, C, 0, mean_sd);
    hipLaunchKernelGGL(gene_moment_partial_kernel, dim3(tiles, nsplit), dim3(GS_TILE), 0, stream, in, G, C, nsplit, (const double *)mean_sd, part);
    hipLaunchKernelGGL(gene_moment_finish_kernel, dim3(tiles), dim3(GS_TILE), 0, stream, part, G, nsplit, C, 1, mean_sd + G);
    hipLaunchKernelGGL(scale_genes_kernel, dim3((unsigned)(C < 8192 ? C : 8192)), dim3(256), 0, stream, in, out, G, C, mean_sd, mean_sd + G);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_select_genes(const double *in, int32_t G_in, int64_t C, const int32_t *keep_dev, int32_t G_out, double *out,
                        hipStream_t stream) {
    if (C <= 0 || G_out <= 0) return ICNV_OK;
    KernelTimer kt("select_genes", stream);
    hipLaunchKernelGGL(select_genes_kernel, dim3((unsigned)(C < 8192 ? C : 8192)), dim3(256), 0, stream, in, G_in, C,
                       keep_dev, G_out, out);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_block_cell_reduce(int pass, const double *x, int32_t G, const int32_t *gene_idx_dev, int32_t n_genes,
                             const int32_t *cell_idx_dev, int32_t n_cells, double mean, double *out, hipStream_t stream) {
    if (n_cells <= 0) return ICNV_OK;
    KernelTimer kt("cells_moments", stream);
    if (pass == 0)
        hipLaunchKernelGGL(block_cell_reduce_kernel<0>, dim3(n_cells), dim3(256), 0, stream, x, G, gene_idx_dev, n_genes,
                           cell_idx_dev, mean, out);
    else
        hipLaunchKernelGGL(block_cell_reduce_kernel<1>, dim3(n_cells), dim3(256), 0, stream, x, G, gene_idx_dev, n_genes,
                           cell_idx_dev, mean, out);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

}  // namespace icnv
