// Fused smoothing-chain kernel for gfx950 (steps 8,9,10,11,12,14,22 of
// infercnv::run(), R/inferCNV_ops.R:771-1589) and its small helper kernels.
//
// One persistent workgroup per CU streams cells: a cell's G-vector (cell-major
// == R's column-major, contiguous) is read once with coalesced 16-B loads,
// lives in LDS + registers for the whole chain, and is written once.
//
//   phase 1 (S layout, thread t owns gene pairs 2(t + NT*k)): step 8 (bounds
//           from L2-resident lo/hi vectors), step 9 clamp -> LDS
//   phase 2 (C layout, thread t owns the contiguous chunk [t*L, (t+1)*L)):
//           step 10 pyramid smoothing as an O(1)/gene sliding update
//           (A += R - L; box sums R, L slide) with a direct 2T+1-tap init at
//           the chunk start / chromosome starts, edge-renormalised
//           denominators in closed form; step 11 exact per-cell median by a
//           value-binned histogram select in LDS (2048 bins over the value
//           range of the last cell the workgroup measured -- any range gives
//           an exact selection --, refined until <= 1024 candidates, then ranked)
//   phase 3 (S layout again, via LDS): step 12, step 14 (2^x), step 22 denoise,
//           coalesced 16-B stores -- or, in the statistics modes used by the
//           reference rounds, per-gene sums / per-cell (sum, sd).
//
// L is odd so that chunk-strided 8-byte LDS accesses are bank-conflict free
// (lane stride 2L dwords, gcd(2L, 64) = 2).
#include <cstdlib>

#include "icnv_internal.h"
#include "icnv_exp2_coef.h"
#include <algorithm>
#include <cmath>
#include <map>

namespace icnv {

// one translation unit per (threads, chunk length) so the variants compile in parallel
int launch_chain_m7(const ChainArgs &a, int mode, hipStream_t stream);    // 768 threads,  <=  5376 positions
int launch_chain_m15(const ChainArgs &a, int mode, hipStream_t stream);   // 768 threads,  <= 11520
int launch_chain_m15s(const ChainArgs &a, int mode, hipStream_t stream);  // ... and <= 10752 genes: seven gene-pair slots
int launch_chain_m15t(const ChainArgs &a, int mode, hipStream_t stream);  // ... and half window 50 at compile time (-1000: not in this form)
int launch_chain_w11(const ChainArgs &a, int mode, hipStream_t stream);   // 1024 threads, <= 11264 positions, <= 10240 (even) genes
int launch_chain_w11t(const ChainArgs &a, int mode, hipStream_t stream);  // ... half window 50 at compile time (-1000: not in this form)
int launch_chain_m17(const ChainArgs &a, int mode, hipStream_t stream);   // 768 threads,  <= 13056 positions (even G: <= 12288 genes, eight slots)
int launch_chain_m19(const ChainArgs &a, int mode, hipStream_t stream);   // 768 threads,  <= 14592 (<= 13824 genes, nine slots)
int launch_chain_m21(const ChainArgs &a, int mode, hipStream_t stream);   // 768 threads,  <= 16128 (<= 15360 genes, ten slots)
int launch_chain_m23(const ChainArgs &a, int mode, hipStream_t stream);   // 768 threads,  <= 17664 (even G <= 16896: eleven slots)
int launch_chain_l35(const ChainArgs &a, int mode, hipStream_t stream);   // 512 threads,  <= 17920


namespace {

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// sums[g] = sum_b partial[b*G + g] in a fixed order; also stores the cell count.  64 genes per workgroup, the rows dealt
// to four wavefronts (row b to wavefront b mod 4, four interleaved partial sums each so that the loads pipeline), the four
// wavefront sums combined in LDS in wavefront order: 16 accumulators per gene, always added in the same order.
__global__ void __launch_bounds__(256) reduce_partials_kernel(const double *partial, int nblk, int G, double *out, double count,
                                                              double *count_out) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = blockIdx.x * 64 + lane;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (g < G) {
        int b = w;
        for (; b + 12 < nblk; b += 16) {
            s0 += partial[(int64_t)(b + 0) * G + g];
            s1 += partial[(int64_t)(b + 4) * G + g];
            s2 += partial[(int64_t)(b + 8) * G + g];
            s3 += partial[(int64_t)(b + 12) * G + g];
        }
        for (; b < nblk; b += 4) s0 += partial[(int64_t)b * G + g];
    }
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && g < G) out[g] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (blockIdx.x == 0 && threadIdx.x == 0 && count_out) *count_out = count;
}

// Raw per-gene sums of the reference groups -- the first reference round when no stage precedes step 8 (run()'s case:
// R/inferCNV_ops.R:1708-1735 takes rowMeans of the step-7 matrix): a plain streaming reduction, every group in one
// launch.  Workgroup (tile, split, group): 256 threads x the split's share of the group's cells, four interleaved sums per
// gene; partial[(group * S + split) * G + g].  Fixed assignment and order -> deterministic.
template <int VEC>   // VEC = 2: G even, every thread adds 16-byte pieces (two adjacent genes); VEC = 1: any G
__global__ void __launch_bounds__(256) group_gene_sums_kernel(const double *x, int G, const int32_t *cells, const int32_t *off,
                                                              int S, double *partial) {
    typedef double dv2 __attribute__((ext_vector_type(2)));
    const int q = blockIdx.z, sp = blockIdx.y;
    const int b = off[q], e = off[q + 1];
    const int per = (e - b + S - 1) / S;
    const int lo = b + sp * per, hi = min(e, lo + per);
    const int g = (blockIdx.x * 256 + threadIdx.x) * VEC;
    if (g >= G) return;
    double *dst = partial + ((int64_t)q * S + sp) * G + g;
    int i = lo;
    if constexpr (VEC == 2) {
        dv2 a0 = {0.0, 0.0}, a1 = a0, a2 = a0, a3 = a0;
        for (; i + 4 <= hi; i += 4) {
            a0 += __builtin_nontemporal_load(reinterpret_cast<const dv2 *>(x + (int64_t)cells[i] * G + g));
            a1 += __builtin_nontemporal_load(reinterpret_cast<const dv2 *>(x + (int64_t)cells[i + 1] * G + g));
            a2 += __builtin_nontemporal_load(reinterpret_cast<const dv2 *>(x + (int64_t)cells[i + 2] * G + g));
            a3 += __builtin_nontemporal_load(reinterpret_cast<const dv2 *>(x + (int64_t)cells[i + 3] * G + g));
        }
        for (; i < hi; ++i) a0 += __builtin_nontemporal_load(reinterpret_cast<const dv2 *>(x + (int64_t)cells[i] * G + g));
        *reinterpret_cast<dv2 *>(dst) = (a0 + a1) + (a2 + a3);
    } else {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (; i + 4 <= hi; i += 4) {
            a0 += __builtin_nontemporal_load(x + (int64_t)cells[i] * G + g);
            a1 += __builtin_nontemporal_load(x + (int64_t)cells[i + 1] * G + g);
            a2 += __builtin_nontemporal_load(x + (int64_t)cells[i + 2] * G + g);
            a3 += __builtin_nontemporal_load(x + (int64_t)cells[i + 3] * G + g);
        }
        for (; i < hi; ++i) a0 += __builtin_nontemporal_load(x + (int64_t)cells[i] * G + g);
        *dst = (a0 + a1) + (a2 + a3);
    }
}
// sums[q*G + g] = sum over the S splits (in split order), counts[q] = cells of the group
__global__ void __launch_bounds__(256) reduce_group_partials_kernel(const double *partial, int S, int G, int n_grp,
                                                                    const int32_t *off, double *sums) {
    const int g = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y;
    if (g < G) {
        const double *p = partial + (int64_t)q * S * G + g;
        double s = 0.0;
        for (int k = 0; k < S; ++k) s += p[(int64_t)k * G];
        sums[(int64_t)q * G + g] = s;
    }
    if (g == 0) sums[(int64_t)n_grp * G + q] = (double)(off[q + 1] - off[q]);
}

// .get_normal_gene_mean_bounds + the min/max / mean-of-means of .subtract_expr
// (R/inferCNV_ops.R:1708-1735, 1750-1776).  sc = [G*n_grp sums | n_grp counts].
__global__ void bounds_from_sums_kernel(const double *sc, int G, int n_grp, int use_bounds, int inv_log, double *bounds, int32_t *nan_flag) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double lo = 0.0, hi = 0.0, tot = 0.0;
    bool any_nan = false;
    for (int r = 0; r < n_grp; ++r) {
        double m = sc[(int64_t)r * G + g] / sc[(int64_t)n_grp * G + r];
        if (inv_log) m = log2(m + 1.0);   // the sums are over 2^x - 1 (R/inferCNV_ops.R:1716)
        if (r == 0) { lo = m; hi = m; }
        else { lo = fmin(lo, m); hi = fmax(hi, m); }
        any_nan = any_nan || (m != m);
        tot += m;
    }
    // R's min() / max() return NA when a group mean is NA (a reference cell with an NA at this gene): fmin / fmax would drop it.
    // With NaN bounds step 8 / 12 turn the whole gene into 0, as which(x > NA) selects nothing (R/inferCNV_ops.R:1757-1768)
    if (any_nan) { lo = tot; hi = tot; }   // (tot is NaN)
    if (use_bounds) {
        bounds[g] = lo;
        bounds[G + g] = hi;
    } else {
        const double mm = tot / (double)n_grp;
        bounds[g] = mm;
        bounds[G + g] = mm;
        // x - NA = NA for this gene in EVERY cell (R/inferCNV_ops.R:1770-1776); the fused pass computes x - clamp(x, NaN, NaN) = 0:
        // an NA-aware chain sends every cell through the NA pass instead (chain_na.hip)
        if (nan_flag && mm != mm) *nan_flag = 1;
    }
}

// out4 = {sum of cell sums, sum of cell sds, n_cells, n_cells*G}
__global__ void reduce_cell_stats_kernel(const double *cs, int n_cells, int G, double *out4) {
    __shared__ double red[2][4];
    double s = 0.0, d = 0.0;
    // fixed assignment + fixed combine order -> deterministic
    for (int i = threadIdx.x; i < n_cells; i += blockDim.x) { s += cs[2 * (int64_t)i]; d += cs[2 * (int64_t)i + 1]; }
    s = wave_sum(s);
    d = wave_sum(d);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = d; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, td = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ts += red[0][w]; td += red[1][w]; }
        out4[0] = ts;
        out4[1] = td;
        out4[2] = (double)n_cells;
        out4[3] = (double)n_cells * (double)G;
    }
}

// Round C of the reference rounds when the reference cells continue from their cache (steps 8-11 done): steps 12 and 14
// are elementwise, so the per-cell (sum, sd) of R/inferCNV_ops.R:2302-2318 come from one streaming pass over the cache
// instead of a launch of the chain geometry (LDS-resident cell, one 1 024-thread workgroup per CU, the pipeline's barriers).
// Round 6: ONE pass per cell with shifted moments.  Rounds 4-5 kept a cell's 10 000 values in registers between the two
// passes of R's sd() -- 512 threads x 125 registers, two cells in flight per CU, the bound vectors fetched in five dependent
// groups: 0.121 ms for 400 MB (3.3 TB/s).  Here a 256-thread workgroup streams its cell once, four slots in flight per
// thread and eight cells per CU, accumulating S1 = sum(y - c) and S2 = sum (y - c)^2 around the cell's own first value c:
//   sum = S1 + G c,   (G - 1) var = S2 - S1^2 / G.
// c lies within a few sd of the cell's mean, so S1^2 / G is at most a few times S2 (no cancellation to speak of): the sd agrees
// with the two-pass value to ~1e-15 relative (tests/test_gpu_parity.py::test_denoise_round_streaming_kernel_equals_chain_geometry
// holds 1e-12 against the chain geometry's two-pass kernel and against the oracle).  Arithmetic per value as in the chain kernel
// (x - clamp(x, lo, hi); 2^x by the same degree-11 polynomial inside its range); sums in a fixed order (thread-strided partials,
// wavefront butterflies, the wavefront sums in wavefront order).
__device__ inline double exp2_lean_cs(double x) {   // chain_kernel.inc: exp2_lean
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
constexpr int CS_NT = 256;
__global__ void __launch_bounds__(CS_NT) cache_cell_stats_kernel(const double *__restrict__ cache, int G, int n_cells, uint32_t mask,
                                                                  const double *__restrict__ b2, double *__restrict__ cell_stats) {
    __shared__ double red[2][CS_NT / 64];
    typedef double dv2 __attribute__((ext_vector_type(2)));
    const int t = threadIdx.x;
    const bool sub = (mask & ICNV_ST_SUBTRACT_REF_2) != 0, inv = (mask & ICNV_ST_INVERT_LOG2) != 0;
    const int np = G >> 1;
    const dv2 *lo2 = reinterpret_cast<const dv2 *>(b2), *hi2 = reinterpret_cast<const dv2 *>(b2 + G);
    auto value = [&](double x, double lo, double hi) -> double {
        if (sub) x = x - fmin(fmax(x, lo), hi);   // .subtract_expr, R/inferCNV_ops.R:1764-1768 (lo <= hi)
        if (inv) {   // R/inferCNV_ops.R:2818; the library function on a rare wave-uniform branch (|x| >= 1022, NaN, Inf)
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(__builtin_fabs(x) < 1022.0)) != 0, 0)) {
                asm volatile("; exp2 outside the lean range" ::: "memory");
                x = exp2(x);
            } else {
                x = exp2_lean_cs(x);
            }
        }
        return x;
    };
    for (int c = blockIdx.x; c < n_cells; c += gridDim.x) {
        const double *colp = cache + (int64_t)c * G;
        const dv2 *col = reinterpret_cast<const dv2 *>(colp);
        const double shift = value(colp[0], sub ? b2[0] : 0.0, sub ? b2[G] : 0.0);   // the cell's first value (every thread computes it)
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
        for (int q = t; q < np; q += CS_NT) {
            const dv2 x = __builtin_nontemporal_load(col + q);
            dv2 lo = {0.0, 0.0}, hi = {0.0, 0.0};
            if (sub) { lo = lo2[q]; hi = hi2[q]; }
            const double d0 = value(x.x, lo.x, hi.x) - shift, d1 = value(x.y, lo.y, hi.y) - shift;
            s1 += d0;
            s1 += d1;
            s2 = __builtin_fma(d0, d0, s2);
            s2 = __builtin_fma(d1, d1, s2);
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if ((t & 63) == 0) { red[0][t >> 6] = s1; red[1][t >> 6] = s2; }
        __syncthreads();
        if (t == 0) {
            double a1 = red[0][0], a2 = red[1][0];
#pragma unroll
            for (int w = 1; w < CS_NT / 64; ++w) { a1 += red[0][w]; a2 += red[1][w]; }
            const double n = (double)G;
            cell_stats[2 * (int64_t)c] = __builtin_fma(n, shift, a1);
            cell_stats[2 * (int64_t)c + 1] = sqrt(fmax(a2 - a1 * a1 / n, 0.0) / (n - 1.0));   // sample sd (R's sd()), from the shifted moments
        }
        __syncthreads();
    }
}

// clear_noise_via_ref_mean_sd / clear_noise parameters (R/inferCNV_ops.R:2240-2255, 2311-2318)
__global__ void denoise_from_stats_kernel(const double *st, double sd_amplifier, double noise_filter, double *mu_s) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        mu_s[0] = st[0] / st[3];
        mu_s[1] = (noise_filter != noise_filter) ? (st[1] / st[2]) * sd_amplifier : noise_filter;
    }
}

// per-cell min / max over genes -> out[2*c], out[2*c+1]  (get_average_bounds, R/inferCNV_ops.R:2733-2742)
__global__ void minmax_cells_kernel(const double *x, int G, int64_t C, double *out) {
    __shared__ double red[16];
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double *col = x + c * (int64_t)G;
        double lo = __builtin_inf(), hi = -__builtin_inf();
        for (int g = threadIdx.x; g < G; g += blockDim.x) { lo = fmin(lo, col[g]); hi = fmax(hi, col[g]); }
        lo = wave_min(lo);
        hi = wave_max(hi);
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = lo; red[4 + (threadIdx.x >> 6)] = hi; }
        __syncthreads();
        lo = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
        hi = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
        __syncthreads();
        if (threadIdx.x == 0) { out[2 * c] = lo; out[2 * c + 1] = hi; }
    }
}

// colSums(expr.data) (R/inferCNV_ops.R:3089): one workgroup per cell, fixed-order reduction
__global__ void col_sums_kernel(const double *x, int G, int64_t C, double *out) {
    __shared__ double red[4];
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double *col = x + c * (int64_t)G;
        double s = 0.0;
        for (int g = threadIdx.x; g < G; g += blockDim.x) s += col[g];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) out[c] = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
    }
}

// .normalize_data_matrix_by_seq_depth (R/inferCNV_ops.R:3082-3111): x / colSum * factor, then
// log2xplus1 (:2756-2769): log2(x + 1).  Operation order as in R (divide, multiply, add, log2).
__global__ void normalize_log2_kernel(const double *in, double *out, int G, int64_t C, const double *col_sums,
                                      double factor, int do_norm, int do_log) {
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const double *src = in + c * (int64_t)G;
        double *dst = out + c * (int64_t)G;
        const double cs = do_norm ? col_sums[c] : 1.0;
        for (int g = threadIdx.x; g < G; g += blockDim.x) {
            double v = src[g];
            if (do_norm) v = v / cs * factor;
            if (do_log) v = log2(v + 1.0);
            dst[g] = v;
        }
    }
}

}  // namespace

int launch_col_sums(const double *x, int32_t G, int64_t C, double *out, hipStream_t stream) {
    if (C <= 0) return ICNV_OK;
    KernelTimer kt("col_sums", stream);
    hipLaunchKernelGGL(col_sums_kernel, dim3((unsigned)(C < 8192 ? C : 8192)), dim3(256), 0, stream, x, G, C, out);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_normalize_log2(const double *in, double *out, int32_t G, int64_t C, const double *col_sums, double factor,
                          int do_norm, int do_log, hipStream_t stream) {
    if (C <= 0) return ICNV_OK;
    KernelTimer kt("normalize_log2", stream);
    hipLaunchKernelGGL(normalize_log2_kernel, dim3((unsigned)(C < 8192 ? C : 8192)), dim3(256), 0, stream, in, out, G, C,
                       col_sums, factor, do_norm, do_log);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int chain_max_genes() { return 512 * 35; }

// Geometry of the LDS-resident cell vector: threads x chunk length, chosen by the padded position count.
// (768 threads = 3 wavefronts per SIMD = 168 VGPRs per lane; 1024 = 4 per SIMD, 128 VGPRs: the fastest where it fits.)
struct ChainGeom { int nt, lmax, pad; };
static bool chain_geom(int64_t G, int n_chr, int T, ChainGeom &g) {
    g.pad = T >= 1 ? ((T + 3) & ~1) : 0;  // even(T+2)
    // padded positions: genes + PAD zeros before every chromosome and after the last
    const int64_t npos = G + (int64_t)(n_chr + 1) * g.pad;
    if (npos <= 768 * 7) { g.nt = 768; g.lmax = 7; return true; }
    // 1024 threads x 11 positions, five gene-pair slots per thread: 16 wavefronts (4 per SIMD, 128 VGPRs) hide the
    // barrier-separated phases better than 12 (chain_apply 2.45 -> 2.30 ms at 10 000 genes)
    if (npos <= 1024 * 11 && G <= 1024 * 5 * 2 && G >= 4) { g.nt = 1024; g.lmax = 11; return true; }
    if (npos <= 768 * 15) { g.nt = 768; g.lmax = 15; return true; }
    // between the 10 000-gene geometries and 768 x 23: chunk lengths 17 / 19 / 21 with (L - 1) / 2 gene-pair slots -- a
    // thread's work follows its chunk length and its slots, so the smallest geometry that holds the cell is the fastest
    for (int L = 17; L <= 21; L += 2)
        if (npos <= 768 * L && G <= 768 * ((L - 1) / 2) * 2) { g.nt = 768; g.lmax = L; return true; }
    if (npos <= 768 * 23) { g.nt = 768; g.lmax = 23; return true; }
    if (npos <= 512 * 35) { g.nt = 512; g.lmax = 35; return true; }
    return false;
}

bool chain_fused_fits(int64_t G, int32_t n_chr, int32_t T) {
    ChainGeom g;
    if (!chain_geom(G, n_chr, T, g)) return false;
    return !(T >= 1 && T / g.lmax > 64);   // CS_GUARD of chain_kernel.inc
}

int chain_build_inv_table(const int32_t *chr_start, int32_t n_chr, int32_t G, int32_t T, std::vector<double> &tab,
                          std::vector<uint32_t> &codes, std::vector<double> &dict, bool &coded, bool view_1024x11) {
    ChainGeom g;
    coded = true;
    tab.clear();
    codes.clear();
    dict.clear();
    if (T < 1) return ICNV_OK;
    if (view_1024x11) {   // a strided view (launch_chain_strided) always runs the 1024 x 11 geometry, however few genes it holds
        g.nt = 1024; g.lmax = 11; g.pad = (T + 3) & ~1;
        if (!chain_view_fits(G, n_chr, T)) ICNV_FAIL(ICNV_ERR_ARG, "normalisation table of a view that does not fit the 1024 x 11 geometry");
    } else if (!chain_geom(G, n_chr, T, g))
        ICNV_FAIL(ICNV_ERR_UNSUPPORTED,
                  "fused smoothing chain: genes + (n_chr+1)*(window/2+2) padding exceed the LDS-resident limit of 17920 positions");
    tab.assign((size_t)g.nt * (g.lmax + 1), 0.0);
    const int64_t full = (int64_t)(T + 1) * (T + 1);
    for (int k = 0; k < n_chr; ++k) {
        const int n = chr_start[k + 1] - chr_start[k];
        const int64_t base = (int64_t)chr_start[k] + (int64_t)(k + 1) * g.pad;
        for (int i = 0; i < n; ++i) {
            double inv;
            if (n <= 1) {
                inv = 1.0 / (double)(T + 1);  // single-gene chromosome: untouched (R/inferCNV_ops.R:2417); A = (T+1) x
            } else {
                const int64_t rl = std::max(T - i, 0), rr = std::max(T - (n - 1 - i), 0);
                inv = 1.0 / (double)(full - rl * (rl + 1) / 2 - rr * (rr + 1) / 2);
            }
            const int64_t p = base + i;
            const int64_t t = p / g.lmax, q = p % g.lmax;
            tab[(size_t)(((q >> 1) * g.nt + t) * 2 + (q & 1))] = inv;
        }
    }
    // byte-coded form: dictionary of the distinct values (entry 0 = padding), one code per position.  Plans
    // with more than 254 distinct values (many very short contigs of different lengths) are not coded.
    dict.assign(256, 0.0);
    codes.assign((size_t)g.nt * ((g.lmax + 3) / 4), 0u);
    std::map<double, int64_t> freq;
    for (double v : tab)
        if (v != 0.0) ++freq[v];
    std::vector<std::pair<int64_t, double>> order;
    for (auto &kv : freq) order.emplace_back(-kv.second, kv.first);
    std::sort(order.begin(), order.end());
    std::map<double, uint32_t> code_of;
    for (size_t i = 0; i < order.size() && i < 254; ++i) {
        code_of[order[i].second] = (uint32_t)i + 1;
        dict[i + 1] = order[i].second;
    }
    for (int64_t p = 0; p < (int64_t)g.nt * g.lmax; ++p) {
        const int64_t t = p / g.lmax, q = p % g.lmax;
        const double inv = tab[(size_t)(((q >> 1) * g.nt + t) * 2 + (q & 1))];
        if (inv == 0.0) continue;
        auto it = code_of.find(inv);
        if (it == code_of.end()) { coded = false; continue; }
        const uint32_t code = it->second;
        codes[(size_t)((q >> 2) * g.nt + t)] |= code << (8 * (q & 3));
    }
    return ICNV_OK;
}

int launch_chain(const ChainArgs &a0, int mode, hipStream_t stream) {
    ChainArgs a = a0;
    if (a.n_chr > 510) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "more than 510 chromosomes/contigs");
    if (a.mask & ICNV_ST_MAX_THRESH) {
        if (a.max_thresh != a.max_thresh) ICNV_FAIL(ICNV_ERR_ARG, "apply_max_threshold_bounds: the threshold is NaN");
        if (std::isinf(a.max_thresh)) a.mask &= ~(uint32_t)ICNV_ST_MAX_THRESH;   // an infinite threshold clamps nothing
    }
    const bool smooth = (a.mask & ICNV_ST_SMOOTH) && a.T >= 1;
    if (!smooth) a.T = 0;
    ChainGeom g;
    if (!chain_geom(a.G, a.n_chr, a.T, g))
        ICNV_FAIL(ICNV_ERR_UNSUPPORTED,
                  "fused smoothing chain: genes + (n_chr+1)*(window/2+2) padding exceed the LDS-resident limit of 17920 positions");
    if (smooth && a.T / g.lmax > 64)   // CS_GUARD of chain_kernel.inc
        ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "smoothing window too long for the fused chain kernel (half window / chunk length > 64)");
    if (smooth && !(a.inv_pos && a.inv_codes && a.inv_dict)) ICNV_FAIL(ICNV_ERR_ARG, "smoothing launch without its normalisation table");
    a.pad = g.pad;
    if (g.lmax == 7) return launch_chain_m7(a, mode, stream);
    if (g.nt == 1024) {
        if (smooth && a.T == 50) {
            const int rc = launch_chain_w11t(a, mode, stream);
            if (rc != -1000) return rc;
        }
        return launch_chain_w11(a, mode, stream);
    }
    if (g.lmax == 15) {
        if (a.G >= 4 && a.G <= 768 * 7 * 2) {
            if (smooth && a.T == 50) {
                const int rc = launch_chain_m15t(a, mode, stream);
                if (rc != -1000) return rc;
            }
            return launch_chain_m15s(a, mode, stream);
        }
        return launch_chain_m15(a, mode, stream);
    }
    if (g.lmax == 17) return launch_chain_m17(a, mode, stream);
    if (g.lmax == 19) return launch_chain_m19(a, mode, stream);
    if (g.lmax == 21) return launch_chain_m21(a, mode, stream);
    if (g.lmax == 23) return launch_chain_m23(a, mode, stream);
    return launch_chain_l35(a, mode, stream);
}

int launch_reduce_partials(const double *partial, int nblk, int32_t G, double *out, double count,
                           double *count_out, hipStream_t stream) {
    KernelTimer kt("reduce_partials", stream);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((G + 63) / 64), dim3(256), 0, stream, partial, nblk, G, out,
                       count, count_out);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_group_gene_sums(const double *x, int32_t G, const int32_t *cells_dev, const int32_t *off_dev, int32_t n_grp,
                           double *partial, int32_t partial_rows, double *sums_counts, hipStream_t stream) {
    if (n_grp <= 0) return ICNV_OK;
    if (n_grp > partial_rows) ICNV_FAIL(ICNV_ERR_ARG, "more reference groups than rows of the partial-sum buffer");
    const int S = std::max(1, std::min(32, partial_rows / n_grp));
    {
        KernelTimer kt("chain_gene_sums", stream);
        if ((G & 1) == 0)
            hipLaunchKernelGGL(group_gene_sums_kernel<2>, dim3((G / 2 + 255) / 256, S, n_grp), dim3(256), 0, stream, x, G, cells_dev,
                               off_dev, S, partial);
        else
            hipLaunchKernelGGL(group_gene_sums_kernel<1>, dim3((G + 255) / 256, S, n_grp), dim3(256), 0, stream, x, G, cells_dev,
                               off_dev, S, partial);
        ICNV_HIP(hipGetLastError());
    }
    KernelTimer kt("reduce_partials", stream);
    hipLaunchKernelGGL(reduce_group_partials_kernel, dim3((G + 255) / 256, n_grp), dim3(256), 0, stream, partial, S, G, n_grp,
                       off_dev, sums_counts);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_bounds_from_sums(const double *sums_counts, int32_t G, int32_t n_grp, int32_t use_bounds, int32_t inv_log,
                            double *bounds, int32_t *nan_flag, hipStream_t stream) {
    KernelTimer kt("bounds_from_sums", stream);
    hipLaunchKernelGGL(bounds_from_sums_kernel, dim3((G + 255) / 256), dim3(256), 0, stream, sums_counts, G, n_grp,
                       use_bounds, inv_log, bounds, nan_flag);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_reduce_cell_stats(const double *cell_stats, int32_t n_cells, int32_t G, double *out4,
                             hipStream_t stream) {
    KernelTimer kt("reduce_cell_stats", stream);
    hipLaunchKernelGGL(reduce_cell_stats_kernel, dim3(1), dim3(256), 0, stream, cell_stats, n_cells, G, out4);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

bool cache_cell_stats_covers(int32_t G) { return (G & 1) == 0 && G >= 2; }
int launch_cache_cell_stats(const double *cache, int32_t G, int32_t n_cells, uint32_t mask, const double *b2, double *cell_stats,
                            hipStream_t stream) {
    if (n_cells <= 0) return ICNV_OK;
    KernelTimer kt("chain_cell_stats", stream);
    hipLaunchKernelGGL(cache_cell_stats_kernel, dim3((unsigned)std::min(n_cells, num_cus() * 32)), dim3(CS_NT), 0, stream, cache, (int)G,
                       (int)n_cells, mask, b2, cell_stats);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_denoise_from_stats(const double *stats4, double sd_amplifier, double noise_filter, double *mu_s,
                              hipStream_t stream) {
    hipLaunchKernelGGL(denoise_from_stats_kernel, dim3(1), dim3(64), 0, stream, stats4, sd_amplifier, noise_filter,
                       mu_s);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_minmax_cells(const double *x, int32_t G, int64_t C, double *out2_dev, hipStream_t stream) {
    int grid = (int)(C < 4096 ? C : 4096);
    if (grid < 1) return ICNV_OK;
    hipLaunchKernelGGL(minmax_cells_kernel, dim3(grid), dim3(256), 0, stream, x, G, C, out2_dev);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

}  // namespace icnv
