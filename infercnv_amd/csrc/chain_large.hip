// Smoothing chain for gene counts beyond the fused kernel's LDS-resident limit (genes + chromosome padding
// > 17 920 positions): the same steps 8, 9, 10, 11, 12, 14, 22 of infercnv::run() (R/inferCNV_ops.R:771-1589) as
// three plain passes over HBM instead of one fused pass,
//
//   S  steps 8, 9, 10   one workgroup per (cell, chromosome): the chromosome in LDS, direct (2T+1)-tap pyramid with
//                       the reference's renormalisation at the chromosome ends (R/inferCNV_ops.R:2406-2532)
//   M  step 11          one workgroup per cell: exact median by the value-binned histogram select of chain_kernel,
//                       the cell re-read from L2 for every pass instead of living in registers (or the mean)
//   E  steps 12, 14, 22 elementwise; optionally the per-cell (sum, sd) of the denoise round
//
// ~3x the fused kernel's HBM traffic and no register/LDS residency tricks: this is the fallback that keeps large
// gene sets working (only a single chromosome of more than ~19 000 genes is refused), and -- forced with
// ICNV_CHAIN_LARGE=1 -- an independent second implementation the parity tests compare the fused kernel with.
#include "icnv_internal.h"

namespace icnv {

namespace {

constexpr int LG_NT = 1024;      // threads of the per-cell passes
constexpr int LG_BINS = 2048;
constexpr int LG_CAND = 1024;

__device__ inline double lg_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline double lg_wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline double lg_wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
// deterministic block reductions (fixed combine order); red holds >= 2 * NT / 64 doubles
template <int NT>
__device__ inline double lg_block_sum(double v, double *red) {
    v = lg_wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < NT / 64; ++w) r += red[w];
    __syncthreads();
    return r;
}
template <int NT>
__device__ inline void lg_block_minmax(double &lo, double &hi, double *red) {
    lo = lg_wave_min(lo);
    hi = lg_wave_max(hi);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = lo; red[NT / 64 + (threadIdx.x >> 6)] = hi; }
    __syncthreads();
    double a = red[0], b = red[NT / 64];
    for (int w = 1; w < NT / 64; ++w) { a = fmin(a, red[w]); b = fmax(b, red[NT / 64 + w]); }
    __syncthreads();
    lo = a;
    hi = b;
}

__device__ inline double lg_subtract_ref(double x, double lo, double hi) {   // .subtract_expr, R/inferCNV_ops.R:1742-1786
    return x - fmin(fmax(x, lo), hi);   // lo == hi == mean of the group means when use_bounds = FALSE
}

// ---------------------------------------------------------------- S: steps 8, 9, 10
// grid (n_chr, rows'): workgroup (c, y) handles chromosome c of rows y, y + gridDim.y, ...
__global__ void __launch_bounds__(256) large_smooth_kernel(const LargeChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sx[];   // [T+1 zeros][chromosome][T+1 zeros]
    const int c = blockIdx.x;
    const int g0 = a.chr_start[c], n = a.chr_start[c + 1] - g0;
    const int T = a.T, H = T + 1;
    const bool smooth = (a.mask & ICNV_ST_SMOOTH) && T >= 1 && n > 1;   // single-gene chromosomes stay (R/inferCNV_ops.R:2417)
    for (int row = blockIdx.y; row < a.n_rows; row += gridDim.y) {
        const int64_t ri = a.in_rows ? a.in_rows[row] : row, ro = a.out_rows ? a.out_rows[row] : row;
        const double *src = a.in + ri * (int64_t)a.G + g0;
        double *dst = a.out + ro * (int64_t)a.G + g0;
        for (int i = threadIdx.x; i < n + 2 * H; i += blockDim.x) {
            const int g = i - H;
            double x = 0.0;
            if (g >= 0 && g < n) {
                x = src[g];
                if (a.mask & ICNV_ST_SUBTRACT_REF_1) x = lg_subtract_ref(x, a.b1[g0 + g], a.b1[a.G + g0 + g]);
                if (a.mask & ICNV_ST_MAX_THRESH) x = fmax(fmin(x, a.max_thresh), -a.max_thresh);   // R/inferCNV_ops.R:2974-2975
            }
            if (smooth) sx[i] = x;
            else if (g >= 0 && g < n) dst[g] = x;
        }
        if (!smooth) continue;
        __syncthreads();
        const int64_t full = (int64_t)H * H;
        for (int g = threadIdx.x; g < n; g += blockDim.x) {
            const double *p = sx + H + g;
            double acc = (double)H * p[0];
            for (int k = 1; k <= T; ++k) acc = fma((double)(H - k), p[-k] + p[k], acc);
            const int64_t rl = T - g > 0 ? T - g : 0, rr = T - (n - 1 - g) > 0 ? T - (n - 1 - g) : 0;
            const double den = (double)(full - rl * (rl + 1) / 2 - rr * (rr + 1) / 2);
            dst[g] = acc * (1.0 / den);   // the fused kernel multiplies by the same correctly rounded reciprocal
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- M: step 11 (in place)
// value of 0-based rank `rank` among the G values of row p (NaN when fewer than rank + 1 values are comparable)
// (`each(f)`: calls f(x) for every value this thread holds of the row -- read from global memory, or kept in registers)
// `next` (nullable): receives the value of rank + 1 when it lies among the ranked candidates of the same bin (the usual case: the
// two middle values of an even G in one selection instead of two), NaN-boxed "unknown" otherwise: *have_next = 0
template <int NT, class Each>
__device__ __forceinline__ double lg_select_rank_of(Each each, int rank, uint32_t *hist, double *cand, double *red, int32_t *sel,
                                                    double *seld, double *next = nullptr, int *have_next = nullptr) {
    const int t = threadIdx.x;
    if (have_next) *have_next = 0;
    double lo = __builtin_inf(), hi = -__builtin_inf();
    each([&](double x) { lo = fmin(lo, x); hi = fmax(hi, x); });
    lg_block_minmax<NT>(lo, hi, red);
    int base = 0;
    for (int level = 0; level < 80; ++level) {
        if (!(lo < hi)) return lo;
        for (int b = t; b < LG_BINS; b += NT) hist[b] = 0u;
        if (t == 0) { sel[0] = -1; sel[3] = 0; }
        __syncthreads();
        const double scale = fmin((double)LG_BINS / (hi - lo), 0x1p1000);
        each([&](double x) {
            if (x >= lo && x <= hi) atomicAdd(&hist[min((uint32_t)__double2uint_rz((x - lo) * scale), (uint32_t)(LG_BINS - 1))], 1u);
        });
        __syncthreads();
        if (t < 64) {   // one wavefront scans: lane l owns 32 bins
            constexpr int BPL = LG_BINS / 64;
            // (an opaque copy of the lane index: the 32 bin addresses of a lane are invariant across levels and rows, and the compiler
            // would otherwise keep them -- in scratch -- for the whole kernel)
            int tl = t;
            asm volatile("" : "+v"(tl));
            const uint32_t *hb = hist + tl * BPL;
            uint32_t mine = 0;
#pragma unroll 1
            for (int b = 0; b < BPL; b += 4) {
                const uint4 h4 = *reinterpret_cast<const uint4 *>(hb + b);
                mine += (h4.x + h4.y) + (h4.z + h4.w);
            }
            uint32_t inc = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o, 64); if (t >= o) inc += v; }
            const uint32_t before = inc - mine, rel = (uint32_t)(rank - base);
            if (rel >= before && rel < before + mine) {
                uint32_t acc = before;
#pragma unroll 1
                for (int b = 0; b < BPL; ++b) {
                    const uint32_t h = hb[b];
                    if (rel >= acc && rel < acc + h) { sel[0] = tl * BPL + b; sel[1] = (int)acc; sel[2] = (int)h; }
                    acc += h;
                }
            }
        }
        __syncthreads();
        const int sbin = sel[0], sbefore = sel[1], scnt = sel[2];
        if (sbin < 0) return __builtin_nan("");   // the rank lies beyond the comparable values (NaNs in the row)
        if (scnt <= LG_CAND) {
            each([&](double x) {
                if (x >= lo && x <= hi && (int)min((uint32_t)__double2uint_rz((x - lo) * scale), (uint32_t)(LG_BINS - 1)) == sbin) {
                    const int pos = atomicAdd(&sel[3], 1);
                    if (pos < LG_CAND) cand[pos] = x;
                }
            });
            __syncthreads();
            const int want = rank - base - sbefore;
            for (int ci = t; ci < scnt; ci += NT) {
                const double cv = cand[ci];
                int less = 0;
                for (int cj = 0; cj < scnt; ++cj) {
                    const double o = cand[cj];
                    less += (o < cv || (o == cv && cj < ci)) ? 1 : 0;
                }
                if (less == want) seld[0] = cv;
                if (less == want + 1) seld[1] = cv;
            }
            __syncthreads();
            const double r = seld[0];
            if (next && want + 1 < scnt) { *next = seld[1]; *have_next = 1; }
            __syncthreads();
            return r;
        }
        // refine inside the selected bin
        base += sbefore;
        double nlo = __builtin_inf(), nhi = -__builtin_inf();
        each([&](double x) {
            if (x >= lo && x <= hi && (int)min((uint32_t)__double2uint_rz((x - lo) * scale), (uint32_t)(LG_BINS - 1)) == sbin) {
                nlo = fmin(nlo, x);
                nhi = fmax(nhi, x);
            }
        });
        lg_block_minmax<NT>(nlo, nhi, red);
        lo = nlo;
        hi = nhi;
    }
    return lo;
}

template <int NT>
__device__ double lg_select_rank(const double *p, int G, int rank, uint32_t *hist, double *cand, double *red, int32_t *sel,
                                 double *seld) {
    const int t = threadIdx.x;
    return lg_select_rank_of<NT>([&](auto f) { for (int g = t; g < G; g += NT) f(p[g]); }, rank, hist, cand, red, sel, seld);
}

__global__ void __launch_bounds__(LG_NT) large_center_kernel(const LargeChainArgs a) {
    __shared__ uint32_t hist[LG_BINS];
    __shared__ double cand[LG_CAND];
    __shared__ double red[2 * LG_NT / 64];
    __shared__ int32_t sel[4];
    __shared__ double seld[2];
    const int G = a.G, t = threadIdx.x;
    for (int row = blockIdx.x; row < a.n_rows; row += gridDim.x) {
        const int64_t ro = a.out_rows ? a.out_rows[row] : row;
        double *p = a.out + ro * (int64_t)G;
        double center;
        if (a.mask & ICNV_ST_CENTER_MEAN) {
            double s = 0.0;
            for (int g = t; g < G; g += LG_NT) s += p[g];
            center = lg_block_sum<LG_NT>(s, red) / (double)G;
        } else {   // R/inferCNV_ops.R:2098: median, mean of the two middle values for even G
            double m_hi = 0.0;
            int have = 0;
            auto each = [&](auto f) { for (int g = t; g < G; g += LG_NT) f(p[g]); };
            const double m_lo = lg_select_rank_of<LG_NT>(each, (G - 1) >> 1, hist, cand, red, sel, seld, &m_hi, &have);
            center = m_lo;
            if (!(G & 1)) {
                if (!have) m_hi = lg_select_rank_of<LG_NT>(each, G >> 1, hist, cand, red, sel, seld);   // (workgroup-uniform: `have` comes from LDS)
                center = (m_lo + m_hi) * 0.5;
            }
        }
        __syncthreads();
        for (int g = t; g < G; g += LG_NT) p[g] -= center;
        __syncthreads();
    }
}

// ---------------------------------------------------------------- E: steps 12, 14, 22 (+ per-row sum and sd)
__global__ void __launch_bounds__(LG_NT) large_finish_kernel(const LargeChainArgs a) {
    __shared__ double red[2 * LG_NT / 64];
    const int G = a.G, t = threadIdx.x;
    double mu = 0.0, lo_d = 0.0, hi_d = 0.0;
    if (a.mask & ICNV_ST_DENOISE) { mu = a.denoise[0]; lo_d = mu - a.denoise[1]; hi_d = mu + a.denoise[1]; }
    for (int row = blockIdx.x; row < a.n_rows; row += gridDim.x) {
        const int64_t ri = a.in_rows ? a.in_rows[row] : row, ro = a.out_rows ? a.out_rows[row] : row;
        const double *src = a.in + ri * (int64_t)G;
        auto value = [&](int g) {
            double x = src[g];
            if (a.mask & ICNV_ST_SUBTRACT_REF_2) x = lg_subtract_ref(x, a.b2[g], a.b2[G + g]);
            if (a.mask & ICNV_ST_INVERT_LOG2) x = exp2(x);   // R/inferCNV_ops.R:2818
            return x;
        };
        if (a.cell_stats) {   // sum and sample sd over genes, two passes like R's sd()
            double s = 0.0;
            for (int g = t; g < G; g += LG_NT) s += value(g);
            const double tot = lg_block_sum<LG_NT>(s, red);
            const double mean = tot / (double)G;
            double ss = 0.0;
            for (int g = t; g < G; g += LG_NT) { const double d = value(g) - mean; ss += d * d; }
            const double sst = lg_block_sum<LG_NT>(ss, red);
            if (t == 0) { a.cell_stats[2 * (int64_t)row] = tot; a.cell_stats[2 * (int64_t)row + 1] = sqrt(sst / (double)(G - 1)); }
            continue;
        }
        double *dst = a.out + ro * (int64_t)G;
        double *dpre = a.pre ? a.pre + ro * (int64_t)G : nullptr;
        for (int g = t; g < G; g += LG_NT) {
            const double x = value(g);
            if (dpre) dpre[g] = x;
            double o = x;
            if ((a.mask & ICNV_ST_DENOISE) && o > lo_d && o < hi_d) o = mu;   // strict bounds, R/inferCNV_ops.R:2335
            dst[g] = o;
        }
    }
}

// ---------------------------------------------------------------- M + E in one pass (round 6): steps 11, 12, 14, 22
// Pass 2 of the two-pass chain: the smoothed row (pass 1's output) is read ONCE into registers -- NS slots of two genes per thread,
// G <= 2 NS x 1024 --, the median comes from the same histogram select (slots beyond G hold NaN: no comparison selects them, like a NaN
// of the data), then steps 11 - 22 run on the registers and the denoised row and the HMM input are written: 8 B read + 16 B written
// per gene and cell, where the M and E kernels read the row five to six times.  Arithmetic per value exactly as in those two kernels.
template <int NS>
__global__ void __launch_bounds__(LG_NT) large_center_finish_kernel(const LargeChainArgs a) {
    __shared__ uint32_t hist[LG_BINS];
    __shared__ double cand[LG_CAND];
    __shared__ double red[2 * LG_NT / 64];
    __shared__ int32_t sel[4];
    __shared__ double seld[2];
    typedef double dv2 __attribute__((ext_vector_type(2)));
    const int G = a.G, t = threadIdx.x;
    double mu = 0.0, lo_d = 0.0, hi_d = 0.0;
    if (a.mask & ICNV_ST_DENOISE) { mu = a.denoise[0]; lo_d = mu - a.denoise[1]; hi_d = mu + a.denoise[1]; }
    for (int row = blockIdx.x; row < a.n_rows; row += gridDim.x) {
        const int64_t ri = a.in_rows ? a.in_rows[row] : row, ro = a.out_rows ? a.out_rows[row] : row;
        const double *src = a.in + ri * (int64_t)G;
        double v[NS][2];
        // (opaque copies of the thread index, one per phase: computed from `t` itself, the per-slot addresses and validity masks are
        // loop-invariant, get hoisted out of the row loop and spilled -- 600 to 1 200 bytes of scratch per lane before this)
        int t1 = t;
        asm volatile("" : "+v"(t1));
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int g = 2 * (t1 + s * LG_NT);
            v[s][0] = v[s][1] = __builtin_nan("");
            if (g + 1 < G) {
                const dv2 d = __builtin_nontemporal_load(reinterpret_cast<const dv2 *>(src + g));
                v[s][0] = d.x;
                v[s][1] = d.y;
            } else if (g < G) {
                v[s][0] = src[g];
            }
        }
        auto each = [&](auto f) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                f(v[s][0]);
                f(v[s][1]);
            }
        };
        double center = 0.0;
        if (a.mask & ICNV_ST_CENTER_MEAN) {
            double sum = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int g = 2 * (t1 + s * LG_NT);
                if (g < G) sum += v[s][0];
                if (g + 1 < G) sum += v[s][1];
            }
            center = lg_block_sum<LG_NT>(sum, red) / (double)G;
        } else if (a.mask & ICNV_ST_CENTER) {   // R/inferCNV_ops.R:2098: median, mean of the two middle values for even G
            double m_hi = 0.0;
            int have = 0;
            const double m_lo = lg_select_rank_of<LG_NT>(each, (G - 1) >> 1, hist, cand, red, sel, seld, &m_hi, &have);
            center = m_lo;
            if (!(G & 1)) {
                if (!have) m_hi = lg_select_rank_of<LG_NT>(each, G >> 1, hist, cand, red, sel, seld);   // (workgroup-uniform: `have` comes from LDS)
                center = (m_lo + m_hi) * 0.5;
            }
        }
        double *dst = a.out + ro * (int64_t)G;
        double *dpre = a.pre ? a.pre + ro * (int64_t)G : nullptr;
        int t2 = t;
        asm volatile("" : "+v"(t2));
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int g = 2 * (t2 + s * LG_NT);
            double o[2], p2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                double x = v[s][j];
                if (g + j < G) {
                    if (a.mask & (ICNV_ST_CENTER | ICNV_ST_CENTER_MEAN)) x -= center;
                    if (a.mask & ICNV_ST_SUBTRACT_REF_2) x = lg_subtract_ref(x, a.b2[g + j], a.b2[G + g + j]);
                    if (a.mask & ICNV_ST_INVERT_LOG2) x = exp2(x);   // R/inferCNV_ops.R:2818
                }
                p2[j] = x;
                o[j] = ((a.mask & ICNV_ST_DENOISE) && x > lo_d && x < hi_d) ? mu : x;   // strict bounds, R/inferCNV_ops.R:2335
            }
            if (g + 1 < G) {
                dv2 d;
                if (dpre) { d.x = p2[0]; d.y = p2[1]; __builtin_nontemporal_store(d, reinterpret_cast<dv2 *>(dpre + g)); }
                d.x = o[0]; d.y = o[1];
                __builtin_nontemporal_store(d, reinterpret_cast<dv2 *>(dst + g));
            } else if (g < G) {
                if (dpre) dpre[g] = p2[0];
                dst[g] = o[0];
            }
        }
        __syncthreads();
    }
}

// sums[q*G + g] = sum over the rows idx[off[q]..off[q+1]) of x[row][g];  counts[q] = number of rows (fixed order)
__global__ void large_group_sums_kernel(const double *__restrict__ x, int G, const int32_t *__restrict__ idx,
                                        const int32_t *__restrict__ off, int n_grp, double *__restrict__ sums) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = blockIdx.y;
    if (g < G) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int i = off[q];
        const int e = off[q + 1];
        for (; i + 4 <= e; i += 4) {
            s0 += x[(int64_t)(idx ? idx[i] : i) * G + g];
            s1 += x[(int64_t)(idx ? idx[i + 1] : i + 1) * G + g];
            s2 += x[(int64_t)(idx ? idx[i + 2] : i + 2) * G + g];
            s3 += x[(int64_t)(idx ? idx[i + 3] : i + 3) * G + g];
        }
        for (; i < e; ++i) s0 += x[(int64_t)(idx ? idx[i] : i) * G + g];
        sums[(int64_t)q * G + g] = (s0 + s1) + (s2 + s3);
    }
    if (g == 0) sums[(int64_t)n_grp * G + q] = (double)(off[q + 1] - off[q]);
}

}  // namespace

size_t chain_large_lds_bytes(int32_t max_chr_len, int32_t T) { return ((size_t)max_chr_len + 2 * (size_t)(T + 1)) * sizeof(double); }

int launch_chain_large_smooth(const LargeChainArgs &a, int32_t max_chr_len, hipStream_t stream) {
    if (a.n_rows <= 0) return ICNV_OK;
    const size_t lds = chain_large_lds_bytes(max_chr_len, a.T);
    if (lds > 152 * 1024) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "a single chromosome does not fit the 160 KiB LDS (more than ~19 000 genes)");
    static DeviceOnce once;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(large_smooth_kernel), 152 * 1024, once)) return rc;
    int gy = a.n_rows < 4096 ? a.n_rows : 4096;
    KernelTimer kt("chain_large_smooth", stream);
    hipLaunchKernelGGL(large_smooth_kernel, dim3(a.n_chr, gy), dim3(256), lds, stream, a);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_chain_large_center(const LargeChainArgs &a, hipStream_t stream) {
    if (a.n_rows <= 0) return ICNV_OK;
    const int grid = a.n_rows < 4 * num_cus() ? a.n_rows : 4 * num_cus();
    KernelTimer kt("chain_large_center", stream);
    hipLaunchKernelGGL(large_center_kernel, dim3(grid), dim3(LG_NT), 0, stream, a);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_chain_large_finish(const LargeChainArgs &a, hipStream_t stream) {
    if (a.n_rows <= 0) return ICNV_OK;
    const int grid = a.n_rows < 8 * num_cus() ? a.n_rows : 8 * num_cus();
    KernelTimer kt("chain_large_finish", stream);
    hipLaunchKernelGGL(large_finish_kernel, dim3(grid), dim3(LG_NT), 0, stream, a);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

// steps 11 - 22 of `a.in` (the smoothed rows) in one pass; false when the row does not fit the kernel's registers (G > 32 768)
bool chain_large_center_finish_covers(int32_t G) { return G <= 2 * 16 * LG_NT; }
int launch_chain_large_center_finish(const LargeChainArgs &a, hipStream_t stream) {
    if (a.n_rows <= 0) return ICNV_OK;
    const int grid = a.n_rows < 2 * num_cus() ? a.n_rows : 2 * num_cus();
    KernelTimer kt("chain_large_center_finish", stream);
    const int need = (a.G + 2 * LG_NT - 1) / (2 * LG_NT);
    if (need <= 8) hipLaunchKernelGGL(large_center_finish_kernel<8>, dim3(grid), dim3(LG_NT), 0, stream, a);
    else if (need <= 10) hipLaunchKernelGGL(large_center_finish_kernel<10>, dim3(grid), dim3(LG_NT), 0, stream, a);
    else if (need <= 12) hipLaunchKernelGGL(large_center_finish_kernel<12>, dim3(grid), dim3(LG_NT), 0, stream, a);
    else if (need <= 16) hipLaunchKernelGGL(large_center_finish_kernel<16>, dim3(grid), dim3(LG_NT), 0, stream, a);
    else ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "large_center_finish: more than 32 768 genes");
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_chain_large_group_sums(const double *x, int32_t G, const int32_t *idx_dev, const int32_t *off_dev, int32_t n_grp,
                                  double *sums_counts, hipStream_t stream) {
    if (n_grp <= 0) return ICNV_OK;
    hipLaunchKernelGGL(large_group_sums_kernel, dim3((G + 255) / 256, n_grp), dim3(256), 0, stream, x, G, idx_dev, off_dev, n_grp,
                       sums_counts);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

}  // namespace icnv
