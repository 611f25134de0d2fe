// i6 / i3 HMM Viterbi (Viterbi.dthmm.adj, R/inferCNV_HMM.R:1101-1176) for gfx950.
//
// Mapping: one LANE per sequence.  A wavefront takes 64 consecutive columns
// (cells, or group-mean profiles) of one chromosome; every lane runs the whole
// pipeline for its own column -- the K emission scores of a gene (Cody's
// log upper-tail normal + the reference's 1/(-lp) normalisation + log) and the
// K x K max-plus recurrence -- so there is no cross-lane traffic, no idle lanes
// in the sequential DP, and the K x K transition matrix / emission parameters
// are wave-uniform (scalar registers).  (chromosome, column-block) tasks are
// issued longest-chromosome-first so the tail of the grid is short.
//
// Back-pointers (K x 3 bits per gene, first-max like R's which.max) go to a
// gene-major scratch plane bp[(gene)*ncols + col] so that a wavefront's 64
// lanes write 256 contiguous bytes; the traceback re-reads them and emits the
// states, packing 8 consecutive genes into one 8-byte store where the
// alignment allows.
//
// Arithmetic spec: every operation is an individually rounded
// IEEE-754 double operation in the reference's order -- this file is compiled
// with -ffp-contract=off -- so state calls are bit-identical to the CPU oracle.
#include "icnv_internal.h"

#pragma clang fp contract(off)

namespace icnv {

namespace {

__device__ inline double dev_log(double x) {
    // log x = k ln2 + log(1+f),  s = f/(2+f),  log(1+f) = f - hfsq + s (hfsq + R(s^2))
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                     Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                     Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                     Lg7 = 1.479819860511658591e-01;
    int32_t hx = __double2hiint(x);
    uint32_t lx = (uint32_t)__double2loint(x);
    int32_t k = 0;
    if (__builtin_expect(hx < 0x00100000, 0)) {  // zero, negative or subnormal
        if (((hx & 0x7fffffff) | lx) == 0) return -__builtin_inf();
        if (hx < 0) return __builtin_nan("");
        k = -54;
        x *= 1.80143985094819840000e+16;
        hx = __double2hiint(x);
        lx = (uint32_t)__double2loint(x);
    }
    if (__builtin_expect(hx >= 0x7ff00000, 0)) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int32_t i = (hx + 0x95f64) & 0x100000;
    x = __hiloint2double(hx | (i ^ 0x3ff00000), (int32_t)lx);
    k += (i >> 20);
    const double f = x - 1.0;
    const double dk = (double)k;
    if (__builtin_expect((0x000fffff & (2 + hx)) < 3, 0)) {  // |f| < 2^-20
        if (f == 0.0) return (k == 0) ? 0.0 : dk * ln2_hi + dk * ln2_lo;
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        return (k == 0) ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const int32_t ij = (hx - 0x6147a) | (0x6b851 - hx);
    if (ij > 0) {
        const double hfsq = 0.5 * f * f;
        const double v = s * (hfsq + R);
        return (k == 0) ? f - (hfsq - v) : dk * ln2_hi - ((hfsq - (v + dk * ln2_lo)) - f);
    }
    const double v = s * (f - R);
    return (k == 0) ? f - v : dk * ln2_hi - ((v - dk * ln2_lo) - f);
}

// log P(Z > y), y >= 0 -- pnorm_both()'s three ranges (R nmath/pnorm.c, Cody 1969).
// The three ranges share ONE division site and ONE log site (the rational's
// numerator factor is y, 1 or 1/y^2; multiplying by 1.0 is exact), which is
// what a diverged wavefront executes once instead of three times; every lane
// still performs exactly the reference's operation sequence for its range.
__device__ inline double dev_pnorm_log_upper(double y) {
    double num, den, f1, cn, cd;
    const bool r1 = (y <= 0.67448975);
    const bool r3 = !(y <= 5.656854249492380195206754896838);
    if (r1) {
        const double q = y * y;
        num = 0.065682337918207449113 * q; den = q;
        num = (num + 2.2352520354606839287) * q;  den = (den + 47.20258190468824187) * q;
        num = (num + 161.02823106855587881) * q;  den = (den + 976.09855173777669322) * q;
        num = (num + 1067.6894854603709582) * q;  den = (den + 10260.932208618978205) * q;
        f1 = y; cn = 18154.981253343561249; cd = 45507.789335026729956;
    } else if (!r3) {
        num = 1.0765576773720192317e-8 * y; den = y;
        num = (num + 0.39894151208813466764) * y;  den = (den + 22.266688044328115691) * y;
        num = (num + 8.8831497943883759412) * y;   den = (den + 235.38790178262499861) * y;
        num = (num + 93.506656132177855979) * y;   den = (den + 1519.377599407554805) * y;
        num = (num + 597.27027639480026226) * y;   den = (den + 6485.558298266760755) * y;
        num = (num + 2494.5375852903726711) * y;   den = (den + 18615.571640885098091) * y;
        num = (num + 6848.1904505362823326) * y;   den = (den + 34900.952721145977266) * y;
        num = (num + 11602.651437647350124) * y;   den = (den + 38912.003286093271411) * y;
        f1 = 1.0; cn = 9842.7148383839780218; cd = 19685.429676859990727;
    } else {
        const double q = 1.0 / (y * y);
        num = 0.02307344176494017303 * q; den = q;
        num = (num + 0.21589853405795699) * q;       den = (den + 1.28426009614491121) * q;
        num = (num + 0.1274011611602473639) * q;     den = (den + 0.468238212480865118) * q;
        num = (num + 0.022235277870649807) * q;      den = (den + 0.0659881378689285515) * q;
        num = (num + 0.001421619193227893466) * q;   den = (den + 0.00378239633202758244) * q;
        f1 = q; cn = 2.9112874951168792e-5; cd = 7.29751555083966205e-5;
    }
    double tmp = f1 * (num + cn) / (den + cd);
    if (r3) tmp = (0.398942280401432677939946059934 - tmp) / y;
    const double lg = dev_log(r1 ? 0.5 - tmp : tmp);
    if (r1) return lg;
    const double xs = __builtin_trunc(y * 16.0) / 16.0;
    const double del = (y - xs) * (y + xs);
    return (-xs * xs * 0.5) + (-del * 0.5) + lg;
}

// emission scores of one observation (R/inferCNV_HMM.R:1129-1133, 1156-1160)
template <int K>
__device__ inline void emission(double x, const HmmParams &p, double sd, double (&sc)[K]) {
    double e[K];
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double lp = dev_pnorm_log_upper(__builtin_fabs(x - p.mean[k]) / sd);
        e[k] = 1.0 / (-1.0 * lp);
        tot = (k == 0) ? e[0] : tot + e[k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) sc[k] = dev_log(e[k] / tot);
}

constexpr int VIT_NT = 256;  // 4 wavefronts = 4 tasks per block

template <int K>
__global__ void __launch_bounds__(VIT_NT) viterbi_kernel(const double *__restrict__ x, uint8_t *__restrict__ states,
                                                         int G, int64_t ncols, const int32_t *__restrict__ chr_start,
                                                         const int32_t *__restrict__ chr_order, int n_chr,
                                                         const HmmParams p, const double *__restrict__ sd_per_col,
                                                         double sd_shared, uint32_t *__restrict__ bp,
                                                         int32_t *n_underflow) {
    const int64_t ncg = (ncols + 63) >> 6;  // column blocks
    const int64_t task = (int64_t)blockIdx.x * (VIT_NT / 64) + (threadIdx.x >> 6);
    if (task >= ncg * n_chr) return;
    const int chr = chr_order[task / ncg];
    const int64_t col = (task % ncg) * 64 + (threadIdx.x & 63);
    if (col >= ncols) return;
    const int s = chr_start[chr];
    const int n = chr_start[chr + 1] - s;
    const double *xc = x + col * (int64_t)G + s;
    uint8_t *st = states + col * (int64_t)G + s;
    if (n < 2) {  // R/inferCNV_HMM.R:1104-1107: neutral state 3 (also under i3)
        if (n == 1) st[0] = 3;
        return;
    }
    const double sd = sd_per_col ? sd_per_col[col] : sd_shared;
    uint32_t *bpc = bp + (int64_t)s * ncols + col;

    double nu[K], sc[K];
    double xn = xc[0];
#pragma unroll
    for (int k = 0; k < K; ++k) nu[k] = 0.0;
    // one emission site for the whole sequence keeps the kernel's code footprint small
    for (int i = 0; i < n; ++i) {
        const double xv = xn;
        if (i + 1 < n) xn = xc[i + 1];
        emission<K>(xv, p, sd, sc);
        if (i == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) nu[k] = p.logDelta[k] + sc[k];
            continue;
        }
        double nn[K];
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double best = nu[0] + p.logPi[0 + K * k];
            uint32_t bj = 0;
#pragma unroll
            for (int j = 1; j < K; ++j) {
                const double v = nu[j] + p.logPi[j + K * k];
                if (v > best) { best = v; bj = j; }  // strict > keeps the first maximum
            }
            nn[k] = best + sc[k];
            word |= bj << (3 * k);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) nu[k] = nn[k];
        bpc[(int64_t)i * ncols] = word;
    }

    bool bad = false;
    int cur = 0;
    double bv = nu[0];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        bad |= (nu[k] == -__builtin_inf());
        if (k > 0 && nu[k] > bv) { bv = nu[k]; cur = k; }
    }
    if (bad && n_underflow) atomicAdd(n_underflow, 1);

    // traceback; states packed 8 genes per 64-bit store where aligned
    const uint64_t base = (uint64_t)(uintptr_t)st;
    uint64_t word = 0;
    for (int i = n - 1; i >= 0; --i) {
        const uint64_t addr = base + (uint64_t)i;
        const int b = (int)(addr & 7);
        word |= (uint64_t)(cur + 1) << (8 * b);
        if (b == 0 || i == 0) {
            const uint64_t w0 = addr - (uint64_t)b;
            const int hi = (int)((base + (uint64_t)(n - 1) - w0) < 7 ? (base + (uint64_t)(n - 1) - w0) : 7);
            if (b == 0 && hi == 7) {
                *reinterpret_cast<uint64_t *>(st + i) = word;
            } else {
                for (int bb = b; bb <= hi; ++bb) *reinterpret_cast<uint8_t *>(w0 + bb) = (uint8_t)(word >> (8 * bb));
            }
            word = 0;
        }
        if (i > 0) cur = (int)((bpc[(int64_t)i * ncols] >> (3 * cur)) & 7u);
    }
}

// rowMeans over a group's cells, two-stage for determinism + parallelism:
// part[(q*nsplit + sp)*G + g] = sum over the sp-th slice of group q's cells.
__global__ void group_partial_sums_kernel(const double *__restrict__ x, int G, const int32_t *__restrict__ idx,
                                          const int32_t *__restrict__ off, int nsplit, double *__restrict__ part) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = blockIdx.y, sp = blockIdx.z;
    if (g >= G) return;
    const int b = off[q], e = off[q + 1];
    const int n = e - b;
    const int per = (n + nsplit - 1) / nsplit;
    int lo = b + sp * per, hi = lo + per;
    if (hi > e) hi = e;
    double s = 0.0;
    for (int i = lo; i < hi; ++i) s += x[(int64_t)idx[i] * G + g];
    part[((int64_t)q * nsplit + sp) * G + g] = s;
}
__global__ void group_means_finish_kernel(const double *__restrict__ part, int G, const int32_t *__restrict__ off,
                                          int nsplit, double *__restrict__ out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = blockIdx.y;
    if (g >= G) return;
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += part[((int64_t)q * nsplit + sp) * G + g];
    out[(int64_t)q * G + g] = s / (double)(off[q + 1] - off[q]);
}

// states[:, c] = grp_states[:, cell_to_grp[c]]  (0xFF where the cell is in no group)
__global__ void broadcast_states_kernel(const uint8_t *__restrict__ gs, int G, int64_t C,
                                        const int32_t *__restrict__ cell_to_grp, uint8_t *__restrict__ states) {
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        const int q = cell_to_grp[c];
        uint8_t *dst = states + c * (int64_t)G;
        if ((G & 15) == 0) {
            const uint4 *src = q >= 0 ? reinterpret_cast<const uint4 *>(gs + (int64_t)q * G) : nullptr;
            uint4 *d4 = reinterpret_cast<uint4 *>(dst);
            const uint4 ff = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            for (int i = threadIdx.x; i < (G >> 4); i += blockDim.x) d4[i] = src ? src[i] : ff;
        } else {
            const uint8_t *src = q >= 0 ? gs + (int64_t)q * G : nullptr;
            for (int i = threadIdx.x; i < G; i += blockDim.x) dst[i] = src ? src[i] : (uint8_t)0xFF;
        }
    }
}

__global__ void states_to_proxy_kernel(const uint8_t *__restrict__ st, double *__restrict__ out, int64_t n, int K) {
    const double nan = __builtin_nan("");
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int s = st[i];
        double v = nan;
        if (K == 3) {  // R/inferCNV_i3HMM.R:405-417
            v = s == 1 ? 0.5 : s == 2 ? 1.0 : s == 3 ? 1.5 : nan;
        } else {       // R/inferCNV_HMM.R:1191-1206
            v = s == 1 ? 0.0 : s == 2 ? 0.5 : s == 3 ? 1.0 : s == 4 ? 1.5 : s == 5 ? 2.0 : s == 6 ? 3.0 : nan;
        }
        out[i] = v;
    }
}

}  // namespace

size_t viterbi_scratch_bytes(int32_t G, int64_t n_cols) { return (size_t)G * (size_t)n_cols * sizeof(uint32_t); }

int launch_viterbi(const double *x, uint8_t *states, int32_t G, int64_t n_cols, const int32_t *chr_start_dev,
                   const int32_t *chr_order_dev, int32_t n_chr, int32_t max_chr_len, const HmmParams &p,
                   const double *sd_per_col_dev, double sd_shared, uint32_t *bp_scratch, int32_t *n_underflow,
                   hipStream_t stream) {
    (void)max_chr_len;
    if (n_cols <= 0 || n_chr <= 0) return ICNV_OK;
    const int64_t ncg = (n_cols + 63) / 64;
    const int64_t tasks = ncg * n_chr;
    const int64_t blocks = (tasks + (VIT_NT / 64) - 1) / (VIT_NT / 64);
    if (blocks > 0x7fffffff) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "too many Viterbi tasks for one launch");
    KernelTimer kt("viterbi", stream);
    if (p.K == 6) {
        hipLaunchKernelGGL(viterbi_kernel<6>, dim3((unsigned)blocks), dim3(VIT_NT), 0, stream, x, states, G, n_cols,
                           chr_start_dev, chr_order_dev, n_chr, p, sd_per_col_dev, sd_shared, bp_scratch, n_underflow);
    } else if (p.K == 3) {
        hipLaunchKernelGGL(viterbi_kernel<3>, dim3((unsigned)blocks), dim3(VIT_NT), 0, stream, x, states, G, n_cols,
                           chr_start_dev, chr_order_dev, n_chr, p, sd_per_col_dev, sd_shared, bp_scratch, n_underflow);
    } else {
        ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "HMM kernels are built for K = 6 (i6) and K = 3 (i3)");
    }
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_broadcast_states(const uint8_t *grp_states, int32_t G, int64_t C, const int32_t *cell_to_grp_dev,
                            uint8_t *states, hipStream_t stream) {
    if (C <= 0) return ICNV_OK;
    KernelTimer kt("broadcast_states", stream);
    const int grid = (int)(C < 65536 ? C : 65536);
    hipLaunchKernelGGL(broadcast_states_kernel, dim3(grid), dim3(256), 0, stream, grp_states, G, C, cell_to_grp_dev,
                       states);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_states_to_proxy(const uint8_t *states, double *out, int64_t n, int32_t K, hipStream_t stream) {
    if (n <= 0) return ICNV_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(states_to_proxy_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, states, out, n, K);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

// exported helpers for api.hip (group means need a partial buffer supplied by the caller)
int group_means_nsplit(int32_t G, int32_t n_grp) {
    const int tiles = (G + 255) / 256;
    int ns = (2048 + tiles * n_grp - 1) / (tiles * n_grp);
    if (ns < 1) ns = 1;
    if (ns > 64) ns = 64;
    return ns;
}

int launch_group_means_ws(const double *x, int32_t G, const int32_t *grp_idx_dev, const int32_t *grp_off_dev,
                          int32_t n_grp, int nsplit, double *part, double *out, hipStream_t stream) {
    if (n_grp <= 0) return ICNV_OK;
    KernelTimer kt("group_means", stream);
    const int tiles = (G + 255) / 256;
    hipLaunchKernelGGL(group_partial_sums_kernel, dim3(tiles, n_grp, nsplit), dim3(256), 0, stream, x, G, grp_idx_dev,
                       grp_off_dev, nsplit, part);
    hipLaunchKernelGGL(group_means_finish_kernel, dim3(tiles, n_grp), dim3(256), 0, stream, part, G, grp_off_dev, nsplit,
                       out);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

}  // namespace icnv
