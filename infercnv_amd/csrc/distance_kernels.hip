// Euclidean distances between the cells of one group: parallelDist(t(expr[, cells])) as the reference calls
// it before hclust (R/inferCNV_tumor_subclusters.R:191, R/inferCNV_ops.R:1930, 3242) -- the
// one dense contraction next to the hot path, and the only kernel of this library that belongs on the matrix cores.
//
//   y_i = x_i - mean over the group's cells        (per gene; distances are translation invariant, and centring
//                                                    removes the cancellation of the Gram formulation)
//   S   = Y Y^T   (n x n Gram matrix)               v_mfma_f64_16x16x4_f64, 64 x 64 output tile per workgroup,
//                                                    upper-triangular tiles only, mirrored on the way out
//   D_ij = sqrt(max(0, S_ii + S_jj - 2 S_ij)),  D_ii = 0
//
// fp64 throughout (the reference's distances are doubles and feed a hierarchical clustering whose merge order
// depends on them); the fp64 MFMA rate equals the fp64 vector rate on gfx950 (78.6 TFLOP/s), the matrix cores'
// gain here is the operand reuse: one 8-byte LDS read per lane feeds 2 x 16 x 16 x 4 multiply-adds.
#include "icnv_internal.h"

namespace icnv {

namespace {

typedef double dbl4_t __attribute__((ext_vector_type(4)));

constexpr int DT = 64;        // output tile (cells x cells) per workgroup
constexpr int KC = 32;        // genes per LDS stage
constexpr int LDR = KC + 2;   // LDS row stride in doubles: (4 row + 2 k) dwords mod 64 are distinct within a 32-lane group

// One workgroup = 4 wavefronts = 2 x 2 sub-tiles of 32 x 32; a wavefront holds 2 x 2 MFMA accumulators (16 x 16 each).
__global__ void __launch_bounds__(256) gram_tiles_kernel(const double *__restrict__ x, int G, const int32_t *__restrict__ idx,
                                                         int n, const double *__restrict__ mean, double *__restrict__ S) {
    __shared__ double As[DT * LDR];
    __shared__ double Bs[DT * LDR];
    // upper-triangular tile pair (bi <= bj) of this workgroup
    const int nt = (n + DT - 1) / DT;
    int bi = 0, rem = blockIdx.x;
    while (rem >= nt - bi) { rem -= nt - bi; ++bi; }
    const int bj = bi + rem;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    dbl4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (dbl4_t){0.0, 0.0, 0.0, 0.0};

    // staging: thread t loads 8 consecutive genes of row t / 4 (64 rows x 32 genes per tile and stage)
    const int lrow = t >> 2, lseg = (t & 3) * 8;
    const int ra = bi * DT + lrow, rb = bj * DT + lrow;
    const double *pa = ra < n ? x + (int64_t)idx[ra] * G : nullptr;
    const double *pb = rb < n ? x + (int64_t)idx[rb] * G : nullptr;

    for (int k0 = 0; k0 < G; k0 += KC) {
        if ((G & 1) == 0) {   // 16-byte loads: every row starts at an even element and g is even
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const int g = k0 + lseg + j;
                const bool in = g < G;
                const double2 m = in ? *reinterpret_cast<const double2 *>(mean + g) : make_double2(0.0, 0.0);
                const double2 va = (pa && in) ? *reinterpret_cast<const double2 *>(pa + g) : m;
                const double2 vb = (pb && in) ? *reinterpret_cast<const double2 *>(pb + g) : m;
                As[lrow * LDR + lseg + j] = va.x - m.x;
                As[lrow * LDR + lseg + j + 1] = va.y - m.y;
                Bs[lrow * LDR + lseg + j] = vb.x - m.x;
                Bs[lrow * LDR + lseg + j + 1] = vb.y - m.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int g = k0 + lseg + j;
                const double m = g < G ? mean[g] : 0.0;
                As[lrow * LDR + lseg + j] = (pa && g < G) ? pa[g] - m : 0.0;
                Bs[lrow * LDR + lseg + j] = (pb && g < G) ? pb[g] - m : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const int k = kk + (lane >> 4), r = lane & 15;
            const double a0 = As[(wr * 32 + r) * LDR + k], a1 = As[(wr * 32 + 16 + r) * LDR + k];
            const double b0 = Bs[(wc * 32 + r) * LDR + k], b1 = Bs[(wc * 32 + 16 + r) * LDR + k];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = bi * DT + wr * 32 + a * 16 + (lane >> 4) + 4 * reg;
                const int col = bj * DT + wc * 32 + b * 16 + (lane & 15);
                if (row < n && col < n) {
                    const double v = acc[a][b][reg];
                    S[(int64_t)row * n + col] = v;
                    if (bi != bj) S[(int64_t)col * n + row] = v;
                }
            }
}

__global__ void gram_diag_kernel(const double *__restrict__ S, int n, double *__restrict__ diag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) diag[i] = S[(int64_t)i * n + i];
}

// in place: S -> D
__global__ void gram_to_dist_kernel(double *__restrict__ S, int n, const double *__restrict__ diag) {
    const int64_t total = (int64_t)n * n;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(p / n), j = (int)(p - (int64_t)i * n);
        const double d2 = diag[i] + diag[j] - 2.0 * S[p];
        S[p] = (i == j) ? 0.0 : sqrt(fmax(d2, 0.0));
    }
}

}  // namespace

int launch_cell_distances(const double *x, int32_t G, const int32_t *idx_dev, int32_t n, const double *mean_dev,
                          double *diag_dev, double *out, hipStream_t stream) {
    if (n <= 0) return ICNV_OK;
    const int64_t nt = (n + DT - 1) / DT;
    const int64_t tiles = nt * (nt + 1) / 2;
    if (tiles > 0x7fffffff) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "too many cells for one distance matrix");
    {
        KernelTimer kt("cell_distances_gram", stream);
        hipLaunchKernelGGL(gram_tiles_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, x, G, idx_dev, n, mean_dev, out);
    }
    hipLaunchKernelGGL(gram_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, out, n, diag_dev);
    int64_t blocks = ((int64_t)n * n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(gram_to_dist_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, n, diag_dev);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

}  // namespace icnv
