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

constexpr int KC = 32;        // genes per LDS stage
constexpr int LDR = KC + 2;   // LDS row stride in doubles: (4 row + 2 k) dwords mod 64 are distinct within a 32-lane group

// One workgroup = 4 wavefronts = 2 x 2 sub-tiles; a wavefront holds WM x WM MFMA accumulators (16 x 16 each), so the
// workgroup's output tile is DT = 32 WM cells square: WM = 2 for small groups (enough tiles to fill 256 CUs),
// WM = 4 for large ones (each 8-byte LDS operand read then feeds four matrix instructions instead of two).
template <int WM>
__global__ void __launch_bounds__(256, (WM == 4 ? 2 : 4)) gram_tiles_kernel(const double *__restrict__ x, int G, const int32_t *__restrict__ idx,
                                                         int n, const double *__restrict__ mean, double *__restrict__ S) {
    constexpr int DT = 32 * WM;
    constexpr int RPT = DT / 64;          // rows staged per thread and tile
    extern __shared__ __attribute__((aligned(16))) double smem_d[];
    double *As = smem_d;
    double *Bs = smem_d + DT * LDR;
    // upper-triangular tile pair (bi <= bj) of this workgroup
    const int nt = (n + DT - 1) / DT;
    int bi = 0, rem = blockIdx.x;
    while (rem >= nt - bi) { rem -= nt - bi; ++bi; }
    const int bj = bi + rem;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    dbl4_t acc[WM][WM];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WM; ++b) acc[a][b] = (dbl4_t){0.0, 0.0, 0.0, 0.0};

    // staging: thread t loads 8 consecutive genes of rows t / 4 (+ 64) of each tile.  The next stage is requested into
    // registers before the current stage's MFMAs and parked in LDS after them: its HBM/L2 latency hides behind the
    // matrix instructions instead of standing between barriers.
    const int lrow = t >> 2, lseg = (t & 3) * 8;
    const double *pa[RPT], *pb[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int ra = bi * DT + lrow + 64 * r, rb = bj * DT + lrow + 64 * r;
        pa[r] = ra < n ? x + (int64_t)idx[ra] * G : nullptr;
        pb[r] = rb < n ? x + (int64_t)idx[rb] * G : nullptr;
    }
    const bool even = (G & 1) == 0;   // 16-byte loads: every row starts at an even element and g is even

    double ra_v[RPT][8], rb_v[RPT][8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (even) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const int g = k0 + lseg + j;
                    const bool in = g < G;
                    const double2 m = in ? *reinterpret_cast<const double2 *>(mean + g) : make_double2(0.0, 0.0);
                    const double2 va = (pa[r] && in) ? *reinterpret_cast<const double2 *>(pa[r] + g) : m;
                    const double2 vb = (pb[r] && in) ? *reinterpret_cast<const double2 *>(pb[r] + g) : m;
                    ra_v[r][j] = va.x - m.x; ra_v[r][j + 1] = va.y - m.y;
                    rb_v[r][j] = vb.x - m.x; rb_v[r][j + 1] = vb.y - m.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int g = k0 + lseg + j;
                    const double m = g < G ? mean[g] : 0.0;
                    ra_v[r][j] = (pa[r] && g < G) ? pa[r][g] - m : 0.0;
                    rb_v[r][j] = (pb[r] && g < G) ? pb[r][g] - m : 0.0;
                }
            }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int r = 0; r < RPT; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                As[(lrow + 64 * r) * LDR + lseg + j] = ra_v[r][j];
                Bs[(lrow + 64 * r) * LDR + lseg + j] = rb_v[r][j];
            }
    };
    fetch(0);
    park();
    __syncthreads();
    for (int k0 = 0; k0 < G; k0 += KC) {
        const bool more = k0 + KC < G;
        if (more) fetch(k0 + KC);
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) {
            const int k = kk + (lane >> 4), r = lane & 15;
            double av[WM], bv[WM];
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                av[a] = As[(wr * 16 * WM + 16 * a + r) * LDR + k];
                bv[a] = Bs[(wc * 16 * WM + 16 * a + r) * LDR + k];
            }
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();   // every wavefront is done with this stage's tiles
        if (more) park();
        __syncthreads();
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WM; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = bi * DT + wr * 16 * WM + a * 16 + (lane >> 4) + 4 * reg;
                const int col = bj * DT + wc * 16 * WM + b * 16 + (lane & 15);
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
    // 128-cell tiles when they still fill the chip twice over, 64-cell tiles otherwise
    const int64_t nt128 = (n + 127) / 128;
    const bool big = nt128 * (nt128 + 1) / 2 >= 2 * (int64_t)num_cus();
    const int DT = big ? 128 : 64;
    const int64_t nt = (n + DT - 1) / DT;
    const int64_t tiles = nt * (nt + 1) / 2;
    if (tiles > 0x7fffffff) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "too many cells for one distance matrix");
    {
        KernelTimer kt("cell_distances_gram", stream);
        const size_t lds = (size_t)2 * DT * LDR * sizeof(double);
        if (big) {
            static DeviceOnce once;
            if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(gram_tiles_kernel<4>), 80 * 1024, once)) return rc;
            hipLaunchKernelGGL(gram_tiles_kernel<4>, dim3((unsigned)tiles), dim3(256), lds, stream, x, G, idx_dev, n, mean_dev, out);
        } else {
            hipLaunchKernelGGL(gram_tiles_kernel<2>, dim3((unsigned)tiles), dim3(256), lds, stream, x, G, idx_dev, n, mean_dev, out);
        }
    }
    hipLaunchKernelGGL(gram_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, out, n, diag_dev);
    int64_t blocks = ((int64_t)n * n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(gram_to_dist_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, n, diag_dev);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

}  // namespace icnv
