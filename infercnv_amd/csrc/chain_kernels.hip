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
//           value-binned histogram select in LDS (2048 bins between the
//           cell's min and max, refined until <= 1024 candidates, then ranked)
//   phase 3 (S layout again, via LDS): step 12, step 14 (2^x), step 22 denoise,
//           coalesced 16-B stores -- or, in the statistics modes used by the
//           reference rounds, per-gene sums / per-cell (sum, sd).
//
// L is odd so that chunk-strided 8-byte LDS accesses are bank-conflict free
// (lane stride 2L dwords, gcd(2L, 64) = 2).
#include "icnv_internal.h"

namespace icnv {

namespace {

constexpr int NB_HIST = 2048;   // histogram bins of the median select
constexpr int CAND_CAP = 1024;  // candidates ranked directly (aliases the histogram)
constexpr int MAX_CHR = 510;
constexpr int MAX_LEVELS = 80;

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
__device__ inline uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// All-thread block reductions; every thread combines the per-wave partials in
// the same fixed order, so the result is deterministic and uniform.
template <int NT>
__device__ inline double block_sum(double v, double *red) {
    constexpr int NW = NT / 64;
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r += red[w];
    __syncthreads();
    return r;
}
template <int NT>
__device__ inline void block_minmax(double &lo, double &hi, double *red) {
    constexpr int NW = NT / 64;
    lo = wave_min(lo);
    hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = lo;
        red[NW + (threadIdx.x >> 6)] = hi;
    }
    __syncthreads();
    double a = red[0], b = red[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        a = fmin(a, red[w]);
        b = fmax(b, red[NW + w]);
    }
    __syncthreads();
    lo = a;
    hi = b;
}
template <int NT>
__device__ inline uint32_t block_sum_u32(uint32_t v, uint32_t *red) {
    constexpr int NW = NT / 64;
    v = wave_sum_u32(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t r = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) r += red[w];
    __syncthreads();
    return r;
}
// Exclusive prefix sum over the block's threads (in thread order).
template <int NT>
__device__ inline uint32_t block_excl_scan_u32(uint32_t v, uint32_t *red) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
    }
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) base += (w < wave) ? red[w] : 0u;
    __syncthreads();
    return base + inc - v;
}

template <int VEC>
struct VecT;
template <>
struct VecT<1> {
    using type = double;
};
template <>
struct VecT<2> {
    using type = double2;
};

template <int VEC>
__device__ inline void load_vec(const double *p, double (&d)[VEC]) {
    if constexpr (VEC == 2) {
        const double2 t = *reinterpret_cast<const double2 *>(p);
        d[0] = t.x;
        d[1] = t.y;
    } else {
        d[0] = *p;
    }
}
template <int VEC>
__device__ inline void store_vec(double *p, const double (&d)[VEC]) {
    if constexpr (VEC == 2) {
        *reinterpret_cast<double2 *>(p) = make_double2(d[0], d[1]);
    } else {
        *p = d[0];
    }
}

// .subtract_expr (R/inferCNV_ops.R:1742-1786): strict comparisons, values
// between the bounds become 0; without bounds lo == hi == mean of group means.
__device__ inline double subtract_ref(double x, double lo, double hi, int use_bounds) {
    if (use_bounds) {
        double o = 0.0;
        if (x > hi) o = x - hi;
        if (x < lo) o = x - lo;
        return o;
    }
    return x - lo;
}

template <int NT, int LMAX, int VEC, int MODE>
__global__ void __launch_bounds__(NT) chain_kernel(const ChainArgs a) {
    constexpr int NS = (LMAX + VEC - 1) / VEC;  // S-layout slots per thread
    constexpr int BPT = NB_HIST / NT;           // histogram bins per thread in the scan
    static_assert(NB_HIST % NT == 0, "bins must divide evenly");
    static_assert(LMAX % 2 == 1, "chunk length must be odd (LDS banking)");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *buf = reinterpret_cast<double *>(smem);                 // [NT*LMAX]
    uint32_t *hist = reinterpret_cast<uint32_t *>(buf + NT * LMAX);  // [NB_HIST]
    double *cand = reinterpret_cast<double *>(hist);                 // [CAND_CAP] aliases hist
    double *red = reinterpret_cast<double *>(hist + NB_HIST);        // [64]
    uint32_t *redu = reinterpret_cast<uint32_t *>(red + 64);         // [32]
    int32_t *sel = reinterpret_cast<int32_t *>(redu + 32);           // [8]
    double *seld = reinterpret_cast<double *>(sel + 8);              // [2]
    int32_t *s_chr = reinterpret_cast<int32_t *>(seld + 2);          // [n_chr+1]

    const int t = threadIdx.x;
    const uint32_t G = (uint32_t)a.G;
    const int Gi = a.G;
    const int T = a.T;
    const uint32_t mask = a.mask;
    const bool use_lds = (mask & (ICNV_ST_SMOOTH | ICNV_ST_CENTER)) != 0;
    const bool do_smooth = (mask & ICNV_ST_SMOOTH) && T >= 1;

    for (int i = t; i <= a.n_chr; i += NT) s_chr[i] = a.chr_start[i];
    __syncthreads();

    // chromosome of this thread's chunk start (same for every cell)
    const int c0 = t * LMAX;
    int k0 = 0;
    if (c0 < Gi) {
        int lo = 0, hi = a.n_chr - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (s_chr[mid] <= c0) lo = mid; else hi = mid - 1;
        }
        k0 = lo;
    }

    double acc[NS][VEC];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[s][v] = 0.0;

    double xin[NS][VEC];
    // VEC == 2 only when G is even, and slot starts are multiples of VEC, so a
    // vector access never straddles G (nor the NT*LMAX LDS buffer).
    auto load_cell = [&](int it) {
        const int64_t col = a.cells ? (int64_t)a.cells[it] : (int64_t)it;
        const double *src = a.in + col * (int64_t)G;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint32_t g = VEC * ((uint32_t)t + NT * s);
            if (g < G) {
                load_vec<VEC>(src + g, xin[s]);
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) xin[s][v] = 0.0;
            }
        }
    };

    int it = blockIdx.x;
    if (it < a.n_cells) load_cell(it);

    for (; it < a.n_cells; it += gridDim.x) {
        const int64_t col = a.cells ? (int64_t)a.cells[it] : (int64_t)it;
        double w[NS][VEC];

        // ---------------- phase 1: steps 8, 9 in the S layout ----------------
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint32_t g = VEC * ((uint32_t)t + NT * s);
            double lo1[VEC], hi1[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) { lo1[v] = 0.0; hi1[v] = 0.0; }
            if ((mask & ICNV_ST_SUBTRACT_REF_1) && g < G) {
                load_vec<VEC>(a.b1 + g, lo1);
                load_vec<VEC>(a.b1 + G + g, hi1);
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                double x = xin[s][v];
                if (mask & ICNV_ST_SUBTRACT_REF_1) x = subtract_ref(x, lo1[v], hi1[v], a.use_bounds);
                if (mask & ICNV_ST_MAX_THRESH) {  // R/inferCNV_ops.R:2974-2975
                    if (x > a.max_thresh) x = a.max_thresh;
                    if (x < -a.max_thresh) x = -a.max_thresh;
                }
                w[s][v] = x;
            }
        }

        if (use_lds) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t g = VEC * ((uint32_t)t + NT * s);
                if (g < (uint32_t)(NT * LMAX)) store_vec<VEC>(buf + g, w[s]);
            }
        }
        // prefetch the next cell while this one is processed
        {
            const int nx = it + gridDim.x;
            if (nx < a.n_cells) load_cell(nx);
        }

        if (use_lds) {
            __syncthreads();
            // ---------------- phase 2: steps 10, 11 in the chunk layout ----------------
            double r[LMAX];
            if (do_smooth) {
                int i = c0, k = k0;
                int cs = s_chr[k], ce = s_chr[k + 1];
                bool fresh = true;
                double A = 0.0, Lb = 0.0, Rb = 0.0;
                const int full = (T + 1) * (T + 1);
#pragma unroll
                for (int q = 0; q < LMAX; ++q, ++i) {
                    double o = 0.0;
                    if (i < Gi) {
                        if (i >= ce) {
                            do { ++k; cs = ce; ce = s_chr[k + 1]; } while (i >= ce);
                            fresh = true;
                        }
                        if (ce - cs <= 1) {  // R/inferCNV_ops.R:2417: single-gene chr untouched
                            o = buf[i];
                        } else {
                            if (fresh) {
                                // direct init: A = sum w(d) x[i+d], Lb = sum x[i-T..i], Rb = sum x[i+1..i+T+1]
                                A = 0.0; Lb = 0.0; Rb = 0.0;
                                for (int d = -T; d <= T + 1; ++d) {
                                    const int j = i + d;
                                    const double xv = (j >= cs && j < ce) ? buf[j] : 0.0;
                                    const int ad = d < 0 ? -d : d;
                                    A += (double)(T + 1 - ad) * xv;   // weight 0 at d = T+1
                                    if (d <= 0) Lb += xv; else Rb += xv;
                                }
                                fresh = false;
                            }
                            int rl = T - (i - cs); rl = rl > 0 ? rl : 0;
                            int rr = T - (ce - 1 - i); rr = rr > 0 ? rr : 0;
                            const int den = full - ((rl * (rl + 1)) >> 1) - ((rr * (rr + 1)) >> 1);
                            o = A / (double)den;
                            // slide to i+1
                            const int j1 = i + 1, jr = i + T + 2, jl = i - T;
                            const double x1 = (j1 < ce) ? buf[j1] : 0.0;
                            const double xr = (jr < ce) ? buf[jr] : 0.0;
                            const double xl = (jl >= cs) ? buf[jl] : 0.0;
                            A += Rb - Lb;
                            Rb += xr - x1;
                            Lb += x1 - xl;
                        }
                    }
                    r[q] = o;
                }
            } else {
#pragma unroll
                for (int q = 0; q < LMAX; ++q) r[q] = (c0 + q < Gi) ? buf[c0 + q] : 0.0;
            }

            if (mask & ICNV_ST_CENTER) {
                double center;
                if (mask & ICNV_ST_CENTER_MEAN) {
                    double s = 0.0;
#pragma unroll
                    for (int q = 0; q < LMAX; ++q) s += (c0 + q < Gi) ? r[q] : 0.0;
                    center = block_sum<NT>(s, red) / (double)G;
                } else {
                    // ---- exact median over the G values (R/inferCNV_ops.R:2098) ----
                    double lo = __builtin_inf(), hi = -__builtin_inf();
#pragma unroll
                    for (int q = 0; q < LMAX; ++q)
                        if (c0 + q < Gi) { lo = fmin(lo, r[q]); hi = fmax(hi, r[q]); }
                    block_minmax<NT>(lo, hi, red);
                    if (!(lo < hi)) {
                        center = lo;
                    } else {
                        const int target = (Gi - 1) >> 1;  // 0-based rank of the lower middle
                        uint64_t member = 0;
#pragma unroll
                        for (int q = 0; q < LMAX; ++q)
                            if (c0 + q < Gi) member |= (1ull << q);
                        int base = 0;
                        double vlo = lo;
                        for (int level = 0; level < MAX_LEVELS; ++level) {
                            const double scale = (double)NB_HIST / (hi - lo);
#pragma unroll
                            for (int b = 0; b < BPT; ++b) hist[t * BPT + b] = 0u;
                            __syncthreads();
                            auto bin_of = [&](double x) -> int {
                                if (x == lo) return 0;
                                const double tt = (x - lo) * scale;
                                return (tt >= (double)(NB_HIST - 1)) ? (NB_HIST - 1) : (int)tt;
                            };
#pragma unroll
                            for (int q = 0; q < LMAX; ++q)
                                if ((member >> q) & 1ull) atomicAdd(&hist[bin_of(r[q])], 1u);
                            __syncthreads();
                            uint32_t own[BPT];
                            uint32_t mine = 0;
#pragma unroll
                            for (int b = 0; b < BPT; ++b) { own[b] = hist[t * BPT + b]; mine += own[b]; }
                            uint32_t before = block_excl_scan_u32<NT>(mine, redu);
                            const uint32_t rel = (uint32_t)(target - base);
                            if (rel >= before && rel < before + mine) {
#pragma unroll
                                for (int b = 0; b < BPT; ++b) {
                                    if (rel >= before && rel < before + own[b]) {
                                        sel[0] = t * BPT + b;
                                        sel[1] = (int)before;
                                        sel[2] = (int)own[b];
                                    }
                                    before += own[b];
                                }
                            }
                            if (t == 0) sel[3] = 0;
                            __syncthreads();
                            const int sbin = sel[0], sbefore = sel[1], scnt = sel[2];
                            if (scnt <= CAND_CAP) {
                                // collect the bin's members (hist is dead now -> cand aliases it)
                                __syncthreads();
#pragma unroll
                                for (int q = 0; q < LMAX; ++q)
                                    if (((member >> q) & 1ull) && bin_of(r[q]) == sbin) {
                                        const int p = atomicAdd(&sel[3], 1);
                                        cand[p] = r[q];
                                    }
                                __syncthreads();
                                const int want = target - base - sbefore;
                                for (int ci = t; ci < scnt; ci += NT) {
                                    const double cv = cand[ci];
                                    int less = 0;
                                    for (int cj = 0; cj < scnt; ++cj) {
                                        const double o = cand[cj];
                                        less += (o < cv || (o == cv && cj < ci)) ? 1 : 0;
                                    }
                                    if (less == want) seld[0] = cv;
                                }
                                __syncthreads();
                                vlo = seld[0];
                                __syncthreads();
                                break;
                            }
                            // refine inside the selected bin
                            base += sbefore;
                            double nlo = __builtin_inf(), nhi = -__builtin_inf();
#pragma unroll
                            for (int q = 0; q < LMAX; ++q) {
                                if ((member >> q) & 1ull) {
                                    if (bin_of(r[q]) == sbin) { nlo = fmin(nlo, r[q]); nhi = fmax(nhi, r[q]); }
                                    else member &= ~(1ull << q);
                                }
                            }
                            block_minmax<NT>(nlo, nhi, red);
                            lo = nlo; hi = nhi;
                            if (!(lo < hi)) { vlo = lo; break; }
                        }
                        if (G & 1) {
                            center = vlo;
                        } else {
                            // upper middle: same value if duplicates cover rank target+1,
                            // otherwise the smallest value above vlo
                            uint32_t le = 0;
                            double above = __builtin_inf();
#pragma unroll
                            for (int q = 0; q < LMAX; ++q)
                                if (c0 + q < Gi) {
                                    le += (r[q] <= vlo) ? 1u : 0u;
                                    if (r[q] > vlo) above = fmin(above, r[q]);
                                }
                            const uint32_t cle = block_sum_u32<NT>(le, redu);
                            double dummy = -__builtin_inf();
                            block_minmax<NT>(above, dummy, red);
                            const double vhi = (cle > (uint32_t)(target + 1)) ? vlo : above;
                            center = (vlo + vhi) * 0.5;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < LMAX; ++q) r[q] -= center;
            }

            __syncthreads();  // every halo read of buf is done
#pragma unroll
            for (int q = 0; q < LMAX; ++q) buf[c0 + q] = r[q];
            __syncthreads();
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t g = VEC * ((uint32_t)t + NT * s);
                if (g < (uint32_t)(NT * LMAX)) load_vec<VEC>(buf + g, w[s]);
                else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) w[s][v] = 0.0;
                }
            }
            __syncthreads();  // buf may be overwritten by the next cell's phase 1
        }

        // ---------------- phase 3: steps 12, 14, 22 in the S layout ----------------
        double mu = 0.0, lo_d = 0.0, hi_d = 0.0;
        if (mask & ICNV_ST_DENOISE) {
            mu = a.denoise[0];
            const double sdv = a.denoise[1];
            lo_d = mu - sdv;
            hi_d = mu + sdv;
        }
        double *dst = (MODE == MODE_APPLY) ? a.out + col * (int64_t)G : nullptr;
        double *dpre = (MODE == MODE_APPLY && a.pre_out) ? a.pre_out + col * (int64_t)G : nullptr;
        double csum = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint32_t g = VEC * ((uint32_t)t + NT * s);
            if (g >= G) continue;
            double lo2[VEC], hi2[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) { lo2[v] = 0.0; hi2[v] = 0.0; }
            if (mask & ICNV_ST_SUBTRACT_REF_2) {
                load_vec<VEC>(a.b2 + g, lo2);
                load_vec<VEC>(a.b2 + G + g, hi2);
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                double x = w[s][v];
                if (mask & ICNV_ST_SUBTRACT_REF_2) x = subtract_ref(x, lo2[v], hi2[v], a.use_bounds);
                if (mask & ICNV_ST_INVERT_LOG2) x = exp2(x);  // R/inferCNV_ops.R:2818
                w[s][v] = x;
            }
            if (MODE == MODE_APPLY) {
                if (dpre) store_vec<VEC>(dpre + g, w[s]);
                if (mask & ICNV_ST_DENOISE) {  // strict bounds, R/inferCNV_ops.R:2335
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        if (w[s][v] > lo_d && w[s][v] < hi_d) w[s][v] = mu;
                }
                store_vec<VEC>(dst + g, w[s]);
            } else if (MODE == MODE_GENE_SUMS) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[s][v] += w[s][v];
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) csum += w[s][v];
            }
        }
        if (MODE == MODE_CELL_STATS) {
            // per-cell sum and sample sd over genes (two-pass, R's sd())
            const double tot = block_sum<NT>(csum, red);
            const double mean = tot / (double)G;
            double ss = 0.0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t g = VEC * ((uint32_t)t + NT * s);
                if (g < G) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) { const double d = w[s][v] - mean; ss += d * d; }
                }
            }
            const double sst = block_sum<NT>(ss, red);
            if (t == 0) {
                a.cell_stats[2 * (int64_t)it] = tot;
                a.cell_stats[2 * (int64_t)it + 1] = sqrt(sst / (double)(G - 1));
            }
        }
    }

    if (MODE == MODE_GENE_SUMS) {
        double *dst = a.partial + (int64_t)blockIdx.x * G;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint32_t g = VEC * ((uint32_t)t + NT * s);
            if (g < G) store_vec<VEC>(dst + g, acc[s]);
        }
    }
}

// sums[g] = sum_b partial[b*G + g] in fixed order; also stores the cell count.
__global__ void reduce_partials_kernel(const double *partial, int nblk, int G, double *out, double count,
                                       double *count_out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * G + g];
        out[g] = s;
    }
    if (g == 0 && count_out) *count_out = count;
}

// .get_normal_gene_mean_bounds + the min/max / mean-of-means of .subtract_expr
// (R/inferCNV_ops.R:1708-1735, 1750-1776).  sc = [G*n_grp sums | n_grp counts].
__global__ void bounds_from_sums_kernel(const double *sc, int G, int n_grp, int use_bounds, double *bounds) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double lo = 0.0, hi = 0.0, tot = 0.0;
    for (int r = 0; r < n_grp; ++r) {
        const double m = sc[(int64_t)r * G + g] / sc[(int64_t)n_grp * G + r];
        if (r == 0) { lo = m; hi = m; }
        else { lo = fmin(lo, m); hi = fmax(hi, m); }
        tot += m;
    }
    if (use_bounds) {
        bounds[g] = lo;
        bounds[G + g] = hi;
    } else {
        const double mm = tot / (double)n_grp;
        bounds[g] = mm;
        bounds[G + g] = mm;
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
        block_minmax<256>(lo, hi, red);
        if (threadIdx.x == 0) { out[2 * c] = lo; out[2 * c + 1] = hi; }
    }
}

template <int NT, int LMAX>
size_t chain_lds_bytes(int n_chr) {
    return (size_t)NT * LMAX * 8 + NB_HIST * 4 + 64 * 8 + 32 * 4 + 8 * 4 + 2 * 8 + (size_t)(n_chr + 1) * 4 + 16;
}

template <int NT, int LMAX, int VEC, int MODE>
int launch_chain_t(const ChainArgs &a, hipStream_t stream) {
    const size_t lds = chain_lds_bytes<NT, LMAX>(a.n_chr);
    static bool attr_set = false;
    if (!attr_set) {
        ICNV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<NT, LMAX, VEC, MODE>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    int blocks_per_cu = (int)((160 * 1024) / lds);
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (blocks_per_cu * NT > 2048) blocks_per_cu = 2048 / NT;
    int grid = num_cus() * blocks_per_cu;
    if (MODE == MODE_GENE_SUMS && grid > 256) grid = 256;
    if (grid > a.n_cells) grid = a.n_cells;
    if (grid < 1) return ICNV_OK;
    KernelTimer kt(MODE == MODE_APPLY ? "chain_apply" : (MODE == MODE_GENE_SUMS ? "chain_gene_sums" : "chain_cell_stats"),
                   stream);
    hipLaunchKernelGGL((chain_kernel<NT, LMAX, VEC, MODE>), dim3(grid), dim3(NT), lds, stream, a);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

template <int NT, int LMAX, int VEC>
int launch_chain_m(const ChainArgs &a, int mode, hipStream_t stream) {
    switch (mode) {
        case MODE_APPLY: return launch_chain_t<NT, LMAX, VEC, MODE_APPLY>(a, stream);
        case MODE_GENE_SUMS: return launch_chain_t<NT, LMAX, VEC, MODE_GENE_SUMS>(a, stream);
        default: return launch_chain_t<NT, LMAX, VEC, MODE_CELL_STATS>(a, stream);
    }
}

template <int NT, int LMAX>
int launch_chain_v(const ChainArgs &a, int mode, hipStream_t stream) {
    if ((a.G & 1) == 0) return launch_chain_m<NT, LMAX, 2>(a, mode, stream);
    return launch_chain_m<NT, LMAX, 1>(a, mode, stream);
}

}  // namespace

constexpr int CHAIN_NT = 512;

int chain_max_genes() { return CHAIN_NT * 37; }

// Number of blocks the GENE_SUMS mode will use for n cells (partial buffer sizing).
int launch_chain(const ChainArgs &a, int mode, hipStream_t stream) {
    if (a.n_chr > MAX_CHR) ICNV_FAIL(ICNV_ERR_UNSUPPORTED, "more than 510 chromosomes/contigs");
    const int G = a.G;
    if (G <= CHAIN_NT * 7) return launch_chain_v<CHAIN_NT, 7>(a, mode, stream);
    if (G <= CHAIN_NT * 21) return launch_chain_v<CHAIN_NT, 21>(a, mode, stream);
    if (G <= CHAIN_NT * 37) return launch_chain_v<CHAIN_NT, 37>(a, mode, stream);
    ICNV_FAIL(ICNV_ERR_UNSUPPORTED,
              "fused smoothing chain supports at most 18944 genes per matrix (LDS-resident cell vector)");
}

int launch_reduce_partials(const double *partial, int nblk, int32_t G, double *out, double count,
                           double *count_out, hipStream_t stream) {
    KernelTimer kt("reduce_partials", stream);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((G + 255) / 256), dim3(256), 0, stream, partial, nblk, G, out,
                       count, count_out);
    ICNV_HIP(hipGetLastError());
    return ICNV_OK;
}

int launch_bounds_from_sums(const double *sums_counts, int32_t G, int32_t n_grp, int32_t use_bounds,
                            double *bounds, hipStream_t stream) {
    KernelTimer kt("bounds_from_sums", stream);
    hipLaunchKernelGGL(bounds_from_sums_kernel, dim3((G + 255) / 256), dim3(256), 0, stream, sums_counts, G, n_grp,
                       use_bounds, bounds);
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
