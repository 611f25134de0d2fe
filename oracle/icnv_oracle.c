/*
 * icnv_oracle.c -- plain-C CPU restatement of inferCNV's smoothing chain,
 * HMM Viterbi and 2-D median filter.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker and the timed
 * "cpu_baseline" (kind = "port": R is not installed in the image, so the R
 * reference itself cannot be timed).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product library
 * (infercnv_amd/csrc) never links or calls it.
 *
 * Written independently of the HIP kernels: same arithmetic *specification*
 * (DESIGN.md "Arithmetic spec"), different code.  Build with
 *   gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC   (see oracle/Makefile)
 * -ffp-contract=off matters: the Viterbi state calls must be bit-identical to
 * the GPU's, so no FMA contraction on either side.
 *
 * Matrices are column-major genes x cells, x[g + G*c]  (R/inferCNV.R:18), all
 * indices 0-based.  Genes of one chromosome are contiguous
 * (.order_reduce, R/inferCNV.R:407): chr_start[] holds n_chr+1 offsets.
 *
 * Pinning: smoothing chain pinned by data/infercnv_object_example.rda
 * (tests/golden); HMM + median filter: PARITY UNPINNED in the reference
 * (no known-answer test exists there) -- cross-checked against the NumPy
 * restatement oracle/oracle_np.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef long double ld_t; /* R accumulates sum()/mean() in long double */

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ */
/* base-R primitives                                                   */
/* ------------------------------------------------------------------ */
/* base::mean on doubles (R src/main/summary.c): LDOUBLE sum / n, then one
 * refinement pass. */
static double r_mean_strided(const double *x, int64_t n, int64_t stride) {
    ld_t s = 0;
    for (int64_t i = 0; i < n; i++) s += x[i * stride];
    s /= (ld_t)n;
    ld_t t = 0;
    for (int64_t i = 0; i < n; i++) t += (x[i * stride] - s);
    s += t / (ld_t)n;
    return (double)s;
}

static double r_mean_idx(const double *x, const int32_t *idx, int64_t n, int64_t stride) {
    ld_t s = 0;
    for (int64_t i = 0; i < n; i++) s += x[(int64_t)idx[i] * stride];
    s /= (ld_t)n;
    ld_t t = 0;
    for (int64_t i = 0; i < n; i++) t += (x[(int64_t)idx[i] * stride] - s);
    s += t / (ld_t)n;
    return (double)s;
}

/* base::rowMeans on doubles (base R src/main/array.c, do_colsum, OP == 3, na.rm = FALSE; outside /root/reference):
 * one LDOUBLE accumulator per row, the columns added in order, divided by the column count in LDOUBLE, then cast.
 * No refinement pass (that is mean()).  On x86-64 LDOUBLE is the 80-bit x87 format: the last bit of the result is
 * platform-dependent in R itself ("group means"). */
static double r_rowmean_idx(const double *x, const int32_t *idx, int64_t n, int64_t stride) {
    ld_t s = 0;
    for (int64_t i = 0; i < n; i++) s += x[(int64_t)idx[i] * stride];
    s /= (ld_t)n;
    return (double)s;
}

/* k-th smallest (0-based) by quickselect on a scratch copy; also returns the
 * (k+1)-th through *next when want_next (min of the upper partition). */
static double select_kth(double *a, int64_t n, int64_t k) {
    int64_t lo = 0, hi = n - 1;
    while (lo < hi) {
        double p = a[lo + (hi - lo) / 2];
        int64_t i = lo, j = hi;
        while (i <= j) {
            while (a[i] < p) i++;
            while (a[j] > p) j--;
            if (i <= j) { double t = a[i]; a[i] = a[j]; a[j] = t; i++; j--; }
        }
        if (k <= j) hi = j;
        else if (k >= i) lo = i;
        else break;
    }
    return a[k];
}

/* stats::median: odd n -> middle; even n -> mean of the two middle values. */
static double r_median_scratch(double *a, int64_t n) {
    int64_t h = n / 2;
    if (n & 1) return select_kth(a, n, h);
    double hi = select_kth(a, n, h);
    double lo = a[0];
    for (int64_t i = 1; i < h; i++) if (a[i] > lo) lo = a[i]; /* max of lower part */
    return (lo + hi) * 0.5;
}

/* ------------------------------------------------------------------ */
/* steps 3-4: normalize_counts_by_seq_depth + log2xplus1               */
/* (R/inferCNV_ops.R:3064-3111, 2756-2769)                             */
/* ------------------------------------------------------------------ */
/* in place; normalize_factor NaN -> median(colSums).  Returns the factor used. */
double orc_normalize_log2(double *x, int64_t G, int64_t C, double normalize_factor, int32_t do_norm, int32_t do_log) {
    double factor = normalize_factor;
    double *cs = (double *)malloc(sizeof(double) * (C > 0 ? C : 1));
    if (do_norm) {
        for (int64_t c = 0; c < C; c++) { /* colSums: long double accumulation like R */
            ld_t s = 0;
            for (int64_t g = 0; g < G; g++) s += x[g + G * c];
            cs[c] = (double)s;
        }
        if (isnan(factor)) {
            double *tmp = (double *)malloc(sizeof(double) * C);
            memcpy(tmp, cs, sizeof(double) * C);
            factor = r_median_scratch(tmp, C);
            free(tmp);
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < C; c++)
        for (int64_t g = 0; g < G; g++) {
            double v = x[g + G * c];
            if (do_norm) v = v / cs[c] * factor;
            if (do_log) v = log2(v + 1.0);
            x[g + G * c] = v;
        }
    free(cs);
    return factor;
}

/* ------------------------------------------------------------------ */
/* A.1 reference subtraction  (R/inferCNV_ops.R:1678-1786)             */
/* ------------------------------------------------------------------ */
/* means[g + G*r] = mean over cells of ref group r (R/inferCNV_ops.R:1708-1735). */
int orc_ref_group_means(const double *x, int64_t G, int64_t C, const int32_t *ref_idx,
                        const int32_t *ref_off, int32_t n_grp, int32_t inv_log, double *means) {
    (void)C;
    for (int32_t r = 0; r < n_grp; r++) {
        const int32_t *idx = ref_idx + ref_off[r];
        int64_t n = ref_off[r + 1] - ref_off[r];
        if (n <= 0) return 1;
#pragma omp parallel for schedule(static)
        for (int64_t g = 0; g < G; g++) {
            if (!inv_log) {
                means[g + G * r] = r_mean_idx(x + g, idx, n, G);
            } else { /* log2(mean(2^x - 1) + 1)  (:1714-1717) */
                ld_t s = 0;
                for (int64_t i = 0; i < n; i++) s += (exp2(x[g + G * (int64_t)idx[i]]) - 1.0);
                s /= (ld_t)n;
                ld_t t = 0;
                for (int64_t i = 0; i < n; i++) t += ((exp2(x[g + G * (int64_t)idx[i]]) - 1.0) - s);
                s += t / (ld_t)n;
                means[g + G * r] = log2((double)s + 1.0);
            }
        }
    }
    return 0;
}

/* .subtract_expr (R/inferCNV_ops.R:1742-1786), in place. */
void orc_subtract_ref(double *x, int64_t G, int64_t C, const double *means, int32_t n_grp,
                      int32_t use_bounds) {
    double *lo = (double *)malloc(sizeof(double) * G), *hi = (double *)malloc(sizeof(double) * G),
           *mm = (double *)malloc(sizeof(double) * G);
    for (int64_t g = 0; g < G; g++) {
        double l = means[g], h = means[g];
        for (int32_t r = 1; r < n_grp; r++) {
            double m = means[g + G * r];
            if (m < l) l = m;
            if (m > h) h = m;
        }
        lo[g] = l; hi[g] = h;
        mm[g] = r_mean_strided(means + g, n_grp, G);
    }
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < C; c++) {
        double *col = x + G * c;
        for (int64_t g = 0; g < G; g++) {
            double v = col[g];
            if (use_bounds) {
                double o = 0.0;
                if (v > hi[g]) o = v - hi[g];
                if (v < lo[g]) o = v - lo[g];
                col[g] = o;
            } else {
                col[g] = v - mm[g];
            }
        }
    }
    free(lo); free(hi); free(mm);
}

/* apply_max_threshold_bounds (R/inferCNV_ops.R:2970-2983). */
void orc_clamp(double *x, int64_t G, int64_t C, double thr) {
    int64_t n = G * C;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        if (x[i] > thr) x[i] = thr;
        if (x[i] < -thr) x[i] = -thr;
    }
}

/* .get_average_bounds (R/inferCNV_ops.R:2733-2742): mean over cells of the
 * per-cell min and max. */
void orc_average_bounds(const double *x, int64_t G, int64_t C, double *out2) {
    double *mn = (double *)malloc(sizeof(double) * C), *mx = (double *)malloc(sizeof(double) * C);
    for (int64_t c = 0; c < C; c++) {
        double a = x[G * c], b = a;
        for (int64_t g = 1; g < G; g++) {
            double v = x[g + G * c];
            if (v < a) a = v;
            if (v > b) b = v;
        }
        mn[c] = a; mx[c] = b;
    }
    out2[0] = r_mean_strided(mn, C, 1);
    out2[1] = r_mean_strided(mx, C, 1);
    free(mn); free(mx);
}

/* ------------------------------------------------------------------ */
/* A.2 pyramid smoothing (R/inferCNV_ops.R:2406-2532, 2640-2661)       */
/* ------------------------------------------------------------------ */
/* One cell, one chromosome: v[0..n) -> out[0..n), the reference's own
 * evaluation order (filter() interior in double, ends via long-double sum). */
static void smooth_helper(const double *v, double *out, int64_t n, int32_t W, const double *numer,
                          const double *filt) {
    int32_t T = (W - 1) / 2;
    double full_den = (double)T * (double)T + (double)W;
    memcpy(out, v, sizeof(double) * n);
    if (n >= W) { /* .smooth_center_helper: stats::filter(vals, filt, sides=2) */
        for (int64_t i = T; i < n - T; i++) {
            double z = 0.0;
            for (int32_t j = 0; j < W; j++) z += filt[j] * v[i + T - j];
            out[i] = z;
        }
    }
    int64_t it_range = (n > W) ? T : (n + 1) / 2;
    for (int64_t tail_end = 1; tail_end <= it_range; tail_end++) {
        int64_t end_tail = n - tail_end + 1;
        int64_t d_left = tail_end - 1;
        int64_t d_right = n - tail_end;
        if (d_right > T) d_right = T;
        int64_t r_left = T - d_left, r_right = T - d_right;
        double den = full_den - (double)(r_left * (r_left + 1)) / 2.0 - (double)(r_right * (r_right + 1)) / 2.0;
        int64_t len = tail_end + d_right;
        const double *nr = numer + (T - d_left); /* numer[(T+1-d_left)..(T+1+d_right)] 1-based */
        ld_t sl = 0, sr = 0;
        for (int64_t q = 0; q < len; q++) sl += (ld_t)(v[q] * nr[q]);
        const double *rv = v + (end_tail - d_right - 1);
        for (int64_t q = 0; q < len; q++) sr += (ld_t)(rv[q] * nr[len - 1 - q]);
        out[tail_end - 1] = (double)sl / den;
        out[end_tail - 1] = (double)sr / den;
    }
}

void orc_smooth_by_chr(double *x, int64_t G, int64_t C, const int32_t *chr_start, int32_t n_chr,
                       int32_t W) {
    if (W < 2) return; /* R/inferCNV_ops.R:2444-2447 */
    int32_t T = (W - 1) / 2;
    double *numer = (double *)malloc(sizeof(double) * W), *filt = (double *)malloc(sizeof(double) * W);
    double full_den = (double)T * (double)T + (double)W;
    for (int32_t j = 0; j < W; j++) {
        int32_t d = j - T; if (d < 0) d = -d;
        numer[j] = (double)(T + 1 - d);
        filt[j] = numer[j] / full_den;
    }
#pragma omp parallel
    {
        double *tmp = (double *)malloc(sizeof(double) * G);
#pragma omp for schedule(static)
        for (int64_t c = 0; c < C; c++) {
            double *col = x + G * c;
            for (int32_t k = 0; k < n_chr; k++) {
                int64_t s = chr_start[k], n = chr_start[k + 1] - s;
                if (n > 1) { /* :2417 */
                    smooth_helper(col + s, tmp, n, W, numer, filt);
                    memcpy(col + s, tmp, sizeof(double) * n);
                }
            }
        }
        free(tmp);
    }
    free(numer); free(filt);
}

/* .center_columns (R/inferCNV_ops.R:2094-2109): method 0 = median, 1 = mean. */
void orc_center(double *x, int64_t G, int64_t C, int32_t method) {
#pragma omp parallel
    {
        double *tmp = (double *)malloc(sizeof(double) * G);
#pragma omp for schedule(static)
        for (int64_t c = 0; c < C; c++) {
            double *col = x + G * c, m;
            if (method == 0) {
                memcpy(tmp, col, sizeof(double) * G);
                m = r_median_scratch(tmp, G);
            } else {
                m = r_mean_strided(col, G, 1);
            }
            for (int64_t g = 0; g < G; g++) col[g] -= m;
        }
        free(tmp);
    }
}

/* invert_log2 (R/inferCNV_ops.R:2814-2826). */
void orc_exp2(double *x, int64_t G, int64_t C) {
    int64_t n = G * C;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) x[i] = exp2(x[i]);
}

/* clear_noise_via_ref_mean_sd parameters (R/inferCNV_ops.R:2311-2318):
 * mu = mean(all ref values), s = mean_c(sd(x[,c])) * sd_amplifier. */
void orc_denoise_params(const double *x, int64_t G, int64_t C, const int32_t *ref_idx, int64_t n_ref,
                        double sd_amplifier, double *mu, double *s) {
    (void)C;
    ld_t tot = 0;
    for (int64_t i = 0; i < n_ref; i++) {
        const double *col = x + G * (int64_t)ref_idx[i];
        for (int64_t g = 0; g < G; g++) tot += col[g];
    }
    ld_t m = tot / (ld_t)(n_ref * G);
    ld_t t = 0;
    for (int64_t i = 0; i < n_ref; i++) {
        const double *col = x + G * (int64_t)ref_idx[i];
        for (int64_t g = 0; g < G; g++) t += (col[g] - m);
    }
    m += t / (ld_t)(n_ref * G);
    *mu = (double)m;
    double *sds = (double *)malloc(sizeof(double) * n_ref);
    for (int64_t i = 0; i < n_ref; i++) {
        const double *col = x + G * (int64_t)ref_idx[i];
        double cm = r_mean_strided(col, G, 1);
        ld_t v = 0;
        for (int64_t g = 0; g < G; g++) { ld_t d = (ld_t)col[g] - (ld_t)cm; v += d * d; }
        sds[i] = sqrt((double)(v / (ld_t)(G - 1)));
    }
    *s = r_mean_strided(sds, n_ref, 1) * sd_amplifier;
    free(sds);
}

/* clear_noise() centre (R/inferCNV_ops.R:2240-2246): mean of all ref values. */
double orc_mean_of_cells(const double *x, int64_t G, const int32_t *idx, int64_t n) {
    double mu, s;
    orc_denoise_params(x, G, 0, idx, n, 1.0, &mu, &s);
    return mu;
}

/* x <- center where center-hw < x < center+hw  (strict; :2270-2278, :2335). */
void orc_denoise_apply(double *x, int64_t G, int64_t C, double center, double hw) {
    int64_t n = G * C;
    double lo = center - hw, hi = center + hw;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
        if (x[i] > lo && x[i] < hi) x[i] = center;
}

/* Steps 8,9,10,11,12,14,22 of run() (R/inferCNV_ops.R:771-1589), in place on x.
 * max_thresh NaN -> skip step 9.  noise_filter NaN -> sd-based denoise.
 * pre_denoise (may be NULL) receives the step-14 matrix (the HMM's input). */
enum { ST8 = 1, ST9 = 2, ST10 = 4, ST11 = 8, ST12 = 16, ST14 = 32, ST22 = 64 };

int orc_smooth_chain(double *x, int64_t G, int64_t C, const int32_t *chr_start, int32_t n_chr,
                     const int32_t *ref_idx, const int32_t *ref_off, int32_t n_grp, int32_t W,
                     double max_thresh, int32_t use_bounds, double sd_amplifier, double noise_filter,
                     uint32_t stage_mask, double *pre_denoise, double *denoise_mu_s) {
    double *means = (double *)malloc(sizeof(double) * G * n_grp);
    if (stage_mask & ST8) {
        if (orc_ref_group_means(x, G, C, ref_idx, ref_off, n_grp, 0, means)) { free(means); return 1; }
        orc_subtract_ref(x, G, C, means, n_grp, use_bounds);
    }
    if ((stage_mask & ST9) && !isnan(max_thresh)) orc_clamp(x, G, C, max_thresh);
    if (stage_mask & ST10) orc_smooth_by_chr(x, G, C, chr_start, n_chr, W);
    if (stage_mask & ST11) orc_center(x, G, C, 0);
    if (stage_mask & ST12) {
        if (orc_ref_group_means(x, G, C, ref_idx, ref_off, n_grp, 0, means)) { free(means); return 1; }
        orc_subtract_ref(x, G, C, means, n_grp, use_bounds);
    }
    if (stage_mask & ST14) orc_exp2(x, G, C);
    if (pre_denoise) memcpy(pre_denoise, x, sizeof(double) * G * C);
    if (stage_mask & ST22) {
        double mu, s;
        int64_t n_ref = ref_off[n_grp];
        if (isnan(noise_filter)) {
            orc_denoise_params(x, G, C, ref_idx, n_ref, sd_amplifier, &mu, &s);
        } else {
            mu = orc_mean_of_cells(x, G, ref_idx, n_ref);
            s = noise_filter;
        }
        if (denoise_mu_s) { denoise_mu_s[0] = mu; denoise_mu_s[1] = s; }
        if (!(s == 0.0 && !isnan(noise_filter))) orc_denoise_apply(x, G, C, mu, s);
    }
    free(means);
    return 0;
}

/* ------------------------------------------------------------------ */
/* A.5 log and pnorm(q>=0, log.p=TRUE, lower.tail=FALSE)               */
/* ------------------------------------------------------------------ */
/* Natural log with a FIXED operation sequence so that gcc, NumPy and the GPU agree
 * bit for bit: table-driven (128 sub-intervals of [0.6875, 1.375), constants generated
 * by oracle/gen_log_table.py) with explicit, correctly rounded fma():
 *   x = 2^k z;  r = fma(z, invc, -1) (exact);  w = k*LN2HI + logc_hi (exact);
 *   hi = w + r;  lo = ((w - hi) + r) + (k*LN2LO + logc_lo);
 *   q = B0 + r(B1 + ... + r B6) by fma Horner;  result = hi + fma(r*r, q, lo).
 * Measured accuracy: <= 0.52 ulp (tests/test_oracle.py).  R itself calls the
 * platform libm log, which is implementation-defined at this level. */
#include "icnv_log_table.h"
static const struct { double invc, logc_hi, logc_lo; } orc_log_tab[ICNV_LOG_N] = ICNV_LOG_TABLE_INIT;

double orc_log(double x) {
    union { double d; uint64_t u; } b;
    b.d = x;
    uint64_t ix = b.u;
    if (ix - 0x0010000000000000ull >= 0x7fe0000000000000ull) { /* zero, subnormal, negative, inf, nan */
        if ((ix << 1) == 0) return -INFINITY;
        if (ix == 0x7ff0000000000000ull) return x;
        if ((ix >> 63) || (ix & 0x7ff0000000000000ull) == 0x7ff0000000000000ull) return NAN;
        b.d = x * 0x1p52; /* subnormal: scale up */
        ix = b.u - (52ull << 52);
    }
    const uint64_t tmp = ix - ICNV_LOG_OFF;
    const int i = (int)((tmp >> 45) & 127);
    const int64_t k = (int64_t)tmp >> 52;
    b.u = ix - (tmp & 0xfff0000000000000ull);
    const double z = b.d;
    const double r = fma(z, orc_log_tab[i].invc, -1.0);
    const double kd = (double)k;
    const double w = fma(kd, ICNV_LOG_LN2HI, orc_log_tab[i].logc_hi);
    const double hi = w + r;
    const double lo = ((w - hi) + r) + (kd * ICNV_LOG_LN2LO + orc_log_tab[i].logc_lo);
    const double r2 = r * r;
    double q = fma(r, ICNV_LOG_B6, ICNV_LOG_B5);
    q = fma(r, q, ICNV_LOG_B4);
    q = fma(r, q, ICNV_LOG_B3);
    q = fma(r, q, ICNV_LOG_B2);
    q = fma(r, q, ICNV_LOG_B1);
    q = fma(r, q, ICNV_LOG_B0);
    return hi + fma(r2, q, lo);
}

/* correctly rounded fma primitive for the NumPy restatement (NumPy has none) */
void orc_fma_array(const double *a, const double *b, const double *c, double *out, int64_t n) {
    for (int64_t i = 0; i < n; i++) out[i] = fma(a[i], b[i], c[i]);
}

void orc_log_array(const double *x, double *out, int64_t n) {
    for (int64_t i = 0; i < n; i++) out[i] = orc_log(x[i]);
}

/* log P(Z > y), y >= 0: pnorm_both()'s branches and operation order
 * (R src/nmath/pnorm.c, Cody 1969; call sites R/inferCNV_HMM.R:1129,1156). */
double orc_pnorm_log_upper(double y) {
    static const double a[5] = {2.2352520354606839287, 161.02823106855587881, 1067.6894854603709582,
                                18154.981253343561249, 0.065682337918207449113};
    static const double b[4] = {47.20258190468824187, 976.09855173777669322, 10260.932208618978205,
                                45507.789335026729956};
    static const double c[9] = {0.39894151208813466764, 8.8831497943883759412, 93.506656132177855979,
                                597.27027639480026226, 2494.5375852903726711, 6848.1904505362823326,
                                11602.651437647350124, 9842.7148383839780218, 1.0765576773720192317e-8};
    static const double d[8] = {22.266688044328115691, 235.38790178262499861, 1519.377599407554805,
                                6485.558298266760755, 18615.571640885098091, 34900.952721145977266,
                                38912.003286093271411, 19685.429676859990727};
    static const double p[6] = {0.21589853405795699, 0.1274011611602473639, 0.022235277870649807,
                                0.001421619193227893466, 2.9112874951168792e-5, 0.02307344176494017303};
    static const double q[5] = {1.28426009614491121, 0.468238212480865118, 0.0659881378689285515,
                                0.00378239633202758244, 7.29751555083966205e-5};
    double xnum, xden, tmp, xsq;
    if (y <= 0.67448975) {
        xsq = y * y;
        xnum = a[4] * xsq;
        xden = xsq;
        for (int i = 0; i < 3; i++) { xnum = (xnum + a[i]) * xsq; xden = (xden + b[i]) * xsq; }
        tmp = y * (xnum + a[3]) / (xden + b[3]);
        return orc_log(0.5 - tmp);
    }
    if (y <= 5.656854249492380195206754896838) {
        xnum = c[8] * y;
        xden = y;
        for (int i = 0; i < 7; i++) { xnum = (xnum + c[i]) * y; xden = (xden + d[i]) * y; }
        tmp = (xnum + c[7]) / (xden + d[7]);
    } else {
        xsq = 1.0 / (y * y);
        xnum = p[5] * xsq;
        xden = xsq;
        for (int i = 0; i < 4; i++) { xnum = (xnum + p[i]) * xsq; xden = (xden + q[i]) * xsq; }
        tmp = xsq * (xnum + p[4]) / (xden + q[4]);
        tmp = (0.398942280401432677939946059934 - tmp) / y;
    }
    xsq = trunc(y * 16.0) / 16.0;
    double del = (y - xsq) * (y + xsq);
    return (-xsq * xsq * 0.5) + (-del * 0.5) + orc_log(tmp);
}

/* ------------------------------------------------------------------ */
/* A.4 Viterbi.dthmm.adj (R/inferCNV_HMM.R:1101-1176)                  */
/* ------------------------------------------------------------------ */
#define ORC_MAXK 16

/* One sequence.  Keeps the full nu matrix and does the reference's own
 * traceback (which.max(logPi[, y[i+1]] + nu[i, ])).  Returns 1 when the
 * reference would stop("Problems With Underflow"). */
static int viterbi_seq(const double *x, int64_t n, int32_t K, const double *mean, double sd,
                       const double *logPi, const double *logDelta, double *nu, uint8_t *y,
                       int64_t ystride) {
    if (n < 2) { /* :1104-1107 */
        for (int64_t i = 0; i < n; i++) y[i * ystride] = 3;
        return 0;
    }
    double e[ORC_MAXK];
    for (int64_t i = 0; i < n; i++) {
        double tot = 0.0;
        for (int32_t k = 0; k < K; k++) {
            double lp = orc_pnorm_log_upper(fabs(x[i] - mean[k]) / sd);
            e[k] = 1.0 / (-1.0 * lp);
            tot = (k == 0) ? e[0] : tot + e[k];
        }
        double *row = nu + (int64_t)i * K;
        for (int32_t k = 0; k < K; k++) {
            double sc = orc_log(e[k] / tot);
            if (i == 0) {
                row[k] = logDelta[k] + sc;
            } else {
                const double *prev = row - K;
                double best = prev[0] + logPi[0 + K * k];
                for (int32_t j = 1; j < K; j++) {
                    double v = prev[j] + logPi[j + K * k];
                    if (v > best) best = v;
                }
                row[k] = best + sc;
            }
        }
    }
    int bad = 0;
    const double *last = nu + (n - 1) * K;
    for (int32_t k = 0; k < K; k++) if (last[k] == -INFINITY) bad = 1;
    int32_t cur = 0;
    for (int32_t k = 1; k < K; k++) if (last[k] > last[cur]) cur = k; /* which.max: first max */
    y[(n - 1) * ystride] = (uint8_t)(cur + 1);
    for (int64_t i = n - 2; i >= 0; i--) {
        const double *row = nu + i * K;
        int32_t bj = 0;
        double bv = logPi[0 + K * cur] + row[0];
        for (int32_t j = 1; j < K; j++) {
            double v = logPi[j + K * cur] + row[j];
            if (v > bv) { bv = v; bj = j; }
        }
        cur = bj;
        y[i * ystride] = (uint8_t)(cur + 1);
    }
    return bad;
}

/* predict_CNV_via_HMM_on_indiv_cells (R/inferCNV_HMM.R:284-324; K=3:
 * R/inferCNV_i3HMM.R:180-225).  sd = median(pm$sd) prepared by the caller.
 * states[g + G*c] in 1..K.  Returns number of underflowing sequences, <0 on
 * bad arguments. */
int orc_viterbi_cells(const double *x, uint8_t *states, int64_t G, int64_t C, const int32_t *chr_start,
                      int32_t n_chr, int32_t K, const double *mean, double sd, const double *logPi,
                      const double *logDelta) {
    if (K < 1 || K > ORC_MAXK) return -1;
    int64_t maxn = 0;
    for (int32_t k = 0; k < n_chr; k++)
        if (chr_start[k + 1] - chr_start[k] > maxn) maxn = chr_start[k + 1] - chr_start[k];
    int bad = 0;
#pragma omp parallel reduction(+ : bad)
    {
        double *nu = (double *)malloc(sizeof(double) * (maxn + 1) * K);
#pragma omp for schedule(static)
        for (int64_t c = 0; c < C; c++)
            for (int32_t k = 0; k < n_chr; k++) {
                int64_t s = chr_start[k], n = chr_start[k + 1] - s;
                bad += viterbi_seq(x + G * c + s, n, K, mean, sd, logPi, logDelta, nu, states + G * c + s, 1);
            }
        free(nu);
    }
    return bad;
}

/* rowMeans(expr.data[, group_cells]) for every group (R/inferCNV_HMM.R:383):
 * out[g + G*q]. */
void orc_group_means(const double *x, int64_t G, int64_t C, const int32_t *grp_idx,
                     const int32_t *grp_off, int32_t n_grp, double *out) {
    (void)C;
    for (int32_t q = 0; q < n_grp; q++) {
        const int32_t *idx = grp_idx + grp_off[q];
        int64_t n = grp_off[q + 1] - grp_off[q];
#pragma omp parallel for schedule(static)
        for (int64_t g = 0; g < G; g++) out[g + G * q] = r_rowmean_idx(x + g, idx, n, G);
    }
}

/* predict_CNV_via_HMM_on_tumor_subclusters / _whole_tumor_samples
 * (R/inferCNV_HMM.R:345-408, 509-567; i3: R/inferCNV_i3HMM.R:249-389): Viterbi
 * on the group-mean profile with that group's shared sd, trace broadcast to
 * the member cells.  Cells in no group keep 0xFF (R: -1). */
int orc_viterbi_groups(const double *x, uint8_t *states, int64_t G, int64_t C, const int32_t *chr_start,
                       int32_t n_chr, const int32_t *grp_idx, const int32_t *grp_off, int32_t n_grp,
                       int32_t K, const double *mean, const double *sd_per_grp, const double *logPi,
                       const double *logDelta) {
    if (K < 1 || K > ORC_MAXK) return -1;
    double *gm = (double *)malloc(sizeof(double) * G * n_grp);
    uint8_t *gs = (uint8_t *)malloc((size_t)G * n_grp);
    orc_group_means(x, G, C, grp_idx, grp_off, n_grp, gm);
    int64_t maxn = 0;
    for (int32_t k = 0; k < n_chr; k++)
        if (chr_start[k + 1] - chr_start[k] > maxn) maxn = chr_start[k + 1] - chr_start[k];
    double *nu = (double *)malloc(sizeof(double) * (maxn + 1) * K);
    int bad = 0;
    memset(states, 0xFF, (size_t)G * C);
    for (int32_t q = 0; q < n_grp; q++) {
        for (int32_t k = 0; k < n_chr; k++) {
            int64_t s = chr_start[k], n = chr_start[k + 1] - s;
            bad += viterbi_seq(gm + G * q + s, n, K, mean, sd_per_grp[q], logPi, logDelta, nu, gs + G * q + s, 1);
        }
        for (int64_t i = grp_off[q]; i < grp_off[q + 1]; i++)
            memcpy(states + G * (int64_t)grp_idx[i], gs + G * q, (size_t)G);
    }
    free(nu); free(gm); free(gs);
    return bad;
}

/* assign_HMM_states_to_proxy_expr_vals (R/inferCNV_HMM.R:1191-1206; i3:
 * R/inferCNV_i3HMM.R:405-417). */
void orc_states_to_proxy(const uint8_t *states, double *out, int64_t n, int32_t K) {
    static const double i6[7] = {NAN, 0.0, 0.5, 1.0, 1.5, 2.0, 3.0};
    static const double i3[4] = {NAN, 0.5, 1.0, 1.5};
    for (int64_t i = 0; i < n; i++) {
        uint8_t s = states[i];
        out[i] = (K == 3) ? (s <= 3 ? i3[s] : NAN) : (s <= 6 ? i6[s] : NAN);
    }
}

/* ------------------------------------------------------------------ */
/* A.6 2-D median filter (R/noise_reduction.R:43-113)                  */
/* ------------------------------------------------------------------ */
/* tiles: concatenated cell index vectors (stored order) with offsets; every
 * (tile, chr) block is filtered independently from a pre-filter copy. */
void orc_median_filter(const double *in, double *out, int64_t G, int64_t C, const int32_t *chr_start,
                       int32_t n_chr, const int32_t *tile_idx, const int32_t *tile_off, int32_t n_tiles,
                       int32_t window_size) {
    if (out != in) memcpy(out, in, sizeof(double) * G * C);
    int64_t half = (window_size - 1) / 2;
    int64_t wmax = (2 * (half + 1) + 1);
#pragma omp parallel
    {
        double *buf = (double *)malloc(sizeof(double) * wmax * wmax);
#pragma omp for schedule(dynamic) collapse(2)
        for (int32_t t = 0; t < n_tiles; t++)
            for (int32_t k = 0; k < n_chr; k++) {
                const int32_t *idx = tile_idx + tile_off[t];
                int64_t ydim = tile_off[t + 1] - tile_off[t];
                int64_t s = chr_start[k], xdim = chr_start[k + 1] - s;
                for (int64_t px = 1; px <= xdim; px++) {
                    int64_t xa = (px <= half + 1) ? 1 : px - (half + 1);
                    int64_t xb = (px >= xdim - (half + 1)) ? xdim : px + (half + 1);
                    for (int64_t py = 1; py <= ydim; py++) {
                        int64_t ya = (py <= half + 1) ? 1 : py - (half + 1);
                        int64_t yb = (py >= ydim - (half + 1)) ? ydim : py + (half + 1);
                        int64_t m = 0;
                        for (int64_t yy = ya; yy <= yb; yy++) {
                            const double *col = in + G * (int64_t)idx[yy - 1] + s;
                            for (int64_t xx = xa; xx <= xb; xx++) buf[m++] = col[xx - 1];
                        }
                        out[G * (int64_t)idx[py - 1] + s + (px - 1)] = r_median_scratch(buf, m);
                    }
                }
            }
        free(buf);
    }
}

/* i3 parameters (R/inferCNV_i3HMM.R:17-80): mean and sd over ALL values of the
 * given cells. */
void orc_mean_sd_of_cells(const double *x, int64_t G, const int32_t *idx, int64_t n, double *mu,
                          double *sigma) {
    double m, s;
    orc_denoise_params(x, G, 0, idx, n, 1.0, &m, &s);
    ld_t v = 0;
    for (int64_t i = 0; i < n; i++) {
        const double *col = x + G * (int64_t)idx[i];
        for (int64_t g = 0; g < G; g++) { ld_t d = (ld_t)col[g] - (ld_t)m; v += d * d; }
    }
    *mu = m;
    *sigma = sqrt((double)(v / (ld_t)(n * G - 1)));
}
