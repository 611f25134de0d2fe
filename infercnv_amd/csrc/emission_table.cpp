// Host-side builder of the emission-score table used by the certified fast Viterbi path.
// See emission_table.h for what the table is; DESIGN.md "Certified fast Viterbi" for the error budget.
// Plain C++ (g++), no HIP.
#include "emission_table.h"

#include <cmath>
#include <cstring>

namespace icnv {

namespace {

constexpr int NC = EMIS_DEG + 1;
constexpr long double PI_L = 3.14159265358979323846264338327950288L;
constexpr long double SQRT1_2_L = 0.70710678118654752440084436210484904L;

void exact_ld(int K, const double *mean, double sd, long double x, long double *s) {
    long double e[8], tot = 0.0L;
    for (int k = 0; k < K; ++k) {
        const long double z = fabsl(x - (long double)mean[k]) / (long double)sd;
        const long double q = 0.5L * erfcl(z * SQRT1_2_L);
        e[k] = -1.0L / logl(q);
        tot += e[k];
    }
    for (int k = 0; k < K; ++k) s[k] = logl(e[k] / tot);
}

// inverse of the Vandermonde matrix of the NC Chebyshev nodes on [-0.5, 0.5] (same nodes in every interval)
struct Nodes {
    long double t[NC];
    long double vinv[NC][NC];
    Nodes() {
        long double a[NC][2 * NC];
        for (int i = 0; i < NC; ++i) {
            t[i] = 0.5L * cosl(PI_L * (2 * i + 1) / (2.0L * NC));
            long double p = 1.0L;
            for (int j = 0; j < NC; ++j) { a[i][j] = p; p *= t[i]; }
            for (int j = 0; j < NC; ++j) a[i][NC + j] = (i == j) ? 1.0L : 0.0L;
        }
        for (int c = 0; c < NC; ++c) {   // Gauss-Jordan with partial pivoting
            int piv = c;
            for (int r = c + 1; r < NC; ++r)
                if (fabsl(a[r][c]) > fabsl(a[piv][c])) piv = r;
            if (piv != c)
                for (int j = 0; j < 2 * NC; ++j) { const long double tmp = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = tmp; }
            const long double d = a[c][c];
            for (int j = 0; j < 2 * NC; ++j) a[c][j] /= d;
            for (int r = 0; r < NC; ++r) {
                if (r == c) continue;
                const long double f = a[r][c];
                if (f != 0.0L)
                    for (int j = 0; j < 2 * NC; ++j) a[r][j] -= f * a[c][j];
            }
        }
        for (int i = 0; i < NC; ++i)
            for (int j = 0; j < NC; ++j) vinv[i][j] = a[i][NC + j];
    }
};

}  // namespace

void emission_scores_exact(int K, const double *mean, double sd, double x, double *s_out) {
    long double s[8];
    exact_ld(K, mean, sd, (long double)x, s);
    for (int k = 0; k < K; ++k) s_out[k] = (double)s[k];
}

bool emission_table_eval(const EmisTable &t, const double *mean, double x, double *s_out) {
    if (!(x >= t.x_lo && x <= t.x_hi)) return false;
    (void)mean;
    int ci = (int)((x - t.cell_lo) * t.inv_wc);
    if (ci > t.n_cells - 1) ci = t.n_cells - 1;
    const int s = t.cell[ci].seg_below + ((x >= t.cell[ci].boundary) ? 1 : 0);
    const EmisSegment &sg = t.seg[s];
    const double u = (x - sg.lo) * sg.inv_w;
    int fi = (int)u;   // u >= 0: truncation == floor
    if (fi > sg.n_m1) fi = sg.n_m1;
    if (fi < 0) fi = 0;
    const double tn = (u - (double)fi) - 0.5;
    const double *c = t.coef.data() + (size_t)(sg.base + fi) * t.K * NC;
    s_out[0] = 0.0;   // the table holds the scores relative to state 1
    for (int k = 1; k < t.K; ++k) {
        double p = c[k * NC + EMIS_DEG];
        for (int j = EMIS_DEG - 1; j >= 0; --j) p = std::fma(p, tn, c[k * NC + j]);
        s_out[k] = p;
    }
    return true;
}

int build_emission_table(int K, const double *mean, double sd, int max_intervals, EmisTable &out, const char **why) {
    static const char *dummy;
    if (!why) why = &dummy;
    *why = "";
    if (K < 2 || K > 6) { *why = "K outside 2..6"; return 1; }
    if (!(sd >= 0x1p-40 && sd <= 0x1p40)) { *why = "sd outside the supported range"; return 1; }
    for (int k = 0; k < K; ++k) {
        if (!std::isfinite(mean[k]) || std::fabs(mean[k]) >= 0x1p40) { *why = "non-finite or huge state mean"; return 1; }
        if (k && !(mean[k] > mean[k - 1])) { *why = "state means are not strictly increasing"; return 1; }
    }
    static const Nodes nodes;
    out = EmisTable();
    out.K = K;
    out.n_seg = K + 1;
    // Degree 4 on intervals of sd / 15: eps_tab 1.7e-12 for the i6 and i3 models, inside the 2e-12 budget (sd / 14: 2.5e-12,
    // sd / 12: 5.6e-12).  Rounds 1-2 used degree 5 on sd / 12 (4e-14; sd / 16: 9e-15, sd / 8: 4e-13): one fused multiply-add per
    // state and gene more, 240- instead of 208-byte records.  The decision band 4 (n + 1) (eps_tab + 2 eps_spec + 6 u B) is
    // dominated by its other two terms, so the coarser table widens it by a third; the launch is 4 % shorter (A/B on one box,
    // profiles/r03_viterbi_degree4.txt), and 727 instead of 629 intervals fit the LDS, so smaller sd stay eligible.
    out.width_sigma = 1.0 / EMIS_WIDTH_DIV;
    const double w_target = out.width_sigma * sd;

    // inner segments [mean_k, mean_{k+1}): whole numbers of intervals
    int n_inner = 0;
    int n_of[EMIS_MAX_SEG] = {0};
    for (int s = 1; s < K; ++s) {
        const double len = mean[s] - mean[s - 1];
        const double nn = std::ceil(len / w_target);
        if (!(nn >= 1.0) || nn > (double)max_intervals) { *why = "state means too far apart (in units of sd) for the table"; return 1; }
        n_of[s] = (int)nn;
        n_inner += n_of[s];
    }
    // tails: what the budget leaves, split evenly, and never further than z = 36 from the far mean
    int n_tail = (max_intervals - n_inner) / 2;
    const double span = mean[K - 1] - mean[0];
    const int tail_cap = (int)std::floor((36.0 * sd - span) / w_target);
    if (n_tail > tail_cap) n_tail = tail_cap;
    if (n_tail < 8) { *why = "not enough table budget left for the tails"; return 1; }
    n_of[0] = n_of[K] = n_tail;

    int base = 0;
    for (int s = 0; s <= K; ++s) {
        EmisSegment &sg = out.seg[s];
        std::memset(&sg, 0, sizeof(sg));
        if (s == 0) {
            sg.inv_w = 1.0 / w_target;
            sg.lo = mean[0] - (double)n_of[0] / sg.inv_w;
        } else if (s == K) {
            sg.inv_w = 1.0 / w_target;
            sg.lo = mean[K - 1];
        } else {
            sg.lo = mean[s - 1];
            sg.inv_w = (double)n_of[s] / (mean[s] - mean[s - 1]);
        }
        sg.base = base;
        sg.n_m1 = n_of[s] - 1;
        base += n_of[s];
    }
    out.n_int = base;
    // the lower tail must end where segment 1 begins: (mean[0] - lo) * inv_w may round to n - epsilon or n + epsilon,
    // the clamp of the interval index takes care of either; x_lo is nudged inside so that u >= 0 always
    out.x_lo = std::nextafter(out.seg[0].lo, mean[0]);
    out.x_hi = (double)((long double)out.seg[K].lo + (long double)n_of[K] / (long double)out.seg[K].inv_w);
    out.x_hi = std::nextafter(out.x_hi, mean[K - 1]);
    out.coef.assign((size_t)out.n_int * K * NC, 0.0);
    {   // segment lookup cells
        double gap = mean[1] - mean[0];
        for (int k = 2; k < K; ++k) gap = std::fmin(gap, mean[k] - mean[k - 1]);
        out.cell_lo = out.seg[0].lo;
        const double width = out.x_hi - out.cell_lo;
        int nc = (int)std::ceil(width / (0.75 * gap)) + 1;
        if (nc < 1) nc = 1;
        if (nc > EMIS_MAX_CELLS) { *why = "state means too close together for the segment lookup"; return 1; }
        out.n_cells = nc;
        out.inv_wc = (double)nc / width;
        for (int c = 0; c < nc; ++c) { out.cell[c].boundary = INFINITY; out.cell[c].seg_below = 0; out.cell[c].pad = 0; }
        int ck[8];
        for (int k = 0; k < K; ++k) {
            ck[k] = (int)((mean[k] - out.cell_lo) * out.inv_wc);
            if (ck[k] > nc - 1) ck[k] = nc - 1;
            if (k && ck[k] <= ck[k - 1]) { *why = "two state means share a lookup cell"; return 1; }
            out.cell[ck[k]].boundary = mean[k];
        }
        for (int c = 0; c < nc; ++c) {
            int below = 0;
            for (int k = 0; k < K; ++k) below += (ck[k] < c) ? 1 : 0;
            out.cell[c].seg_below = below;
        }
    }

    // check positions: the extrema of T_{NC} (where the interpolation error peaks) and two more per gap
    long double chk[4 * NC + 1];
    int n_chk = 0;
    for (int j = 0; j <= 2 * NC; ++j) chk[n_chk++] = 0.5L * cosl(PI_L * j / (2.0L * NC));

    long double max_err = 0.0L, s_max = 0.0L;
    for (int s = 0; s <= K; ++s) {
        const EmisSegment &sg = out.seg[s];
        for (int fi = 0; fi <= sg.n_m1; ++fi) {
            long double val[NC][8];
            for (int i = 0; i < NC; ++i) {
                const long double x = (long double)sg.lo + (nodes.t[i] + (long double)fi + 0.5L) / (long double)sg.inv_w;
                exact_ld(K, mean, sd, x, val[i]);
            }
            double *c = out.coef.data() + (size_t)(sg.base + fi) * K * NC;
            for (int k = 0; k < K; ++k)
                for (int j = 0; j < NC; ++j) {
                    long double acc = 0.0L;
                    for (int i = 0; i < NC; ++i) acc += nodes.vinv[j][i] * (val[i][k] - val[i][0]);
                    c[k * NC + j] = (double)acc;   // state 1's own row is all zeros
                }
        }
    }
    // verification through the kernel's own double arithmetic (second pass: a check point on an interval's
    // edge is evaluated with whichever neighbour the kernel's index arithmetic picks)
    for (int s = 0; s <= K; ++s) {
        const EmisSegment &sg = out.seg[s];
        for (int fi = 0; fi <= sg.n_m1; ++fi) {
            for (int q = 0; q < n_chk; ++q) {
                const long double xl = (long double)sg.lo + (chk[q] + (long double)fi + 0.5L) / (long double)sg.inv_w;
                double x = (double)xl;
                if (x < out.x_lo) x = out.x_lo;
                if (x > out.x_hi) x = out.x_hi;
                // stay inside this segment (the end points of inner segments are the means themselves)
                if (s > 0 && x < mean[s - 1]) x = mean[s - 1];
                if (s < K && x >= mean[s]) x = std::nextafter(mean[s], -INFINITY);
                double got[8];
                long double want[8];
                if (!emission_table_eval(out, mean, x, got)) { *why = "internal: check point outside the domain"; return 2; }
                exact_ld(K, mean, sd, (long double)x, want);
                for (int k = 0; k < K; ++k) {
                    const long double d = want[k] - want[0];
                    const long double e = fabsl((long double)got[k] - d);
                    if (e > max_err) max_err = e;
                    if (fabsl(want[k]) > s_max) s_max = fabsl(want[k]);
                    if (fabsl(d) > s_max) s_max = fabsl(d);
                }
            }
        }
    }
    out.eps_tab = (double)(1.5L * max_err) + 1e-15;
    out.s_max = (double)s_max * 1.01 + 0.01;
    if (!(out.eps_tab <= EMIS_EPS_MAX)) { *why = "table accuracy target (2e-12) not met for these parameters"; return 1; }
    return 0;
}

}  // namespace icnv
