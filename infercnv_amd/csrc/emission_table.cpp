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

// side[k] = 0: |x - mean_k| (the function itself); +1 / -1: the branch x >= mean_k / x < mean_k continued analytically
// across the mean (z may then be negative: P(Z > z) > 1/2) -- what a record of the table is fitted to
void exact_ld(int K, const double *mean, double sd, long double x, long double *s, const int *side = nullptr) {
    long double e[8], tot = 0.0L;
    for (int k = 0; k < K; ++k) {
        const long double d = x - (long double)mean[k];
        const long double z = ((side && side[k]) ? (side[k] > 0 ? d : -d) : fabsl(d)) / (long double)sd;
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
    const double u = (x - t.x_lo) * t.inv_w;
    const int j = (int)u;   // u >= 0: truncation == floor; j <= n_grid - 1 is a property of x_hi the builder establishes
    if (j < 0 || j >= t.n_grid) return false;
    const EmisGridEntry &g = t.grid[(size_t)j];
    const int r = g.rec + ((x >= g.boundary) ? 1 : 0);
    const double tn = (u - (double)j) - 0.5;
    const double *c = t.coef.data() + (size_t)r * t.K * NC;
    s_out[0] = 0.0;   // the table holds the scores relative to state 1
    for (int k = 1; k < t.K; ++k) {
        double p = c[k * NC + EMIS_DEG];
        for (int j2 = EMIS_DEG - 1; j2 >= 0; --j2) p = std::fma(p, tn, c[k * NC + j2]);
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
    // Degree 4 on a grid of sd / 16: eps_tab <= 1.4e-12 over 400 random i3 / i6 models (sd / 15: 1.9e-12, sd / 14: 2.5e-12,
    // sd / 12: 5.6e-12), inside the 2e-12 the builder accepts.  Rounds 1-2 used degree 5 on sd / 12 intervals anchored at
    // the means (4e-14): one fused multiply-add per state and gene more, 240- instead of 208-byte records, two dependent
    // LDS lookups (cell -> segment) in front of the coefficient reads instead of one.  The decision band
    // 4 (n + 1) (eps_tab + 2 eps_spec + 6 u B) is dominated by its other two terms, so the coarser table widens it by a
    // third (profiles/r03_viterbi_degree4.txt: -4 % on the launch, A/B on one box).
    out.width_sigma = 1.0 / EMIS_WIDTH_DIV;
    const double w = out.width_sigma * sd;
    out.inv_w = 1.0 / w;
    const double span = mean[K - 1] - mean[0];
    const double inner = std::ceil(span / w) + 1.0;     // grid intervals that reach from the first to the last mean
    if (!(inner >= 1.0) || inner + (double)K > (double)max_intervals) { *why = "state means too far apart (in units of sd) for the table"; return 1; }
    // tails: what the budget leaves, split evenly, and never further than z = 36 from the far mean
    int n_tail = (max_intervals - K - (int)inner) / 2;
    const int tail_cap = (int)std::floor((36.0 * sd - span) / w);
    if (n_tail > tail_cap) n_tail = tail_cap;
    if (n_tail < 8) { *why = "not enough table budget left for the tails"; return 1; }
    out.n_grid = 2 * n_tail + (int)inner;
    out.n_int = out.n_grid + K;
    out.x_lo = (double)((long double)mean[0] - (long double)n_tail * (long double)w);
    // the kernel does not clamp the interval index: x_hi is the largest double whose index, computed the kernel's way, is
    // still n_grid - 1
    out.x_hi = (double)((long double)out.x_lo + (long double)out.n_grid * (long double)w);
    for (int guard = 0; guard < 64 && (int)((out.x_hi - out.x_lo) * out.inv_w) > out.n_grid - 1; ++guard)
        out.x_hi = std::nextafter(out.x_hi, -INFINITY);
    if ((int)((out.x_hi - out.x_lo) * out.inv_w) > out.n_grid - 1 || !(out.x_hi > mean[K - 1])) { *why = "internal: table domain"; return 2; }

    // the grid: which interval every mean falls into (the kernel's arithmetic), the record numbers
    int jm[8];
    for (int k = 0; k < K; ++k) {
        jm[k] = (int)((mean[k] - out.x_lo) * out.inv_w);
        if (jm[k] < 0 || jm[k] >= out.n_grid) { *why = "internal: a state mean outside the grid"; return 2; }
        if (k && jm[k] <= jm[k - 1]) { *why = "two state means share a table interval (closer than sd / 16)"; return 1; }
    }
    out.grid.assign((size_t)out.n_grid, EmisGridEntry{INFINITY, 0, 0});
    for (int j = 0; j < out.n_grid; ++j) {
        int below = 0;
        for (int k = 0; k < K; ++k) below += (jm[k] < j) ? 1 : 0;
        out.grid[(size_t)j].rec = j + below;
    }
    for (int k = 0; k < K; ++k) out.grid[(size_t)jm[k]].boundary = mean[k];
    out.coef.assign((size_t)out.n_int * K * NC, 0.0);

    // the records: Chebyshev interpolation of the branch that is valid in the record's part of the interval.  Every x the
    // kernel sends to interval j lies on one side of every mean outside it (the interval index is monotone in x and the
    // means are placed by the same arithmetic), so each state's branch is fixed per record and analytic over the whole
    // interval -- also where a mean sits within rounding of an interval edge
    auto fit = [&](int j, int rec, const int *side) {
        long double val[NC][8];
        for (int i = 0; i < NC; ++i) {
            const long double x = (long double)out.x_lo + (nodes.t[i] + (long double)j + 0.5L) * (long double)w;
            exact_ld(K, mean, sd, x, val[i], side);
        }
        double *c = out.coef.data() + (size_t)rec * K * NC;
        for (int k = 0; k < K; ++k)
            for (int q = 0; q < NC; ++q) {
                long double acc = 0.0L;
                for (int i = 0; i < NC; ++i) acc += nodes.vinv[q][i] * (val[i][k] - val[i][0]);
                c[k * NC + q] = (double)acc;   // state 1's own row is all zeros
            }
    };
    for (int j = 0; j < out.n_grid; ++j) {
        int side[8], here = -1;
        for (int k = 0; k < K; ++k) {
            side[k] = (j > jm[k]) ? 1 : -1;
            if (j == jm[k]) here = k;
        }
        const int rec = out.grid[(size_t)j].rec;
        if (here < 0) {
            fit(j, rec, side);
        } else {
            side[here] = -1;
            fit(j, rec, side);
            side[here] = 1;
            fit(j, rec + 1, side);
        }
    }

    // verification through the kernel's own double arithmetic against the function itself (|x - mean_k|): the extrema of
    // T_{NC} in every interval (where the interpolation error peaks; a check point on an edge is evaluated with whichever
    // neighbour the kernel's index arithmetic picks), and every mean with its two neighbours
    long double chk[4 * NC + 1];
    int n_chk = 0;
    for (int q = 0; q <= 2 * NC; ++q) chk[n_chk++] = 0.5L * cosl(PI_L * q / (2.0L * NC));
    long double max_err = 0.0L, s_max = 0.0L;
    auto check_at = [&](double x) -> bool {
        if (x < out.x_lo) x = out.x_lo;
        if (x > out.x_hi) x = out.x_hi;
        double got[8];
        long double want[8];
        if (!emission_table_eval(out, mean, x, got)) return false;
        exact_ld(K, mean, sd, (long double)x, want);
        for (int k = 0; k < K; ++k) {
            const long double d = want[k] - want[0];
            const long double e = fabsl((long double)got[k] - d);
            if (e > max_err) max_err = e;
            if (fabsl(want[k]) > s_max) s_max = fabsl(want[k]);
            if (fabsl(d) > s_max) s_max = fabsl(d);
        }
        return true;
    };
    for (int j = 0; j < out.n_grid; ++j)
        for (int q = 0; q < n_chk; ++q)
            if (!check_at((double)((long double)out.x_lo + (chk[q] + (long double)j + 0.5L) * (long double)w))) { *why = "internal: check point outside the domain"; return 2; }
    for (int k = 0; k < K; ++k) {
        double x = mean[k];
        for (int q = 0; q < 3; ++q) x = std::nextafter(x, -INFINITY);
        for (int q = 0; q < 7; ++q, x = std::nextafter(x, INFINITY))
            if (!check_at(x)) { *why = "internal: check point outside the domain"; return 2; }
        // the far end of the shorter part of the mean's interval, where its record is used furthest from its fit's centre
        const long double lo_edge = (long double)out.x_lo + (long double)jm[k] * (long double)w;
        if (!check_at(std::nextafter((double)lo_edge, INFINITY)) || !check_at(std::nextafter((double)(lo_edge + (long double)w), -INFINITY))) { *why = "internal: check point outside the domain"; return 2; }
    }
    out.eps_tab = (double)(1.5L * max_err) + 1e-15;
    out.s_max = (double)s_max * 1.01 + 0.01;
    if (!(out.eps_tab <= EMIS_EPS_MAX)) { *why = "table accuracy target (2e-12) not met for these parameters"; return 1; }
    return 0;
}

}  // namespace icnv
