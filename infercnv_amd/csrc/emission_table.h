// Emission-score table of the certified fast Viterbi path (viterbi_fast.hip).
//
// The reference's emission scores (R/inferCNV_HMM.R:1129-1133, 1156-1160)
//     lp_k = log P(Z > |x - mean_k| / sd),  e_k = 1 / (-lp_k),  s_k = log(e_k / sum_j e_j)
// are K smooth functions of the single observation x between consecutive state means (|x - mean_k|
// kinks at every mean).  Only the differences between the states' scores enter the decisions of the
// max-plus recurrence (a term common to all states shifts every candidate alike), so the table holds
//     d_k(x) = s_k(x) - s_1(x) = log e_k - log e_1,   k = 2..K     (d_1 = 0 is not stored on the device)
// on a UNIFORM grid over [x_lo, x_hi] (interval j = floor((x - x_lo) / w), w = sd / 16): one degree-DEG polynomial per
// state and grid interval in the normalised position tn in [-0.5, 0.5] inside the interval.  A grid interval that holds a
// state mean (|x - mean_k| kinks there) carries TWO records -- the branch below the mean and the branch from the mean
// on, each the smooth continuation of its side fitted over the whole interval -- and the kernel picks by one exact
// comparison with the mean.  It is built on the host in 80-bit long double from the
// mathematically exact functions (erfcl / logl) and verified against them through the very double
// operations the kernel executes; eps_tab is the certified bound the kernel's margin test uses.
//
// Host-only, no HIP types: compiled by g++ and also exported through the C ABI
// (icnv_hmm_emission_table) so that the CPU tests can check the bound independently.
#pragma once
#include <stdint.h>
#include <vector>

namespace icnv {

#ifndef ICNV_EMIS_DEG
#define ICNV_EMIS_DEG 4
#endif
#ifndef ICNV_EMIS_EPS_MAX
#define ICNV_EMIS_EPS_MAX 2e-12
#endif
constexpr int EMIS_DEG = ICNV_EMIS_DEG;   // polynomial degree (4 in the product; scripts/viterbi_variants.py builds others)
#ifndef ICNV_EMIS_WIDTH_DIV
#define ICNV_EMIS_WIDTH_DIV 16.0
#endif
constexpr double EMIS_WIDTH_DIV = ICNV_EMIS_WIDTH_DIV;   // interval width = sd / this
constexpr double EMIS_EPS_MAX = ICNV_EMIS_EPS_MAX;   // a table whose certified error exceeds this is refused
// One entry per grid interval, 16 bytes, read by the kernel with one LDS load: the record of an observation x in interval
// j is rec + (x >= boundary).  boundary = the state mean inside the interval (+inf when there is none); which interval a
// mean falls into is computed with the very double operations the kernel applies to an observation (both are monotone
// in x), so the lookup is exact whatever the rounding at the interval edges.
struct EmisGridEntry {
    double boundary;
    int32_t rec;
    int32_t pad;
};

struct EmisTable {
    int K = 0;
    int n_grid = 0;                  // grid intervals
    int n_int = 0;                   // records = n_grid + K (every mean splits its interval)
    double x_lo = 0, x_hi = 0;       // covered domain; observations outside take the exact path
    double inv_w = 0;                // 1 / interval width: interval of x = (int)((x - x_lo) * inv_w)
    double eps_tab = 0;              // certified bound on |table value - (s_k - s_1)| over the domain
    double s_max = 0;                // max over the domain of |s_k| and |s_k - s_1| (bounds the magnitude of the DP values of both kernels)
    double width_sigma = 0;          // interval width in units of sd
    std::vector<EmisGridEntry> grid; // [n_grid]
    std::vector<double> coef;        // [n_int][K][EMIS_DEG + 1], c0 first; row k = 0 (state 1) is zero
};

// Build the table for K states with strictly increasing means and a shared sd.  max_intervals is the
// LDS budget in records (a grid entry rides with every record).  Returns 0 on success; non-zero (with a reason in *why) when the parameters are not
// eligible (unsorted means, non-finite values, accuracy target not met): callers then use the exact kernel.
int build_emission_table(int K, const double *mean, double sd, int max_intervals, EmisTable &out, const char **why);

// The exact scores in long double (the function the table approximates); used by the build's own
// verification and exported for the tests.
void emission_scores_exact(int K, const double *mean, double sd, double x, double *s_out);

// The kernel's evaluation of the table restated on the host with the same double operations
// (explicit fma, same order): s_out[k] = d_k(x), s_out[0] = 0.  Returns false when x is outside the
// domain / not finite.
bool emission_table_eval(const EmisTable &t, const double *mean, double x, double *s_out);

}  // namespace icnv
