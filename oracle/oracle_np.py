"""NumPy restatement of inferCNV's smoothing chain + HMM + median filter.

TEST INFRASTRUCTURE ONLY -- this is the parity *checker*.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
the product package `infercnv_amd` never does.

Every function cites the reference lines it restates (paths relative to
/root/reference).  Conventions: `expr` is a float64 ndarray of shape
(G genes, C cells) -- the same orientation as the reference's `expr.data`
(R/inferCNV.R:18) -- and all index vectors are 0-based (R's are 1-based).

Pinning status
  * smoothing chain: pinned by the reference's own golden
    data/infercnv_object_example.rda (tests/golden/, 1e-12) and by the literal
    goldens of tests/testthat/test_infer_cnv.R.
  * HMM (Viterbi.dthmm.adj) and median filter: the reference holds no
    known-answer test -> PARITY UNPINNED; this file is the
    restatement, cross-checked against the independent C restatement
    oracle/icnv_oracle.c and, loosely, against data/HMM_states.rda.
"""
from __future__ import annotations

import math

import numpy as np

LD = np.longdouble  # R accumulates sum()/mean() in long double (x86-64: 80-bit)


# --------------------------------------------------------------------------
# base-R primitives restated
# --------------------------------------------------------------------------
def r_mean(x, axis=None):
    """base::mean on doubles: long-double sum / n, then one refinement pass
    (R src/main/summary.c, real_mean).  Within ~1 ulp of np.mean."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size if axis is None else x.shape[axis]
    s = x.astype(LD).sum(axis=axis) / LD(n)
    if axis is None:
        t = (x.astype(LD) - s).sum() / LD(n)
    else:
        t = (x.astype(LD) - np.expand_dims(s, axis)).sum(axis=axis) / LD(n)
    return np.asarray(s + t, dtype=np.float64)


def r_row_means(x):
    """base::rowMeans on a double matrix (base R src/main/array.c, do_colsum with OP == 3): one LDOUBLE accumulator
    per row, columns added in order, divided by the column count in LDOUBLE, cast to double -- no refinement pass
    (that is mean()).  LD is numpy's longdouble: the 80-bit x87 format on x86-64, like R there."""
    x = np.asarray(x, dtype=np.float64)
    acc = np.zeros(x.shape[0], dtype=LD)
    for j in range(x.shape[1]):
        acc += x[:, j]
    return np.asarray(acc / LD(x.shape[1]), dtype=np.float64)


def r_sum(x, axis=None):
    return np.asarray(np.asarray(x, dtype=np.float64).astype(LD).sum(axis=axis), dtype=np.float64)


def r_sd(x, axis=0):
    """stats::sd = sqrt(var), two-pass with n-1 (R src/library/stats/src/cov.c)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[axis]
    m = r_mean(x, axis=axis)
    d = x.astype(LD) - np.expand_dims(m.astype(LD), axis)
    v = (d * d).sum(axis=axis) / LD(n - 1)
    return np.sqrt(np.asarray(v, dtype=np.float64))


def r_median(x, axis=0):
    """stats::median: odd n -> middle order statistic; even n -> mean of the
    two middle ones."""
    xs = np.sort(np.asarray(x, dtype=np.float64), axis=axis)
    n = xs.shape[axis]
    h = n // 2
    if n % 2:
        return np.take(xs, h, axis=axis)
    return (np.take(xs, h - 1, axis=axis) + np.take(xs, h, axis=axis)) * 0.5


# --------------------------------------------------------------------------
# A.1 reference subtraction
# --------------------------------------------------------------------------
def get_normal_gene_mean_bounds(expr, ref_groups, inv_log=False):
    """R/inferCNV_ops.R:1708-1735 -> (G, n_groups) per-group gene means."""
    cols = []
    for idx in ref_groups:
        sub = expr[:, np.asarray(idx, dtype=np.int64)]
        if inv_log:
            cols.append(np.log2(r_mean(np.exp2(sub) - 1.0, axis=1) + 1.0))
        else:
            cols.append(r_mean(sub, axis=1))
    return np.stack(cols, axis=1)


def subtract_expr(expr, grp_means, use_bounds=False):
    """R/inferCNV_ops.R:1742-1786 (strict comparisons, in-between -> 0)."""
    if use_bounds:
        lo = grp_means.min(axis=1)[:, None]
        hi = grp_means.max(axis=1)[:, None]
        out = np.zeros_like(expr)
        above = expr > hi
        below = expr < lo
        out[above] = (expr - hi)[above]
        out[below] = (expr - lo)[below]
        return out
    return expr - r_mean(grp_means, axis=1)[:, None]


def subtract_ref_expr_from_obs(expr, ref_groups, inv_log=False, use_bounds=True):
    """R/inferCNV_ops.R:1678-1702.  `ref_groups`: list of 0-based index arrays;
    pass [all observation indices] when there are no reference cells (:1686)."""
    return subtract_expr(expr, get_normal_gene_mean_bounds(expr, ref_groups, inv_log), use_bounds)


# --------------------------------------------------------------------------
# A.3 clamp / centre / 2^x / denoise
# --------------------------------------------------------------------------
def apply_max_threshold_bounds(expr, threshold):
    """R/inferCNV_ops.R:2970-2983."""
    out = expr.copy()
    out[out > threshold] = threshold
    out[out < -threshold] = -threshold
    return out


def get_average_bounds(expr):
    """R/inferCNV_ops.R:2733-2742: quantile()[[1]] / [[5]] are min / max."""
    return np.array([float(r_mean(expr.min(axis=0))), float(r_mean(expr.max(axis=0)))])


def center_columns(expr, method="median"):
    """R/inferCNV_ops.R:2094-2109: per cell subtract median (or mean) over ALL genes."""
    c = r_median(expr, axis=0) if method == "median" else r_mean(expr, axis=0)
    return expr - c[None, :]


def invert_log2(expr):
    """R/inferCNV_ops.R:2814-2826."""
    return np.exp2(expr)


def clear_noise_params_via_ref_mean_sd(expr, ref_idx, sd_amplifier=1.5):
    """R/inferCNV_ops.R:2311-2318 -> (mean_ref_vals, mean_ref_sd*amplifier)."""
    vals = expr[:, np.asarray(ref_idx, dtype=np.int64)]
    mu = float(r_mean(vals))
    s = float(r_mean(r_sd(vals, axis=0))) * sd_amplifier
    return mu, s


def clear_noise_bounds(expr, center, halfwidth):
    """R/inferCNV_ops.R:2270-2278 / :2335 (strict comparisons)."""
    out = expr.copy()
    out[(expr > center - halfwidth) & (expr < center + halfwidth)] = center
    return out


def scale_rows(expr):
    """t(scale(t(x))) (R/inferCNV_ops.R:3177; base R scale.default: centre = colMeans, scale = sqrt(sum(centred^2) / max(1, n - 1)),
    R's long-double accumulation)."""
    x = np.asarray(expr, dtype=np.float64)
    LD = np.longdouble
    n = x.shape[1]
    m = np.asarray(x.astype(LD).sum(axis=1) / LD(n), dtype=np.float64)
    cen = x - m[:, None]
    sc = np.asarray(np.sqrt((cen.astype(LD) ** 2).sum(axis=1) / LD(max(1, n - 1))), dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        return cen / sc[:, None]


def remove_tails(chr_idx_1based, tail_length):
    """.remove_tails (R/inferCNV_ops.R:2370-2386), 1-based like the reference's tests."""
    chr_idx = list(chr_idx_1based)
    n = len(chr_idx)
    if tail_length < 3 or n < 3:
        return []
    if n < tail_length * 2:
        tail_length = n // 3
    tail_length = int(tail_length)
    return chr_idx[:tail_length] + chr_idx[n - tail_length:]


def remove_outliers_norm(data, out_method="average_bound", lower_bound=None, upper_bound=None):
    """.remove_outliers_norm (R/inferCNV_ops.R:1998-2054)."""
    data = np.asarray(data, dtype=np.float64)
    if lower_bound is None or upper_bound is None:
        assert out_method == "average_bound"
        lower_bound, upper_bound = get_average_bounds(data)
    out = data.copy()
    out[out < lower_bound] = lower_bound
    out[out > upper_bound] = upper_bound
    return out


def apply_logistic_val_adj(expr, expr_mean, delta_midpt, slope=20.0):
    """.apply_logistic_val_adj (R/inferCNV_heatmap.R:2791-2810; .logistic, R/SplatterScrape.R:210-212) element by element:
    val = |x - mean|, p = 1 / (1 + exp(-slope (val - midpt))), x -> mean + p val above the mean, mean - p val below it."""
    out = np.ascontiguousarray(expr, dtype=np.float64).copy()
    flat = out.reshape(-1)                      # a view: C-contiguous by construction
    for i in range(flat.size):
        x = flat[i]
        val = abs(x - expr_mean)
        p = 1.0 / (1.0 + math.exp(-slope * (val - delta_midpt)))
        if x > expr_mean:
            flat[i] = expr_mean + p * val
        elif x < expr_mean:
            flat[i] = expr_mean - p * val
    return out


def clear_noise_via_ref_mean_sd(expr, ref_idx, sd_amplifier=1.5, noise_logistic=False):
    """R/inferCNV_ops.R:2302-2346."""
    mu, s = clear_noise_params_via_ref_mean_sd(expr, ref_idx, sd_amplifier)
    if noise_logistic:
        return apply_logistic_val_adj(expr, mu, s)            # depress_log_signal_midpt_val(obj, mean_ref_vals, threshold), :2326-2330
    return clear_noise_bounds(expr, mu, s)


def clear_noise(expr, ref_idx, threshold, noise_logistic=False):
    """R/inferCNV_ops.R:2232-2262."""
    if threshold == 0:
        return expr.copy()
    mu = float(r_mean(expr[:, np.asarray(ref_idx, dtype=np.int64)]))
    if noise_logistic:
        return apply_logistic_val_adj(expr, mu, threshold)   # :2249-2252
    return clear_noise_bounds(expr, mu, threshold)


# --------------------------------------------------------------------------
# A.2 pyramid smoothing, in the reference's own evaluation order
# --------------------------------------------------------------------------
def smooth_window(data, window_length):
    """R/inferCNV_ops.R:2440-2532, 2640-2661 on an (n, C) block of one chr.
    Interior: stats::filter(vals, pyramid/denominator, sides=2) -- double
    accumulation over taps j=0..W-1 of filt[j]*x[i+T-j].  Ends: truncated
    pyramid sums (long-double sum()) divided by the truncated denominator."""
    data = np.asarray(data, dtype=np.float64)
    if window_length < 2:
        return data.copy()
    n = data.shape[0]
    W = int(window_length)
    T = (W - 1) // 2
    out = data.copy()
    numer = np.concatenate([np.arange(1, T + 1), [T + 1], np.arange(T, 0, -1)]).astype(np.float64)
    full_den = float(T * T + W)
    if n >= W:
        filt = numer / full_den
        acc = np.zeros((n - W + 1,) + data.shape[1:], dtype=np.float64)
        for j in range(W):
            lo = 2 * T - j          # x[i + T - j] for i = T .. n-T-1
            acc = acc + filt[j] * data[lo:lo + n - W + 1]
        out[T:n - T] = acc
    it_range = T if n > W else int(np.ceil(n / 2))
    for tail_end in range(1, it_range + 1):
        end_tail = n - tail_end + 1
        d_left = tail_end - 1
        d_right = min(n - tail_end, T)
        r_left = T - d_left
        r_right = T - d_right
        den = full_den - (r_left * (r_left + 1)) / 2 - (r_right * (r_right + 1)) / 2
        left = data[:tail_end + d_right]
        right = data[end_tail - d_right - 1:n]
        nr = numer[T - d_left:T + 1 + d_right]
        shape = (-1,) + (1,) * (data.ndim - 1)
        out[tail_end - 1] = np.asarray((left * nr.reshape(shape)).astype(LD).sum(axis=0), dtype=np.float64) / den
        out[end_tail - 1] = np.asarray((right * nr[::-1].reshape(shape)).astype(LD).sum(axis=0), dtype=np.float64) / den
    return out


def smooth_window_na(col, window_length):
    """.smooth_helper on ONE cell's chromosome with missing values (R/inferCNV_ops.R:2487-2489, 2529): the NAs are taken
    out, the shortened sequence is smoothed as if the genes either side of a gap were neighbours, and the NAs are put back
    at their positions.  run() never reaches this (its chain input is log2(x + 1) of counts); restated so that the
    library's own policy for non-finite input -- which is NOT this one, see tests/test_gpu_parity.py -- can be stated
    against it."""
    col = np.asarray(col, dtype=np.float64)
    nas = np.isnan(col)
    out = col.copy()
    if (~nas).any():
        out[~nas] = smooth_window(col[~nas][:, None], window_length)[:, 0]
    return out


def center_columns_na(expr):
    """.center_columns with method "median": median(x, na.rm = TRUE) per cell, NAs stay NA (R/inferCNV_ops.R:2098)."""
    expr = np.asarray(expr, dtype=np.float64)
    return expr - np.nanmedian(expr, axis=0)[None, :]


def chr_segments(chr_codes):
    """Order of first appearance, like unique(gene_order$chr); returns a list of
    index arrays (which(chr == c)) -- genes of one chr need not be contiguous
    here, although `.order_reduce` (R/inferCNV.R:407) makes them so."""
    chr_codes = np.asarray(chr_codes)
    _, first = np.unique(chr_codes, return_index=True)
    order = chr_codes[np.sort(first)]
    return [np.nonzero(chr_codes == c)[0] for c in order]


def smooth_by_chromosome(expr, chr_codes, window_length):
    """R/inferCNV_ops.R:2406-2434 (chr with <=1 gene untouched)."""
    out = expr.copy()
    for idx in chr_segments(chr_codes):
        if idx.size > 1:
            out[idx] = smooth_window(expr[idx], window_length)
    return out


def smooth_direct(expr, chr_codes, window_length):
    """Closed form of A.2: edge-renormalised pyramid.  Used only to show that
    the reference's two-branch evaluation equals this formula to rounding."""
    W = int(window_length)
    T = (W - 1) // 2
    out = expr.copy()
    if W < 2:
        return out
    for idx in chr_segments(chr_codes):
        n = idx.size
        if n <= 1:
            continue
        x = expr[idx]
        for i in range(n):
            a, b = max(0, i - T), min(n - 1, i + T)
            w = (T + 1 - np.abs(np.arange(a, b + 1) - i)).astype(np.float64)
            out[idx[i]] = (w[:, None] * x[a:b + 1]).sum(axis=0) / w.sum()
    return out


# --------------------------------------------------------------------------
# steps 3/4 (needed only to replay the reference's golden object)
# --------------------------------------------------------------------------
def normalize_counts_by_seq_depth(counts):
    """R/inferCNV_ops.R:3064-3111: x / colSums * median(colSums)."""
    counts = np.asarray(counts, dtype=np.float64)
    cs = r_sum(counts, axis=0)
    return counts / cs[None, :] * float(r_median(cs, axis=0))


def log2xplus1(expr):
    """R/inferCNV_ops.R:2756-2769."""
    return np.log2(expr + 1.0)


def run_chain(expr, chr_codes, ref_groups, window_length=101, max_centered_threshold=3.0,
              sd_amplifier=1.5, denoise=True, return_pre_denoise=False):
    """Steps 8,9,10,11,12,14,(22) of run() -- R/inferCNV_ops.R:771-1589."""
    x = subtract_ref_expr_from_obs(expr, ref_groups, use_bounds=True)          # step 8
    if max_centered_threshold is not None:
        x = apply_max_threshold_bounds(x, max_centered_threshold)             # step 9
    x = smooth_by_chromosome(x, chr_codes, window_length)                     # step 10
    x = center_columns(x, "median")                                           # step 11
    x = subtract_ref_expr_from_obs(x, ref_groups, use_bounds=True)            # step 12
    x = invert_log2(x)                                                        # step 14
    pre = x
    if denoise:
        ref_idx = np.concatenate([np.asarray(g) for g in ref_groups])
        x = clear_noise_via_ref_mean_sd(x, ref_idx, sd_amplifier)             # step 22
    return (x, pre) if return_pre_denoise else x


def smooth_by_chromosome_na(expr, chr_codes, window_length):
    """smooth_by_chromosome on a matrix with NAs: every chromosome of more than one gene goes through .smooth_helper cell by
    cell, which strips and re-inserts the cell's NAs (R/inferCNV_ops.R:2406-2434, 2487-2489, 2529)."""
    out = np.array(expr, dtype=np.float64, copy=True)
    for idx in chr_segments(chr_codes):
        if idx.size > 1:
            for c in range(out.shape[1]):
                out[idx, c] = smooth_window_na(expr[idx, c], window_length)
    return out


def run_chain_na(expr, chr_codes, ref_groups, window_length=101, max_centered_threshold=3.0, sd_amplifier=1.5,
                 use_bounds=True, denoise=True, return_pre_denoise=False):
    """run_chain on a matrix that holds NAs, the way the reference's step functions treat them: `.subtract_expr` with bounds
    selects by which(x > hi) / which(x < lo) -- an NA (value or bound) is never selected, the entry comes out as 0
    (R/inferCNV_ops.R:1757-1768; the boolean masks of subtract_expr above do exactly that) --, `x[x > thr] <- thr` leaves an
    NA alone (:2974-2975), the smoothing strips and re-inserts NAs, the centre is median(na.rm = TRUE) (:2098), 2^NA = NA and
    step 22's which(x > lo & x < hi) never selects one (:2335)."""
    with np.errstate(invalid="ignore"):
        x = subtract_expr(expr, get_normal_gene_mean_bounds(expr, ref_groups), use_bounds)       # step 8
        if max_centered_threshold is not None:
            x = apply_max_threshold_bounds(x, max_centered_threshold)                            # step 9
        x = smooth_by_chromosome_na(x, chr_codes, window_length)                                 # step 10
        x = center_columns_na(x)                                                                 # step 11
        x = subtract_expr(x, get_normal_gene_mean_bounds(x, ref_groups), use_bounds)             # step 12
        x = invert_log2(x)                                                                       # step 14
        pre = x
        if denoise:
            ref_idx = np.concatenate([np.asarray(g) for g in ref_groups])
            mu, s = clear_noise_params_via_ref_mean_sd(x, ref_idx, sd_amplifier)
            x = clear_noise_bounds(x, mu, s)                                                     # step 22
    return (x, pre) if return_pre_denoise else x


# --------------------------------------------------------------------------
# A.5 pnorm(q>=0, log.p=TRUE, lower.tail=FALSE)  (R nmath pnorm_both, Cody 1969)
# --------------------------------------------------------------------------
_A = (2.2352520354606839287, 161.02823106855587881, 1067.6894854603709582,
      18154.981253343561249, 0.065682337918207449113)
_B = (47.20258190468824187, 976.09855173777669322, 10260.932208618978205,
      45507.789335026729956)
_C = (0.39894151208813466764, 8.8831497943883759412, 93.506656132177855979,
      597.27027639480026226, 2494.5375852903726711, 6848.1904505362823326,
      11602.651437647350124, 9842.7148383839780218, 1.0765576773720192317e-8)
_D = (22.266688044328115691, 235.38790178262499861, 1519.377599407554805,
      6485.558298266760755, 18615.571640885098091, 34900.952721145977266,
      38912.003286093271411, 19685.429676859990727)
_P = (0.21589853405795699, 0.1274011611602473639, 0.022235277870649807,
      0.001421619193227893466, 2.9112874951168792e-5, 0.02307344176494017303)
_Q = (1.28426009614491121, 0.468238212480865118, 0.0659881378689285515,
      0.00378239633202758244, 7.29751555083966205e-5)
_M_1_SQRT_2PI = 0.398942280401432677939946059934
_SQRT32 = 5.656854249492380195206754896838

def _fma(a, b, c):
    """NumPy has no fused multiply-add; the correctly rounded primitive comes from
    libm through the C oracle's helper (the *algorithm* below stays independent)."""
    import oracle_c
    return oracle_c.fma(a, b, c)


def icnv_log(x):
    """Natural log with a FIXED operation sequence (DESIGN.md "Arithmetic spec"):
    table-driven, constants from oracle/gen_log_table.py, explicit correctly
    rounded fma.  x = 2^k z; r = fma(z, invc, -1) (exact); w = k LN2HI + logc_hi
    (exact); hi = w + r; lo = ((w - hi) + r) + (k LN2LO + logc_lo);
    result = hi + fma(r r, B0 + r(B1 + ... + r B6), lo).  <= 0.52 ulp.
    R itself calls the platform libm log (implementation-defined at this level)."""
    import icnv_log_table as T
    x = np.asarray(x, dtype=np.float64)
    shp = x.shape
    x = np.atleast_1d(x).ravel().copy()
    ix = x.view(np.uint64).copy()
    special = (ix - np.uint64(0x0010000000000000)) >= np.uint64(0x7FE0000000000000)
    zero = (ix << np.uint64(1)) == 0
    pinf = ix == np.uint64(0x7FF0000000000000)
    bad = ((ix >> np.uint64(63)) != 0) | ((ix & np.uint64(0x7FF0000000000000)) == np.uint64(0x7FF0000000000000))
    sub = special & ~zero & ~pinf & ~bad
    with np.errstate(all="ignore"):
        xs = np.where(sub, x * 2.0 ** 52, x)
    ix = np.where(sub, xs.view(np.uint64) - np.uint64(52 << 52), ix)
    tmp = ix - np.uint64(T.OFF)
    i = ((tmp >> np.uint64(45)) & np.uint64(127)).astype(np.int64)
    k = (tmp.view(np.int64) >> 52)
    z = (ix - (tmp & np.uint64(0xFFF0000000000000))).view(np.float64)
    tab = np.array(T.TABLE)
    invc, lchi, lclo = tab[i, 0], tab[i, 1], tab[i, 2]
    with np.errstate(all="ignore"):
        r = _fma(z, invc, -1.0)
        kd = k.astype(np.float64)
        w = _fma(kd, T.LN2HI, lchi)
        hi = w + r
        lo = ((w - hi) + r) + (kd * T.LN2LO + lclo)
        r2 = r * r
        q = _fma(r, T.B[6], T.B[5])
        for j in (4, 3, 2, 1, 0):
            q = _fma(r, q, T.B[j])
        res = hi + _fma(r2, q, lo)
    res = np.where(zero, -np.inf, res)
    res = np.where(pinf, np.inf, res)
    res = np.where(bad & ~pinf, np.nan, res)
    return res.reshape(shp)


def pnorm_log_upper(y, log=icnv_log):
    """log P(Z > y) for y >= 0, following pnorm_both()'s three branches and its
    exact operation order (R src/nmath/pnorm.c; called at
    R/inferCNV_HMM.R:1129,1156 as pnorm(q, log.p=TRUE, lower.tail=FALSE))."""
    y = np.asarray(y, dtype=np.float64)
    out = np.empty_like(y)
    with np.errstate(all="ignore"):
        # branch (i): y <= 0.67448975
        xsq = y * y
        xnum = _A[4] * xsq
        xden = xsq
        for i in range(3):
            xnum = (xnum + _A[i]) * xsq
            xden = (xden + _B[i]) * xsq
        tmp1 = y * (xnum + _A[3]) / (xden + _B[3])
        r1 = log(0.5 - tmp1)
        # branch (ii): y <= sqrt(32)
        xnum = _C[8] * y
        xden = y
        for i in range(7):
            xnum = (xnum + _C[i]) * y
            xden = (xden + _D[i]) * y
        tmp2 = (xnum + _C[7]) / (xden + _D[7])
        # branch (iii)
        xsq3 = 1.0 / (y * y)
        xnum = _P[5] * xsq3
        xden = xsq3
        for i in range(4):
            xnum = (xnum + _P[i]) * xsq3
            xden = (xden + _Q[i]) * xsq3
        tmp3 = xsq3 * (xnum + _P[4]) / (xden + _Q[4])
        tmp3 = (_M_1_SQRT_2PI - tmp3) / y
        tmp = np.where(y <= _SQRT32, tmp2, tmp3)
        xs = np.trunc(y * 16.0) / 16.0
        dl = (y - xs) * (y + xs)
        r23 = (-xs * xs * 0.5) + (-dl * 0.5) + log(tmp)
        out = np.where(y <= 0.67448975, r1, r23)
    return out


# --------------------------------------------------------------------------
# A.4 HMM
# --------------------------------------------------------------------------
def get_HMM_i6(t=1e-6):
    """R/inferCNV_HMM.R:230-265 -> (Pi 6x6, delta)."""
    Pi = np.full((6, 6), t, dtype=np.float64)
    np.fill_diagonal(Pi, 1 - 5 * t)
    delta = np.array([t, t, 1 - 5 * t, t, t, t], dtype=np.float64)
    return Pi, delta


def get_HMM_i3(t=1e-6):
    """R/inferCNV_i3HMM.R:108-114 (diag is 1-5t although K=3; rows sum to 1-3t)."""
    Pi = np.full((3, 3), t, dtype=np.float64)
    np.fill_diagonal(Pi, 1 - 5 * t)
    delta = np.array([t, 1 - 5 * t, t], dtype=np.float64)
    return Pi, delta


def emission_scores(x, means, sd, log=icnv_log, long_double_sum=False):
    """R/inferCNV_HMM.R:1129-1133 / 1156-1160 for a vector of x: (len(x), K).
    long_double_sum: accumulate sum(emission) in the platform's long double (80-bit on x86) and round the total to
    double once, as R's sum() does (rsum, src/main/summary.c); the default is the spec'd left-to-right double sum
    that the C oracle and the HIP kernels execute (DESIGN.md "Arithmetic spec")."""
    x = np.asarray(x, dtype=np.float64)
    z = np.abs(x[..., None] - means) / sd
    lp = pnorm_log_upper(z, log=log)
    e = 1.0 / (-1.0 * lp)
    if long_double_sum:
        tot = e[..., 0].astype(np.longdouble)
        for k in range(1, e.shape[-1]):
            tot = tot + e[..., k].astype(np.longdouble)
        tot = tot.astype(np.float64)
    else:
        tot = e[..., 0].copy()
        for k in range(1, e.shape[-1]):      # sum() left to right (long double in R;
            tot = tot + e[..., k]            # the double sum is the spec'd order here)
    e = e / tot[..., None]
    return log(e)


def viterbi_dthmm_adj(x, means, sd_vec, Pi, delta, log=icnv_log, long_double_sum=False):
    """R/inferCNV_HMM.R:1101-1176 for a batch: x is (n, S) -- S independent
    sequences of one chromosome.  Returns 1-based states (n, S) int8 and a
    flag array (S,) that is True where the reference would stop() with
    'Problems With Underflow' (:1165)."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    n, S = x.shape
    if n < 2:
        return np.full((n, S), 3, dtype=np.int8), np.zeros(S, dtype=bool)   # :1104-1107
    means = np.asarray(means, dtype=np.float64)
    K = means.size
    sd = float(r_median(np.asarray(sd_vec, dtype=np.float64)))               # :1122
    with np.errstate(divide="ignore"):
        logPi = np.log(np.asarray(Pi, dtype=np.float64))
        logdelta = np.log(np.asarray(delta, dtype=np.float64))
    return viterbi_core(x, means, sd, logPi, logdelta, log=log, long_double_sum=long_double_sum)


def viterbi_core(x, means, sd, logPi, logdelta, log=icnv_log, long_double_sum=False, return_margin=False):
    """The DP itself, with log(Pi), log(delta) and the shared sd already
    prepared on the host (this is what the C-ABI entry point receives)."""
    n, S = x.shape
    K = means.size
    if n < 2:
        return np.full((n, S), 3, dtype=np.int8), np.zeros(S, dtype=bool)
    bp = np.zeros((n, S, K), dtype=np.int8)
    nu = logdelta[None, :] + emission_scores(x[0], means, sd, log=log, long_double_sum=long_double_sum)   # (S, K)
    margin = np.full(S, np.inf)     # smallest lead of a winning candidate over the runner-up, any gene / state
    for i in range(1, n):
        sc = emission_scores(x[i], means, sd, log=log, long_double_sum=long_double_sum)
        cand = nu[:, :, None] + logPi[None, :, :]                            # [s, j, k]
        bp[i] = np.argmax(cand, axis=1)                                      # first max over j
        if return_margin:
            top2 = np.sort(cand, axis=1)[:, -2:, :]
            with np.errstate(invalid="ignore"):
                margin = np.fmin(margin, np.min(top2[:, 1, :] - top2[:, 0, :], axis=1))
        nu = np.max(cand, axis=1) + sc
    bad = np.any(nu == -np.inf, axis=1)
    if return_margin:
        top2 = np.sort(nu, axis=1)[:, -2:]
        with np.errstate(invalid="ignore"):
            margin = np.fmin(margin, top2[:, 1] - top2[:, 0])
    y = np.zeros((n, S), dtype=np.int64)
    y[n - 1] = np.argmax(nu, axis=1)
    rows = np.arange(S)
    for i in range(n - 2, -1, -1):
        # which.max(logPi[, y[i+1]] + nu[i, ]) == argmax_j(nu[i-th row][j] + logPi[j, y]) ==
        # the back-pointer recorded when row i+1 was computed (same addends, first max)
        y[i] = bp[i + 1, rows, y[i + 1]]
    if return_margin:
        return (y + 1).astype(np.int8), bad, margin
    return (y + 1).astype(np.int8), bad


def viterbi_scalar(x, means, sd, logPi, logdelta, log=icnv_log):
    """Literal single-sequence transcription of R/inferCNV_HMM.R:1101-1176 that
    keeps the full nu matrix and does the reference's own traceback."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    if n < 2:
        return np.full(n, 3, dtype=np.int8)
    K = len(means)
    nu = np.zeros((n, K))
    nu[0] = logdelta + emission_scores(x[0:1], means, sd, log=log)[0]
    for i in range(1, n):
        m = nu[i - 1][:, None] + logPi            # [j, k]
        nu[i] = m.max(axis=0) + emission_scores(x[i:i + 1], means, sd, log=log)[0]
    y = np.zeros(n, dtype=np.int64)
    y[n - 1] = int(np.argmax(nu[n - 1]))
    for i in range(n - 2, -1, -1):
        y[i] = int(np.argmax(logPi[:, y[i + 1]] + nu[i]))
    return (y + 1).astype(np.int8)


def predict_cnv_on_indiv_cells(expr, chr_codes, means, sd_vec, Pi, delta, log=icnv_log):
    """R/inferCNV_HMM.R:284-324 (i6) and R/inferCNV_i3HMM.R:180-225 (i3)."""
    G, C = expr.shape
    states = np.full((G, C), -1, dtype=np.int8)
    for idx in chr_segments(chr_codes):
        st, _ = viterbi_dthmm_adj(expr[idx], means, sd_vec, Pi, delta, log=log)
        states[idx] = st
    return states


def group_means(expr, groups):
    """rowMeans(expr.data[chr_gene_idx, group_cells]) -- R/inferCNV_HMM.R:383."""
    return np.stack([r_row_means(expr[:, np.asarray(g, dtype=np.int64)]) for g in groups], axis=1)


def predict_cnv_on_groups(expr, chr_codes, groups, means, sd_vec_per_group, Pi, delta, log=icnv_log):
    """R/inferCNV_HMM.R:345-408, 509-567 and R/inferCNV_i3HMM.R:249-389: Viterbi
    on the per-group mean profile, trace broadcast to every member cell."""
    G, C = expr.shape
    states = np.full((G, C), -1, dtype=np.int8)
    gm = group_means(expr, groups)
    for idx in chr_segments(chr_codes):
        for gi, g in enumerate(groups):
            st, _ = viterbi_dthmm_adj(gm[idx, gi], means, sd_vec_per_group[gi], Pi, delta, log=log)
            states[np.ix_(idx, np.asarray(g, dtype=np.int64))] = st
    return states


_I6_PROXY = np.array([np.nan, 0.0, 0.5, 1.0, 1.5, 2.0, 3.0])
_I3_PROXY = np.array([np.nan, 0.5, 1.0, 1.5])


def assign_HMM_states_to_proxy_expr_vals(states):
    """R/inferCNV_HMM.R:1191-1206."""
    return _I6_PROXY[np.asarray(states, dtype=np.int64)]


def i3HMM_assign_HMM_states_to_proxy_expr_vals(states):
    """R/inferCNV_i3HMM.R:405-417."""
    return _I3_PROXY[np.asarray(states, dtype=np.int64)]


def i3_params(expr, ref_idx, i3_p_val=0.05):
    """R/inferCNV_i3HMM.R:17-80, 435-445 (use_KS=FALSE): mu, sigma over all
    values of the reference cells; delta = |qnorm(p, 0, sigma)|."""
    from scipy.special import ndtri
    vals = expr[:, np.asarray(ref_idx, dtype=np.int64)].ravel()
    mu = float(r_mean(vals))
    sigma = float(r_sd(vals[:, None], axis=0)[0])
    delta = abs(float(ndtri(i3_p_val)) * sigma)
    return mu, sigma, delta


# --------------------------------------------------------------------------
# A.6 median filter
# --------------------------------------------------------------------------
def median_filter(data, window_size):
    """R/noise_reduction.R:92-113 on one (n_chr_genes, n_tile_cells) tile."""
    half = (window_size - 1) // 2
    xdim, ydim = data.shape
    out = data.copy()
    for px in range(1, xdim + 1):
        xa = 1 if px <= half + 1 else px - (half + 1)
        xb = xdim if px >= xdim - (half + 1) else px + (half + 1)
        for py in range(1, ydim + 1):
            ya = 1 if py <= half + 1 else py - (half + 1)
            yb = ydim if py >= ydim - (half + 1) else py + (half + 1)
            out[px - 1, py - 1] = r_median(data[xa - 1:xb, ya - 1:yb].ravel(), axis=0)
    return out


def apply_median_filtering(expr, chr_codes, tiles, window_size=7):
    """R/noise_reduction.R:43-89.  `tiles` = list of 0-based cell index vectors
    (each tumour subcluster, each whole reference group), in stored order."""
    out = expr.copy()
    for idx in chr_segments(chr_codes):
        for t in tiles:
            t = np.asarray(t, dtype=np.int64)
            out[np.ix_(idx, t)] = median_filter(expr[np.ix_(idx, t)], window_size)
    return out


# --------------------------------------------------------------------------
# CNV region consensus
# --------------------------------------------------------------------------
def state_consensus(states, groups):
    """.get_state_consensus (R/inferCNV_HMM.R:977-987): per gene the most frequent state among a group's
    cells; table() orders states ascending and order(decreasing=TRUE)[1] keeps the first maximum."""
    states = np.asarray(states)
    out = np.zeros((states.shape[0], len(groups)))
    for gi, g in enumerate(groups):
        sub = states[:, np.asarray(g, dtype=np.int64)]
        for r in range(states.shape[0]):
            vals, cnt = np.unique(sub[r], return_counts=True)      # ascending values
            out[r, gi] = vals[np.argmax(cnt)]                      # first maximum
    return out


def define_cnv_gene_regions(state_consensus_vec, chrs, counter):
    """.define_cnv_gene_regions (R/inferCNV_HMM.R:1005-1057) as the plain loop the reference runs:
    returns [(region name, state, [0-based gene indices])] and the advanced counter."""
    regions = []
    seen = []
    for c in chrs:
        if c not in seen:
            seen.append(c)
    for c in seen:
        gene_idx = [i for i, cc in enumerate(chrs) if cc == c]
        if len(gene_idx) < 2:
            continue
        prev = state_consensus_vec[gene_idx[0]]
        counter += 1
        regions.append(["%s-region_%d" % (c, counter), prev, [gene_idx[0]]])
        for i in gene_idx[1:]:
            st = state_consensus_vec[i]
            if st != prev:
                counter += 1
                regions.append(["%s-region_%d" % (c, counter), st, [i]])
            else:
                regions[-1][2].append(i)
            prev = st
    return regions, counter


# --------------------------------------------------------------------------
# gene filters (step 2) and spike-in emission statistics
# --------------------------------------------------------------------------
def below_min_mean_expr_cutoff(expr, min_mean_expr):
    """.below_min_mean_expr_cutoff (R/inferCNV_ops.R:2154-2163): which(rowMeans(expr) < cutoff), 0-based."""
    return np.nonzero(r_sum(expr, axis=1) / expr.shape[1] < min_mean_expr)[0]


def genes_passing_min_cells(expr, min_cells_per_gene):
    """require_above_min_cells_ref (R/inferCNV_ops.R:2182-2184): sum(x > 0 & !is.na(x)) >= min_cells, 0-based."""
    with np.errstate(invalid="ignore"):
        return np.nonzero((np.asarray(expr) > 0).sum(axis=1) >= min_cells_per_gene)[0]


def gene_expr_mean_sd(expr, gene_idx, cell_idx):
    """.get_gene_expr_mean_sd_by_cnv (R/inferCNV_HMM.R:84-99): mean() and sd() of c(expr[genes, cells])."""
    v = np.asarray(expr, dtype=np.float64)[np.ix_(np.asarray(gene_idx), np.asarray(cell_idx))].ravel(order="F")
    return float(r_mean(v)), float(r_sd(v.reshape(-1, 1), axis=0)[0])


# ----------------------------------------------------------------------------
# cell-cell distances
def cell_distances(expr, cells):
    """parallelDist(t(expr[, cells])), method "euclidean" (R/inferCNV_tumor_subclusters.R:191): direct sums of squared
    differences, full symmetric matrix."""
    x = np.asarray(expr, dtype=np.float64)[:, np.asarray(cells, dtype=np.int64)].T      # (n, G)
    n = x.shape[0]
    d = np.zeros((n, n))
    for i in range(n):
        diff = x[i + 1:] - x[i]
        d[i, i + 1:] = np.sqrt((diff * diff).sum(axis=1))
    return d + d.T


# ---------------------------------------------------------------------------------------------------------------------
# R's RNG and the sd-vs-cell-count resampling fit (R/inferCNV_HMM.R:154-212), restated independently of
# infercnv_amd/r_rng.py: the Mersenne-Twister recurrence is written out here (Matsumoto & Nishimura's genrand_int32, what
# base R's MT_genrand runs), the R-specific parts -- seed scrambling, rbits / rejection sampling, sample(), replicate,
# rowMeans, sd, lm -- as scalar loops.
class RMersenne:
    """`set.seed(seed)` of base R (src/main/RNG.c: RNG_Init + FixupSeeds), Mersenne-Twister, one output at a time."""
    N, M = 624, 397

    def __init__(self, seed):
        s = int(seed) & 0xFFFFFFFF
        for _ in range(50):
            s = (69069 * s + 1) & 0xFFFFFFFF
        seedvec = []
        for _ in range(625):
            s = (69069 * s + 1) & 0xFFFFFFFF
            seedvec.append(s)
        self.mt = np.array(seedvec[1:], dtype=np.uint64)     # seedvec[0] is mti, set to N by FixupSeeds
        self.mti = self.N
        self.buf = np.zeros(0, dtype=np.uint64)
        self.pos = 0

    def _regenerate(self):
        mt, N, M = self.mt, self.N, self.M
        UP, LO, A = np.uint64(0x80000000), np.uint64(0x7FFFFFFF), np.uint64(0x9908B0DF)
        def twist(u, v):
            y = (u & UP) | (v & LO)
            return (y >> np.uint64(1)) ^ np.where((y & np.uint64(1)) != 0, A, np.uint64(0))
        # the recurrence reads words it has already renewed once kk >= N - M: three dependent segments
        mt[0:N - M] = mt[M:N] ^ twist(mt[0:N - M], mt[1:N - M + 1])
        for a in range(N - M, N - 1, N - M):                  # segments of at most N - M words, each reading renewed ones
            b = min(a + (N - M), N - 1)
            mt[a:b] = mt[a - (N - M):b - (N - M)] ^ twist(mt[a:b], mt[a + 1:b + 1])
        y = (mt[N - 1] & UP) | (mt[0] & LO)
        mt[N - 1] = mt[M - 1] ^ (y >> np.uint64(1)) ^ (A if (int(y) & 1) else np.uint64(0))
        y = mt.copy()
        y ^= y >> np.uint64(11)
        y ^= (y << np.uint64(7)) & np.uint64(0x9D2C5680)
        y ^= (y << np.uint64(15)) & np.uint64(0xEFC60000)
        y ^= y >> np.uint64(18)
        self.buf = y & np.uint64(0xFFFFFFFF)
        self.pos = 0

    def genrand_int32(self):
        if self.pos >= self.buf.size:
            self._regenerate()
        v = int(self.buf[self.pos])
        self.pos += 1
        return v

    def unif_rand(self):
        u = self.genrand_int32() * 2.3283064365386963e-10
        if u <= 0.0:
            return 0.5 * 2.328306437080797e-10
        if 1.0 - u <= 0.0:
            return 1.0 - 0.5 * 2.328306437080797e-10
        return u

    def unif_index(self, dn):
        """R_unif_index, sample.kind = "Rejection" (src/main/RNG.c)."""
        if dn <= 0:
            return 0
        bits = int(np.ceil(np.log2(dn)))
        while True:
            v, n = 0, 0
            while n <= bits:
                v = 65536 * v + int(np.floor(self.unif_rand() * 65536))
                n += 16
            if bits < 64:
                v &= (1 << bits) - 1
            if v < dn:
                return v


def r_norm_rand(rng):
    """norm_rand() of base R with N01_kind = INVERSION (src/nmath/snorm.c): BIG = 2^27; u = unif_rand();
    u = (int)(BIG * u) + unif_rand(); qnorm5(u / BIG, 0, 1, 1, 0).  qnorm5 is Wichura's AS 241 (PPND16) -- the routine CPython's
    statistics.NormalDist.inv_cdf implements too, used here as the independent evaluation."""
    import statistics
    u = float(int(134217728.0 * rng.unif_rand())) + rng.unif_rand()
    return statistics.NormalDist().inv_cdf(u / 134217728.0)


def r_rnorm(rng, n, mean, sd):
    return np.array([mean + sd * r_norm_rand(rng) for _ in range(n)], dtype=np.float64)


def r_ks_test_p_value(x, y):
    """ks.test(x, y)$p.value, two-sided, no ties (src/library/stats/R/ks.test.R; C code src/library/stats/src/ks.c:
    psmirnov2x for n.x * n.y < 10000, pKS2 with tol = 1e-6 otherwise) -- scalar loops, test infrastructure."""
    nx, ny = len(x), len(y)
    tagged = sorted([(v, 0) for v in x] + [(v, 1) for v in y])
    z, stat = 0.0, 0.0
    for _, which in tagged:
        z += (1.0 / nx) if which == 0 else (-1.0 / ny)
        stat = max(stat, abs(z))
    if nx * ny < 10000:
        m, n = min(nx, ny), max(nx, ny)
        md, nd = float(m), float(n)
        q = (0.5 + math.floor(stat * md * nd - 1e-7)) / (md * nd)
        u = [0.0 if (j / nd) > q else 1.0 for j in range(n + 1)]
        for i in range(1, m + 1):
            w = i / (i + nd)
            u[0] = 0.0 if (i / md) > q else w * u[0]
            for j in range(1, n + 1):
                u[j] = 0.0 if abs(i / md - j / nd) > q else w * u[j] + u[j - 1]
        p = 1.0 - u[n]
    else:
        xs = math.sqrt(nx * ny / (nx + ny)) * stat
        if xs <= 0:
            cdf = 0.0
        elif xs < 1:
            k_max = int(math.sqrt(2 - math.log(1e-6)))
            cdf = sum(math.exp(k * k * (-(math.pi / 2 * math.pi / 4) / (xs * xs)) - math.log(xs)) for k in range(1, k_max, 2)) \
                / 0.398942280401432677939946059934
        else:
            s, k, old, new = -1.0, 1, 0.0, 1.0
            while abs(old - new) > 1e-6:
                old = new
                new += 2 * s * math.exp(-2.0 * xs * xs * k * k)
                s, k = -s, k + 1
            cdf = new
        p = 1.0 - cdf
    return min(1.0, max(0.0, p))


def honeybadger_set_gexp_dev(gexp_sd, alpha, k_cells, seed, n_iter=100):
    """get_HoneyBADGER_setGexpDev (R/inferCNV_i3HMM.R:469-493) after set.seed(seed): scalar restatement (RMersenne stream)."""
    rng = RMersenne(seed)
    k_cells = max(int(k_cells), 2)
    by = gexp_sd / 10.0
    n = int(gexp_sd / by + 1e-10)
    devs = [min(i * by, gexp_sd) for i in range(n + 1)]
    pvs = []
    for dev in devs:
        acc = 0.0
        for _ in range(n_iter):
            a = r_rnorm(rng, k_cells, 0.0, gexp_sd)
            b = r_rnorm(rng, k_cells, dev, gexp_sd)
            acc += r_ks_test_p_value(a, b)
        pvs.append(acc / n_iter)
    px, dy = np.asarray(pvs), np.asarray(devs)
    slope = ((px - px.mean()) * (dy - dy.mean())).sum() / ((px - px.mean()) ** 2).sum()
    return float(dy.mean() - slope * px.mean() + slope * alpha)


def hspike_sd_trend_fit(expr_vals_by_level, seed, nrounds=100, max_cells=100):
    """R/inferCNV_HMM.R:154-212 on {level: vector of residuals}: returns {level: (sds[max_cells], (intercept, slope))}."""
    rng = RMersenne(seed)
    out = {}
    for level, ev in expr_vals_by_level.items():
        ev = np.asarray(ev, dtype=np.float64)
        sds = np.full(max_cells, np.nan)
        for ncells in range(1, max_cells + 1):
            vals = np.empty((ncells, nrounds))                 # replicate(): one column per round
            for r in range(nrounds):
                for i in range(ncells):
                    vals[i, r] = ev[rng.unif_index(ev.size)]   # sample(expr_vals, size = ncells, replace = TRUE)
            if ncells > 1:
                sds[ncells - 1] = float(r_sd(r_row_means(vals), axis=0))
        ok = ~np.isnan(sds)                                     # lm(): na.omit
        X = np.stack([np.ones(int(ok.sum())), np.log(np.arange(1, max_cells + 1)[ok])], axis=1)
        coef, *_ = np.linalg.lstsq(X, np.log(sds[ok]), rcond=None)
        out[level] = (sds, (float(coef[0]), float(coef[1])))
    return out
