"""Host-side mirror of the reference's HMM predictors (R/inferCNV_HMM.R,
R/inferCNV_i3HMM.R): same names and argument meaning; parameters are prepared
on the host exactly as the reference prepares them in R, the Viterbi runs in
libicnv_hip.so.
"""
from __future__ import annotations

import ctypes as ct
import math
import statistics

import numpy as np

from . import _lib
from ._lib import check, f64, i32, pack_groups
from .infercnv_object import InfercnvObject

CNV_LEVELS = ("cnv:0.01", "cnv:0.5", "cnv:1", "cnv:1.5", "cnv:2", "cnv:3")


def _get_HMM(cnv_mean_sd, t):
    """.get_HMM (R/inferCNV_HMM.R:230-265)."""
    Pi = np.full((6, 6), t, dtype=np.float64)
    np.fill_diagonal(Pi, 1 - 5 * t)
    delta = np.array([t, t, 1 - 5 * t, t, t, t], dtype=np.float64)
    mean = np.array([cnv_mean_sd[k]["mean"] for k in CNV_LEVELS], dtype=np.float64)
    sd = np.array([cnv_mean_sd[k]["sd"] for k in CNV_LEVELS], dtype=np.float64)
    return {"state_transitions": Pi, "delta": delta, "state_emission_params": {"mean": mean, "sd": sd}}


def _i3HMM_get_HMM(sd_trend, t, i3_p_val=0.05, use_KS=False):
    """.i3HMM_get_HMM (R/inferCNV_i3HMM.R:99-156): note the 1-5t diagonal with K=3."""
    Pi = np.full((3, 3), t, dtype=np.float64)
    np.fill_diagonal(Pi, 1 - 5 * t)
    delta = np.array([t, 1 - 5 * t, t], dtype=np.float64)
    mu, sigma = sd_trend["mu"], sd_trend["sigma"]
    if use_KS and sd_trend.get("KS_delta") is None:
        raise NotImplementedError(
            "use_KS=TRUE: the KS-based mean delta (get_HoneyBADGER_setGexpDev, R/inferCNV_i3HMM.R:469-493) is estimated "
            "from rnorm() draws of R's RNG stream: give its state (seed=..., as in set.seed), pass sd_trend['KS_delta'], or "
            "use use_KS=False")
    mean_delta = sd_trend["KS_delta"] if use_KS else sd_trend["mean_delta"]
    return {"state_transitions": Pi, "delta": delta,
            "state_emission_params": {"mean": np.array([mu - mean_delta, mu, mu + mean_delta]),
                                      "sd": np.array([sigma, sigma, sigma])}}


def determine_mean_delta_via_Z(sigma, p):
    """R/inferCNV_i3HMM.R:435-445: abs(qnorm(p, 0, sigma))."""
    return abs(statistics.NormalDist(0.0, sigma).inv_cdf(p))


HSPIKE_CHR_INFO = (("chrA", 1), ("chr_0", 0.01), ("chr_B", 1), ("chr_0pt5", 0.5), ("chr_C", 1), ("chr_1pt5", 1.5),
                   ("chr_D", 1), ("chr_2pt0", 2.0), ("chr_E", 1), ("chr_3pt0", 3), ("chr_F", 1))
"""(name, cnv) of the fake chromosomes of the hidden spike-in (.get_hspike_chr_info, R/inferCNV_hidden_spike.R:170-215)."""


def get_spike_dists(hspike_obj: InfercnvObject, chr_info=HSPIKE_CHR_INFO):
    """get_spike_dists (R/inferCNV_HMM.R:15-99): per CNV level the mean and sd of the residual expression of the
    spiked (observation) cells over the genes of that level's fake chromosomes -- one block reduction on the
    device per level.  Returns {"cnv:1": {"mean":, "sd":}, ...} in order of first appearance."""
    if hspike_obj is None:
        raise ValueError("get_spike_dists(hspike_obj): Error, hspike obj is null")
    L = _lib.load()
    x = np.asfortranarray(hspike_obj.expr_data, dtype=np.float64)
    G, C = x.shape
    cells = np.concatenate([np.asarray(v, dtype=np.int32) for v in hspike_obj.observation_grouped_cell_indices.values()])
    chrs = np.asarray(hspike_obj.gene_order.chr)
    genes_by_cnv = {}
    for name, cnv in chr_info:
        key = "cnv:%g" % cnv
        genes_by_cnv.setdefault(key, []).append(np.nonzero(chrs == name)[0])
    out = {}
    for key, parts in genes_by_cnv.items():
        gi, gp = i32(np.concatenate(parts))
        if gi.size == 0:
            continue
        ci, cp = i32(cells)
        buf = (ct.c_double * 2)()
        check(L.icnv_block_mean_sd(x.ctypes.data_as(ct.c_void_p), G, C, gp, gi.size, cp, ci.size, buf))
        out[key] = {"mean": buf[0], "sd": buf[1]}
    return out


def _hspike_level_offsets(hspike_obj, chr_info=HSPIKE_CHR_INFO):
    """.get_gene_expr_by_cnv (R/inferCNV_HMM.R:45-68) as element offsets: level -> int64 offsets (g + G c) of
    c(spike.expr.data[chr_gene_idx, ]) for every fake chromosome of the level, concatenated in chr_info order."""
    G = hspike_obj.expr_data.shape[0]
    cells = np.concatenate([np.asarray(v, dtype=np.int64) for v in hspike_obj.observation_grouped_cell_indices.values()])
    chrs = np.asarray(hspike_obj.gene_order.chr)
    out = {}
    for name, cnv in chr_info:
        genes = np.nonzero(chrs == name)[0].astype(np.int64)
        block = (genes[:, None] + G * cells[None, :]).ravel(order="F")       # column-major flatten: genes fastest
        key = "cnv:%g" % cnv
        out[key] = np.concatenate([out[key], block]) if key in out else block
    return out


def get_hspike_cnv_mean_sd_trend_by_num_cells_fit(hspike_obj: InfercnvObject, seed=None, rng=None, chr_info=HSPIKE_CHR_INFO,
                                                  nrounds=100, max_cells=100):
    """get_hspike_cnv_mean_sd_trend_by_num_cells_fit (R/inferCNV_HMM.R:154-212): for every CNV level of the hidden
    spike-in and every ncells in 1..100, `vals <- replicate(100, sample(expr_vals, size = ncells, replace = TRUE))`,
    `sd(rowMeans(vals))` (for ncells = 1 `vals` is a vector, `means` a single number and its sd NA), then
    `lm(log(sd) ~ log(num_cells))` per level.  Reference behaviour kept: rowMeans runs over the ROUNDS, not over the cells
    of a round, so every sd estimates sigma / sqrt(100) and the fitted slope is ~0.

    The draws are R's own stream -- `seed` as in set.seed(seed) right before the call (infercnv_amd/r_rng.py); the
    reference never seeds, so an unseeded run of it is not reproducible either.  The 3 M sampled residuals are gathered
    from the hidden-spike matrix on the device in one call (icnv_gather_values); sd and the 99-point regressions are
    host arithmetic in R's precision (long double accumulation).  Returns {level: (intercept, slope)} -- what
    `.get_state_emission_params` predicts from -- plus the sd vectors under the key "_sd"."""
    from .r_rng import RRandom
    if rng is None:
        if seed is None:
            raise ValueError("give the RNG state: seed (as in set.seed(seed)) or an RRandom")
        rng = RRandom(seed)
    offs = _hspike_level_offsets(hspike_obj, chr_info)
    draws = {}                                                   # (nrounds, max_cells: the reference's constants, :160-162)
    for level, o in offs.items():                                # names(gene_expr_by_cnv) order; ncells 1..100; round; draw
        if o.size == 0:
            raise ValueError(f"no hidden-spike values for {level}")
        draws[level] = o[rng.sample_replace(o.size, nrounds * (max_cells * (max_cells + 1) // 2))]
    allo = np.ascontiguousarray(np.concatenate(list(draws.values())), dtype=np.int64)
    vals = np.empty(allo.size, dtype=np.float64)
    L = _lib.load()
    x = np.asfortranarray(hspike_obj.expr_data, dtype=np.float64)
    check(L.icnv_gather_values(x.ctypes.data_as(ct.c_void_p), x.shape[0], x.shape[1], allo.ctypes.data_as(ct.POINTER(ct.c_int64)),
                               allo.size, vals.ctypes.data_as(ct.POINTER(ct.c_double))))
    LD = np.longdouble
    fits, sds_all, pos = {}, {}, 0
    for level in offs:
        sds = np.full(max_cells, np.nan)
        for ncells in range(1, max_cells + 1):
            v = vals[pos:pos + nrounds * ncells].reshape(nrounds, ncells)   # round r, draw i  (R: vals[i, r])
            pos += nrounds * ncells
            if ncells == 1:
                continue                                          # means is one number: sd(means) is NA
            means = np.asarray(v.astype(LD).sum(axis=0) / LD(nrounds), dtype=np.float64)    # rowMeans(vals): over the rounds
            m = means.astype(LD).sum() / LD(ncells)
            m = m + (means.astype(LD) - m).sum() / LD(ncells)                              # mean(): one refinement pass
            sds[ncells - 1] = float(np.sqrt(((means.astype(LD) - m) ** 2).sum() / LD(ncells - 1)))
        ok = ~np.isnan(sds)                                       # lm drops the NA row (na.action = na.omit)
        lx, ly = np.log(np.arange(1, max_cells + 1, dtype=np.float64)[ok]).astype(LD), np.log(sds[ok]).astype(LD)
        mx, my = lx.mean(), ly.mean()
        slope = ((lx - mx) * (ly - my)).sum() / ((lx - mx) ** 2).sum()
        fits[level] = (float(my - slope * mx), float(slope))
        sds_all[level] = sds
    fits["_sd"] = sds_all
    return fits


def _log(a):
    with np.errstate(divide="ignore"):
        return np.log(np.asarray(a, dtype=np.float64))


def _median(v):
    """stats::median of the sd vector (R/inferCNV_HMM.R:1122)."""
    v = np.sort(np.asarray(v, dtype=np.float64))
    n = v.size
    return float(v[n // 2]) if n % 2 else float((v[n // 2 - 1] + v[n // 2]) * 0.5)


def _layout(obj):
    perm, chr_start = obj.chr_layout()
    x = np.asfortranarray(obj.expr_data if perm is None else obj.expr_data[perm], dtype=np.float64)
    return perm, chr_start, x


def _unpermute(states, perm):
    if perm is None:
        return states
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    return states[inv]


def Viterbi_dthmm_adj(x, Pi, delta, mean, sd):
    """Viterbi.dthmm.adj (R/inferCNV_HMM.R:1101-1176) for one observation vector."""
    x = np.asarray(x, dtype=np.float64).reshape(-1, 1)
    st = _viterbi_cells(np.asfortranarray(x), np.array([0, x.shape[0]], dtype=np.int32), mean, _median(sd), Pi, delta)
    return st[:, 0].astype(np.int64)


def _viterbi_cells(x, chr_start, mean, sd_shared, Pi, delta):
    L = _lib.load()
    G, C = x.shape
    cs, cp = i32(chr_start)
    m, mp = f64(mean)
    lp = np.asfortranarray(_log(Pi))
    ld, ldp = f64(_log(delta))
    st = np.empty((G, C), dtype=np.uint8, order="F")
    check(L.icnv_viterbi_cells(x.ctypes.data_as(ct.c_void_p), st.ctypes.data_as(ct.c_void_p), G, C, cp, cs.size - 1,
                               m.size, mp, float(sd_shared), lp.ctypes.data_as(ct.POINTER(ct.c_double)), ldp))
    return st


def _viterbi_groups(x, chr_start, groups, mean, sd_per_group, Pi, delta):
    L = _lib.load()
    G, C = x.shape
    cs, cp = i32(chr_start)
    idx, off = pack_groups(groups)
    idx, ip = i32(idx)
    off, op = i32(off)
    m, mp = f64(mean)
    sd, sdp = f64(sd_per_group)
    lp = np.asfortranarray(_log(Pi))
    ld, ldp = f64(_log(delta))
    st = np.empty((G, C), dtype=np.uint8, order="F")
    check(L.icnv_viterbi_groups(x.ctypes.data_as(ct.c_void_p), st.ctypes.data_as(ct.c_void_p), G, C, cp, cs.size - 1,
                                ip, op, len(groups), m.size, mp, sdp, lp.ctypes.data_as(ct.POINTER(ct.c_double)), ldp))
    return st


def _states_obj(obj, st, perm):
    """The reference stores states as numeric with -1 for untouched entries
    (R/inferCNV_HMM.R:294-295)."""
    out = _unpermute(st, perm).astype(np.float64)
    out[out == 255] = -1.0
    new = obj.copy()
    new.expr_data = out
    return new


# ------------------------------------------------------------------ i6
def predict_CNV_via_HMM_on_indiv_cells(infercnv_obj: InfercnvObject, cnv_mean_sd, t=1e-6) -> InfercnvObject:
    """R/inferCNV_HMM.R:284-324.  `cnv_mean_sd` = get_spike_dists(hspike) as a dict
    {"cnv:0.01": {"mean":, "sd":}, ...} (host-side, R/inferCNV_HMM.R:15-31)."""
    hmm = _get_HMM(cnv_mean_sd, t)
    perm, chr_start, x = _layout(infercnv_obj)
    pm = hmm["state_emission_params"]
    st = _viterbi_cells(x, chr_start, pm["mean"], _median(pm["sd"]), hmm["state_transitions"], hmm["delta"])
    return _states_obj(infercnv_obj, st, perm)


def _group_sd(num_cells, cnv_mean_sd, cnv_level_to_mean_sd_fit):
    """.get_state_emission_params (R/inferCNV_HMM.R:586-614): sd_k =
    exp(predict(lm(log(sd) ~ log(num_cells)))).  `fit[level]` = (intercept, slope)
    of that lm (the RNG-driven fit itself stays on the host, :154-212)."""
    sds = []
    for k in CNV_LEVELS:
        if cnv_level_to_mean_sd_fit is None:
            sds.append(cnv_mean_sd[k]["sd"])
        else:
            b0, b1 = cnv_level_to_mean_sd_fit[k]
            sds.append(math.exp(b0 + b1 * math.log(num_cells)))
    return _median(sds)


def _predict_groups_i6(obj, groups, cnv_mean_sd, fit, t):
    hmm = _get_HMM(cnv_mean_sd, t)
    perm, chr_start, x = _layout(obj)
    sd = [_group_sd(len(g), cnv_mean_sd, fit) for g in groups]
    st = _viterbi_groups(x, chr_start, groups, hmm["state_emission_params"]["mean"], sd,
                         hmm["state_transitions"], hmm["delta"])
    return _states_obj(obj, st, perm)


def _flatten_subclusters(obj):
    """unlist(tumor_subclusters[["subclusters"]], recursive=FALSE) (R/inferCNV_HMM.R:371)."""
    out = []
    for grp in obj.tumor_subclusters["subclusters"].values():
        for idx in grp.values():
            out.append(np.asarray(idx, dtype=np.int32))
    return out


def _whole_sample_groups(obj, cluster_by_groups):
    """`tumor_samples` of R/inferCNV_HMM.R:528-533 (i3: R/inferCNV_i3HMM.R:351-356).  With cluster_by_groups = FALSE the
    reference writes `c(all_observations = unlist(obs_indices), reference_grouped_cell_indices)`: c() of an integer
    VECTOR with a list yields a list with one element per vector entry, so every observation cell becomes a "sample"
    of its own (num_cells = 1: the sd of one cell, the Viterbi on the cell's own profile) next to the reference groups.
    Reference behaviour, kept (SURVEY.md appendix A.7 lists the others)."""
    if cluster_by_groups:
        groups = list(obj.observation_grouped_cell_indices.values())
    else:
        groups = [np.asarray([c]) for v in obj.observation_grouped_cell_indices.values() for c in np.asarray(v).ravel()]
    groups += list(obj.reference_grouped_cell_indices.values())
    return [np.asarray(g, dtype=np.int32) for g in groups]


def predict_CNV_via_HMM_on_whole_tumor_samples(infercnv_obj, cluster_by_groups, cnv_mean_sd,
                                               cnv_level_to_mean_sd_fit=None, t=1e-6):
    """R/inferCNV_HMM.R:509-567."""
    return _predict_groups_i6(infercnv_obj, _whole_sample_groups(infercnv_obj, cluster_by_groups), cnv_mean_sd,
                              cnv_level_to_mean_sd_fit, t)


def predict_CNV_via_HMM_on_tumor_subclusters(infercnv_obj, cnv_mean_sd, cnv_level_to_mean_sd_fit=None, t=1e-6):
    """R/inferCNV_HMM.R:345-408 (falls back to whole samples when no subclusters, :358-361)."""
    if infercnv_obj.tumor_subclusters is None:
        return predict_CNV_via_HMM_on_whole_tumor_samples(infercnv_obj, True, cnv_mean_sd, cnv_level_to_mean_sd_fit, t)
    return _predict_groups_i6(infercnv_obj, _flatten_subclusters(infercnv_obj), cnv_mean_sd,
                              cnv_level_to_mean_sd_fit, t)


def predict_CNV_via_HMM_on_tumor_subclusters_per_chr(infercnv_obj, subclusters_per_chr, cnv_mean_sd,
                                                     cnv_level_to_mean_sd_fit=None, t=1e-6):
    """R/inferCNV_HMM.R:412-487: `subclusters_per_chr[chr]` lists the cell index vectors of the subclusters
    defined on that chromosome; one device call per chromosome, then (:473-483) every cell of each GLOBAL
    subcluster (infercnv_obj.tumor_subclusters) receives that subcluster's per-gene consensus state."""
    if subclusters_per_chr is None:
        return predict_CNV_via_HMM_on_whole_tumor_samples(infercnv_obj, True, cnv_mean_sd, cnv_level_to_mean_sd_fit, t)
    from .cnv_regions import overwrite_with_consensus
    hmm = _get_HMM(cnv_mean_sd, t)
    chrs = np.asarray(infercnv_obj.gene_order.chr)
    out = np.full(infercnv_obj.expr_data.shape, -1.0)
    for chr_name, groups in subclusters_per_chr.items():
        rows = np.nonzero(chrs == chr_name)[0]
        if rows.size == 0:
            continue
        groups = [np.asarray(g, dtype=np.int32) for g in (groups.values() if isinstance(groups, dict) else groups)]
        x = np.asfortranarray(infercnv_obj.expr_data[rows], dtype=np.float64)
        sd = [_group_sd(len(g), cnv_mean_sd, cnv_level_to_mean_sd_fit) for g in groups]
        st = _viterbi_groups(x, np.array([0, rows.size], dtype=np.int32), groups,
                             hmm["state_emission_params"]["mean"], sd, hmm["state_transitions"], hmm["delta"])
        st = st.astype(np.float64)
        st[st == 255] = -1.0
        out[rows] = st
    new = infercnv_obj.copy()
    new.expr_data = out
    if infercnv_obj.tumor_subclusters is not None:   # get_predicted_CNV_regions(by="subcluster") consensus overwrite
        new = overwrite_with_consensus(new, _flatten_subclusters(infercnv_obj))
    return new


def assign_HMM_states_to_proxy_expr_vals(infercnv_obj: InfercnvObject) -> InfercnvObject:
    """R/inferCNV_HMM.R:1191-1206."""
    return _proxy(infercnv_obj, 6)


# ------------------------------------------------------------------ i3
def _ks_two_sample_p_value(x, y):
    """`ks.test(x, y)$p.value` (two-sided) as base R computes it (src/library/stats/R/ks.test.R, src/library/stats/src/ks.c):
    D = max |F_x - F_y|; the exact distribution of D (psmirnov2x: the lattice-path recursion) when n.x * n.y < 10000, else
    the limiting Kolmogorov distribution at sqrt(n.x n.y / (n.x + n.y)) D (pKS2, series cut at tol = 1e-6); clipped to [0, 1].
    Continuous samples: no ties."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    nx, ny = x.size, y.size
    w = np.concatenate([x, y])
    order = np.argsort(w, kind="stable")
    z = np.cumsum(np.where(order < nx, 1.0 / nx, -1.0 / ny))
    stat = float(np.abs(z).max())
    if nx * ny < 10000:
        m, n = (nx, ny) if nx <= ny else (ny, nx)
        md, nd = float(m), float(n)
        q = (0.5 + math.floor(stat * md * nd - 1e-7)) / (md * nd)
        jn = np.arange(n + 1, dtype=np.float64) / nd
        u = np.where(jn > q, 0.0, 1.0)
        for i in range(1, m + 1):
            wgt = i / (i + nd)
            im = i / md
            keep = ~(np.abs(im - jn) > q)
            u0 = 0.0 if im > q else wgt * u[0]
            # u[j] = keep[j] ? wgt * u[j] + u[j - 1] : 0, left to right (u[j - 1] is the NEW value): a first-order recurrence
            nu = np.empty_like(u)
            nu[0] = u0
            acc = u0
            for j in range(1, n + 1):
                acc = (wgt * u[j] + acc) if keep[j] else 0.0
                nu[j] = acc
            u = nu
        p = 1.0 - float(u[n])
    else:
        xs = math.sqrt(nx * ny / (nx + ny)) * stat
        tol = 1e-6
        if xs <= 0.0:
            cdf = 0.0
        elif xs < 1.0:
            k_max = int(math.sqrt(2.0 - math.log(tol)))
            zz = -(math.pi / 2.0 * math.pi / 4.0) / (xs * xs)
            lw = math.log(xs)
            sacc = 0.0
            for k in range(1, k_max, 2):
                sacc += math.exp(k * k * zz - lw)
            cdf = sacc / 0.398942280401432677939946059934   # M_1_SQRT_2PI
        else:
            zz = -2.0 * xs * xs
            sgn, k, old, new = -1.0, 1, 0.0, 1.0
            while abs(old - new) > tol:
                old = new
                new += 2.0 * sgn * math.exp(zz * k * k)
                sgn = -sgn
                k += 1
            cdf = new
        p = 1.0 - cdf
    return min(1.0, max(0.0, p))


def get_HoneyBADGER_setGexpDev(gexp_sd, alpha, k_cells=2, n_iter=100, seed=None, rng=None):
    """get_HoneyBADGER_setGexpDev (R/inferCNV_i3HMM.R:469-493): for dev in seq(0, sd, sd / 10) the mean over n_iter rounds of
    ks.test(rnorm(k_cells, 0, sd), rnorm(k_cells, dev, sd))$p.value, then lm(devs ~ pvs) evaluated at pvs = alpha.  The
    draws are R's own stream (`seed` as in set.seed(seed) right before the call; infercnv_amd/r_rng.py) -- the reference
    never seeds, so its own value differs from session to session."""
    from .r_rng import RRandom
    if rng is None:
        if seed is None:
            raise ValueError("give the RNG state: seed (as in set.seed(seed)) or an RRandom")
        rng = RRandom(seed)
    k_cells = max(int(k_cells), 2)                                  # (:471-474)
    by = gexp_sd / 10.0
    n = int((gexp_sd - 0.0) / by + 1e-10)                           # seq.default
    devs = np.minimum(0.0 + np.arange(n + 1, dtype=np.float64) * by, gexp_sd)
    pvs = np.empty(devs.size)
    for d, dev in enumerate(devs):
        acc = []
        for _ in range(int(n_iter)):
            a = rng.rnorm(k_cells, 0.0, gexp_sd)                    # ks.test forces x, then y
            b = rng.rnorm(k_cells, float(dev), gexp_sd)
            acc.append(_ks_two_sample_p_value(a, b))
        pvs[d] = float(np.mean(np.asarray(acc, dtype=np.longdouble)))
    LD = np.longdouble
    px, dy = pvs.astype(LD), devs.astype(LD)
    mx, my = px.mean(), dy.mean()
    slope = ((px - mx) * (dy - my)).sum() / ((px - mx) ** 2).sum()
    return float(my - slope * mx + slope * LD(alpha))


def i3HMM_get_sd_trend(infercnv_obj: InfercnvObject, i3_p_val=0.05, seed=None, rng=None):
    """.i3HMM_get_sd_trend_by_num_cells_fit (R/inferCNV_i3HMM.R:17-80): mu and sigma over all values of the reference
    cells (the device), mean_delta = |qnorm(p, 0, sigma)|, and -- when the RNG state is given (`seed` as in
    set.seed(seed) before the call) -- KS_delta = get_HoneyBADGER_setGexpDev(sigma, p, k_cells = number of those cells)."""
    idx = (infercnv_obj.get_reference_grouped_cell_indices() if infercnv_obj.has_reference_cells()
           else np.concatenate([np.asarray(v) for v in infercnv_obj.observation_grouped_cell_indices.values()]))
    L = _lib.load()
    x = np.asfortranarray(infercnv_obj.expr_data, dtype=np.float64)
    ci, cp = i32(idx)
    buf = (ct.c_double * 2)()
    check(L.icnv_cells_mean_sd(x.ctypes.data_as(ct.c_void_p), x.shape[0], x.shape[1], cp, ci.size, buf))   # mean / sd on the device
    mu, sigma = float(buf[0]), float(buf[1])
    ks = None
    if seed is not None or rng is not None:
        ks = get_HoneyBADGER_setGexpDev(sigma, i3_p_val, k_cells=int(ci.size), seed=seed, rng=rng)
    return {"mu": mu, "sigma": sigma, "mean_delta": determine_mean_delta_via_Z(sigma, i3_p_val), "KS_delta": ks}


def i3HMM_predict_CNV_via_HMM_on_indiv_cells(infercnv_obj, i3_p_val=0.05, sd_trend=None, t=1e-6, use_KS=True, seed=None):
    """R/inferCNV_i3HMM.R:180-225 (use_KS = TRUE is the reference's default; it needs the RNG state: `seed`, or a
    sd_trend that carries KS_delta)."""
    sd_trend = sd_trend or i3HMM_get_sd_trend(infercnv_obj, i3_p_val, seed=seed if use_KS else None)
    hmm = _i3HMM_get_HMM(sd_trend, t, i3_p_val, use_KS)
    perm, chr_start, x = _layout(infercnv_obj)
    pm = hmm["state_emission_params"]
    st = _viterbi_cells(x, chr_start, pm["mean"], _median(pm["sd"]), hmm["state_transitions"], hmm["delta"])
    return _states_obj(infercnv_obj, st, perm)


def _predict_groups_i3(obj, groups, i3_p_val, sd_trend, t, use_KS, seed=None):
    sd_trend = sd_trend or i3HMM_get_sd_trend(obj, i3_p_val, seed=seed if use_KS else None)
    hmm = _i3HMM_get_HMM(sd_trend, t, i3_p_val, use_KS)
    perm, chr_start, x = _layout(obj)
    pm = hmm["state_emission_params"]
    st = _viterbi_groups(x, chr_start, groups, pm["mean"], [_median(pm["sd"])] * len(groups),
                         hmm["state_transitions"], hmm["delta"])
    return _states_obj(obj, st, perm)


def i3HMM_predict_CNV_via_HMM_on_whole_tumor_samples(infercnv_obj, cluster_by_groups, i3_p_val=0.05, sd_trend=None,
                                                     t=1e-6, use_KS=True, seed=None):
    """R/inferCNV_i3HMM.R:332-389."""
    return _predict_groups_i3(infercnv_obj, _whole_sample_groups(infercnv_obj, cluster_by_groups), i3_p_val, sd_trend,
                              t, use_KS, seed)


def i3HMM_predict_CNV_via_HMM_on_tumor_subclusters(infercnv_obj, i3_p_val=0.05, sd_trend=None, t=1e-6, use_KS=True, seed=None):
    """R/inferCNV_i3HMM.R:249-308."""
    if infercnv_obj.tumor_subclusters is None:
        return i3HMM_predict_CNV_via_HMM_on_whole_tumor_samples(infercnv_obj, True, i3_p_val, sd_trend, t, use_KS, seed)
    return _predict_groups_i3(infercnv_obj, _flatten_subclusters(infercnv_obj), i3_p_val, sd_trend, t, use_KS, seed)


def i3HMM_assign_HMM_states_to_proxy_expr_vals(infercnv_obj: InfercnvObject) -> InfercnvObject:
    """R/inferCNV_i3HMM.R:405-417."""
    return _proxy(infercnv_obj, 3)


def _proxy(obj, K):
    L = _lib.load()
    st = np.asfortranarray(np.where(obj.expr_data < 0, 255, obj.expr_data).astype(np.uint8))
    out = np.empty(st.shape, dtype=np.float64, order="F")
    check(L.icnv_states_to_proxy(st.ctypes.data_as(ct.c_void_p), out.ctypes.data_as(ct.c_void_p), st.size, K))
    # entries that were not HMM states (e.g. -1) stay as they were, like R's masked assignment
    out = np.where(np.isnan(out), obj.expr_data, out)
    new = obj.copy()
    new.expr_data = out
    return new
