"""Host-side mirror of the reference's smoothing-chain step functions
(R/inferCNV_ops.R), same names / argument meaning / error behaviour, each one a
thin call into libicnv_hip.so.  Every function takes an InfercnvObject and
returns a new InfercnvObject whose `expr_data` was replaced -- and mirrors
itself onto `hspike` where the reference does.

These wrappers take host (numpy) matrices, like the R shim would hand over
`expr.data`; use `infercnv_amd.device` for device-resident tensors.
"""
from __future__ import annotations

import ctypes as ct
import math

import numpy as np

from . import _lib
from ._lib import Cfg, check
from .infercnv_object import InfercnvObject


def _as_f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _run_chain(obj: InfercnvObject, stage_mask, *, window_length=101, max_thresh=None, use_bounds=True,
               sd_amplifier=1.5, noise_filter=None, want_pre_denoise=False, inv_log=False, noise_logistic=False):
    L = _lib.load()
    perm, chr_start = obj.chr_layout()
    x = _as_f(obj.expr_data if perm is None else obj.expr_data[perm])
    G, C = x.shape
    if np.isnan(x).any():      # NA values: the cells that hold one are recomputed with the reference's NA semantics (csrc/chain_na.hip)
        stage_mask = int(stage_mask) | _lib.ST_NA_AWARE
    cfg = Cfg(G, C, chr_start, obj.ref_groups_or_proxy(), window_length, max_thresh, use_bounds, sd_amplifier,
              noise_filter, stage_mask, inv_log, noise_logistic)
    out = np.empty_like(x, order="F")
    pre = np.empty_like(x, order="F") if want_pre_denoise else None
    check(L.icnv_smooth_chain(x.ctypes.data_as(ct.c_void_p), out.ctypes.data_as(ct.c_void_p),
                              pre.ctypes.data_as(ct.c_void_p) if pre is not None else None, cfg.ptr()))
    if perm is not None:
        inv = np.empty_like(perm)
        inv[perm] = np.arange(perm.size)
        out = out[inv]
        pre = pre[inv] if pre is not None else None
    return out, pre


def _with_expr(obj, expr, hspike=None):
    new = obj.copy()
    new.expr_data = expr
    if hspike is not None:
        new.hspike = hspike
    return new


# ------------------------------------------------------------------ step 2 (gene filters)
def _gene_stats(infercnv_obj):
    """(rowSums(expr), per-gene number of cells with expr > 0) from the HIP path."""
    L = _lib.load()
    x = _as_f(infercnv_obj.expr_data)
    G, C = x.shape
    sums = np.empty(G, dtype=np.float64)
    nnz = np.empty(G, dtype=np.int32)
    check(L.icnv_gene_stats(x.ctypes.data_as(ct.c_void_p), G, C, sums.ctypes.data_as(ct.c_void_p),
                            nnz.ctypes.data_as(ct.c_void_p)))
    return sums, nnz


def _keep_gene_rows(infercnv_obj, keep):
    """The per-gene slots of remove_genes (R/inferCNV.R:445-457) restricted to rows `keep`; expr.data is the caller's."""
    new = infercnv_obj.copy()
    if infercnv_obj.count_data is not None:
        new.count_data = np.asarray(infercnv_obj.count_data)[keep]
    go = infercnv_obj.gene_order
    new.gene_order = type(go)(np.asarray(go.chr)[keep], None if go.start is None else np.asarray(go.start)[keep],
                              None if go.stop is None else np.asarray(go.stop)[keep])
    if infercnv_obj.gene_names is not None:
        new.gene_names = np.asarray(infercnv_obj.gene_names)[keep]
    return new


def remove_genes(infercnv_obj: InfercnvObject, gene_indices_to_remove) -> InfercnvObject:
    """remove_genes (R/inferCNV.R:445-457): drops the rows (0-based here) from expr.data (on the device),
    count.data and gene_order."""
    L = _lib.load()
    x = _as_f(infercnv_obj.expr_data)
    G, C = x.shape
    drop = np.zeros(G, dtype=bool)
    drop[np.asarray(gene_indices_to_remove, dtype=np.int64)] = True
    keep = np.nonzero(~drop)[0].astype(np.int32)
    out = np.empty((keep.size, C), dtype=np.float64, order="F")
    if keep.size:                                        # (no gene left: a 0-row object, as the reference returns)
        check(L.icnv_select_genes(x.ctypes.data_as(ct.c_void_p), G, C, keep.ctypes.data_as(ct.POINTER(ct.c_int32)), keep.size,
                                  out.ctypes.data_as(ct.c_void_p)))
    new = _keep_gene_rows(infercnv_obj, keep)
    new.expr_data = out
    new.validate()
    return new


def require_above_min_mean_expr_cutoff(infercnv_obj: InfercnvObject, min_mean_expr_cutoff) -> InfercnvObject:
    """R/inferCNV_ops.R:2128-2163: removes genes with rowMeans(expr.data) < cutoff."""
    sums, _ = _gene_stats(infercnv_obj)
    indices = np.nonzero(sums / infercnv_obj.expr_data.shape[1] < min_mean_expr_cutoff)[0]
    return remove_genes(infercnv_obj, indices) if indices.size else infercnv_obj


def require_above_min_cells_ref(infercnv_obj: InfercnvObject, min_cells_per_gene) -> InfercnvObject:
    """R/inferCNV_ops.R:2182-2213: keeps genes expressed (> 0, not NA) in at least min_cells_per_gene cells;
    stops when no gene passes (the reference's stop(998))."""
    _, nnz = _gene_stats(infercnv_obj)
    passed = nnz >= min_cells_per_gene
    if passed.all():
        return infercnv_obj
    if not passed.any():
        raise RuntimeError("All genes removed! Must revisit your data..., cannot continue here.")
    return remove_genes(infercnv_obj, np.nonzero(~passed)[0])


# ------------------------------------------------------------------ steps 3 / 4 (ingest)
def _normalize_log2(infercnv_obj, normalize_factor, do_norm, do_log):
    L = _lib.load()
    x = _as_f(infercnv_obj.expr_data)
    out = np.empty_like(x, order="F")
    used = ct.c_double()
    nf = float("nan") if normalize_factor is None or (isinstance(normalize_factor, float) and math.isnan(normalize_factor)) \
        else float(normalize_factor)
    check(L.icnv_normalize_log2(x.ctypes.data_as(ct.c_void_p), out.ctypes.data_as(ct.c_void_p), x.shape[0], x.shape[1],
                                nf, int(do_norm), int(do_log), ct.byref(used)))
    return out


def normalize_counts_by_seq_depth(infercnv_obj: InfercnvObject, normalize_factor=None) -> InfercnvObject:
    """R/inferCNV_ops.R:3064-3111 (normalize_factor=NA -> median library size).  No hspike mirror in the reference."""
    return _with_expr(infercnv_obj, _normalize_log2(infercnv_obj, normalize_factor, True, False))


def log2xplus1(infercnv_obj: InfercnvObject) -> InfercnvObject:
    """R/inferCNV_ops.R:2756-2769."""
    hs = log2xplus1(infercnv_obj.hspike) if infercnv_obj.hspike is not None else None
    return _with_expr(infercnv_obj, _normalize_log2(infercnv_obj, None, False, True), hs)


def ingest_counts(infercnv_obj: InfercnvObject, min_mean_expr_cutoff=None, min_cells_per_gene=0, normalize_factor=None,
                  sparse=None):
    """Steps 2, 3, 4 of run() in ONE call from the integer count matrix (`infercnv_obj.expr_data` holding raw counts --
    a dense integer array or a scipy.sparse matrix): require_above_min_mean_expr_cutoff + require_above_min_cells_ref
    (R/inferCNV_ops.R:2128-2213), normalize_counts_by_seq_depth (:3064-3111), log2xplus1 (:2756-2769).  The counts are
    uploaded once as int32 (dense) or CSC; the result equals the four step functions applied in run()'s order, bit for
    bit.  Returns (object with the log-scale matrix of the kept genes, bytes uploaded)."""
    L = _lib.load()
    x = infercnv_obj.expr_data
    is_sparse = hasattr(x, "tocsc")
    if sparse is None:
        sparse = is_sparse
    G, C = x.shape
    if sparse:
        m = x.tocsc() if is_sparse else __import__("scipy.sparse", fromlist=["csc_matrix"]).csc_matrix(np.asarray(x))
        m.sum_duplicates()
        vals = np.ascontiguousarray(m.data)
        if not np.array_equal(vals, np.rint(vals)):
            raise ValueError("ingest_counts wants integer counts")
        colptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
        rowidx = np.ascontiguousarray(m.indices, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.int32)
        cnt = _lib.Counts(None, colptr.ctypes.data, rowidx.ctypes.data, vals.ctypes.data, int(vals.size))
        keepalive = (colptr, rowidx, vals)
    else:
        xd = np.asarray(x.todense() if is_sparse else x)
        if not np.array_equal(xd, np.rint(xd)):
            raise ValueError("ingest_counts wants integer counts")
        dense = np.asfortranarray(xd, dtype=np.int32)
        cnt = _lib.Counts(dense.ctypes.data, None, None, None, 0)
        keepalive = (dense,)
    keep = np.empty(G, dtype=np.int32)
    n, used, up = ct.c_int64(), ct.c_double(), ct.c_int64()
    out = np.empty((G, C), dtype=np.float64, order="F")
    check(L.icnv_ingest_counts(ct.byref(cnt), G, C, float("nan") if min_mean_expr_cutoff is None else float(min_mean_expr_cutoff),
                               int(min_cells_per_gene), float("nan") if normalize_factor is None else float(normalize_factor),
                               keep.ctypes.data_as(ct.POINTER(ct.c_int32)), ct.byref(n), out.ctypes.data_as(ct.c_void_p), ct.byref(used),
                               ct.byref(up)))
    del keepalive
    g_out = n.value
    keep = keep[:g_out]
    expr = np.asfortranarray(out.ravel(order="F")[: g_out * C].reshape((g_out, C), order="F"))
    new = _keep_gene_rows(infercnv_obj, keep)
    new.expr_data = expr
    return new, up.value


# ------------------------------------------------------------------ step 8 / 12
def subtract_ref_expr_from_obs(infercnv_obj: InfercnvObject, inv_log=False, use_bounds=True) -> InfercnvObject:
    """R/inferCNV_ops.R:1678-1702.  `inv_log=TRUE` (not used by run(): :771 and :952 pass FALSE) takes the
    group means as log2(mean(2^x - 1) + 1) (:1714-1717)."""
    out, _ = _run_chain(infercnv_obj, _lib.ST_SUBTRACT_REF_1, use_bounds=use_bounds, inv_log=inv_log)
    hs = None
    if infercnv_obj.hspike is not None:  # :1695-1698
        hs = subtract_ref_expr_from_obs(infercnv_obj.hspike, inv_log=inv_log, use_bounds=use_bounds)
    return _with_expr(infercnv_obj, out, hs)


# ------------------------------------------------------------------ step 9
def get_average_bounds(infercnv_obj: InfercnvObject):
    """R/inferCNV_ops.R:2723-2742 -> (lower, upper)."""
    L = _lib.load()
    x = _as_f(infercnv_obj.expr_data)
    out = (ct.c_double * 2)()
    check(L.icnv_average_bounds(x.ctypes.data_as(ct.c_void_p), x.shape[0], x.shape[1], out))
    return out[0], out[1]


def scale_infercnv_expr(infercnv_obj: InfercnvObject) -> InfercnvObject:
    """Step 5 of run() (scale_data, off by default; R/inferCNV_ops.R:3174-3185): t(scale(t(expr.data))), mirrored on the hspike."""
    L = _lib.load()
    x = _as_f(infercnv_obj.expr_data)
    out = np.empty_like(x, order="F")
    check(L.icnv_scale_genes(x.ctypes.data_as(ct.c_void_p), out.ctypes.data_as(ct.c_void_p), x.shape[0], x.shape[1]))
    hs = scale_infercnv_expr(infercnv_obj.hspike) if infercnv_obj.hspike is not None else None
    return _with_expr(infercnv_obj, out, hs)


def _remove_tails(chr_idx, tail_length):
    """.remove_tails (R/inferCNV_ops.R:2370-2386): the first and last tail_length positions of a chromosome's gene index
    vector (0-based here); nothing when the tail or the chromosome is shorter than 3; a chromosome shorter than two
    tails loses floor(n / 3) genes at either end."""
    chr_idx = np.asarray(chr_idx, dtype=np.int64)
    n = chr_idx.size
    if tail_length < 3 or n < 3:
        return np.zeros(0, dtype=np.int64)
    if n < tail_length * 2:
        tail_length = n // 3
    tail_length = int(tail_length)
    return np.concatenate([chr_idx[:tail_length], chr_idx[n - tail_length:]])


def remove_genes_at_ends_of_chromosomes(infercnv_obj: InfercnvObject, window_length) -> InfercnvObject:
    """Step 13 of run() (R/inferCNV_ops.R:3000-3033; remove_genes_at_chr_ends, off by default): drops (window_length - 1) / 2
    genes at either end of every chromosome through remove_genes (the row selection runs on the device)."""
    contig_tail = (window_length - 1) / 2
    chrs = np.asarray(infercnv_obj.gene_order.chr)
    drop = []
    seen = []
    for c in chrs:                                       # unique(), order of first appearance
        if c not in seen:
            seen.append(c)
    for c in seen:
        drop.append(_remove_tails(np.nonzero(chrs == c)[0], contig_tail))
    drop = np.concatenate(drop) if drop else np.zeros(0, dtype=np.int64)
    if not drop.size:                                    # the reference stops here (flog.error + stop(1234), :3029-3031)
        raise ValueError("No genes removed at chr ends.... something wrong here")
    out = remove_genes(infercnv_obj, drop)
    if infercnv_obj.hspike is not None:                  # "-mirroring for hspike" (:3035-3038): its own gene order, its own tails
        hs = remove_genes_at_ends_of_chromosomes(infercnv_obj.hspike, window_length)
        out = _with_expr(out, out.expr_data, hs)
    return out


def remove_outliers_norm(infercnv_obj: InfercnvObject, out_method="average_bound", lower_bound=None, upper_bound=None) -> InfercnvObject:
    """Step 16 of run() (R/inferCNV_ops.R:1969-2054, prune_outliers): clamp to [lower, upper]; both bounds given = hard
    thresholds, otherwise out_method "average_bound" (.get_average_bounds of the matrix); mirrored on the hspike (:1985-1988)."""
    na = lambda v: v is None or (isinstance(v, float) and math.isnan(v))
    if na(lower_bound) or na(upper_bound):
        if na(out_method):
            raise ValueError("must specify outmethod or define exact bounds")            # stop(992)
        if out_method != "average_bound":
            raise ValueError("please provide an approved method for outlier removal")    # stop(991)
        lower_bound = upper_bound = float("nan")
    L = _lib.load()
    x = _as_f(infercnv_obj.expr_data)
    out = np.empty_like(x, order="F")
    check(L.icnv_remove_outliers(x.ctypes.data_as(ct.c_void_p), out.ctypes.data_as(ct.c_void_p), x.shape[0], x.shape[1],
                                 float(lower_bound), float(upper_bound), None))
    hs = None
    if infercnv_obj.hspike is not None:
        hs = remove_outliers_norm(infercnv_obj.hspike, out_method, None if math.isnan(lower_bound) else lower_bound,
                                  None if math.isnan(upper_bound) else upper_bound)
    return _with_expr(infercnv_obj, out, hs)


def apply_max_threshold_bounds(infercnv_obj: InfercnvObject, threshold) -> InfercnvObject:
    """R/inferCNV_ops.R:2970-2983; threshold may be "auto" like run()'s
    max_centered_threshold (:802-817: mean(abs(get_average_bounds())))."""
    if isinstance(threshold, str):
        if threshold != "auto":
            raise ValueError('threshold must be numeric or "auto"')
        lo, hi = get_average_bounds(infercnv_obj)
        threshold = (abs(lo) + abs(hi)) / 2.0
    out, _ = _run_chain(infercnv_obj, _lib.ST_MAX_THRESH, max_thresh=float(threshold))
    hs = None
    if infercnv_obj.hspike is not None:  # :2977-2980
        hs = apply_max_threshold_bounds(infercnv_obj.hspike, threshold)
    return _with_expr(infercnv_obj, out, hs)


# ------------------------------------------------------------------ step 10
def smooth_by_chromosome(infercnv_obj: InfercnvObject, window_length, smooth_ends=True) -> InfercnvObject:
    """R/inferCNV_ops.R:2406-2434 (pyramid weights, ends renormalised).  Like the
    reference, window_length < 2 returns the data unchanged (:2444-2447)."""
    if window_length >= 2 and int(window_length) % 2 == 0:
        raise ValueError("window_length must be odd")
    out, _ = _run_chain(infercnv_obj, _lib.ST_SMOOTH, window_length=int(window_length))
    hs = None
    if infercnv_obj.hspike is not None:  # :2427-2430
        hs = smooth_by_chromosome(infercnv_obj.hspike, window_length, smooth_ends)
    return _with_expr(infercnv_obj, out, hs)


# ------------------------------------------------------------------ step 11
def center_cell_expr_across_chromosome(infercnv_obj: InfercnvObject, method="mean") -> InfercnvObject:
    """R/inferCNV_ops.R:2074-2088; method "median" (what run() passes, :911) or "mean"."""
    mask = _lib.ST_CENTER | (0 if method == "median" else _lib.ST_CENTER_MEAN)
    out, _ = _run_chain(infercnv_obj, mask)
    hs = None
    if infercnv_obj.hspike is not None:  # :2081-2084
        hs = center_cell_expr_across_chromosome(infercnv_obj.hspike, method)
    return _with_expr(infercnv_obj, out, hs)


# ------------------------------------------------------------------ step 14
def invert_log2(infercnv_obj: InfercnvObject) -> InfercnvObject:
    """R/inferCNV_ops.R:2814-2826."""
    out, _ = _run_chain(infercnv_obj, _lib.ST_INVERT_LOG2)
    hs = invert_log2(infercnv_obj.hspike) if infercnv_obj.hspike is not None else None
    return _with_expr(infercnv_obj, out, hs)


# ------------------------------------------------------------------ step 22
def clear_noise_via_ref_mean_sd(infercnv_obj: InfercnvObject, sd_amplifier=1.5, noise_logistic=False) -> InfercnvObject:
    """R/inferCNV_ops.R:2302-2346.  hspike is NOT mirrored (commented out in the reference, :2340-2343).
    noise_logistic: depress_log_signal_midpt_val around the same centre with the same half width (:2326-2330;
    .apply_logistic_val_adj, R/inferCNV_heatmap.R:2791-2810) instead of the select."""
    out, _ = _run_chain(infercnv_obj, _lib.ST_DENOISE, sd_amplifier=sd_amplifier, noise_logistic=noise_logistic)
    return _with_expr(infercnv_obj, out)


def clear_noise(infercnv_obj: InfercnvObject, threshold, noise_logistic=False) -> InfercnvObject:
    """R/inferCNV_ops.R:2232-2262 (threshold == 0 -> unchanged; noise_logistic as in clear_noise_via_ref_mean_sd)."""
    if threshold == 0:
        return infercnv_obj
    out, _ = _run_chain(infercnv_obj, _lib.ST_DENOISE, noise_filter=float(threshold), noise_logistic=noise_logistic)
    return _with_expr(infercnv_obj, out)


# ------------------------------------------------------------------ fused entry
def hip_smooth_chain(infercnv_obj: InfercnvObject, window_length=101, max_centered_threshold=3.0,
                     sd_amplifier=1.5, noise_filter=None, denoise=True, return_hmm_input=False, noise_logistic=False):
    """Steps 8,9,10,11,12,14(,22) of run() back to back in one fused device pass
    (SURVEY.md 8b.1).  Equivalent to calling the stand-alone wrappers in run()'s
    order.  With return_hmm_input=True also returns the object before step 22
    (what step 17's HMM reads, R/inferCNV_ops.R:1237-1309)."""
    thr = max_centered_threshold
    mask = _lib.ST_ALL if denoise else (_lib.ST_ALL & ~_lib.ST_DENOISE)
    if isinstance(thr, str):
        if thr != "auto":
            raise ValueError('max_centered_threshold must be numeric, NA/None or "auto"')
        tmp_obj = infercnv_obj.copy()
        tmp_obj.hspike = None
        lo, hi = get_average_bounds(subtract_ref_expr_from_obs(tmp_obj))   # R/inferCNV_ops.R:802-806
        thr = (abs(lo) + abs(hi)) / 2.0
    if thr is None or (isinstance(thr, float) and math.isnan(thr)):
        mask &= ~_lib.ST_MAX_THRESH
        thr = None
    out, pre = _run_chain(infercnv_obj, mask, window_length=window_length, max_thresh=thr,
                          sd_amplifier=sd_amplifier, noise_filter=noise_filter, want_pre_denoise=return_hmm_input,
                          noise_logistic=noise_logistic)
    hs = None
    if infercnv_obj.hspike is not None:
        # the hspike mirrors steps 8..14 but not the denoise (reference: commented out)
        hs = hip_smooth_chain(infercnv_obj.hspike, window_length, thr, sd_amplifier, noise_filter, denoise=False)
    res = _with_expr(infercnv_obj, out, hs)
    if return_hmm_input:
        return res, _with_expr(infercnv_obj, pre if denoise else out, hs)
    return res
