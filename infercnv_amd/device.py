"""Device-resident entry points: thin wrappers that hand torch CUDA tensors'
data pointers to the C ABI (`*_dev` functions of include/icnv.h).

PyTorch is used only for device memory, streams and (in `sharded.py`)
torch.distributed -- all arithmetic happens in libicnv_hip.so.

Layout: an expression matrix is a contiguous float64 tensor of shape
(C cells, G genes): row-major (C, G) is byte-identical to R's column-major
genes x cells `expr.data` (R/inferCNV.R:18), i.e. element (gene g, cell c) sits
at offset g + G*c.  State matrices are uint8 (C, G).
"""
from __future__ import annotations

import ctypes as ct
import os

import numpy as np
import torch

from . import _lib
from ._lib import Cfg, check, f64, i32, pack_groups


def _stream():
    return ct.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ct.c_void_p(t.data_ptr()) if t is not None else ct.c_void_p(0)


def _check_matrix(x, dtype=torch.float64):
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == dtype and x.dim() == 2 and x.is_contiguous()):
        raise TypeError(f"expected a contiguous CUDA {dtype} tensor of shape (cells, genes)")
    return x.shape[0], x.shape[1]


def _check_matrix_ld(x, dtype=torch.float64):
    """(C, G, ld) of a CUDA matrix whose rows -- the cells -- are contiguous and lie `ld` elements apart: a contiguous
    (cells, genes) tensor (ld = G) or the [:, :G] view of a wider one (padded_matrix)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == dtype and x.dim() == 2 and (x.shape[1] <= 1 or x.stride(1) == 1)
            and (x.shape[0] <= 1 or x.stride(0) >= x.shape[1])):
        raise TypeError(f"expected a CUDA {dtype} tensor of shape (cells, genes) with contiguous rows")
    return x.shape[0], x.shape[1], (x.stride(0) if x.shape[0] > 1 else x.shape[1])


def padded_matrix(C, G, dtype=torch.float64, device="cuda", multiple=16):
    """A (C, G) matrix whose rows start `ld = G rounded up to a multiple of 16` elements apart -- every cell on a cache line
    (float64) / a 16-byte word (uint8) of its own, whatever G is.  The per-cell Viterbi reads and writes such matrices at the
    speed of a gene count that is a multiple of 16 (icnv_viterbi_cells_ld_dev); ChainPlan.apply / smooth_chain write their
    HMM input into one when it is passed as `pre` (icnv_chain_apply_ld_dev).  The padding columns are never read or written."""
    ld = (int(G) + multiple - 1) // multiple * multiple
    return torch.empty((int(C), ld), dtype=dtype, device=device)[:, :int(G)]


def init(device=None):
    """Bind the calling thread to `device` (defaults to torch's current device)."""
    L = _lib.load()
    if device is None:
        device = torch.cuda.current_device()
    check(L.icnv_init(int(device)))
    if os.environ.get("ICNV_VITERBI_MODE"):     # developer switch: 1 = exact Viterbi kernel only (see viterbi_set_mode)
        check(L.icnv_viterbi_set_mode(int(os.environ["ICNV_VITERBI_MODE"])))


def release_pool():
    """Hand the library's cached device blocks (workspace pool, resident matrices, Viterbi tables) back to the driver
    (icnv_shutdown; the library stays usable, the next call allocates again).  For callers that need the whole HBM for
    one large matrix after smaller runs -- the 1 M-cell case holds 270 of the 288 GB."""
    torch.cuda.synchronize()
    _lib.load().icnv_shutdown()


# ------------------------------------------------------------------ smoothing chain
def smooth_chain(x, chr_start, ref_groups, window_length=101, max_thresh=3.0, use_bounds=True,
                 sd_amplifier=1.5, noise_filter=None, stage_mask=_lib.ST_ALL, out=None, want_pre_denoise=False,
                 inv_log=False):
    """Steps 8,9,10,11,12,14,22 of run() (R/inferCNV_ops.R:771-1589) fused on the
    GPU.  Returns (out, pre_denoise or None).  inv_log: the stand-alone
    subtract_ref_expr_from_obs(inv_log=TRUE) (stage_mask must be ST_SUBTRACT_REF_1 alone)."""
    L = _lib.load()
    C, G = _check_matrix(x)
    cfg = Cfg(G, C, chr_start, ref_groups, window_length, max_thresh, use_bounds, sd_amplifier, noise_filter,
              stage_mask, inv_log)
    if out is None:
        out = torch.empty_like(x)
    pre = torch.empty_like(x) if want_pre_denoise else None
    check(L.icnv_smooth_chain_dev(_ptr(x), _ptr(out), _ptr(pre), cfg.ptr(), _stream()))
    return out, pre


class ChainPlan:
    """Split-phase chain (icnv_chain_* in include/icnv.h) for cell-sharded runs:
    for each reference round r: partial(r) -> all-reduce(sum) -> finish(r); then apply()."""

    def __init__(self, G, C, chr_start, ref_groups_local, **kw):
        self.L = _lib.load()
        self.cfg = Cfg(G, C, chr_start, ref_groups_local, **kw)
        h = ct.c_void_p()
        check(self.L.icnv_chain_begin(ct.byref(h), self.cfg.ptr()))
        self.h = h
        self.G, self.C = G, C

    @property
    def num_rounds(self):
        return self.L.icnv_chain_num_rounds(self.h)

    def round_partial(self, r, x):
        """Enqueue this rank's partial statistic; returns a float64 CUDA tensor
        *view* of the library's buffer (all-reduce it in place)."""
        p, n = ct.c_void_p(), ct.c_int64()
        check(self.L.icnv_chain_round_partial_dev(self.h, r, _ptr(x), ct.byref(p), ct.byref(n), _stream()))
        return _wrap_f64(p.value, n.value, x.device)

    def round_finish(self, r):
        check(self.L.icnv_chain_round_finish_dev(self.h, r, _stream()))

    def apply(self, x, out=None, want_pre_denoise=False, pre=None):
        """`pre`: a preallocated tensor for the matrix before step 22 (implies want_pre_denoise); a padded_matrix() is
        written with its leading dimension (icnv_chain_apply_ld_dev)."""
        if out is None:
            out = torch.empty_like(x)
        if pre is None and want_pre_denoise:
            pre = torch.empty_like(x)
        if pre is not None and not pre.is_contiguous():
            _, G, ld = _check_matrix_ld(pre)
            assert G == self.G and pre.shape[0] == self.C
            check(self.L.icnv_chain_apply_ld_dev(self.h, _ptr(x), _ptr(out), _ptr(pre), int(ld), _stream()))
            return out, pre
        check(self.L.icnv_chain_apply_dev(self.h, _ptr(x), _ptr(out), _ptr(pre), _stream()))
        return out, pre

    def denoise_params(self):
        buf = (ct.c_double * 2)()
        check(self.L.icnv_chain_get_denoise(self.h, buf, _stream()))
        return buf[0], buf[1]

    def close(self):
        if self.h:
            self.L.icnv_chain_end(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DevView:
    """Expose a raw device pointer through __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def _wrap_f64(ptr, n, device):
    return torch.as_tensor(_DevView(ptr, n), device=device)


def remove_outliers(x, lower=None, upper=None, out=None):
    """remove_outliers_norm (R/inferCNV_ops.R:1969-2054) on a device-resident (C, G) matrix; bounds None = "average_bound".
    Returns (matrix, (lower, upper) used)."""
    L = _lib.load()
    C, G = x.shape
    if out is None:
        out = torch.empty_like(x)
    used = (ct.c_double * 2)()
    nan = float("nan")
    check(L.icnv_remove_outliers_dev(_ptr(x), _ptr(out), G, C, nan if lower is None else float(lower),
                                     nan if upper is None else float(upper), used, _stream()))
    return out, (used[0], used[1])


def average_bounds(x):
    """get_average_bounds (R/inferCNV_ops.R:2723-2742) -> (lower, upper)."""
    L = _lib.load()
    C, G = _check_matrix(x)
    out = (ct.c_double * 2)()
    check(L.icnv_average_bounds_dev(_ptr(x), G, C, out, _stream()))
    return out[0], out[1]


# ------------------------------------------------------------------ ingest (steps 3-4)
def col_sums(x):
    """colSums per cell (R/inferCNV_ops.R:3089) -> float64 (C,) tensor."""
    L = _lib.load()
    C, G = _check_matrix(x)
    out = torch.empty(C, dtype=torch.float64, device=x.device)
    check(L.icnv_col_sums_dev(_ptr(x), G, C, _ptr(out), _stream()))
    return out


def normalize_log2(x, col_sums_t=None, normalize_factor=None, do_normalize=True, do_log2=True, out=None):
    """normalize_counts_by_seq_depth + log2xplus1 (R/inferCNV_ops.R:3064-3111, 2756-2769).  With
    normalize_factor=None the factor is median(colSums) of THIS matrix; a cell-sharded caller
    all-gathers the column sums and passes the global median instead."""
    L = _lib.load()
    C, G = _check_matrix(x)
    if do_normalize and col_sums_t is None:
        col_sums_t = col_sums(x)
    if do_normalize and normalize_factor is None:
        srt = torch.sort(col_sums_t).values
        n = srt.numel()
        normalize_factor = float(srt[n // 2]) if n % 2 else float((srt[n // 2 - 1] + srt[n // 2]) * 0.5)
    if out is None:
        out = torch.empty_like(x)
    check(L.icnv_normalize_log2_dev(_ptr(x), _ptr(out), G, C, _ptr(col_sums_t), float(normalize_factor or 0.0),
                                    int(do_normalize), int(do_log2), _stream()))
    return out


# ------------------------------------------------------------------ HMM
def viterbi_cells(x, chr_start, means, sd_shared, logPi, logDelta, states=None):
    """predict_CNV_via_HMM_on_indiv_cells (R/inferCNV_HMM.R:284-324) / i3 variant
    (R/inferCNV_i3HMM.R:180-225).  Returns (states uint8 (C, G), n_underflow int32[1] tensor)."""
    L = _lib.load()
    C, G, ld_x = _check_matrix_ld(x)
    cs, cp = i32(chr_start)
    m, mp = f64(means)
    lp = np.asfortranarray(logPi, dtype=np.float64)
    ld, ldp = f64(logDelta)
    if states is None:
        # (a padded input gets padded states: the traceback then writes aligned 16-byte words for every cell)
        states = torch.empty((C, G), dtype=torch.uint8, device=x.device) if ld_x == G else padded_matrix(C, G, torch.uint8, x.device)
    _, Gs, ld_st = _check_matrix_ld(states, torch.uint8)
    assert Gs == G and states.shape[0] == C
    bad = torch.zeros(1, dtype=torch.int32, device=x.device)
    if ld_x == G and ld_st == G:
        check(L.icnv_viterbi_cells_dev(_ptr(x), _ptr(states), G, C, cp, cs.size - 1, m.size, mp, float(sd_shared),
                                       lp.ctypes.data_as(ct.POINTER(ct.c_double)), ldp, _ptr(bad), _stream()))
    else:
        check(L.icnv_viterbi_cells_ld_dev(_ptr(x), int(ld_x), _ptr(states), int(ld_st), G, C, cp, cs.size - 1, m.size, mp,
                                          float(sd_shared), lp.ctypes.data_as(ct.POINTER(ct.c_double)), ldp, _ptr(bad), _stream()))
    return states, bad


def viterbi_set_mode(mode):
    """0 = auto (certified fast path when the parameters are eligible), 1 = exact kernel only, 2 = auto without the
    staged fast kernel (developer A/B)."""
    check(_lib.load().icnv_viterbi_set_mode(int(mode)))


def viterbi_last_stats():
    """{path, sequences, flagged, table_intervals, fallback} of the last per-cell Viterbi call on this device
    (synchronises with it).  path "fast": the certified fast kernel ran; fallback: its last column batch had so many
    flagged sequences that the exact kernel recomputed the whole batch (decided on the device, per batch)."""
    buf = (ct.c_int64 * 4)()
    check(_lib.load().icnv_viterbi_last_stats(buf))
    return {"path": "fast" if buf[0] >= 1 else "exact", "sequences": int(buf[1]), "flagged": int(buf[2]),
            "table_intervals": int(buf[3]), "fallback": buf[0] == 2,
            # which fast kernel: "staged" (observations through LDS, short table), "register" (full table), "staged+register"
            # (the staged kernel's batch left its table and was redone with the full one)
            "kernel": {0: "exact", 1: "register", 2: "exact", 3: "staged", 4: "staged+register"}[int(buf[0])]}


def viterbi_groups(x, chr_start, groups, means, sd_shared_per_group, logPi, logDelta, states=None):
    """predict_CNV_via_HMM_on_tumor_subclusters / _whole_tumor_samples
    (R/inferCNV_HMM.R:345-408, 509-567; i3: R/inferCNV_i3HMM.R:249-389)."""
    L = _lib.load()
    C, G = _check_matrix(x)
    cs, cp = i32(chr_start)
    idx, off = pack_groups(groups)
    idx, ip = i32(idx)
    off, op = i32(off)
    m, mp = f64(means)
    sd, sdp = f64(sd_shared_per_group)
    lp = np.asfortranarray(logPi, dtype=np.float64)
    ld, ldp = f64(logDelta)
    if states is None:
        states = torch.empty((C, G), dtype=torch.uint8, device=x.device)
    bad = torch.zeros(1, dtype=torch.int32, device=x.device)
    check(L.icnv_viterbi_groups_dev(_ptr(x), _ptr(states), G, C, cp, cs.size - 1, ip, op, len(groups), m.size, mp,
                                    sdp, lp.ctypes.data_as(ct.POINTER(ct.c_double)), ldp, _ptr(bad), _stream()))
    return states, bad


class GroupHMMPlan:
    """The i3 HMM at group level with device-resident parameters (icnv_group_hmm_* in include/icnv.h): the group structure is
    uploaded once; `i3_partial(x)` returns the three shifted moments as a CUDA tensor view (all-reduce it in place in a
    cell-sharded run), `i3_finish(states, ...)` derives mu / sigma / delta on the device and runs the Viterbi + broadcast.
    No host round trip inside a step."""

    def __init__(self, G, C, chr_start, groups, ref_cells):
        self.L = _lib.load()
        self.G, self.C = int(G), int(C)
        self.cs, cp = i32(chr_start)
        idx, off = pack_groups(groups)
        self.idx, ip = i32(idx)
        self.off, op = i32(off)
        self.ref, rp = i32(ref_cells)
        h = ct.c_void_p()
        check(self.L.icnv_group_hmm_begin(ct.byref(h), self.G, self.C, cp, self.cs.size - 1, ip, op, len(groups), rp, self.ref.size))
        self.h = h

    def i3_partial(self, x):
        C, G = _check_matrix(x)
        assert (C, G) == (self.C, self.G)
        p = ct.c_void_p()
        check(self.L.icnv_group_hmm_i3_partial_dev(self.h, _ptr(x), ct.byref(p), _stream()))
        return _wrap_f64(p.value, 3, x.device)

    def i3_finish(self, logPi, logDelta, z_abs, delta_abs=None, states=None, device=None):
        if states is None:
            states = torch.empty((self.C, self.G), dtype=torch.uint8, device=device or "cuda")
        lp = np.asfortranarray(logPi, dtype=np.float64)
        ld, ldp = f64(logDelta)
        check(self.L.icnv_group_hmm_i3_finish_dev(self.h, _ptr(states), lp.ctypes.data_as(ct.POINTER(ct.c_double)), ldp, float(z_abs),
                                                  float("nan") if delta_abs is None else float(delta_abs), None, _stream()))
        return states

    def i3_params(self):
        """(mu, sigma, delta) of the last finish (synchronises)."""
        buf = (ct.c_double * 3)()
        check(self.L.icnv_group_hmm_get_i3_params(self.h, buf, _stream()))
        return buf[0], buf[1], buf[2]

    def close(self):
        if self.h:
            self.L.icnv_group_hmm_end(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def group_means(x, groups):
    """rowMeans(expr.data[, group]) per group -> (n_groups, G) tensor."""
    L = _lib.load()
    C, G = _check_matrix(x)
    idx, off = pack_groups(groups)
    idx, ip = i32(idx)
    off, op = i32(off)
    out = torch.empty((len(groups), G), dtype=torch.float64, device=x.device)
    check(L.icnv_group_means_dev(_ptr(x), G, C, ip, op, len(groups), _ptr(out), _stream()))
    return out


def cell_distances(x, cells):
    """parallelDist(t(expr.data[, cells])) (Euclidean; R/inferCNV_tumor_subclusters.R:191) -> (n, n) tensor."""
    L = _lib.load()
    C, G = _check_matrix(x)
    idx, ip = i32(cells)
    out = torch.empty((idx.size, idx.size), dtype=torch.float64, device=x.device)
    check(L.icnv_cell_distances_dev(_ptr(x), G, C, ip, idx.size, _ptr(out), _stream()))
    return out


def state_consensus(states, groups, overwrite=False):
    """.get_state_consensus (R/inferCNV_HMM.R:977-987) per group -> (n_groups, G) uint8; with
    overwrite=True also returns the state matrix with every member cell set to its group's consensus."""
    L = _lib.load()
    C, G = _check_matrix(states, torch.uint8)
    idx, off = pack_groups(groups)
    idx, ip = i32(idx)
    off, op = i32(off)
    cons = torch.empty((len(groups), G), dtype=torch.uint8, device=states.device)
    out = torch.empty_like(states) if overwrite else None
    check(L.icnv_state_consensus_dev(_ptr(states), G, C, ip, op, len(groups), _ptr(cons), _ptr(out), _stream()))
    return (cons, out) if overwrite else cons


def states_to_proxy(states, K):
    """assign_HMM_states_to_proxy_expr_vals (R/inferCNV_HMM.R:1191-1206) / i3 (R/inferCNV_i3HMM.R:405-417)."""
    L = _lib.load()
    _check_matrix(states, torch.uint8)
    out = torch.empty(states.shape, dtype=torch.float64, device=states.device)
    check(L.icnv_states_to_proxy_dev(_ptr(states), _ptr(out), states.numel(), int(K), _stream()))
    return out


def cells_mean_sd(x, cell_idx):
    """mean and sd over ALL values of the listed cells (R/inferCNV_i3HMM.R:17-80)."""
    L = _lib.load()
    C, G = _check_matrix(x)
    idx, ip = i32(cell_idx)
    out = (ct.c_double * 2)()
    check(L.icnv_cells_mean_sd_dev(_ptr(x), G, C, ip, idx.size, out, _stream()))
    return out[0], out[1]


# ------------------------------------------------------------------ ingest from integer counts (steps 2-4)
class DeviceCounts:
    """The raw count matrix on the device: dense int32 (C, G) tensor -- row-major (C, G) is R's column-major G x C -- or
    CSC (colptr int64 [C + 1], rowidx int32 [nnz], vals int32 [nnz]) tensors."""

    def __init__(self, G, C, dense=None, colptr=None, rowidx=None, vals=None):
        self.G, self.C = int(G), int(C)
        self.t = (dense, colptr, rowidx, vals)          # keeps the tensors alive
        if dense is not None:
            assert dense.is_cuda and dense.dtype == torch.int32 and dense.is_contiguous() and tuple(dense.shape) == (C, G)
            self.c = _lib.Counts(dense.data_ptr(), None, None, None, 0)
        else:
            assert colptr.dtype == torch.int64 and rowidx.dtype == torch.int32 and vals.dtype == torch.int32
            assert colptr.numel() == C + 1 and rowidx.numel() == vals.numel()
            self.c = _lib.Counts(None, colptr.data_ptr(), rowidx.data_ptr() if rowidx.numel() else None,
                                 vals.data_ptr() if vals.numel() else None, int(vals.numel()))

    @property
    def device(self):
        return next(t for t in self.t if t is not None).device


def ingest_gene_stats(counts):
    """[G sums | G counts of cells with a positive count] as one float64 tensor (all-reduce it in a sharded run)."""
    L = _lib.load()
    out = torch.empty(2 * counts.G, dtype=torch.float64, device=counts.device)
    check(L.icnv_ingest_gene_stats_dev(ct.byref(counts.c), counts.G, counts.C, _ptr(out), _stream()))
    return out


def ingest_select(stats_host, G, C_total, min_mean_expr_cutoff=None, min_cells_per_gene=0):
    """Step 2's decision from the gene statistics (R/inferCNV_ops.R:2128-2213) -> kept gene indices."""
    L = _lib.load()
    st, sp = f64(stats_host)
    keep = np.empty(G, dtype=np.int32)
    n = ct.c_int64()
    check(L.icnv_ingest_select(sp, G, int(C_total), float("nan") if min_mean_expr_cutoff is None else float(min_mean_expr_cutoff),
                               int(min_cells_per_gene), keep.ctypes.data_as(ct.POINTER(ct.c_int32)), ct.byref(n)))
    return keep[:n.value].copy()


def ingest_col_sums(counts, keep_idx):
    L = _lib.load()
    mask = torch.zeros(counts.G, dtype=torch.uint8, device=counts.device)
    mask[torch.as_tensor(np.asarray(keep_idx, dtype=np.int64), device=counts.device)] = 1
    out = torch.empty(counts.C, dtype=torch.float64, device=counts.device)
    check(L.icnv_ingest_col_sums_dev(ct.byref(counts.c), counts.G, counts.C, _ptr(mask), _ptr(out), _stream()))
    torch.cuda.current_stream().synchronize()           # (the mask is released when this returns)
    return out


def ingest_apply(counts, keep_idx, col_sums, factor):
    """log2(count / colSum * factor + 1) of the kept genes -> (C, G_out) float64 tensor."""
    L = _lib.load()
    keep = torch.as_tensor(np.asarray(keep_idx, dtype=np.int32), device=counts.device)
    out = torch.empty((counts.C, int(keep.numel())), dtype=torch.float64, device=counts.device)
    check(L.icnv_ingest_apply_dev(ct.byref(counts.c), counts.G, counts.C, _ptr(keep), int(keep.numel()), _ptr(col_sums), float(factor),
                                  1, 1, _ptr(out), _stream()))
    torch.cuda.current_stream().synchronize()
    return out


def ingest_counts(counts, min_mean_expr_cutoff=None, min_cells_per_gene=0, normalize_factor=None):
    """Steps 2-4 of run() in one call on one device -> (expr (C, G_out) float64, kept gene indices, factor used)."""
    L = _lib.load()
    keep = np.empty(counts.G, dtype=np.int32)
    n, used = ct.c_int64(), ct.c_double()
    buf = torch.empty((counts.C * counts.G,), dtype=torch.float64, device=counts.device)
    check(L.icnv_ingest_counts_dev(ct.byref(counts.c), counts.G, counts.C,
                                   float("nan") if min_mean_expr_cutoff is None else float(min_mean_expr_cutoff), int(min_cells_per_gene),
                                   float("nan") if normalize_factor is None else float(normalize_factor),
                                   keep.ctypes.data_as(ct.POINTER(ct.c_int32)), ct.byref(n), _ptr(buf), ct.byref(used), _stream()))
    g_out = n.value
    return buf[: counts.C * g_out].view(counts.C, g_out), keep[:g_out].copy(), used.value


def cells_moments_partial(x, cell_idx, phase, mean=0.0):
    """One rank's share of the split-phase mean / sd over all values of the listed cells (icnv_cells_moments_partial_dev):
    phase 0 -> (sum of values, number of values), phase 1 -> (sum of (x - mean)^2, number of values)."""
    L = _lib.load()
    C, G = _check_matrix(x)
    idx, ip = i32(cell_idx)
    out = (ct.c_double * 3)()
    check(L.icnv_cells_moments_partial_dev(_ptr(x), G, C, ip, idx.size, int(phase), float(mean), out, _stream()))
    return out[0], out[1]


# ------------------------------------------------------------------ median filter
def median_filter(x, chr_start, tiles, window_size=7, out=None):
    """apply_median_filtering (R/noise_reduction.R:43-113) on device tensors."""
    L = _lib.load()
    C, G = _check_matrix(x)
    cs, cp = i32(chr_start)
    idx, off = pack_groups(tiles)
    idx, ip = i32(idx)
    off, op = i32(off)
    if out is None:
        out = torch.empty_like(x)
    check(L.icnv_median_filter_dev(_ptr(x), _ptr(out), G, C, cp, cs.size - 1, ip, op, len(tiles), int(window_size),
                                   _stream()))
    return out


# ------------------------------------------------------------------ timing hooks
def timing_enable(on=True):
    """on: False / 0 off, True / 1 every kernel family, 2 only the hot launches "chain_apply" and "viterbi"."""
    _lib.load().icnv_timing_enable(2 if on == 2 and on is not True else int(bool(on)))


def timing_reset():
    _lib.load().icnv_timing_reset()


def timing_get(kernel):
    ms, n = ct.c_double(), ct.c_int64()
    check(_lib.load().icnv_timing_get(kernel.encode(), ct.byref(ms), ct.byref(n)))
    return ms.value, n.value
