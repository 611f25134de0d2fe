"""ctypes binding of the C ABI in include/icnv.h (libicnv_hip.so).

The library is built in-tree (`infercnv_amd/libicnv_hip.so`, see
`__graft_entry__.build()` / `infercnv_amd/csrc/Makefile`).  There is NO CPU
fallback: if the shared object is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as ct
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICNV_LIB", os.path.join(_HERE, "libicnv_hip.so"))

ST_SUBTRACT_REF_1 = 0x01
ST_MAX_THRESH = 0x02
ST_SMOOTH = 0x04
ST_CENTER = 0x08
ST_SUBTRACT_REF_2 = 0x10
ST_INVERT_LOG2 = 0x20
ST_DENOISE = 0x40
ST_ALL = 0x7F
ST_CENTER_MEAN = 0x80
ST_NA_AWARE = 0x100     # the matrix may hold NaN: cells that do are recomputed with the reference's NA semantics

OK, ERR_ARG, ERR_HIP, ERR_UNSUPPORTED, ERR_UNDERFLOW, ERR_NOMEM = 0, 1, 2, 3, 4, 5


class IcnvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libicnv_hip error {code}: {msg}")
        self.code = code


class ChainCfg(ct.Structure):
    _fields_ = [
        ("G", ct.c_int64), ("C", ct.c_int64),
        ("chr_start", ct.POINTER(ct.c_int32)), ("n_chr", ct.c_int32),
        ("window_length", ct.c_int32),
        ("max_thresh", ct.c_double),
        ("use_bounds", ct.c_int32),
        ("inv_log", ct.c_int32),
        ("sd_amplifier", ct.c_double),
        ("noise_filter", ct.c_double),
        ("stage_mask", ct.c_uint32),
        ("noise_logistic", ct.c_int32),
        ("ref_idx", ct.POINTER(ct.c_int32)),
        ("ref_off", ct.POINTER(ct.c_int32)),
        ("n_ref_grp", ct.c_int32),
    ]


class Counts(ct.Structure):
    """icnv_counts of include/icnv.h: the raw count matrix, dense int32 or CSC (pointers: host or device, by entry point)."""
    _fields_ = [("dense", ct.c_void_p), ("colptr", ct.c_void_p), ("rowidx", ct.c_void_p), ("vals", ct.c_void_p), ("nnz", ct.c_int64)]


_vp, _i64, _i32, _dbl = ct.c_void_p, ct.c_int64, ct.c_int32, ct.c_double
_ip = ct.POINTER(ct.c_int32)
_dp = ct.POINTER(ct.c_double)

# name -> (restype, argtypes); every symbol include/icnv.h declares
PROTOTYPES = {
    "icnv_version": (ct.c_int, []),
    "icnv_last_error": (ct.c_char_p, []),
    "icnv_init": (ct.c_int, [ct.c_int]),
    "icnv_shutdown": (None, []),
    "icnv_set_devices": (ct.c_int, [ct.c_int]),
    "icnv_get_devices": (ct.c_int, []),
    "icnv_residency": (ct.c_int, [ct.c_int]),
    "icnv_residency_drop": (None, []),
    "icnv_residency_stats": (ct.c_int, [ct.POINTER(_i64)]),
    "icnv_host_path_stats": (ct.c_int, [_dp, _i32]),
    "icnv_host_path_stats_reset": (None, []),
    "icnv_smooth_chain": (ct.c_int, [_vp, _vp, _vp, ct.POINTER(ChainCfg)]),
    "icnv_smooth_chain_dev": (ct.c_int, [_vp, _vp, _vp, ct.POINTER(ChainCfg), _vp]),
    "icnv_chain_begin": (ct.c_int, [ct.POINTER(_vp), ct.POINTER(ChainCfg)]),
    "icnv_chain_num_rounds": (ct.c_int, [_vp]),
    "icnv_chain_round_partial_dev": (ct.c_int, [_vp, ct.c_int, _vp, ct.POINTER(_vp), ct.POINTER(_i64), _vp]),
    "icnv_chain_round_finish_dev": (ct.c_int, [_vp, ct.c_int, _vp]),
    "icnv_chain_apply_dev": (ct.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "icnv_chain_apply_ld_dev": (ct.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "icnv_chain_get_denoise": (ct.c_int, [_vp, _dp, _vp]),
    "icnv_chain_end": (None, [_vp]),
    "icnv_average_bounds": (ct.c_int, [_vp, _i64, _i64, _dp]),
    "icnv_average_bounds_dev": (ct.c_int, [_vp, _i64, _i64, _dp, _vp]),
    "icnv_scale_genes": (ct.c_int, [_vp, _vp, _i64, _i64]),
    "icnv_scale_genes_dev": (ct.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "icnv_remove_outliers": (ct.c_int, [_vp, _vp, _i64, _i64, _dbl, _dbl, _dp]),
    "icnv_remove_outliers_dev": (ct.c_int, [_vp, _vp, _i64, _i64, _dbl, _dbl, _dp, _vp]),
    "icnv_col_sums_dev": (ct.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "icnv_normalize_log2_dev": (ct.c_int, [_vp, _vp, _i64, _i64, _vp, _dbl, _i32, _i32, _vp]),
    "icnv_normalize_log2": (ct.c_int, [_vp, _vp, _i64, _i64, _dbl, _i32, _i32, _dp]),
    "icnv_viterbi_cells": (ct.c_int, [_vp, _vp, _i64, _i64, _ip, _i32, _i32, _dp, _dbl, _dp, _dp]),
    "icnv_viterbi_cells_dev": (ct.c_int, [_vp, _vp, _i64, _i64, _ip, _i32, _i32, _dp, _dbl, _dp, _dp, _vp, _vp]),
    "icnv_group_hmm_begin": (ct.c_int, [_vp, _i64, _i64, _ip, _i32, _ip, _ip, _i32, _ip, _i64]),
    "icnv_group_hmm_i3_partial_dev": (ct.c_int, [_vp, _vp, _vp, _vp]),
    "icnv_group_hmm_i3_finish_dev": (ct.c_int, [_vp, _vp, _dp, _dp, _dbl, _dbl, _vp, _vp]),
    "icnv_group_hmm_get_i3_params": (ct.c_int, [_vp, _dp, _vp]),
    "icnv_group_hmm_end": (None, [_vp]),
    "icnv_viterbi_cells_ld_dev": (ct.c_int, [_vp, _i64, _vp, _i64, _i64, _i64, _ip, _i32, _i32, _dp, _dbl, _dp, _dp, _vp, _vp]),
    "icnv_viterbi_groups": (ct.c_int, [_vp, _vp, _i64, _i64, _ip, _i32, _ip, _ip, _i32, _i32, _dp, _dp, _dp, _dp]),
    "icnv_viterbi_groups_dev": (ct.c_int, [_vp, _vp, _i64, _i64, _ip, _i32, _ip, _ip, _i32, _i32, _dp, _dp, _dp, _dp,
                                           _vp, _vp]),
    "icnv_viterbi_set_mode": (ct.c_int, [ct.c_int]),
    "icnv_viterbi_last_stats": (ct.c_int, [ct.POINTER(_i64)]),
    "icnv_hmm_emission_table": (ct.c_int, [_i32, _dp, _dbl, _dp, _dp, _dp, _i64]),
    "icnv_hmm_emission_scores": (ct.c_int, [_i32, _dp, _dbl, _dp, _i64, _i32, _dp, _vp]),
    "icnv_cell_distances": (ct.c_int, [_vp, _i64, _i64, _ip, _i64, _vp]),
    "icnv_cell_distances_dev": (ct.c_int, [_vp, _i64, _i64, _ip, _i64, _vp, _vp]),
    "icnv_group_means_dev": (ct.c_int, [_vp, _i64, _i64, _ip, _ip, _i32, _vp, _vp]),
    "icnv_gene_stats": (ct.c_int, [_vp, _i64, _i64, _vp, _vp]),
    "icnv_gene_stats_dev": (ct.c_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "icnv_select_genes": (ct.c_int, [_vp, _i64, _i64, _ip, _i64, _vp]),
    "icnv_select_genes_dev": (ct.c_int, [_vp, _i64, _i64, _ip, _i64, _vp, _vp]),
    "icnv_block_mean_sd": (ct.c_int, [_vp, _i64, _i64, _ip, _i64, _ip, _i64, _dp]),
    "icnv_block_mean_sd_dev": (ct.c_int, [_vp, _i64, _i64, _ip, _i64, _ip, _i64, _dp, _vp]),
    "icnv_state_consensus": (ct.c_int, [_vp, _i64, _i64, _ip, _ip, _i32, _vp, _vp]),
    "icnv_state_consensus_dev": (ct.c_int, [_vp, _i64, _i64, _ip, _ip, _i32, _vp, _vp, _vp]),
    "icnv_states_to_proxy": (ct.c_int, [_vp, _vp, _i64, _i32]),
    "icnv_states_to_proxy_dev": (ct.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "icnv_cells_mean_sd_dev": (ct.c_int, [_vp, _i64, _i64, _ip, _i64, _dp, _vp]),
    "icnv_cells_mean_sd": (ct.c_int, [_vp, _i64, _i64, _ip, _i64, _dp]),
    "icnv_ingest_counts": (ct.c_int, [ct.POINTER(Counts), _i64, _i64, ct.c_double, ct.c_int32, ct.c_double, _ip, ct.POINTER(_i64), _vp,
                                      _dp, ct.POINTER(_i64)]),
    "icnv_ingest_counts_dev": (ct.c_int, [ct.POINTER(Counts), _i64, _i64, ct.c_double, ct.c_int32, ct.c_double, _ip, ct.POINTER(_i64),
                                          _vp, _dp, _vp]),
    "icnv_ingest_gene_stats_dev": (ct.c_int, [ct.POINTER(Counts), _i64, _i64, _vp, _vp]),
    "icnv_ingest_select": (ct.c_int, [_dp, _i64, _i64, ct.c_double, ct.c_int32, _ip, ct.POINTER(_i64)]),
    "icnv_ingest_col_sums_dev": (ct.c_int, [ct.POINTER(Counts), _i64, _i64, _vp, _vp, _vp]),
    "icnv_ingest_apply_dev": (ct.c_int, [ct.POINTER(Counts), _i64, _i64, _vp, _i64, _vp, ct.c_double, ct.c_int32, ct.c_int32, _vp, _vp]),
    "icnv_gather_values_dev": (ct.c_int, [_vp, _i64, ct.POINTER(_i64), _i64, _dp, _vp]),
    "icnv_gather_values": (ct.c_int, [_vp, _i64, _i64, ct.POINTER(_i64), _i64, _dp]),
    "icnv_cells_moments_partial_dev": (ct.c_int, [_vp, _i64, _i64, _ip, _i64, ct.c_int32, ct.c_double, _dp, _vp]),
    "icnv_median_filter": (ct.c_int, [_vp, _vp, _i64, _i64, _ip, _i32, _ip, _ip, _i32, _i32]),
    "icnv_median_filter_dev": (ct.c_int, [_vp, _vp, _i64, _i64, _ip, _i32, _ip, _ip, _i32, _i32, _vp]),
    "icnv_timing_enable": (None, [ct.c_int]),
    "icnv_timing_reset": (None, []),
    "icnv_timing_get": (ct.c_int, [ct.c_char_p, _dp, ct.POINTER(_i64)]),
}

_lib = None


def load():
    """Load libicnv_hip.so and bind every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C infercnv_amd/csrc`).  infercnv_amd has no CPU fallback.")
    L = ct.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(L, name)  # AttributeError here == a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise IcnvError(rc, load().icnv_last_error().decode("utf-8", "replace"))


def i32(a):
    """Contiguous int32 array + pointer (keep the array alive while the pointer is in use)."""
    arr = np.ascontiguousarray(a, dtype=np.int32)
    return arr, arr.ctypes.data_as(_ip)


def f64(a):
    arr = np.ascontiguousarray(a, dtype=np.float64)
    return arr, arr.ctypes.data_as(_dp)


def pack_groups(groups):
    """list of 0-based index vectors -> (concatenated int32, int32 offsets)."""
    off = np.zeros(len(groups) + 1, dtype=np.int32)
    for k, g in enumerate(groups):
        off[k + 1] = off[k] + len(g)
    idx = (np.concatenate([np.asarray(g, dtype=np.int32).ravel() for g in groups])
           if len(groups) else np.zeros(0, dtype=np.int32))
    return idx.astype(np.int32), off


class Cfg:
    """Owns the numpy arrays a ChainCfg points to."""

    def __init__(self, G, C, chr_start, ref_groups, window_length=101, max_thresh=3.0, use_bounds=True,
                 sd_amplifier=1.5, noise_filter=None, stage_mask=ST_ALL, inv_log=False, noise_logistic=False):
        self.chr_start, cp = i32(chr_start)
        idx, off = pack_groups(list(ref_groups) if ref_groups is not None else [])
        self.ref_idx, ip = i32(idx)
        self.ref_off, op = i32(off)
        c = ChainCfg()
        c.G, c.C = int(G), int(C)
        c.chr_start, c.n_chr = cp, self.chr_start.size - 1
        c.window_length = int(window_length)
        c.max_thresh = float("nan") if max_thresh is None else float(max_thresh)
        c.use_bounds = int(bool(use_bounds))
        c.inv_log = int(bool(inv_log))
        c.sd_amplifier = float(sd_amplifier)
        c.noise_filter = float("nan") if noise_filter is None else float(noise_filter)
        c.stage_mask = int(stage_mask)
        c.noise_logistic = int(bool(noise_logistic))
        c.ref_idx, c.ref_off = ip, op
        c.n_ref_grp = len(off) - 1
        self.c = c

    def ptr(self):
        return ct.byref(self.c)
