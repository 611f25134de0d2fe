"""ctypes binding of oracle/libicnv_oracle.so (the plain-C CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Matrices are passed as Fortran-ordered (G, C)
float64 arrays == the reference's column-major genes x cells expr.data.
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libicnv_oracle.so")

ST8, ST9, ST10, ST11, ST12, ST14, ST22 = 1, 2, 4, 8, 16, 32, 64
ST_ALL = 127


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "icnv_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libicnv_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_dp = ct.POINTER(ct.c_double)
_ip = ct.POINTER(ct.c_int32)
_bp = ct.POINTER(ct.c_uint8)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ct.CDLL(_SO)
        i64, i32, dbl = ct.c_int64, ct.c_int32, ct.c_double
        L.orc_num_threads.restype = ct.c_int
        L.orc_set_num_threads.argtypes = [ct.c_int]
        L.orc_normalize_log2.argtypes = [_dp, i64, i64, dbl, i32, i32]
        L.orc_normalize_log2.restype = dbl
        L.orc_ref_group_means.argtypes = [_dp, i64, i64, _ip, _ip, i32, i32, _dp]
        L.orc_ref_group_means.restype = ct.c_int
        L.orc_subtract_ref.argtypes = [_dp, i64, i64, _dp, i32, i32]
        L.orc_clamp.argtypes = [_dp, i64, i64, dbl]
        L.orc_average_bounds.argtypes = [_dp, i64, i64, _dp]
        L.orc_smooth_by_chr.argtypes = [_dp, i64, i64, _ip, i32, i32]
        L.orc_center.argtypes = [_dp, i64, i64, i32]
        L.orc_exp2.argtypes = [_dp, i64, i64]
        L.orc_denoise_params.argtypes = [_dp, i64, i64, _ip, i64, dbl, _dp, _dp]
        L.orc_denoise_apply.argtypes = [_dp, i64, i64, dbl, dbl]
        L.orc_smooth_chain.argtypes = [_dp, i64, i64, _ip, i32, _ip, _ip, i32, i32, dbl, i32, dbl, dbl,
                                       ct.c_uint32, _dp, _dp]
        L.orc_smooth_chain.restype = ct.c_int
        L.orc_log.argtypes = [dbl]
        L.orc_log.restype = dbl
        L.orc_log_array.argtypes = [_dp, _dp, i64]
        L.orc_fma_array.argtypes = [_dp, _dp, _dp, _dp, i64]
        L.orc_pnorm_log_upper.argtypes = [dbl]
        L.orc_pnorm_log_upper.restype = dbl
        L.orc_viterbi_cells.argtypes = [_dp, _bp, i64, i64, _ip, i32, i32, _dp, dbl, _dp, _dp]
        L.orc_viterbi_cells.restype = ct.c_int
        L.orc_group_means.argtypes = [_dp, i64, i64, _ip, _ip, i32, _dp]
        L.orc_viterbi_groups.argtypes = [_dp, _bp, i64, i64, _ip, i32, _ip, _ip, i32, i32, _dp, _dp, _dp, _dp]
        L.orc_viterbi_groups.restype = ct.c_int
        L.orc_states_to_proxy.argtypes = [_bp, _dp, i64, i32]
        L.orc_median_filter.argtypes = [_dp, _dp, i64, i64, _ip, i32, _ip, _ip, i32, i32]
        L.orc_mean_sd_of_cells.argtypes = [_dp, i64, _ip, i64, _dp, _dp]
        _lib = L
    return _lib


def _f(a):
    a = np.asfortranarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def pack_groups(groups):
    """list of index arrays -> (concatenated int32 idx, int32 offsets)."""
    off = np.zeros(len(groups) + 1, dtype=np.int32)
    for k, g in enumerate(groups):
        off[k + 1] = off[k] + len(g)
    idx = np.concatenate([np.asarray(g, dtype=np.int32) for g in groups]) if groups else np.zeros(0, np.int32)
    return idx.astype(np.int32), off


def chr_starts_from_codes(chr_codes):
    """Contiguous chromosome blocks -> n_chr+1 offsets (raises if a chr is split)."""
    chr_codes = np.asarray(chr_codes)
    cut = np.nonzero(chr_codes[1:] != chr_codes[:-1])[0] + 1
    starts = np.concatenate([[0], cut, [chr_codes.size]]).astype(np.int32)
    seen = chr_codes[starts[:-1]]
    if len(set(seen.tolist())) != seen.size:
        raise ValueError("genes of one chromosome must be contiguous")
    return starts


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def num_threads():
    return lib().orc_num_threads()


def log(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().orc_log_array(x.ctypes.data_as(_dp), out.ctypes.data_as(_dp), x.size)
    return out


def fma(a, b, c):
    """Correctly rounded fused multiply-add (libm), broadcast like NumPy."""
    a, b, c = np.broadcast_arrays(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64),
                                  np.asarray(c, dtype=np.float64))
    a, b, c = (np.ascontiguousarray(v) for v in (a, b, c))
    out = np.empty(a.shape, dtype=np.float64)
    lib().orc_fma_array(a.ctypes.data_as(_dp), b.ctypes.data_as(_dp), c.ctypes.data_as(_dp),
                        out.ctypes.data_as(_dp), out.size)
    return out


def pnorm_log_upper(y):
    L = lib()
    return np.array([L.orc_pnorm_log_upper(float(v)) for v in np.ravel(y)]).reshape(np.shape(y))


def normalize_log2(expr, normalize_factor=None, do_norm=True, do_log=True):
    """Steps 3-4 -> (matrix, factor used)."""
    x = np.array(expr, dtype=np.float64, order="F", copy=True)
    f = lib().orc_normalize_log2(x.ctypes.data_as(_dp), x.shape[0], x.shape[1],
                                 float("nan") if normalize_factor is None else float(normalize_factor),
                                 int(do_norm), int(do_log))
    return x, f


def ref_group_means(expr, ref_groups, inv_log=False):
    x, xp = _f(expr)
    G, C = x.shape
    idx, off = pack_groups(ref_groups)
    idx, ip = _i(idx)
    off, op = _i(off)
    out = np.zeros((G, len(ref_groups)), dtype=np.float64, order="F")
    rc = lib().orc_ref_group_means(xp, G, C, ip, op, len(ref_groups), int(inv_log), out.ctypes.data_as(_dp))
    if rc:
        raise ValueError("empty reference group")
    return out


def subtract_ref_expr_from_obs(expr, ref_groups, inv_log=False, use_bounds=True):
    x = np.array(expr, dtype=np.float64, order="F", copy=True)
    G, C = x.shape
    m = ref_group_means(x, ref_groups, inv_log)
    lib().orc_subtract_ref(x.ctypes.data_as(_dp), G, C, m.ctypes.data_as(_dp), m.shape[1], int(use_bounds))
    return x


def apply_max_threshold_bounds(expr, thr):
    x = np.array(expr, dtype=np.float64, order="F", copy=True)
    lib().orc_clamp(x.ctypes.data_as(_dp), x.shape[0], x.shape[1], float(thr))
    return x


def get_average_bounds(expr):
    x, xp = _f(expr)
    out = np.zeros(2)
    lib().orc_average_bounds(xp, x.shape[0], x.shape[1], out.ctypes.data_as(_dp))
    return out


def smooth_by_chromosome(expr, chr_start, window_length):
    x = np.array(expr, dtype=np.float64, order="F", copy=True)
    cs, cp = _i(chr_start)
    lib().orc_smooth_by_chr(x.ctypes.data_as(_dp), x.shape[0], x.shape[1], cp, cs.size - 1, int(window_length))
    return x


def center_columns(expr, method="median"):
    x = np.array(expr, dtype=np.float64, order="F", copy=True)
    lib().orc_center(x.ctypes.data_as(_dp), x.shape[0], x.shape[1], 0 if method == "median" else 1)
    return x


def invert_log2(expr):
    x = np.array(expr, dtype=np.float64, order="F", copy=True)
    lib().orc_exp2(x.ctypes.data_as(_dp), x.shape[0], x.shape[1])
    return x


def denoise_params(expr, ref_idx, sd_amplifier=1.5):
    x, xp = _f(expr)
    idx, ip = _i(ref_idx)
    mu, s = ct.c_double(), ct.c_double()
    lib().orc_denoise_params(xp, x.shape[0], x.shape[1], ip, idx.size, float(sd_amplifier),
                             ct.byref(mu), ct.byref(s))
    return mu.value, s.value


def denoise_apply(expr, center, hw):
    x = np.array(expr, dtype=np.float64, order="F", copy=True)
    lib().orc_denoise_apply(x.ctypes.data_as(_dp), x.shape[0], x.shape[1], float(center), float(hw))
    return x


def smooth_chain(expr, chr_start, ref_groups, window_length=101, max_thresh=3.0, use_bounds=True,
                 sd_amplifier=1.5, noise_filter=float("nan"), stage_mask=ST_ALL, want_pre_denoise=False):
    """Returns (out, pre_denoise or None, (mu, s))."""
    x = np.array(expr, dtype=np.float64, order="F", copy=True)
    G, C = x.shape
    cs, cp = _i(chr_start)
    idx, off = pack_groups(ref_groups)
    idx, ip = _i(idx)
    off, op = _i(off)
    pre = np.zeros((G, C), dtype=np.float64, order="F") if want_pre_denoise else None
    ms = np.zeros(2)
    rc = lib().orc_smooth_chain(x.ctypes.data_as(_dp), G, C, cp, cs.size - 1, ip, op, len(ref_groups),
                                int(window_length), float("nan") if max_thresh is None else float(max_thresh),
                                int(use_bounds), float(sd_amplifier), float(noise_filter), int(stage_mask),
                                pre.ctypes.data_as(_dp) if pre is not None else None, ms.ctypes.data_as(_dp))
    if rc:
        raise ValueError("orc_smooth_chain failed")
    return x, pre, (ms[0], ms[1])


def viterbi_cells(expr, chr_start, means, sd, logPi, logDelta):
    """-> (states uint8 (G, C) F-order, n_underflow)."""
    x, xp = _f(expr)
    G, C = x.shape
    cs, cp = _i(chr_start)
    m, mp = _d(means)
    lp = np.asfortranarray(logPi, dtype=np.float64)
    ld, ldp = _d(logDelta)
    st = np.zeros((G, C), dtype=np.uint8, order="F")
    bad = lib().orc_viterbi_cells(xp, st.ctypes.data_as(_bp), G, C, cp, cs.size - 1, m.size, mp, float(sd),
                                  lp.ctypes.data_as(_dp), ldp)
    if bad < 0:
        raise ValueError("bad K")
    return st, bad


def group_means(expr, groups):
    x, xp = _f(expr)
    G, C = x.shape
    idx, off = pack_groups(groups)
    idx, ip = _i(idx)
    off, op = _i(off)
    out = np.zeros((G, len(groups)), dtype=np.float64, order="F")
    lib().orc_group_means(xp, G, C, ip, op, len(groups), out.ctypes.data_as(_dp))
    return out


def viterbi_groups(expr, chr_start, groups, means, sd_per_group, logPi, logDelta):
    x, xp = _f(expr)
    G, C = x.shape
    cs, cp = _i(chr_start)
    idx, off = pack_groups(groups)
    idx, ip = _i(idx)
    off, op = _i(off)
    m, mp = _d(means)
    sdv, sdp = _d(sd_per_group)
    lp = np.asfortranarray(logPi, dtype=np.float64)
    ld, ldp = _d(logDelta)
    st = np.zeros((G, C), dtype=np.uint8, order="F")
    bad = lib().orc_viterbi_groups(xp, st.ctypes.data_as(_bp), G, C, cp, cs.size - 1, ip, op, len(groups),
                                   m.size, mp, sdp, lp.ctypes.data_as(_dp), ldp)
    if bad < 0:
        raise ValueError("bad K")
    return st, bad


def states_to_proxy(states, K):
    s = np.asfortranarray(states, dtype=np.uint8)
    out = np.zeros(s.shape, dtype=np.float64, order="F")
    lib().orc_states_to_proxy(s.ctypes.data_as(_bp), out.ctypes.data_as(_dp), s.size, int(K))
    return out


def median_filter(expr, chr_start, tiles, window_size=7):
    x, xp = _f(expr)
    G, C = x.shape
    cs, cp = _i(chr_start)
    idx, off = pack_groups(tiles)
    idx, ip = _i(idx)
    off, op = _i(off)
    out = np.zeros((G, C), dtype=np.float64, order="F")
    lib().orc_median_filter(xp, out.ctypes.data_as(_dp), G, C, cp, cs.size - 1, ip, op, len(tiles),
                            int(window_size))
    return out


def mean_sd_of_cells(expr, idx):
    x, xp = _f(expr)
    ii, ip = _i(idx)
    mu, sg = ct.c_double(), ct.c_double()
    lib().orc_mean_sd_of_cells(xp, x.shape[0], ip, ii.size, ct.byref(mu), ct.byref(sg))
    return mu.value, sg.value
