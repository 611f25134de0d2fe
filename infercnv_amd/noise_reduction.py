"""apply_median_filtering (R/noise_reduction.R:43-89), same name and arguments."""
from __future__ import annotations

import ctypes as ct
import logging

import numpy as np

from . import _lib
from ._lib import check, i32, pack_groups
from .infercnv_object import InfercnvObject

log = logging.getLogger("infercnv_amd")


def apply_median_filtering(infercnv_obj: InfercnvObject, window_size=7, on_observations=True,
                           on_references=True) -> InfercnvObject:
    if window_size % 2 != 1 or window_size < 2:
        # the reference logs an error but carries on (R/noise_reduction.R:48-50)
        log.error("::apply_median_filtering: Error, window_size is an even or < 2. Please specify an odd number >= 3.")
    tiles = []
    if on_observations:   # :56-73: every subcluster of every observation group, stored cell order
        for tumor_type in infercnv_obj.observation_grouped_cell_indices:
            for idx in infercnv_obj.tumor_subclusters["subclusters"][tumor_type].values():
                tiles.append(np.asarray(idx, dtype=np.int32))
    if on_references:     # :75-86: each whole reference group
        for idx in infercnv_obj.reference_grouped_cell_indices.values():
            tiles.append(np.asarray(idx, dtype=np.int32))
    L = _lib.load()
    perm, chr_start = infercnv_obj.chr_layout()
    x = np.asfortranarray(infercnv_obj.expr_data if perm is None else infercnv_obj.expr_data[perm], dtype=np.float64)
    G, C = x.shape
    cs, cp = i32(chr_start)
    idx, off = pack_groups(tiles)
    idx, ip = i32(idx)
    off, op = i32(off)
    out = np.empty_like(x, order="F")
    check(L.icnv_median_filter(x.ctypes.data_as(ct.c_void_p), out.ctypes.data_as(ct.c_void_p), G, C, cp, cs.size - 1,
                               ip, op, len(tiles), int(window_size)))
    if perm is not None:
        inv = np.empty_like(perm)
        inv[perm] = np.arange(perm.size)
        out = out[inv]
    new = infercnv_obj.copy()
    new.expr_data = out
    return new
