"""Ingest from integer counts (icnv_ingest_counts): steps 2, 3, 4 of run() in one call from host counts --
dense int32 and CSC -- against the same steps from a double matrix through the stand-alone entry points.  Reports times,
bytes that cross PCIe and the device-resident rate of the pieces.

    python scripts/bench_ingest.py [genes] [cells] [density]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    dens = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
    import ctypes as ct
    import torch
    from infercnv_amd import _lib, device
    from infercnv_amd._lib import check
    device.init(0)
    L = _lib.load()
    rng = np.random.default_rng(5)
    counts = np.zeros((G, C), dtype=np.int32, order="F")
    nnz_per_col = int(G * dens)
    for c in range(C):                                     # scRNA-seq-like: ~10 % of the genes expressed per cell
        rows = rng.choice(G, nnz_per_col, replace=False)
        counts[rows, c] = rng.geometric(0.3, nnz_per_col)
    res = {"genes": G, "cells": C, "density": dens}

    def run(cnt_struct, label):
        keep = np.zeros(G, dtype=np.int32)
        out = np.empty((G, C), dtype=np.float64, order="F")
        g_out, used, up = ct.c_int64(), ct.c_double(), ct.c_int64()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            check(L.icnv_ingest_counts(ct.byref(cnt_struct), G, C, 0.1, 3, float("nan"), keep.ctypes.data_as(ct.POINTER(ct.c_int32)),
                                       ct.byref(g_out), out.ctypes.data_as(ct.c_void_p), ct.byref(used), ct.byref(up)))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[label] = {"ms": best * 1e3, "h2d_bytes": up.value, "d2h_bytes": int(g_out.value) * C * 8, "genes_kept": int(g_out.value),
                      "cells_per_s": C / best}
        return out.reshape(-1, order="F")[:g_out.value * C].copy(), int(g_out.value)      # G_out x C, column-major

    dense = _lib.Counts()
    dense.dense = counts.ctypes.data
    a, ga = run(dense, "dense_int32")
    # CSC
    colptr = np.zeros(C + 1, dtype=np.int64)
    nz = counts != 0
    colptr[1:] = np.cumsum(nz.sum(axis=0))
    rowidx = np.nonzero(nz.T)[1].astype(np.int32)
    vals = counts.T[nz.T].astype(np.int32)
    sp = _lib.Counts()
    sp.colptr = colptr.ctypes.data
    sp.rowidx = rowidx.ctypes.data
    sp.vals = vals.ctypes.data
    sp.nnz = int(vals.size)
    b, gb = run(sp, "csc")
    res["dense_equals_csc"] = bool(ga == gb and np.array_equal(a, b))
    # the same steps from a double matrix: gene stats (down), select + normalise + log2 (matrix up, result down)
    x = np.asfortranarray(counts, dtype=np.float64)
    t0 = time.perf_counter()
    sums = np.empty(G); nnz = np.empty(G, dtype=np.int32)
    check(L.icnv_gene_stats(x.ctypes.data_as(ct.c_void_p), G, C, sums.ctypes.data_as(ct.c_void_p), nnz.ctypes.data_as(ct.c_void_p)))
    keepi = np.nonzero((sums / C >= 0.1) & (nnz >= 3))[0].astype(np.int32)
    sel = np.empty((keepi.size, C), dtype=np.float64, order="F")
    check(L.icnv_select_genes(x.ctypes.data_as(ct.c_void_p), G, C, keepi.ctypes.data_as(ct.POINTER(ct.c_int32)), keepi.size, sel.ctypes.data_as(ct.c_void_p)))
    o2 = np.empty_like(sel)
    used = ct.c_double()
    check(L.icnv_normalize_log2(sel.ctypes.data_as(ct.c_void_p), o2.ctypes.data_as(ct.c_void_p), keepi.size, C, float("nan"), 1, 1, ct.byref(used)))
    dt = time.perf_counter() - t0
    res["from_double_matrix_three_calls"] = {"ms": dt * 1e3, "h2d_bytes": int(G * C * 8 + keepi.size * C * 8 + G * C * 8), "cells_per_s": C / dt}
    res["three_calls_equal_one_call"] = bool(keepi.size == ga and np.array_equal(o2.ravel(order="F"), a))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
