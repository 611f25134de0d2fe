#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (what an R .Call shim hands over; developer tool, run on the GPU box):
  (a) fused: icnv_smooth_chain (out + HMM input down) + icnv_viterbi_cells on host matrices,
  (b) run()'s six stand-alone steps 8, 9, 10, 11, 12, 14 back to back, each taking the matrix the previous one returned,
each without and with icnv_residency(1) (a recognised matrix is not uploaded again), and with the cells split over
`devices` logical devices (ICNV_FAKE_DEVICES=n maps them onto one GPU: what that exercises is the code path, not the
bandwidth of n PCIe links).   usage: bench_host_path.py [cells] [devices]"""
import ctypes as ct, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from infercnv_amd import _lib, synth
from infercnv_amd._lib import Cfg, check, f64, i32


def measure(G=10000, C=20000, devices=1, reps=2):
    L = _lib.load()
    check(L.icnv_init(0))
    check(L.icnv_set_devices(devices))
    x, cs = synth.make_matrix_np(G, C)
    x = np.asfortranarray(x)
    refs, _ = synth.groups(C)
    out = np.empty_like(x); pre = np.empty_like(x)
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    m, mp = f64(means); lp = np.asfortranarray(logPi); ld, ldp = f64(logDelta); csa, csp = i32(cs)
    states = np.empty((G, C), dtype=np.uint8, order="F")
    vp = lambda a: a.ctypes.data_as(ct.c_void_p)

    def fused():
        cfg = Cfg(G, C, cs, refs)
        check(L.icnv_smooth_chain(vp(x), vp(out), vp(pre), cfg.ptr()))
        check(L.icnv_viterbi_cells(vp(pre), vp(states), G, C, csp, csa.size - 1, 6, mp, float(sd),
                                   lp.ctypes.data_as(ct.POINTER(ct.c_double)), ldp))

    bufs = [np.empty_like(x) for _ in range(2)]

    def six_steps():
        cur = x
        for i, mask in enumerate((0x01, 0x02, 0x04, 0x08, 0x10, 0x20)):
            dst = bufs[i & 1]
            cfg = Cfg(G, C, cs, refs, stage_mask=mask)
            check(L.icnv_smooth_chain(vp(cur), vp(dst), None, cfg.ptr()))
            cur = dst
        return cur

    res = {"genes": G, "cells": C, "devices": devices}
    for name, fn, gb in (("fused_chain_plus_hmm", fused, (3 * 8 + 8 + 1) * G * C / 1e9), ("six_standalone_steps", six_steps, 6 * 16 * G * C / 1e9)):
        for resident in (0, 1):
            check(L.icnv_residency(resident))
            for _ in range(4 if resident else 2):   # steady state: staging buffers / the pool's blocks exist (bench.py warms up the same way)
                fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            t = (time.perf_counter() - t0) / reps
            st = (ct.c_int64 * 4)()
            check(L.icnv_residency_stats(st))
            res[f"{name}_residency{resident}"] = {"ms": t * 1e3, "cells_per_s": C / t, "pcie_gb_without_residency": gb,
                                                  "recognised_so_far": int(st[0])}
    check(L.icnv_residency(0))
    check(L.icnv_set_devices(1))
    return res


if __name__ == "__main__":
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    nd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print(json.dumps(measure(C=C, devices=nd)))
