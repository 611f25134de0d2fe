#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (what an R .Call shim hands over): icnv_smooth_chain +
icnv_viterbi_cells on host matrices, uploads and downloads included (developer tool; run on the GPU box)."""
import ctypes as ct, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from infercnv_amd import _lib, synth
from infercnv_amd._lib import Cfg, check, f64, i32
L = _lib.load()
check(L.icnv_init(0))
G, C = 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 20000
x, cs = synth.make_matrix_np(G, C)
x = np.asfortranarray(x)
refs, _ = synth.groups(C)
cfg = Cfg(G, C, cs, refs)
out = np.empty_like(x); pre = np.empty_like(x)
means, sd, logPi, logDelta = synth.hmm_params_i6()
m, mp = f64(means); lp = np.asfortranarray(logPi); ld, ldp = f64(logDelta); csa, csp = i32(cs)
states = np.empty((G, C), dtype=np.uint8, order="F")
def run():
    check(L.icnv_smooth_chain(x.ctypes.data_as(ct.c_void_p), out.ctypes.data_as(ct.c_void_p), pre.ctypes.data_as(ct.c_void_p), cfg.ptr()))
    check(L.icnv_viterbi_cells(pre.ctypes.data_as(ct.c_void_p), states.ctypes.data_as(ct.c_void_p), G, C, csp, csa.size - 1, 6, mp,
                               float(sd), lp.ctypes.data_as(ct.POINTER(ct.c_double)), ldp))
run()
t0 = time.perf_counter(); run(); t = time.perf_counter() - t0
gb = (3 * 8 + 8 + 1) * G * C / 1e9
print(f"host-buffer path, {G} x {C}: {t*1e3:.1f} ms = {C/t/1e6:.3f} M cells/s, {gb:.1f} GB over PCIe ({gb/t:.1f} GB/s incl. compute)")
