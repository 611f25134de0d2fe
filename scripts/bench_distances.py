#!/usr/bin/env python3
"""Time icnv_cell_distances_dev on one tumor group of the bench workload (developer tool; run on the GPU box).
usage: bench_distances.py [cells_in_group] [genes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infercnv_amd import device, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
torch.cuda.set_device(0); device.init(0)
x, cs = synth.make_matrix_torch(G, n, "cuda")
cells = np.arange(n, dtype=np.int32)
d = device.cell_distances(x, cells); torch.cuda.synchronize()
device.timing_reset(); device.timing_enable(True)
reps = 5
for _ in range(reps): d = device.cell_distances(x, cells)
torch.cuda.synchronize(); device.timing_enable(False)
ms, k = device.timing_get("cell_distances_gram")
ms /= k
nt128 = (n + 127) // 128
DT = 128 if nt128 * (nt128 + 1) // 2 >= 2 * torch.cuda.get_device_properties(0).multi_processor_count else 64   # launch_cell_distances' rule
nt = (n + DT - 1) // DT
flops_done = nt * (nt + 1) / 2 * DT * DT * 2.0 * G       # upper-triangular tiles only
print(f"gram kernel {n} cells x {G} genes ({DT}-cell tiles): {ms:.3f} ms, {flops_done / ms / 1e9:.1f} TFLOP/s fp64 MFMA executed "
      f"({2.0 * n * n * G / ms / 1e9:.1f} TFLOP/s counting the full n^2 G product)")
