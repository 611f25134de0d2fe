#!/usr/bin/env python3
"""Single-GPU slices of BASELINE.json configs 3-5 (run on the GPU box):
   config 3: i6 HMM per cell (1M cells / 8 GPUs = 125 000 cells per GPU)
   config 4: i3 HMM at subcluster level (200 000 cells / 4 GPUs = 50 000 cells per GPU, 500-cell subclusters)
   config 5: apply_median_filtering, window 7, 500-cell tiles (slice of 125 000 cells per GPU -> timed on fewer)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infercnv_amd import device, synth

torch.cuda.set_device(0); device.init(0)
G = 10000
res = {}

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

# config 3: per-cell i6 HMM on 125k cells (HMM input = chain output, produced once)
C = 125000
x, cs = synth.make_matrix_torch(G, C, "cuda", C_total=1000000)
refs, _ = synth.groups(C)
out, pre = device.smooth_chain(x, cs, refs, want_pre_denoise=True)
del out
means, sd, logPi, logDelta = synth.hmm_params_i6()
states = torch.empty((C, G), dtype=torch.uint8, device="cuda")
t = timed(lambda: device.viterbi_cells(pre, cs, means, sd, logPi, logDelta, states=states), 2)
res["config3_i6_cells_per_gpu"] = {"cells": C, "ms": t * 1e3, "cells_per_s": C / t, **device.viterbi_last_stats()}
del states

# config 4: i3 at subcluster level, 50k cells per GPU
C4 = 50000
pre4 = pre[:C4].contiguous()
mu, sigma = device.cells_mean_sd(pre4, np.arange(5000, dtype=np.int32))
dm = 1.6448536269514722 * sigma
m3 = np.array([mu - dm, mu, mu + dm])
Pi = np.full((3, 3), 1e-6); np.fill_diagonal(Pi, 1 - 5e-6); dl = np.array([1e-6, 1 - 5e-6, 1e-6])
groups = [np.arange(s, s + 500, dtype=np.int32) for s in range(0, C4, 500)]
st4 = torch.empty((C4, G), dtype=torch.uint8, device="cuda")
t = timed(lambda: device.viterbi_groups(pre4, cs, groups, m3, [sigma] * len(groups), np.log(Pi), np.log(dl), states=st4))
res["config4_i3_subclusters_per_gpu"] = {"cells": C4, "subclusters": len(groups), "ms": t * 1e3, "cells_per_s": C4 / t}
# config 4b: the same i3 model per cell (certified fast path with a 3-state table)
t = timed(lambda: device.viterbi_cells(pre4, cs, m3, sigma, np.log(Pi), np.log(dl), states=st4), 2)
res["config4b_i3_per_cell"] = {"cells": C4, "ms": t * 1e3, "cells_per_s": C4 / t, **device.viterbi_last_stats()}
# distances inside one 2 500-cell tumor group (fp64 MFMA Gram matrix)
cells = np.arange(5000, 7500, dtype=np.int32)
t = timed(lambda: device.cell_distances(pre4, cells), 3)
res["group_distances_2500_cells"] = {"cells": 2500, "ms": t * 1e3, "TFLOPs_full_product": 2.0 * 2500 * 2500 * G / t / 1e12}
del st4, pre4

# config 5: median filter on a 5 000-cell slice (10 tiles of 500 cells)
C5 = 5000
x5 = pre[:C5].contiguous()
tiles = [np.arange(s, s + 500, dtype=np.int32) for s in range(0, C5, 500)]
o5 = torch.empty_like(x5)
t = timed(lambda: device.median_filter(x5, cs, tiles, 7, out=o5), 2)
res["config5_median_filter_slice"] = {"cells": C5, "ms": t * 1e3, "cells_per_s": C5 / t,
                                      "GBps_algorithmic": 2 * 8 * G * C5 / t / 1e9}
print(json.dumps(res))
