#!/usr/bin/env python3
"""Per-kernel times of BASELINE config 4's slice (i3 HMM at subcluster level, 50 000 cells, 100 subclusters of 500)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infercnv_amd import device, synth
torch.cuda.set_device(0); device.init(0)
G, C = 10000, 50000
x, cs = synth.make_matrix_torch(G, C, "cuda", C_total=200000)
mu, sigma = device.cells_mean_sd(x, np.arange(5000, dtype=np.int32))
dm = 1.6448536269514722 * sigma
m3 = np.array([mu - dm, mu, mu + dm])
Pi = np.full((3, 3), 1e-6); np.fill_diagonal(Pi, 1 - 5e-6); dl = np.array([1e-6, 1 - 5e-6, 1e-6])
groups = [np.arange(s, s + 500, dtype=np.int32) for s in range(0, C, 500)]
st = torch.empty((C, G), dtype=torch.uint8, device="cuda")
f = lambda: device.viterbi_groups(x, cs, groups, m3, [sigma] * len(groups), np.log(Pi), np.log(dl), states=st)
f(); torch.cuda.synchronize()
device.timing_reset(); device.timing_enable(True)
t0 = time.perf_counter()
for _ in range(3): f()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 3
print("config 4 slice: %.3f ms per call" % (t * 1e3))
for k in ("group_means", "viterbi", "viterbi_redo", "broadcast_states"):
    ms, n = device.timing_get(k)
    if n: print("  %-18s %.3f ms x %.1f per call" % (k, ms / n, n / 3))
