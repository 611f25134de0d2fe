#!/usr/bin/env python3
"""Stress campaign of the 9 x 9 median filter (run on the GPU box): random chromosome lengths, tile sizes, cell
permutations and tie densities against the CPU oracle, exact equality.
  python tests/campaigns/stress_median_filter.py [first_seed] [n_seeds] [large]
"large": chromosomes of up to 700 genes and tiles of up to 500 cells -- many 56 x 32 classification tiles per (chromosome, tile)
pair, queue segments of several workgroups, dense 32 x 16 patches next to decided ones."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np, torch
import oracle_c as oc
from infercnv_amd import device
torch.cuda.set_device(0); device.init(0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
large = len(sys.argv) > 3 and sys.argv[3] == "large"
t0 = time.time(); outputs = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(1, 700 if large else 90, size=int(rng.integers(1, 7)))
    tsz = rng.integers(1, 500 if large else 70, size=int(rng.integers(1, 6)))
    G, C = int(sizes.sum()), int(tsz.sum()) + int(rng.integers(0, 4))
    cs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    x = rng.normal(size=(G, C))
    if seed % 3 == 0: x = np.round(x, 1)                      # many ties
    if seed % 5 == 0: x[rng.random((G, C)) < 0.5] = 0.0       # half zeros
    if seed % 4 == 1: x[rng.random((G, C)) < rng.uniform(0.3, 0.98)] = 1.012490474117089   # one dominant value (the majority shortcut)
    if seed % 8 == 3: x[:, : C // 2][rng.random((G, C // 2)) < 0.85] = -2.5                  # ... a second one in half of the cells
    if large and seed % 4 == 2:                                                                # the denoised shape: a constant with islands of values
        x[:] = 1.0012
        for _ in range(int(rng.integers(1, 40))):
            g0, c0 = int(rng.integers(0, G)), int(rng.integers(0, C))
            g1, c1 = g0 + int(rng.integers(1, 120)), c0 + int(rng.integers(1, 90))
            x[g0:g1, c0:c1] = rng.normal(size=x[g0:g1, c0:c1].shape) * 0.2 + 1.3
    perm = rng.permutation(C)
    off = np.concatenate([[0], np.cumsum(tsz)])
    tiles = [perm[off[i]:off[i + 1]].astype(np.int32) for i in range(len(tsz))]
    xd = torch.from_numpy(np.ascontiguousarray(x.T)).cuda()
    got = device.median_filter(xd, cs, tiles, 7).cpu().numpy().T
    want = oc.median_filter(x, cs, tiles, 7)
    assert np.array_equal(got, want), "seed %d: mismatch" % seed
    outputs += G * C
print(("large " if large else "") + "seeds %d..%d: %d outputs identical to the oracle (%.0f s)" % (first, first + n - 1, outputs, time.time() - t0))
