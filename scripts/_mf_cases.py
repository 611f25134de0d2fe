
import sys, numpy as np, torch
sys.path[:0] = [__import__('os').environ['GRAFT_REPO_ROOT'], __import__('os').environ['GRAFT_REPO_ROOT'] + '/oracle']
import oracle_c as oc
from infercnv_amd import device
torch.cuda.set_device(0); device.init(0)
sizes = [150, 9, 61, 330, 8, 40, 75, 64, 65]; G = sum(sizes)
cs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
tsz = [300, 41, 9, 8, 130, 129, 17]; C = sum(tsz)
mu = 1.012490474117089
def data(case, rng):
    x = rng.normal(1.0, 0.2, size=(G, C))
    if case == "continuous":
        pass
    elif case == "dominant_islands":                 # the denoised shape: one value with islands of continuous values
        x[:] = mu
        for _ in range(25):
            g0, c0 = int(rng.integers(0, G)), int(rng.integers(0, C))
            g1, c1 = min(G, g0 + int(rng.integers(1, 200))), min(C, c0 + int(rng.integers(1, 200)))
            x[g0:g1, c0:c1] = rng.normal(1.3, 0.2, size=(g1 - g0, c1 - c0))
    elif case == "rounded":                          # discrete data: the probe turns the strip kernel off
        x = np.round(x, 1)
    elif case == "three_values":                     # repeated values below a majority: codes of their own
        r = rng.random((G, C))
        x[r < 0.10] = 1.0; x[(r >= 0.10) & (r < 0.18)] = mu; x[(r >= 0.18) & (r < 0.24)] = 0.75
    elif case == "six_values":                       # more repeated values than codes: strip kernel off, or its queue takes the rest
        r = rng.random((G, C))
        for k, v in enumerate((1.0, mu, 0.75, 1.25, 0.5, 1.5)):
            x[(r >= 0.05 * k) & (r < 0.05 * (k + 1))] = v
    elif case == "rare_ties":                        # a repeated value the probe is unlikely to see (0.3 %): its windows are queued
        x[rng.random((G, C)) < 0.003] = 1.0
        x[100:140, 50:90] = 1.0                      # ... and a block where it is every window's median
    elif case == "outliers":                         # end buckets: infinities and huge values, next to ordinary ones
        r = rng.random((G, C))
        x[r < 0.002] = np.inf; x[(r >= 0.002) & (r < 0.004)] = -np.inf; x[(r >= 0.004) & (r < 0.006)] = 1e300; x[(r >= 0.006) & (r < 0.008)] = -1e300
        x[200:260, 100:180] = np.inf                 # windows whose median is +Inf
    elif case == "tight":                            # a range set by two far values, everything else inside a few buckets: collisions everywhere
        x = 1.0 + 1e-9 * rng.normal(size=(G, C))
        x[0, 0], x[1, 1] = -5e3, 5e3
        x[rng.random((G, C)) < 0.001] = 4e3
    return x
rng = np.random.default_rng(606)
perm = rng.permutation(C); off = np.concatenate([[0], np.cumsum(tsz)])
tiles = [perm[off[i]:off[i + 1]].astype(np.int32) for i in range(len(tsz))]
tiles[0] = np.sort(tiles[0])
for case in ("continuous", "dominant_islands", "rounded", "three_values", "six_values", "rare_ties", "outliers", "tight"):
    x = data(case, rng)
    print("CASE", case, file=sys.stderr, flush=True)
    got = device.median_filter(torch.from_numpy(np.ascontiguousarray(x.T)).cuda(), cs, tiles, 7).cpu().numpy().T
    want = oc.median_filter(x, cs, tiles, 7)
    assert np.array_equal(got, want), (case, int((got != want).sum()))
print("MF9_STRIP_OK")
