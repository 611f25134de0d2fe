"""How long does ONE gene step of the fast Viterbi take when nothing else competes?  A single chromosome of 10 000 genes and
64 x n columns: n wavefronts, all on one CU (one workgroup) -- the per-wavefront dependent latency of a gene step (n = 1) and
the issue-bound round time of a full workgroup (n = 12), without any memory-system contention (0.06-0.6 MB of observations).
    python tests/campaigns/viterbi_latency_probe.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "oracle")]
import numpy as np, torch
from infercnv_amd import device, synth
import oracle_np as onp
torch.cuda.set_device(0); device.init(0)
G = 10000
cs = np.array([0, G], dtype=np.int32)
m6, sd6, lp6, ld6 = synth.hmm_params_i6()
Pi, dl = onp.get_HMM_i3(1e-6)
m3 = np.array([0.93, 1.0, 1.07])
rng = np.random.default_rng(0)
clock_ghz = 2.1
for name, args in (("i6", (m6, sd6, lp6, ld6)), ("i3", (m3, 0.0412, np.log(Pi), np.log(dl)))):
    for waves in (1, 2, 4, 8, 12, 24):
        C = 64 * waves
        x = torch.from_numpy(1.0 + 0.06 * rng.standard_normal((C, G))).cuda()
        st = torch.empty((C, G), dtype=torch.uint8, device="cuda")
        device.viterbi_cells(x, cs, *args, states=st); torch.cuda.synchronize()
        device.timing_reset(); device.timing_enable(True)
        for _ in range(3): device.viterbi_cells(x, cs, *args, states=st)
        torch.cuda.synchronize(); device.timing_enable(False)
        ms, n = device.timing_get("viterbi")
        per = ms / n * 1e-3 * clock_ghz * 1e9 / G
        print(f"{name}: {waves:2d} wavefront(s) of one chromosome of {G} genes: {ms / n:.3f} ms = {per:.0f} cycles per gene step (at {clock_ghz} GHz), path {device.viterbi_last_stats()['path']}")
