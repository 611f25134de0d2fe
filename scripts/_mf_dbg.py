import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infercnv_amd import device, synth
torch.cuda.set_device(0); device.init(0)
for C5 in (20000, 50000):
    G = 10000
    x0, cs = synth.make_matrix_torch(G, C5, "cuda", C_total=50000)
    refs, _ = synth.groups(C5)
    xd, xp = device.smooth_chain(x0, cs, [r for r in refs], want_pre_denoise=True)
    eq = (xp == 1.0)
    print(C5, "fraction == 1.0:", eq.float().mean().item(), "by cell decile:", [round(eq[i * C5 // 10:(i + 1) * C5 // 10].float().mean().item(), 4) for i in range(10)], "refs", [len(r) for r in refs], [int(r[0]) for r in refs])
    del x0, xd, xp, eq
