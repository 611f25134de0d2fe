import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infercnv_amd import device, synth
torch.cuda.set_device(0); device.init(0)
G, C5 = 10000, int(os.environ.get("MF_CELLS", "10000"))
x0, cs = synth.make_matrix_torch(G, C5, "cuda", C_total=50000)
refs, _ = synth.groups(C5)
xd, xp = device.smooth_chain(x0, cs, [r for r in refs], want_pre_denoise=True)
del x0
tiles = [np.arange(s, min(s + 490, C5), dtype=np.int32) for s in range(0, C5, 490)]
o5 = torch.empty_like(xd)
for name, x in (("denoised", xd), ("no_ties", xp)):
    device.median_filter(x, cs, tiles, 7, out=o5); torch.cuda.synchronize()
    device.timing_reset(); device.timing_enable(True)
    for _ in range(3): device.median_filter(x, cs, tiles, 7, out=o5)
    torch.cuda.synchronize()
    ms, n = device.timing_get("median_filter")
    print("%s %d cells: kernel %.3f ms" % (name, C5, ms / max(n, 1)), flush=True)
