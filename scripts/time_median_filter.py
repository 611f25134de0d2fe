import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infercnv_amd import device, synth
torch.cuda.set_device(0); device.init(0)
G, C5 = 10000, 5000
x0, cs = synth.make_matrix_torch(G, C5, "cuda", C_total=50000)
refs, _ = synth.groups(C5)
x, _ = device.smooth_chain(x0, cs, [r for r in refs], want_pre_denoise=False)    # the denoised matrix: what apply_median_filtering runs on
del x0
tiles = [np.arange(s, s + 500, dtype=np.int32) for s in range(0, C5, 500)]
o5 = torch.empty_like(x)
device.median_filter(x, cs, tiles, 7, out=o5); torch.cuda.synchronize()
device.timing_reset(); device.timing_enable(True)
t0 = time.perf_counter()
for _ in range(3): device.median_filter(x, cs, tiles, 7, out=o5)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 3
ms, n = device.timing_get("median_filter")
print("median filter 5000 cells: %.3f ms per call, kernel %.3f ms" % (t * 1e3, ms / max(n, 1)))
