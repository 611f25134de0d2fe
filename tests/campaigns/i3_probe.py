import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "oracle"))
import numpy as np, torch
from infercnv_amd import device, synth
import oracle_np as onp
torch.cuda.set_device(0); device.init(0)
G, C = 10000, 50000
x, cs = synth.make_matrix_torch(G, C, "cuda")
refs, _ = synth.groups(C)
_, pre = device.smooth_chain(x, cs, refs, want_pre_denoise=True)
ref_idx = np.concatenate(refs)
mu, sigma = device.cells_mean_sd(pre, ref_idx)
d = abs(-1.6448536269514722 * sigma)
m3 = np.array([mu - d, mu, mu + d])
Pi, dl = onp.get_HMM_i3(1e-6)
st = torch.empty((C, G), dtype=torch.uint8, device="cuda")
m6, sd6, lp6, ld6 = synth.hmm_params_i6()
for name, args in (("i3", (m3, sigma, np.log(Pi), np.log(dl))), ("i6", (m6, sd6, lp6, ld6))):
    device.viterbi_cells(pre, cs, *args, states=st); torch.cuda.synchronize()
    device.timing_reset(); device.timing_enable(True)
    for _ in range(5): device.viterbi_cells(pre, cs, *args, states=st)
    torch.cuda.synchronize(); device.timing_enable(False)
    print(name, {k: device.timing_get(k) for k in ("viterbi", "viterbi_redo", "viterbi_exact_fallback")}, device.viterbi_last_stats(), "mu %.4f sigma %.4f" % (mu, sigma))
