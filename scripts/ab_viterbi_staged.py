#!/usr/bin/env python3
"""One-call A/B of the two certified fast Viterbi kernels on the bench workload's HMM input (run on the GPU box):
mode 2 = register kernel with the full table (every lane walks its column), mode 0 = staged kernel first (observations by
whole cache lines through LDS-DMA, short table).  Interleaved passes, states compared, also against the exact kernel.
  python scripts/ab_viterbi_staged.py [cells] [genes] [passes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infercnv_amd import device, synth

C = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
torch.cuda.set_device(0); device.init(0)
x, cs = synth.make_matrix_torch(G, C, "cuda")
refs, _ = synth.groups(C)
_, pre = device.smooth_chain(x, cs, refs, want_pre_denoise=True)
del x
means, sd, logPi, logDelta = synth.hmm_params_i6()
st = {m: torch.empty((C, G), dtype=torch.uint8, device="cuda") for m in (0, 1, 2)}


def timed(mode, reps=5):
    device.viterbi_set_mode(mode)
    device.viterbi_cells(pre, cs, means, sd, logPi, logDelta, states=st[mode])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): device.viterbi_cells(pre, cs, means, sd, logPi, logDelta, states=st[mode])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, device.viterbi_last_stats()


for p in range(passes):
    t2, s2 = timed(2)
    t0, s0 = timed(0)
    print(f"pass {p}: register kernel {t2:.3f} ms ({s2['kernel']}, {s2['table_intervals']} records, {s2['flagged']} flagged)   "
          f"staged kernel {t0:.3f} ms ({s0['kernel']}, {s0['table_intervals']} records, {s0['flagged']} flagged)")
print("states identical (staged vs register):", bool(torch.equal(st[0], st[2])))
if C <= 60000:
    t1, _ = timed(1, 1)
    print(f"exact kernel {t1:.3f} ms; states identical (staged vs exact):", bool(torch.equal(st[0], st[1])))
device.viterbi_set_mode(0)
