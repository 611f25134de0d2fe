#!/usr/bin/env python3
"""Run the i6 Viterbi kernel alone on the bench workload's HMM input (for rocprofv3 / PMC passes).
usage: run_viterbi.py [cells] [reps] [uniform]   -- 'uniform' makes all cells identical (no lane divergence)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infercnv_amd import device, synth

G = 10000
C = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0); device.init(0)
x, cs = synth.make_matrix_torch(G, C, "cuda")
refs, _ = synth.groups(C)
_, pre = device.smooth_chain(x, cs, refs, want_pre_denoise=True)
if len(sys.argv) > 3 and sys.argv[3] == "uniform":
    pre = pre[C // 2:C // 2 + 1].expand(C, G).contiguous()
means, sd, logPi, logDelta = synth.hmm_params_i6()
states = torch.empty((C, G), dtype=torch.uint8, device="cuda")
device.viterbi_cells(pre, cs, means, sd, logPi, logDelta, states=states)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    device.viterbi_cells(pre, cs, means, sd, logPi, logDelta, states=states)
e1.record(); torch.cuda.synchronize()
print(f"viterbi {G}x{C}: {e0.elapsed_time(e1) / reps:.3f} ms")
