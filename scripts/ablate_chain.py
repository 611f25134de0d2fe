#!/usr/bin/env python3
"""Time the fused chain kernel per stage mask (ablation; run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infercnv_amd import device, synth

G, C = 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 50000
torch.cuda.set_device(0); device.init(0)
x, cs = synth.make_matrix_torch(G, C, "cuda")
refs, _ = synth.groups(C)
out = torch.empty_like(x)
masks = [("copy(0x00)", 0x00), ("st8", 0x01), ("st9", 0x02), ("st10 smooth", 0x04), ("st11 median", 0x08), ("st12", 0x10),
         ("st14 exp2", 0x20), ("st8-10", 0x07), ("st8-11", 0x0F), ("st8-14", 0x3F), ("all", 0x7F)]
if len(sys.argv) > 2 and sys.argv[2] == "copyonly":
    masks = masks[:1]
if len(sys.argv) > 2 and sys.argv[2] == "allonly":
    masks = masks[-1:]
for name, m in masks:
    plan = device.ChainPlan(G, C, cs, refs, stage_mask=m)
    for r in range(plan.num_rounds):
        plan.round_partial(r, x); plan.round_finish(r)
    for _ in range(2): plan.apply(x, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 5
    for _ in range(n): plan.apply(x, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:14s} mask=0x{m:02x}  {ms:7.3f} ms  {2*8*G*C/ms/1e6:8.1f} GB/s")
    plan.close()
# plain torch copy for reference
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out.copy_(x); torch.cuda.synchronize(); e0.record()
for _ in range(5): out.copy_(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"torch copy_     {ms:7.3f} ms  {2*8*G*C/ms/1e6:8.1f} GB/s")
