#!/usr/bin/env python3
"""Per-phase cycle counts of the fused chain kernel (developer tool; `make -C infercnv_amd/csrc prof` first).
Runs the bench workload's chain_apply with the profiling build and prints s_memtime cycles per phase per cell."""
import ctypes as ct, os, sys
here = os.path.dirname(os.path.abspath(__file__))
os.environ["ICNV_LIB"] = os.path.join(here, "..", "infercnv_amd", "libicnv_hip_prof.so")
sys.path.insert(0, os.path.join(here, ".."))
import torch
from infercnv_amd import _lib, device, synth
G, C = 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 50000
torch.cuda.set_device(0); device.init(0)
L = _lib.load()
x, cs = synth.make_matrix_torch(G, C, "cuda")
refs, _ = synth.groups(C)
plan = device.ChainPlan(G, C, cs, refs)
for r in range(plan.num_rounds):
    plan.round_partial(r, x); plan.round_finish(r)
out = torch.empty_like(x); pre = torch.empty_like(x)
plan.apply(x, out=out); torch.cuda.synchronize()
buf = (ct.c_ulonglong * 32)()
L.icnv_debug_chain_profile(buf, 1)
plan.apply(x, out=out); torch.cuda.synchronize()
L.icnv_debug_chain_profile(buf, 0)
names = ["[A] steps 8,9 -> LDS", "[C] stores", "barrier before smoothing", "smoothing (init, slide, write-back)",
         "median: rest (after rank)", "centre + prefetch + steps 12,14", "median: get + min/max", "median: histogram", "median: scan",
         "median: collect", "median: rank"]
ncell = (C + 255) // 256
tot = sum(buf[i] for i in range(len(names)))
for i, n in enumerate(names):
    print(f"{n:28s} {buf[i] / ncell:9.0f} ticks/cell  {100.0 * buf[i] / tot:5.1f}%")
print(f"total {tot / ncell:.0f} ticks/cell (s_memtime ticks: 100 MHz constant clock on gfx9 -> {tot / ncell * 10:.0f} ns)")
