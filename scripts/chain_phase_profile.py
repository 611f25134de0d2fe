#!/usr/bin/env python3
"""Per-phase cycle counts of the fused chain kernel (developer tool).  Needs a build with -DICNV_CHAIN_PROFILE
(`make -C infercnv_amd/csrc prof`, or scripts/build_variant.sh with that flag): one thread of workgroup 0
(-DICNV_PROF_THREAD=n, default 0) accumulates s_memtime deltas per phase.
usage: chain_phase_profile.py [lib] [cells] [symbol]"""
import ctypes as ct, os, sys
here = os.path.dirname(os.path.abspath(__file__))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "infercnv_amd", "libicnv_hip_prof.so")
os.environ["ICNV_LIB"] = os.path.abspath(lib)
sys.path.insert(0, os.path.join(here, ".."))
import torch
from infercnv_amd import _lib, device, synth
G, C = 10000, int(sys.argv[2]) if len(sys.argv) > 2 else 50000
sym = sys.argv[3] if len(sys.argv) > 3 else "icnv_debug_chain_profile"
torch.cuda.set_device(0); device.init(0)
L = _lib.load()
fn = getattr(ct.CDLL(os.environ["ICNV_LIB"]), sym)
x, cs = synth.make_matrix_torch(G, C, "cuda")
refs, _ = synth.groups(C)
plan = device.ChainPlan(G, C, cs, refs)
for r in range(plan.num_rounds):
    plan.round_partial(r, x); plan.round_finish(r)
out = torch.empty_like(x); pre = torch.empty_like(x)
plan.apply(x, out=out); torch.cuda.synchronize()
buf = (ct.c_ulonglong * 32)()
fn(buf, 1)
plan.apply(x, out=out); torch.cuda.synchronize()
fn(buf, 0)
names = ["[A] steps 8,9 -> LDS", "[C] stores", "barrier before smoothing", "smoothing (init, slide, write-back)",
         "median: rest (after rank)", "centre + prefetch + steps 12,14", "median: get + range", "median: histogram", "median: scan",
         "median: collect", "median: rank"]
n_ref = sum(len(r) for r in refs)
ncell = (C - n_ref + 255) // 256
tot = sum(buf[i] for i in range(len(names)))
print(os.path.basename(lib))
for i, n in enumerate(names):
    print(f"{n:38s} {buf[i] / ncell:9.1f} ticks/cell  {100.0 * buf[i] / tot:5.1f}%")
print(f"total {tot / ncell:.0f} ticks/cell (s_memtime ticks: 100 MHz constant clock on gfx9 -> {tot / ncell * 10:.0f} ns)")
