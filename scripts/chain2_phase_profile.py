#!/usr/bin/env python3
"""Per-phase s_memtime ticks of the chain2 kernel (developer tool).  Needs `make -C infercnv_amd/csrc prof`
(-DICNV_CHAIN_PROFILE; -DICNV_PROF_THREAD=n picks the thread of workgroup 0 that accumulates, default 0).
usage: chain2_phase_profile.py [lib] [cells]"""
import ctypes as ct, os, sys
here = os.path.dirname(os.path.abspath(__file__))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "infercnv_amd", "libicnv_hip_prof.so")
os.environ["ICNV_LIB"] = os.path.abspath(lib)
sys.path.insert(0, os.path.join(here, ".."))
import torch
from infercnv_amd import _lib, device, synth
G, C = 10000, int(sys.argv[2]) if len(sys.argv) > 2 else 50000
torch.cuda.set_device(0); device.init(0)
L = _lib.load()
fn = getattr(ct.CDLL(os.environ["ICNV_LIB"]), "icnv_debug_chain2_profile")
x, cs = synth.make_matrix_torch(G, C, "cuda")
refs, _ = synth.groups(C)
plan = device.ChainPlan(G, C, cs, refs)
out = torch.empty_like(x)
def run():
    for r in range(plan.num_rounds):
        plan.round_partial(r, x); plan.round_finish(r)
    plan.apply(x, out=out, want_pre_denoise=True); torch.cuda.synchronize()
run()
buf = (ct.c_ulonglong * 32)()
fn(buf, 1)
run()
fn(buf, 0)
names = ["sub-block 0: first wait (drains the previous stores) + steps 8,9 -> window", "sub-block 0: smoothing + read-back",
         "sub-block 1: wait + steps 8,9 -> window", "sub-block 1: smoothing", "median: histogram pass + barrier",
         "median: scan + barrier", "median: collect + barrier", "median: rank + rest", "last phase: bounds, steps 12,14,22, stores"]
tot = sum(buf[i] for i in range(len(names)))
n_ref = sum(len(r) for r in refs)
# workgroup 0 sees the staging launch (reference cells) and the apply launch (all others): cells of both
print(os.path.basename(lib), "ticks are s_memtime (100 MHz)")
for i, n in enumerate(names):
    print(f"{n:80s} {buf[i]:10d} ticks  {100.0 * buf[i] / max(tot, 1):5.1f}%")
print("total ticks", tot)
