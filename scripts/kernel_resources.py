#!/usr/bin/env python3
"""Registers / scratch / occupancy of every kernel of one HIP source (developer tool; needs hipcc, no GPU).
usage: kernel_resources.py infercnv_amd/csrc/chain_m15.hip [extra hipcc flags]"""
import os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = sys.argv[1]
out = os.path.join(tempfile.mkdtemp(), "o.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "infercnv_amd", "csrc"),
                "-I" + os.path.join(root, "include"), "-S", "--cuda-device-only", "-o", out, src] + sys.argv[2:], check=True,
               stderr=subprocess.DEVNULL)
name = None
rec = {}
for line in open(out):
    m = re.match(r"^(_Z\S+):", line)
    if m:
        name = m.group(1)
        rec = {"name": name}
    m = re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|TotalNumSgprs): (\d+)", line)
    if m and name:
        rec[m.group(1).replace("Total", "")] = int(m.group(2))
        if m.group(1) == "Occupancy":
            short = subprocess.run(["c++filt", rec["name"]], capture_output=True, text=True).stdout.strip()
            short = short.replace("icnv::(anonymous namespace)::", "").replace("void ", "")
            short = re.sub(r"\(icnv::.*|\(.*\)$", "", short)
            print(f"{short:60s} vgpr={rec.get('NumVgprs')} agpr={rec.get('NumAgprs')} sgpr={rec.get('NumSgprs')} "
                  f"scratch={rec.get('ScratchSize')} occupancy={rec.get('Occupancy')}")
