#!/usr/bin/env python3
"""Developer tool: variant builds of the certified fast Viterbi kernel (no GPU needed to build): the product source compiled
with other launch geometries (-DVF_NT = threads per workgroup, -DVF_CH = genes per observation chunk) into
exp_libs/lib_vf_<name>.so by scripts/build_variant.sh; scripts/bench_libs.sh times them side by side
(LIBS="vf_base vf_nt1024_ch8 ...") and checks the output checksums (every variant must return the product kernel's states).
Round 4's A/Bs (docs/KERNEL_LOG.md) were made this way."""
import os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
V = {"base": "", "nt1024_ch8": "-DVF_NT=1024 -DVF_CH=8", "nt768_ch8": "-DVF_NT=768 -DVF_CH=8", "nt512_ch16": "-DVF_NT=512 -DVF_CH=16",
     "nt1024_ch16": "-DVF_NT=1024 -DVF_CH=16", "nt640_ch16": "-DVF_NT=640 -DVF_CH=16"}
want = sys.argv[1:] or list(V)
src = os.path.join(root, "infercnv_amd", "csrc", "viterbi_fast.hip")
for name in want:
    subprocess.run(["bash", os.path.join(root, "scripts", "build_variant.sh"), "vf_" + name, V[name]], check=True,
                   env=dict(os.environ, REBUILD="viterbi_fast"))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "kernel_resources.py"), src, "-ffp-contract=off"] + V[name].split(),
                       capture_output=True, text=True)
    print(name, [l for l in r.stdout.splitlines() if "kernel<6>" in l])
