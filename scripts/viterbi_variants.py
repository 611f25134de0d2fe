#!/usr/bin/env python3
"""Developer tool: variant builds of the certified fast Viterbi kernel (no GPU needed to build): the product source with a
few textual patches, compiled as a replacement translation unit into exp_libs/lib_vf_<name>.so by
scripts/build_variant.sh; scripts/bench_libs.sh times them side by side (LIBS="vf_base vf_nt768 ...") and checks the
output checksums (every variant must return the product kernel's states)."""
import os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = open(os.path.join(root, "infercnv_amd", "csrc", "viterbi_fast.hip")).read()
def nt(n): return [("constexpr int FAST_NT = 1024;", f"constexpr int FAST_NT = {n};")]
def ch(n): return [("constexpr int FAST_CH = 8; ", f"constexpr int FAST_CH = {n};")]
def tg(n): return [("constexpr int FAST_TG = 8; ", f"constexpr int FAST_TG = {n};")]
V = {"base": [], "nt768": nt(768), "nt512": nt(512), "ch16": ch(16), "nt768_ch16": nt(768) + ch(16), "nt768_ch16_tg16": nt(768) + ch(16) + tg(16),
     "nt768_tg16": nt(768) + tg(16)}
want = sys.argv[1:] or list(V)
for name in want:
    s = src
    for old, new in V[name]:
        assert s.count(old) == 1, (name, old[:60])
        s = s.replace(old, new)
    d = f"/tmp/vfvar/{name}"
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "viterbi_fast.hip"), "w").write(s)
    subprocess.run(["bash", os.path.join(root, "scripts", "build_variant.sh"), "vf_" + name, "-ffp-contract=off", os.path.join(d, "viterbi_fast.hip")],
                   check=True, env=dict(os.environ, REBUILD=" "))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "kernel_resources.py"), os.path.join(d, "viterbi_fast.hip")],
                       capture_output=True, text=True)
    print(name, [l for l in r.stdout.splitlines() if "kernel<6>" in l])
