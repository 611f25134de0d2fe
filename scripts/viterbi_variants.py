#!/usr/bin/env python3
"""Developer tool: variant builds of the certified fast Viterbi kernel (no GPU needed to build): the product source with a
few textual patches, compiled as a replacement translation unit into exp_libs/lib_vf_<name>.so by
scripts/build_variant.sh; scripts/bench_libs.sh times them side by side (LIBS="vf_base vf_nt768 ...") and checks the
output checksums (every variant must return the product kernel's states)."""
import os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = open(os.path.join(root, "infercnv_amd", "csrc", "viterbi_fast.hip")).read()
SPLIT = [("                sc[k] = __builtin_fma(p, tn, c01.x);\n", "                sc[k] = __builtin_fma(p, tn, c01.x);\n                if (k == 2) __builtin_amdgcn_sched_barrier(0);   // two batches of coefficient gathers in flight\n")]
GENEBAR = [("            bpc[i * 64] = (uint16_t)word;   // (no scheduling barrier between the genes of a chunk: neighbours overlap a little)",
            "            bpc[i * 64] = (uint16_t)word;\n            __builtin_amdgcn_sched_barrier(0);")]
NT768 = [("constexpr int FAST_NT = 512;", "constexpr int FAST_NT = 768;")]
NT640 = [("constexpr int FAST_NT = 512;", "constexpr int FAST_NT = 640;")]
V = {"base": [], "split": SPLIT, "genebar": GENEBAR, "split_genebar": SPLIT + GENEBAR,
     "nt768": NT768, "nt768_split": NT768 + SPLIT, "nt768_genebar": NT768 + GENEBAR, "nt768_split_genebar": NT768 + SPLIT + GENEBAR,
     "nt640_split_genebar": NT640 + SPLIT + GENEBAR}
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
