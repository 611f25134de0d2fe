#!/usr/bin/env python3
"""Static instruction mix of the certified fast Viterbi kernel's chunk loop (developer tool; needs hipcc, no GPU):
instructions between the markers around the CH unrolled genes, divided by CH."""
import collections, os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "infercnv_amd", "csrc")
src = open(os.path.join(root, "viterbi_fast.hip")).read()
begin = "                const double x_behind = more ? xnext[0] : xc[(i + CH < n) ? i + CH : i + CH - 1];\n"
end = "                if (use_sum && ((a0u + i + CH) & 15) == 0) {\n"
assert begin in src and end in src
src = src.replace(begin, begin + 'asm volatile("; MARK begin");\n', 1).replace(end, 'asm volatile("; MARK end");\n' + end, 1)
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "vf.hip"), "w").write(src)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + root,
                "-I" + os.path.join(root, "..", "..", "include"), "-S", "--cuda-device-only", "-o", os.path.join(tmp, "o.s"),
                os.path.join(tmp, "vf.hip")] + sys.argv[1:], check=True, stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, "o.s")).read()
for K in (6, 3):
    i = s.index("viterbi_fast_kernelILi%dEEEvNS_15FastViterbiArgsE:" % K)
    j = s.index(".end_amdhsa_kernel", i)
    body = s[i:j]
    a = body.index("; MARK begin"); b = body.index("; MARK end")
    c = collections.Counter()
    for l in body[a:b].split("\n"):
        l = l.strip()
        if not l or l.startswith((";", ".", "//")) or l.endswith(":"): continue
        op = l.split()[0]
        if "readlane" in op or "writelane" in op: c["lane_spill"] += 1
        if op.startswith("v_"): c["valu"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
        else: c["other"] += 1
    ch = 16
    for a_ in sys.argv[1:]:
        if a_.startswith('-DVF_CH='): ch = int(a_.split('=')[1])
    print("K=%d per gene:" % K, {k: round(v / ch, 1) for k, v in sorted(c.items())})
