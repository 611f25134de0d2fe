#!/usr/bin/env python3
"""Static instruction mix of the chain2 kernel per phase (developer tool; needs hipcc, no GPU): inserts `; MARK <phase>`
comments at the phase boundaries of chain2.hip, compiles to assembly and counts VALU / LDS / VMEM / SALU / scratch
instructions between the markers.  usage: chain2_asm_stats.py [mangled-template-args, default Li0ELi127ELi50E]"""
import collections, os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "infercnv_amd", "csrc")
variant = sys.argv[1] if len(sys.argv) > 1 else "Li0ELi127ELi50E"
src = open(os.path.join(root, "chain2.hip")).read()
marks = [("            // padding inside the window's needed range", "A_put"), ("            // ---- chunk layout: this lane owns", "S_chunk"),
         ("            double r[L];", "S_slide"), ("            // back to the S layout: the core slots", "S_readback"),
         ("        // ---------------- step 11: exact median", "MED_hist"), ("                        if (t < 64) {   // one wavefront scans", "MED_scan"),
         ("                        const int sbin = sel[0]", "MED_collect"), ("                            const int want = target - base - sbefore;", "MED_rank"),
         ("                        // refine inside the selected bin", "MED_refine(cold)"),
         ("                center = (G & 1) ? mid_lo", "MED_end"), ("        // ---------------- steps 11 (subtract), 12, 14, 22", "E"), ("        // ---------------- steps 8, 9, 10: wave-private", "A_loads")]
for m, name in marks:
    if m not in src:
        print("marker anchor missing:", name)
        continue
    src = src.replace(m, 'asm volatile("; MARK %s");\n' % name + m)
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "c2m.hip"), "w").write(src)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + root, "-I" + os.path.join(root, "..", "..", "include"),
                "-S", "--cuda-device-only", "-o", os.path.join(tmp, "o.s"), os.path.join(tmp, "c2m.hip")], check=True, stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, "o.s")).read()
i = s.index("chain2_kernelI" + variant + "EEvNS0_10Chain2ArgsE:")
j = s.index(".end_amdhsa_kernel", i)
cur, st = "PRE", collections.OrderedDict()
for l in s[i:j].split("\n"):
    l = l.strip()
    m = re.match(r"; MARK (\S+)", l)
    if m:
        cur = m.group(1) + "#" + str(sum(1 for k in st if k.startswith(m.group(1) + "#")))
        continue
    if not l or l.startswith((".", ";")):
        continue
    op = l.split()[0]
    d = st.setdefault(cur, collections.Counter())
    if op.startswith("v_"):
        d["valu"] += 1
    elif op.startswith("ds_"):
        d["lds"] += 1
    elif op.startswith("scratch_"):
        d["scratch"] += 1
    elif op.startswith(("global_", "buffer_")):
        d["vmem"] += 1
    elif op == "s_barrier":
        d["barrier"] += 1
    elif op.startswith("s_"):
        d["salu"] += 1
    if op in ("v_readlane_b32", "v_writelane_b32"):
        d["lane"] += 1
for k, d in st.items():
    print(f"{k:20s}", " ".join(f"{a}={b}" for a, b in sorted(d.items())))
if os.environ.get("KEEP_ASM"):
    print(os.path.join(tmp, "o.s"))
