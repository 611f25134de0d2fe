#!/usr/bin/env python3
"""Static instruction mix of the fused chain kernel per phase (developer tool; needs hipcc, no GPU).
Inserts `; MARK <phase>` comments at the phase boundaries of chain_kernel.inc, compiles chain_m15.hip to
assembly and counts VALU / LDS / VMEM / SALU / scratch instructions between the markers for one variant."""
import collections, os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "infercnv_amd", "csrc")
variant = sys.argv[1] if len(sys.argv) > 1 else "Li768ELi15ELi2ELi0ELi127E"
src = open(os.path.join(root, "chain_kernel.inc")).read()
marks = [("        // ---------------- [A] steps 8, 9", "A"), ("        // ---------------- [B] prefetch", "B"),
         ("        // ---------------- [C] step 22", "C"), ("        // ---------------- [D] steps 10, 11", "D_INIT"),
         ("                double Lb = Ls + cur;", "D_SLIDE"),
         ("                __syncthreads();  // every halo read of buf is done", "D_WRITE"),
         ("            // back to the S layout;", "D_GET"), ("                // ---- exact median over", "MED_MINMAX"),
         ("                    const int target = ((int)G - 1) >> 1;", "MED_HIST"),
         ("                        if (t < 64) {   // one wavefront scans", "MED_SCAN"),
         ("                        const int sbin = sel[0]", "MED_CAND"),
         ("                            const int want = target - base - sbefore;", "MED_RANK"),
         ("                        // refine inside the selected bin", "MED_REFINE"),
         ("        // steps 12, 14 (the bound vectors", "D_ST12_14")]
for m, name in marks:
    if m not in src:
        print("marker anchor missing:", name); continue
    src = src.replace(m, 'asm volatile("; MARK %s");\n' % name + m, 1)
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "chain_mark.inc"), "w").write(src)
open(os.path.join(tmp, "m15.hip"), "w").write('#include "%s/chain_mark.inc"\nnamespace icnv { int launch_chain_m15(const ChainArgs &a, int mode, hipStream_t s) { return launch_chain_v<768, 15>(a, mode, s); } }\n' % tmp)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + root, "-I" + os.path.join(root, "..", "..", "include"),
                "-S", "--cuda-device-only", "-o", os.path.join(tmp, "o.s"), os.path.join(tmp, "m15.hip")], check=True, stderr=subprocess.DEVNULL)
lines = open(os.path.join(tmp, "o.s")).read().split("\n")
on, cur, stats = False, "PRE", collections.OrderedDict()
for l in lines:
    if re.match(r"^_Z\w+:", l):
        on = variant in l.split(":")[0]
        cur = "PRE"
        continue
    if not on:
        continue
    m = re.search(r"; MARK (\w+)", l)
    if m:
        cur = m.group(1); continue
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
        continue
    op = t[0]
    c = ("scratch" if op.startswith("scratch_") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
         "barrier" if op.startswith("s_barrier") else "wait" if op.startswith("s_waitcnt") else "salu" if op.startswith("s_") else
         "vmem" if op.startswith(("global_", "buffer_", "flat_")) else None)
    if c:
        stats.setdefault(cur, collections.Counter())[c] += 1
    if op == "s_endpgm":
        on = False
for k, v in stats.items():
    print(f"{k:11s}", " ".join(f"{a}={b}" for a, b in sorted(v.items())))
for l in lines:
    if ".name:" in l and variant in l: print(l.strip())
