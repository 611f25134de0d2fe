#!/usr/bin/env python3
"""Static instruction mix of the fused chain kernel per phase (developer tool; needs hipcc, no GPU).
Inserts `; MARK <phase>` comments at the phase boundaries of chain_kernel.inc, compiles the 768 x 15 variant to
assembly and counts VALU / LDS / VMEM / SALU / v_readlane instructions between the markers."""
import collections, os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "infercnv_amd", "csrc")
variant = sys.argv[1] if len(sys.argv) > 1 else "Li1024ELi11ELi2ELi0ELi127ELi5ELi50ELb0E"   # (the last parameter: Lb1E = the odd-gene-count twin)
launch = sys.argv[2] if len(sys.argv) > 2 else "launch_chain_m<1024, 11, 2, 5, 50>"   # e.g. "launch_chain_v<768, 15>" with Li768ELi15ELi2ELi0ELi127ELi0ELi0ELb0E
src = open(os.path.join(root, "chain_kernel.inc")).read()
marks = [("        // ---------------- [A] steps 8, 9", "A"), ("        // ---------------- [C] step 22", "C"),
         ("        // ---------------- [D] steps 10, 11", "D_smooth_init"),
         ("#pragma unroll\n                for (int q = 0; q < LMAX; ++q) {\n                    const int p = p0 + q;\n                    r[q] = A * inv_at(q);", "D_slide"),
         ("                __syncthreads();  // every halo read of buf (and of the chunk moments) is done", "D_writeback"),
         ("            // back to the S layout; steps 11.. run on registers", "D_get_minmax"),
         ("                        const double scale = fmin((double)NB_HIST / (hi - lo), 0x1p1000);", "MED_hist"),
         ("                        if (t < 64) {   // one wavefront scans", "MED_scan"),
         ("                        const int sbin = sel[0], sbefore = sel[1], scnt = sel[2];", "MED_collect"),
         ("                            const int want = target - base - sbefore;", "MED_rank"),
         ("                        // refine inside the selected bin", "MED_refine(cold)"),
         ("                  center = (G & 1) ? mid_lo : (mid_lo + mid_hi) * 0.5;", "MED_end"),
         ("        // steps 12, 14; the step-12 bound vectors", "F_12_14")]
for m, name in marks:
    if m not in src:
        print("marker anchor missing:", name)
        continue
    src = src.replace(m, 'asm volatile("; MARK %s");\n' % name + m, 1)
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "chain_mark.inc"), "w").write(src)
open(os.path.join(tmp, "m15.hip"), "w").write('#include "%s/chain_mark.inc"\nnamespace icnv { int launch_chain_m15(const ChainArgs &a, int mode, hipStream_t s) { return %s(a, mode, s); } }\n' % (tmp, launch))
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + root, "-I" + os.path.join(root, "..", "..", "include"),
                "-S", "--cuda-device-only", "-o", os.path.join(tmp, "o.s"), os.path.join(tmp, "m15.hip")], check=True, stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, "o.s")).read()
i = s.index("chain_kernelI" + variant + "EEvNS_9ChainArgsE:")
j = s.index(".end_amdhsa_kernel", i)
cur, st = "PRE", collections.OrderedDict()
for l in s[i:j].split("\n"):
    l = l.strip()
    m = re.match(r"; MARK (\S+)", l)
    if m:
        cur = m.group(1)
        continue
    if not l or l.startswith((".", ";")):
        continue
    op = l.split()[0]
    d = st.setdefault(cur, collections.Counter())
    if op.startswith("v_"):
        d["valu"] += 1
    elif op.startswith("ds_"):
        d["lds"] += 1
    elif op.startswith(("global_", "scratch_", "buffer_")):
        d["vmem"] += 1
    elif op == "s_barrier":
        d["barrier"] += 1
    elif op.startswith("s_"):
        d["salu"] += 1
    if op == "v_readlane_b32":
        d["readlane"] += 1
for k, d in st.items():
    print(f"{k:18s}", " ".join(f"{a}={b}" for a, b in sorted(d.items())))
if os.environ.get("KEEP_ASM"):
    open(os.environ["KEEP_ASM"], "w").write(s[i:j])
