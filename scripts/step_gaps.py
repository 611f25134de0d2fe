#!/usr/bin/env python3
"""Where the time of a bench step goes that is NOT inside a kernel: the timeline of a rocprofv3 --kernel-trace run of bench.py
(rocpd database), cut into steps at every launch of the step's first kernel, per step the kernel time, the idle time between
consecutive kernels and which kernel each gap follows.  Developer tool (GPU box):
   rocprofv3 --kernel-trace -d gpurun_out/trace -o t -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3
   python scripts/step_gaps.py gpurun_out/trace/*/t_results.db"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
if not cols:
    print("tables:", [r[0] for r in c.execute("select name from sqlite_master")]); sys.exit(1)
s_col = "start" if "start" in cols else [x for x in cols if "start" in x][0]
e_col = "end" if "end" in cols else [x for x in cols if "end" in x][0]
rows = c.execute(f'select name, "{s_col}", "{e_col}" from kernels order by "{s_col}"').fetchall()
short = lambda n: n.replace("icnv::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
first = "group_gene_sums_kernel"
idx = [i for i, r in enumerate(rows) if first in r[0]]
# a step = two launches of the raw gene sums (rounds A and B): cut at every second one
starts = idx[::2]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = steps[len(steps) // 2:]           # the timed half (warm-up and set-up kernels come first)
if not steps:
    print("no steps found; kernels:", collections.Counter(short(r[0]) for r in rows).most_common(12)); sys.exit(1)
tot_k = tot_gap = 0.0
gap_after = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, 0])
span = 0.0
for st, nxt in zip(steps[:-1], steps[1:]):
    seq = st + nxt[:1]
    span += (nxt[0][1] - st[0][1]) / 1e3
    for (n0, s0, e0), (n1, s1, e1) in zip(seq[:-1], seq[1:]):
        tot_k += (e0 - s0) / 1e3
        g = max(0.0, (s1 - e0) / 1e3)
        tot_gap += g
        gap_after[short(n0)][0] += g; gap_after[short(n0)][1] += 1
        dur[short(n0)][0] += (e0 - s0) / 1e3; dur[short(n0)][1] += 1
n = len(steps) - 1
print(f"# {n} steps: {span / n:.1f} us per step = {tot_k / n:.1f} us in kernels + {tot_gap / n:.1f} us between kernels; {len(steps[0])} launches per step")
print(f"{'kernel':72s} {'calls/step':>10} {'us/step':>9} {'gap after, us/step':>19} {'gap per launch':>15}")
for k, (d, m) in sorted(dur.items(), key=lambda kv: -kv[1][0]):
    g, gm = gap_after[k]
    print(f"{k:72s} {m / n:10.1f} {d / n:9.1f} {g / n:19.1f} {g / max(gm, 1):15.2f}")
