#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace --stats) as text:
   python scripts/rocprof_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.txt"""
import os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as _bench
c = sqlite3.connect(sys.argv[1])
# what the statistics belong to (the GPU box has no .git: pass ICNV_COMMIT=$(git rev-parse --short HEAD) in the gpurun command)
print(f"# library sources sha16 {_bench.source_stamp()}  commit {os.environ.get('ICNV_COMMIT') or 'unknown'}")
rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]} (durations in microseconds)")
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>12} {'pct':>7}  kernel")
for name, calls, tot, avg, pct in rows:
    short = name if len(name) < 150 else name[:147] + "..."
    print(f"{calls:6d} {tot:12.1f} {avg:12.1f} {pct:7.2f}  {short}")
try:
    q = ("select name, count(*), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
         "max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels where name like '%icnv%' group by name")
    print("\n# icnv kernels: calls, avg/min/max duration (ns), vgpr, sgpr, lds bytes, scratch bytes, grid_x, workgroup_x")
    for r in c.execute(q):
        print(r[1:], r[0][:120])
except Exception as e:
    print("# (no per-dispatch details:", e, ")")
