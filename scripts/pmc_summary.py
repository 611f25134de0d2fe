#!/usr/bin/env python3
"""gpurun_out/pmc/{FETCH_SIZE,WRITE_SIZE}/*_counter_collection.csv -> profiles/pmc_traffic.json (+ a text table).

FETCH_SIZE / WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports 1/2 of the
bytes of a wide coalesced streaming read, so it is doubled; the calibration launch (chain kernel with stage_mask 0:
exactly 4.0 GB read + 4.0 GB written) checks the correction in our own access pattern."""
import collections, csv, json, os, re, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
def load(path):
    agg = collections.defaultdict(list)
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if "icnv" in r["Kernel_Name"]:
            m = re.search(r"::(\w+_kernel(?:<[^>]*>)?)\(", r["Kernel_Name"])
            agg[m.group(1) if m else r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
f = {t: load(f"{root}/FETCH_SIZE/{t}_counter_collection.csv") for t in ("copy", "bench")}
w = {t: load(f"{root}/WRITE_SIZE/{t}_counter_collection.csv") for t in ("copy", "bench")}
lines = ["kernel  FETCH_SIZE(KiB, raw)  fetch_bytes(x2 corrected)  WRITE_SIZE(KiB)  write_bytes  total_bytes"]
out = {}
for t in ("copy", "bench"):
    for k in sorted(f[t]):
        fb, wb = f[t][k] * 2 * 1024, w[t].get(k, 0.0) * 1024
        lines.append(f"{t}:{k}  {f[t][k]:.0f}  {fb:.4g}  {w[t].get(k, 0):.0f}  {wb:.4g}  {fb + wb:.4g}")
        if t == "bench":
            if re.match(r"chain_kernel<\d+, \d+, \d+, 0, 127[,>]", k): out["chain_apply"] = fb + wb   # MODE_APPLY, full mask
            if k.startswith("viterbi_fast_kernel"): out["viterbi"] = fb + wb
            elif k.startswith("viterbi_kernel") and "viterbi" not in out: out["viterbi"] = fb + wb
        else:
            out["calibration_copy_4e9_read_4e9_write"] = {"fetch_bytes_corrected": fb, "write_bytes": wb}
# configs 4 / 5: bytes per STEP of the step's own kernels (the run is 1 warm-up + 2 steps; the chain that prepares the
# inputs runs once and is left out)
STEP_KERNELS = {"cfg4": ("block_cell_reduce_kernel", "group_partial_sums_kernel", "group_means_finish_kernel", "reduce_moments_kernel", "i3_params_kernel", "viterbi_redo_kernel",
                         "viterbi_kernel", "viterbi_fast_kernel", "broadcast_states_kernel"),
                "cfg5": ("median9_probe_count_kernel", "median9_probe_finish_kernel", "median9_classify_kernel", "median9_sweep_kernel", "median9_border_kernel", "median9_units_kernel", "median9_strip_kernel", "median_filter9_kernel", "median9_sparse_kernel", "median_filter_kernel")}
def load_all(path):
    agg = collections.defaultdict(list)
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        if "icnv" in r["Kernel_Name"]:
            m = re.search(r"::(\w+_kernel)", r["Kernel_Name"])
            agg[m.group(1) if m else r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    return agg
for t, names in STEP_KERNELS.items():
    fa, wa = load_all(f"{root}/FETCH_SIZE/{t}_counter_collection.csv"), load_all(f"{root}/WRITE_SIZE/{t}_counter_collection.csv")
    if not fa:
        continue
    per = {}
    for k in names:
        if k in fa:
            fb, wb = sum(fa[k]) * 2 * 1024 / 3.0, sum(wa.get(k, [0.0])) * 1024 / 3.0
            per[k] = fb + wb
            lines.append(f"{t}:{k}  launches/step {len(fa[k]) / 3.0:.2f}  fetch_bytes/step(x2 corrected) {fb:.4g}  write_bytes/step {wb:.4g}  total/step {fb + wb:.4g}")
    out["config%s_bytes_per_step" % t[-1]] = sum(per.values())
    out["config%s_by_kernel" % t[-1]] = per
# what the counters belong to: the library sources' stamp (bench.py quotes the traffic only for the same stamp) and the commit
# (ICNV_COMMIT: the GPU box has no .git -- pass `ICNV_COMMIT=$(git rev-parse --short HEAD)` in the gpurun command)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as _bench
out["source_sha16"] = _bench.source_stamp()
out["commit"] = os.environ.get("ICNV_COMMIT") or None
lines.append(f"source_sha16 {out['source_sha16']}  commit {out['commit']}")
print("\n".join(lines))
os.makedirs("profiles", exist_ok=True)
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
