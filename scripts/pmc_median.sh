#!/bin/bash
# SQ / LDS counters of the 9 x 9 median filter kernel, one pass per counter group (no other trace domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_median; mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
         "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/g$i -o mf -- python $R/scripts/time_median_filter.py > $OUT/log$i.txt 2>&1
done
python - <<'PY'
import csv, glob, collections, os
out=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc_median"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "median" in r["Kernel_Name"]:
            k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].split("::")[-1]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","."))
import bench as _bench
with open(out+"/summary.txt","w") as fo:
    fo.write(f"# library sources sha16 {_bench.source_stamp()}  commit {os.environ.get('ICNV_COMMIT') or 'unknown'}\n")
    for k,d in agg.items():
        for c,v in sorted(d.items()):
            line=f"{k}  {c}  mean={sum(v)/len(v):.6g}  n={len(v)}"
            print(line); fo.write(line+"\n")
PY
