#!/bin/bash
# SQ / LDS / HBM counters of the certified fast Viterbi kernel, one pass per counter group (no other trace domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_vitfast; mkdir -p $OUT
python $R/scripts/run_viterbi.py 50000 5 > $OUT/time_fast.txt 2>&1
ICNV_VITERBI_MODE=1 python $R/scripts/run_viterbi.py 50000 3 > $OUT/time_exact.txt 2>&1
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/g$i -o vit -- python $R/scripts/run_viterbi.py 50000 1 > $OUT/log$i.txt 2>&1
done
cat $OUT/time_*.txt
python - <<'PY'
import csv, glob, collections, os
out=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc_vitfast"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "viterbi" in r["Kernel_Name"]:
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
