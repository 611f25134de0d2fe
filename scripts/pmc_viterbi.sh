#!/bin/bash
# SQ counters of the Viterbi kernel, one pass per counter group (no other trace domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_vit; mkdir -p $OUT
python $R/scripts/run_viterbi.py 50000 3 > $OUT/time_random.txt 2>&1
python $R/scripts/run_viterbi.py 50000 3 uniform > $OUT/time_uniform.txt 2>&1
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/g$i -o vit -- python $R/scripts/run_viterbi.py 50000 1 > $OUT/log$i.txt 2>&1
done
cat $OUT/time_*.txt
