#!/bin/bash
# SQ counters of the chain kernel (ablation run), one pass per counter group.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_sq; mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/g$i -o abl -- python $R/scripts/ablate_chain.py 50000 allonly > $OUT/log$i.txt 2>&1
done
ls $OUT/*
