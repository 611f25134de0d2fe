#!/bin/bash
# Collect HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) for the
# bench's kernels plus a calibration copy (chain kernel with stage_mask 0 = 4 GB read + 4 GB write).
# Run on the GPU box:  bash scripts/pmc_traffic.sh   -> gpurun_out/pmc/*.csv
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o bench -- \
      python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/bench_$C.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o copy -- \
      python $R/scripts/ablate_chain.py 50000 copyonly > $OUT/copy_$C.log 2>&1
  for c in 4 5; do    # BASELINE configs 4 (group HMM) and 5 (median filter): 1 warm-up + 2 steps = 3 steps per run
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -o cfg$c -- \
        python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side-legs > $OUT/cfg${c}_$C.log 2>&1
  done
done
ls -R $OUT | head -40
