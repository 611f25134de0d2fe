set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
timeout 300 python $R/bench.py > $R/gpurun_out/bench_full.json 2> $R/gpurun_out/bench_full.err
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r01 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_under_rocprof.json 2> $R/gpurun_out/prof/log.txt
ls -la $R/gpurun_out/prof | head
timeout 900 bash $R/scripts/pmc_traffic.sh > $R/gpurun_out/pmc_log.txt 2>&1
tail -2 $R/gpurun_out/bench_full.json | cut -c1-400
