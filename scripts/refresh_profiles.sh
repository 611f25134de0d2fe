# Refresh the evidence under gpurun_out/ (copied into profiles/ afterwards): the default bench line, the
# rocprofv3 kernel-trace statistics of the same command, and the HBM traffic counters (separate --pmc passes).
# usage (on the GPU box): bash scripts/refresh_profiles.sh r02
set -x
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
timeout 300 python $R/bench.py > $R/gpurun_out/bench_full.json 2> $R/gpurun_out/bench_full.err
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_under_rocprof.json 2> $R/gpurun_out/prof/log.txt
python $R/scripts/rocprof_summary.py $R/gpurun_out/prof/${TAG}_results.db > $R/gpurun_out/prof/${TAG}_kernel_stats.txt 2>&1
ls -la $R/gpurun_out/prof | head
timeout 900 bash $R/scripts/pmc_traffic.sh > $R/gpurun_out/pmc_log.txt 2>&1
cd $R && python scripts/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc/summary.txt 2>&1; cp profiles/pmc_traffic.json gpurun_out/pmc/pmc_traffic.json
python scripts/bench_host_path.py 20000 1 > gpurun_out/host_path.json 2> gpurun_out/host_path.err
python scripts/bench_configs.py > gpurun_out/configs.json 2> gpurun_out/configs.err
tail -2 $R/gpurun_out/bench_full.json | cut -c1-600
