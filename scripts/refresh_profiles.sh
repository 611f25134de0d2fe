# Refresh the evidence under gpurun_out/ (copied into profiles/ afterwards): the default bench line, the rocprofv3
# kernel-trace statistics of the same command, the HBM traffic counters (separate --pmc passes), the memory-system and
# fp64 microbenchmarks the ceilings in bench.py come from, the other BASELINE configs, and the opt-in chain2 kernels.
# usage (on the GPU box): bash scripts/refresh_profiles.sh r03
set -x
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O/prof
# ceilings: what HBM gives a 1-read : 1-write and a 1-read : 2-write stream; dependent-issue latency of fp64 VALU
$R/scripts/ubench/stream_1r2w > $O/ubench_stream_1r2w.txt 2>&1
$R/scripts/ubench/f64_latency > $O/ubench_f64_latency.txt 2>&1
cp $O/ubench_stream_1r2w.txt $R/profiles/ubench_stream_1r2w.txt      # bench.py reads its ceiling from profiles/ (this run's, when present)
timeout 300 python $R/bench.py > $O/bench_full.json 2> $O/bench_full.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o $TAG -- python $R/bench.py --no-cpu-baseline > $O/prof/bench_under_rocprof.json 2> $O/prof/log.txt   # the default command without its CPU / host-buffer legs (they launch the same kernels on a 20 000-cell matrix and would mix into the averages)
python $R/scripts/rocprof_summary.py $O/prof/${TAG}_results.db > $O/${TAG}_kernel_stats.txt 2>&1
rm -f $O/prof/*.db      # (gpurun merges at most 64 MiB back: the summaries travel, the raw traces do not)
for c in 4 5; do
  timeout 300 python $R/bench.py --config $c --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o ${TAG}_config$c -- python $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $O/prof/log_config$c.txt
  python $R/scripts/rocprof_summary.py $O/prof/${TAG}_config${c}_results.db > $O/${TAG}_kernel_stats_config$c.txt 2>&1
  rm -f $O/prof/*.db
done
timeout 900 bash $R/scripts/pmc_traffic.sh > $O/pmc_log.txt 2>&1
cd $R && python scripts/pmc_summary.py gpurun_out/pmc > $O/${TAG}_pmc_traffic.txt 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json
find $R/gpurun_out/pmc -name "*.db" -delete; du -sh $R/gpurun_out/pmc
python scripts/bench_host_path.py 20000 1 > $O/host_path.json 2> $O/host_path.err
# the opt-in two-cells-per-CU chain kernels: the kept negative result
tail -2 $O/bench_full.json | cut -c1-600
# SQ / LDS counters of the fast Viterbi kernel, what the launch path costs (eager vs hipGraph replay), the other slices
timeout 900 bash $R/scripts/pmc_viterbi_fast.sh > $O/pmc_viterbi_fast_log.txt 2>&1
cp $R/gpurun_out/pmc_vitfast/summary.txt $O/${TAG}_pmc_viterbi_fast.txt; find $R/gpurun_out/pmc_vitfast -name "*.db" -delete
timeout 300 python $R/scripts/graph_step.py > $O/graph_step.txt 2>&1
timeout 600 python $R/scripts/bench_configs.py > $O/configs_slices.json 2> $O/configs_slices.err
du -sh $R/gpurun_out
