# Refresh the evidence under gpurun_out/<tag>/ (copied into profiles/ afterwards): the default bench line (with its parity
# block), the rocprofv3 kernel-trace statistics of the same command, the HBM traffic counters (separate --pmc passes), the
# memory-system microbenchmarks the ceilings in bench.py come from, the SQ / LDS counters of the fast Viterbi and of the
# median filter, BASELINE configs 4 / 5, the host-buffer path.
# usage (on the GPU box):  ICNV_COMMIT=<short hash> bash scripts/refresh_profiles.sh r05
#   (the box has no .git: the commit travels in the environment and is written into every summary)
set -x
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O/prof
# ceilings: what HBM gives a 1-read : 2-write stream (the smooth pass) and a per-lane column walk (the Viterbi's observations)
$R/scripts/ubench/stream_1r2w > $O/ubench_stream_1r2w.txt 2>&1
$R/scripts/ubench/column_walk > $O/ubench_column_walk.txt 2>&1
cp $O/ubench_stream_1r2w.txt $R/profiles/ubench_stream_1r2w.txt      # bench.py reads its ceiling from profiles/ (this run's, when present)
# counters first: bench.py quotes profiles/pmc_traffic.json only when it carries this tree's source stamp
timeout 900 bash $R/scripts/pmc_traffic.sh > $O/pmc_log.txt 2>&1
cd $R && python scripts/pmc_summary.py gpurun_out/pmc > $O/${TAG}_pmc_traffic.txt 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json
find $R/gpurun_out/pmc -name "*.db" -delete
cd /tmp
timeout 300 python $R/bench.py > $O/bench_full.json 2> $O/bench_full.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o $TAG -- python $R/bench.py --no-cpu-baseline > $O/prof/bench_under_rocprof.json 2> $O/prof/log.txt   # the default command without its CPU / host-buffer legs (they launch the same kernels on other matrices and would mix into the averages)
python $R/scripts/rocprof_summary.py $O/prof/${TAG}_results.db > $O/${TAG}_kernel_stats.txt 2>&1
python $R/scripts/step_gaps.py $O/prof/${TAG}_results.db > $O/${TAG}_step_timeline.txt 2>&1   # the same trace as a timeline: kernel time vs time between kernels, per step
rm -f $O/prof/*.db      # (gpurun merges at most 64 MiB back: the summaries travel, the raw traces do not)
for c in 4 5; do
  timeout 300 python $R/bench.py --config $c --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o ${TAG}_config$c -- python $R/bench.py --config $c --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-side-legs > /dev/null 2> $O/prof/log_config$c.txt
  python $R/scripts/rocprof_summary.py $O/prof/${TAG}_config${c}_results.db > $O/${TAG}_kernel_stats_config$c.txt 2>&1
  rm -f $O/prof/*.db
done
cd $R && python scripts/bench_host_path.py 20000 1 > $O/host_path.json 2> $O/host_path.err
# SQ / LDS counters of the fast Viterbi kernel and of the 9 x 9 median filter
timeout 900 bash $R/scripts/pmc_viterbi_fast.sh > $O/pmc_viterbi_fast_log.txt 2>&1
cp $R/gpurun_out/pmc_vitfast/summary.txt $O/${TAG}_pmc_viterbi_fast.txt; find $R/gpurun_out/pmc_vitfast -name "*.db" -delete
timeout 600 bash $R/scripts/pmc_median.sh > $O/pmc_median_log.txt 2>&1
cp $R/gpurun_out/pmc_median/summary.txt $O/${TAG}_pmc_median.txt; find $R/gpurun_out/pmc_median -name "*.db" -delete
timeout 600 python $R/scripts/bench_configs.py > $O/configs_slices.json 2> $O/configs_slices.err
# the north star's own shape on ONE GPU (BASELINE configs[2]: 10 000 x 1 000 000, parity on the matrix itself) and the (G, sigma) sweep
timeout 400 python $R/bench.py --config 3 --steps 5 --warmup 2 > $O/bench_config3_n1.json 2> $O/bench_config3_n1.err
timeout 600 python $R/scripts/sweep_shapes.py > $O/sweep.json 2> $O/sweep.err
find $R/gpurun_out -name "*.db" -delete; du -sh $R/gpurun_out
tail -1 $O/bench_full.json | cut -c1-400
