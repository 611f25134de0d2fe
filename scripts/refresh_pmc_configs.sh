cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O/prof
timeout 900 bash $R/scripts/pmc_traffic.sh > $O/pmc_log.txt 2>&1
cd $R && python scripts/pmc_summary.py gpurun_out/pmc > $O/r03_pmc_traffic.txt 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json
find $R/gpurun_out/pmc -name "*.db" -delete
cd /tmp
timeout 300 python $R/bench.py --config 4 --no-cpu-baseline > $O/bench_config4.json 2> $O/bench_config4.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o r03_config4 -- python $R/bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $O/prof/log_config4.txt
python $R/scripts/rocprof_summary.py $O/prof/r03_config4_results.db > $O/r03_kernel_stats_config4.txt 2>&1
rm -f $O/prof/*.db
timeout 300 python $R/bench.py --config 5 --no-cpu-baseline > $O/bench_config5.json 2> $O/bench_config5.err
timeout 600 python $R/scripts/bench_ingest.py > $O/ingest.json 2> $O/ingest.err
tail -c 600 $O/ingest.json; tail -3 $O/ingest.err
cat $O/r03_pmc_traffic.txt | tail -12
