cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mf
(timeout 600 python tests/campaigns/stress_median_filter.py 0 30 2>&1 | tail -2) > gpurun_out/mf/stress.log
(ICNV_MF9_DEBUG=1 timeout 300 python scripts/time_median_filter.py 2>&1 | tail -2) > gpurun_out/mf/time.log
(timeout 900 python bench.py --config 5 --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 ms', d['ms_per_step'], 'no_ties ms', d['no_ties_input']['ms_per_step'])") > gpurun_out/mf/bench5.log 2>&1
cat gpurun_out/mf/stress.log gpurun_out/mf/time.log gpurun_out/mf/bench5.log
