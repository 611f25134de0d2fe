cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mf
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "median_filter" 2>&1 | tail -3) > gpurun_out/mf/pytest.log
(timeout 600 python tests/campaigns/stress_median_filter.py 0 100 2>&1 | tail -1) > gpurun_out/mf/stress.log
(timeout 600 python tests/campaigns/stress_median_filter.py 0 30 large 2>&1 | tail -1) >> gpurun_out/mf/stress.log
(timeout 900 python bench.py --config 5 --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 ms', d['ms_per_step'], 'no_ties ms', d['no_ties_input']['ms_per_step'])") > gpurun_out/mf/bench5.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o mf -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1; python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof -name "*.db" | head -1) | grep -i "median" | cut -c1-200) > gpurun_out/mf/prof.log 2>&1
cat gpurun_out/mf/pytest.log gpurun_out/mf/stress.log gpurun_out/mf/bench5.log gpurun_out/mf/prof.log
