cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mf
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "median_filter" 2>&1 | tail -15) > gpurun_out/mf/pytest.log
(timeout 600 python tests/campaigns/stress_median_filter.py 0 60 2>&1 | tail -5) > gpurun_out/mf/stress.log
(timeout 600 python tests/campaigns/stress_median_filter.py 0 24 large 2>&1 | tail -5) >> gpurun_out/mf/stress.log
(ICNV_MF9_DEBUG=1 timeout 300 python scripts/time_median_filter.py 2>&1 | tail -8) > gpurun_out/mf/time.log
(ICNV_MF9_STRIP=0 ICNV_MF9_PROBE=0 timeout 300 python scripts/time_median_filter.py 2>&1 | tail -3) >> gpurun_out/mf/time.log
(timeout 900 python bench.py --config 5 --steps 5 --warmup 2 2>&1 | tail -3) > gpurun_out/mf/bench5.log
cat gpurun_out/mf/pytest.log gpurun_out/mf/stress.log gpurun_out/mf/time.log; tail -c 3000 gpurun_out/mf/bench5.log
