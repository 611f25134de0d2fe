cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mf
(timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "median_filter" 2>&1 | tail -15) > gpurun_out/mf/pytest.log
(timeout 600 python tests/campaigns/stress_median_filter.py 0 200 2>&1 | tail -1) > gpurun_out/mf/stress.log
(timeout 600 python tests/campaigns/stress_median_filter.py 0 60 large 2>&1 | tail -1) >> gpurun_out/mf/stress.log
cat gpurun_out/mf/pytest.log gpurun_out/mf/stress.log
