cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mf
export TMPDIR=/tmp
(timeout 600 python tests/campaigns/stress_median_filter.py 0 30 2>&1 | tail -2) > gpurun_out/mf/stress.log
(ICNV_MF9_DEBUG=1 timeout 300 python scripts/_mf_time2.py 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/mf/time2.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o mf -- python $GRAFT_REPO_ROOT/scripts/_mf_time2.py > /dev/null 2>&1; python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof -name "*.db" | head -1) | grep -i "icnv" | cut -c1-260) > gpurun_out/mf/prof.log 2>&1
cat gpurun_out/mf/stress.log gpurun_out/mf/time2.log gpurun_out/mf/prof.log
