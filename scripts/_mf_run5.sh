cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mf
(MF_CELLS=50000 ICNV_MF9_DEBUG=1 timeout 300 python scripts/_mf_time2.py 2>&1 | grep -v amdgpu.ids | tail -6) > gpurun_out/mf/time2.log
(MF_CELLS=20000 ICNV_MF9_DEBUG=1 timeout 300 python scripts/_mf_time2.py 2>&1 | grep -v amdgpu.ids | tail -6) >> gpurun_out/mf/time2.log
cat gpurun_out/mf/time2.log
