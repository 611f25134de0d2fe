#!/bin/bash
# developer tool (GPU box): bench kernel times + output checksums for several builds of the library, then the chain tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LIBS=${LIBS:-"base new"}
{
for lib in $LIBS; do
  ICNV_LIB=$PWD/exp_libs/lib_$lib.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --checksum 1 2>gpurun_out/exp_$lib.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],3), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()}, d['checksums']['per_part'])"
done
} > gpurun_out/exp.log 2>&1
if [ -n "$TESTS" ]; then
  if [ -n "$TESTLIB" ]; then export ICNV_LIB=$PWD/exp_libs/lib_$TESTLIB.so; fi
  timeout 1200 python -m pytest tests -x -q -m gpu -k "$TESTS" > gpurun_out/exp_pytest.log 2>&1
  tail -5 gpurun_out/exp_pytest.log
fi
cat gpurun_out/exp.log
