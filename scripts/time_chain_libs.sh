#!/bin/bash
# developer tool: chain_apply time of the bench workload for several builds of the library (ICNV_LIB)
for lib in "$@"; do
  ICNV_LIB=$PWD/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', {k: round(v['avg_ms'],3) for k,v in d['kernels'].items() if k.startswith('chain_apply') or k=='viterbi'})"
done
