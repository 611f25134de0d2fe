# After a change of the fast Viterbi kernel / table: tests, stress campaign, A/B against a reference build, counters.
# usage (GPU box): bash scripts/refresh_viterbi.sh   -> gpurun_out/vit/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/vit; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu -k "viterbi or hmm or i3 or i6 or config1 or full_size or smoke" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 python tests/campaigns/stress_viterbi_fast.py 6000 1500 > $O/stress.txt 2>&1; tail -1 $O/stress.txt
if [ -n "$LIBS" ]; then bash scripts/bench_libs.sh > /dev/null 2>&1; cp gpurun_out/exp.log $O/ab.txt; cat $O/ab.txt; fi
timeout 300 python bench.py --no-cpu-baseline | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['ms_per_step'],3), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()}); print(d['roofline_by_kernel']['viterbi']['note'][:120] if 'roofline_by_kernel' in d else '')"
timeout 900 bash scripts/pmc_viterbi_fast.sh > $O/pmc_log.txt 2>&1; cp gpurun_out/pmc_vitfast/summary.txt $O/pmc_viterbi_fast.txt; find gpurun_out/pmc_vitfast -name "*.db" -delete
grep "viterbi_fast_kernel" $O/pmc_viterbi_fast.txt
timeout 300 python scripts/bench_configs.py > $O/configs_slices.json 2>/dev/null; cat $O/configs_slices.json | cut -c1-900
