#!/bin/bash
# Developer ablation of the certified fast Viterbi kernel: build variants with pieces stubbed out (ICNV_VF_EXP bits:
# 1 no traceback, 2 no back-pointer stores (only together with 1), 4 no recurrence, 8 no observation loads, 16 no coefficient reads,
# 32 no interval lookup) and time each on the bench workload.  Results are wrong by construction -- timing only.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R/infercnv_amd/csrc
mkdir -p exp_obj
if [ "$1" = build ]; then
  for e in ${EXPS:-1 2 3 4 8 16 32 48 63}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $(echo $e | sed 's/^/-DICNV_VF_EXP=/; s/_nt/ -DICNV_VF_NT=/; s/_ch/ -DICNV_VF_CH=/; s/_pol/ -DICNV_VF_POLICY=/; s/_sb/ -DICNV_VF_SB=/; s/_tg/ -DICNV_VF_TG=/; s/_pipe/ -DICNV_VF_PIPE=/; s/_tb/ -DICNV_VF_TB=/') -c viterbi_fast.hip -o exp_obj/vf_$e.o &
  done; wait
  for e in ${EXPS:-1 2 3 4 8 16 32 48 63}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libicnv_exp_$e.so api.o chain_kernels.o chain_m7.o chain_m15.o chain_m23.o chain_l35.o chain_large.o viterbi_kernels.o exp_obj/vf_$e.o emission_table.o median_kernels.o stats_kernels.o distance_kernels.o
  done
  exit 0
fi
cd $R
echo "baseline: $(python scripts/run_viterbi.py 50000 5 | tail -1)"
for e in ${EXPS:-1 2 3 4 8 16 32 48 63}; do
  echo "exp $e: $(ICNV_LIB=$R/infercnv_amd/libicnv_exp_$e.so python scripts/run_viterbi.py 50000 5 | tail -1)"
done
