#!/bin/bash
# developer tool: build a variant of the library into exp_libs/lib_<name>.so (travels to the GPU box, git-ignored)
#   scripts/build_variant.sh <name> "<extra hipcc flags>" [extra .hip sources ...]
# Only the translation units the flags can change are rebuilt; the other objects come from the product build.
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/infercnv_amd/csrc
out=/tmp/exp_obj/$name; mkdir -p $out $root/exp_libs
HIPCC=/opt/rocm/bin/hipcc
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I$src -I$root/include"
objs=""
for f in ${REBUILD-chain_kernels chain_m15 chain_m15s chain_m15t chain_w11 chain_w11t}; do
  extra=""; case $f in viterbi_*) extra=-ffp-contract=off;; median_kernels) extra=-fno-honor-nans;; esac
  $HIPCC $COMMON $extra $flags -c $src/$f.hip -o $out/$f.o &
done
for x in "$@"; do b=$(basename $x .hip); $HIPCC $COMMON $flags -c $x -o $out/$b.o & if [ ! -f $src/$b.o ]; then objs="$objs $out/$b.o"; fi; done   # (an extra source named like a product unit replaces it)
wait
for o in $src/*.o; do b=$(basename $o); if [ -f $out/$b ]; then objs="$objs $out/$b"; else objs="$objs $o"; fi; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $root/exp_libs/lib_$name.so $objs
echo built exp_libs/lib_$name.so
