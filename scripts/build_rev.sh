#!/bin/bash
# developer tool: build the library of a git revision into exp_libs/lib_<name>.so (A/B against the working tree on one box:
# scripts/bench_libs.sh)    usage: scripts/build_rev.sh <git-rev> <name>
set -e
rev=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/icnv_rev_$name; rm -rf $tmp; mkdir -p $tmp $root/exp_libs
git -C $root archive $rev infercnv_amd/csrc include | tar -x -C $tmp
make -C $tmp/infercnv_amd/csrc -j8 >/dev/null
cp $tmp/infercnv_amd/libicnv_hip.so $root/exp_libs/lib_$name.so
echo built exp_libs/lib_$name.so from $rev
