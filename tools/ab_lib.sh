#!/bin/bash
# usage (on the GPU box): tools/ab_lib.sh <variant.so> [rounds] — the product library against a variant build of it, alternating on
# the same box (the variant is copied over the product library in the box's scratch copy of the repo only)
R=${GRAFT_REPO_ROOT:-/root/repo}; var=$1; rounds=${2:-2}
cp $R/circom-2-arithc_amd/libc2a_hip.so /tmp/base.so
for i in $(seq $rounds); do
  cp /tmp/base.so $R/circom-2-arithc_amd/libc2a_hip.so; bash $R/tools/bench_stages.sh base
  cp $var $R/circom-2-arithc_amd/libc2a_hip.so; bash $R/tools/bench_stages.sh variant
done
cp /tmp/base.so $R/circom-2-arithc_amd/libc2a_hip.so
