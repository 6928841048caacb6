#!/bin/bash
# usage (on the GPU box): tools/ab_kernels.sh <variant.so> <kernel name pattern> — per-kernel durations of one step, product library vs variant
R=${GRAFT_REPO_ROOT:-/root/repo}; var=$1; pat=$2
cp $R/circom-2-arithc_amd/libc2a_hip.so /tmp/base.so
for i in 1 2; do
  cp /tmp/base.so $R/circom-2-arithc_amd/libc2a_hip.so; bash $R/tools/timeline.sh a > /tmp/a.txt 2>&1; echo "base:"; grep -E "$pat|^step" /tmp/a.txt
  cp $var $R/circom-2-arithc_amd/libc2a_hip.so; bash $R/tools/timeline.sh b > /tmp/b.txt 2>&1; echo "variant:"; grep -E "$pat|^step" /tmp/b.txt
done
cp /tmp/base.so $R/circom-2-arithc_amd/libc2a_hip.so
