#!/bin/bash
# usage (on the GPU box): tools/ab.sh <rounds> <variant> [<variant> ...] — the short checked bench run with build_ab/<variant>.so
# in place of the product library, the variants taken in turn, <rounds> times (same box, same minute: box-to-box variance is
# larger than most of the differences worth measuring)
R=${GRAFT_REPO_ROOT:-/root/repo}; rounds=$1; shift
for r in $(seq $rounds); do
  for v in "$@"; do
    cp $R/build_ab/$v.so $R/circom-2-arithc_amd/libc2a_hip.so
    echo -n "[$v] "; NO_STATS=1 $R/tools/peel_try.sh
  done
done
