#!/bin/bash
# usage (on the GPU box): tools/ab2.sh <rounds> <variant> [<variant> ...] [-- bench args] — ms/step and the stage split with build_ab/<variant>.so in
# place of the product library, the variants taken in turn on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}; rounds=$1; shift
vars=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vars+=("$1"); shift; done
for r in $(seq $rounds); do for v in "${vars[@]}"; do cp $R/build_ab/$v.so $R/circom-2-arithc_amd/libc2a_hip.so; bash $R/tools/bench_stages.sh $v "$@" | cut -c1-330; done; done
