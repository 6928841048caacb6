#!/bin/bash
# usage (on the GPU box): tools/ab_family.sh <n> "<families>" <variant> [<variant> ...] — tools/family_check.py with build_ab/<variant>.so in place of
# the product library, the variants taken in turn on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}; n=$1; fams=$2; shift 2
for v in "$@"; do cp $R/build_ab/$v.so $R/circom-2-arithc_amd/libc2a_hip.so; echo "== $v"; timeout 900 python $R/tools/family_check.py --n $n --reps 3 $fams | cut -c1-330; done
