#!/bin/bash
# usage (on the GPU box): tools/bool_ctx_ab.sh <variant> <variant> — k_boolify / first-boolify time over six fresh contexts per variant library, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for v in "$@"; do cp $R/build_ab/$v.so $R/circom-2-arithc_amd/libc2a_hip.so; echo "== $v"; python $R/tools/cold_try.py 2>&1 | grep "^create" | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$12,$13,$14}'; done; done
