#!/bin/bash
# usage (on the GPU box): tools/ab_configs.sh <variant> [<variant> ...] — the deep and narrow REAL circuits (SHA-256 x 9 blocks, a Merkle tree of SHA-256 blocks) and the
# headline-shaped families through tools/family_check.py with build_ab/<variant>.so in place of the product library, the variants taken in turn on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do cp $R/build_ab/$v.so $R/circom-2-arithc_amd/libc2a_hip.so; echo "== $v"
  timeout 300 python $R/tools/family_check.py --n 31000 --reps 5 sha_chain | cut -c1-215 | head -1
  timeout 300 python $R/tools/family_check.py --n 1000000 --reps 3 sha_tree sha_chain | cut -c1-215 | head -2
done
