#!/bin/bash
# usage (on the GPU box): tools/bool_alloc_ab.sh <variant.so> — tools/bool_alloc.py with the product library, then with a variant
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/circom-2-arithc_amd/libc2a_hip.so /tmp/base.so
for i in 1 2; do
echo "== base"; python $R/tools/bool_alloc.py 2>&1 | grep "k_boolify ms"
cp $1 $R/circom-2-arithc_amd/libc2a_hip.so
echo "== variant"; python $R/tools/bool_alloc.py 2>&1 | grep "k_boolify ms"
cp /tmp/base.so $R/circom-2-arithc_amd/libc2a_hip.so
done
