#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/build_ab/sh.so $R/circom-2-arithc_amd/libc2a_hip.so
for r in 1 2; do for k in 1 2 3 4 6 8 12; do echo -n "[shallow=$k] "; C2A_PEEL_SHALLOW=$k NO_STATS=1 $R/tools/peel_try.sh | cut -c1-60; done; done
python -m pytest $R/tests -m gpu -x -q -k "parity_build or full_size or random" 2>&1 | tail -2
