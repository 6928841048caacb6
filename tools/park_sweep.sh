#!/bin/bash
# usage (on the GPU box): tools/park_sweep.sh <variant> [<variant> ...] — THE RESERVE of the dataflow launch (c2a_peel.h) across the shapes it matters for: wide and
# shallow (tools/extreme_probe.py), the headline's shape and strict layers (tools/window_probe.py), the SHA-256 chains and the 10 M-gate families, with
# build_ab/<variant>.so in place of the product library; C2A_PEEL_WAVES / C2A_PEEL_RESERVE / C2A_PEEL_RELEASE come from the environment
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "$@"; do
  cp build_ab/$v.so circom-2-arithc_amd/libc2a_hip.so; echo "== $v"
  timeout 900 python tools/extreme_probe.py matmul_170 ten_layers_1m "butterfly_20x2^19" narrow_20x100k 2>&1 | cut -c1-120
  timeout 300 python tools/window_probe.py 64 5000 2000; timeout 300 python tools/window_probe.py 1 5000 2000
  timeout 300 python tools/family_check.py --n 31000 --reps 5 sha_chain | cut -c90-200
  timeout 300 python tools/family_check.py --n 10000000 --reps 2 sha_tree hub window_all forest const_hub | cut -c1-12,90-200
done
