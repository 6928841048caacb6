#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for cfg in "C2A_PEEL_SEED_CHUNK=8" "C2A_PEEL_SEED_CHUNK=4" "C2A_PEEL_SEED_CHUNK=2" "C2A_PEEL_SEED_CHUNK=16" "C2A_PEEL_SEED_CHUNK=4 C2A_PEEL_SHALLOW=5" "C2A_PEEL_SEED_CHUNK=2 C2A_PEEL_SHALLOW=6"; do echo -n "[$cfg] "; env $cfg NO_STATS=1 $R/tools/peel_try.sh | cut -c1-60; done; done
