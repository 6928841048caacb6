#!/bin/bash
# usage (on the GPU box): tools/sweep_waves.sh — peel time against the number of waves per CU / reserve waves (headline shape)
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "C2A_PEEL_WAVES=6" "C2A_PEEL_WAVES=8" "C2A_PEEL_WAVES=10" "C2A_PEEL_WAVES=12" "C2A_PEEL_WAVES=16" "C2A_PEEL_WAVES=8 C2A_PEEL_RESERVE=4" "C2A_PEEL_WAVES=8 C2A_PEEL_FIFOS=32"; do
  echo -n "[$cfg] "; env $cfg NO_STATS=1 $R/tools/peel_try.sh | cut -c1-120
done
