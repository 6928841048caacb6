#!/bin/bash
# usage (on the GPU box): tools/stress.sh <rounds> — the full-size parity check over and over, across shapes, wave counts,
# hand-off array counts and the reserve: the hand-off / termination protocol of the dataflow launch has no second chance
R=${GRAFT_REPO_ROOT:-/root/repo}; rounds=${1:-3}; fail=0
for r in $(seq $rounds); do
  for cfg in "5000x2000 8 64 0" "5000x2000 16 64 0" "5000x2000 8 4 0" "5000x2000 8 64 8" "5000x2000 3 1 0" "20000x64 8 64 0" "200x50000 8 64 8" "1000x10000 12 16 4" "50000x8 8 64 0"; do
    set -- $cfg
    out=$(C2A_PEEL_FIFOS=$3 C2A_PEEL_RESERVE=$4 $R/tools/shape_check.sh $1@$2 2>&1 | tail -1)
    case "$out" in *"== oracle"*) ;; *) fail=$((fail+1)); echo "FAIL [$cfg]: $out";; esac
    echo "$r [$cfg] $out" | cut -c1-150
  done
done
# the families beyond layered_dag at 10 M gates (synth.family; tools/family_check.py: every result array against the oracle, k_peel ms per shape)
for r in $(seq $rounds); do
  out=$(timeout 900 python $R/tools/family_check.py --n 10000000 --reps 2 hub_mild hub window_all forest sha_tree const_hub strict 2>&1)
  echo "$out" | sed "s/^/$r /" | cut -c1-330
  case "$out" in *"failures: 0"*) ;; *) fail=$((fail+1));; esac
done
echo "failures: $fail"
