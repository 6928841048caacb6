#!/bin/bash
# usage (on the GPU box): tools/bench_cold_ab.sh — does the cold single shot in front (a context created, used once and destroyed)
# change what k_boolify takes in the timed steps behind it?  (same box, alternating)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  for cold in "--no-cold" ""; do
    python $R/bench.py --steps 10 --warmup 3 --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-cpu-baseline $cold 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('cold run in front:', '${cold}' == '', 'ms/step', round(d['ms_per_step'], 3), 'bool_map', round(d['stages_ms']['bool_map'], 3), 'build', round(d['stages_ms']['build_total'], 3))"
  done
done
