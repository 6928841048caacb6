#!/bin/bash
# usage (on the GPU box): tools/sweep.sh VAR v1 v2 ... — one short bench run per value of the environment knob VAR
R=${GRAFT_REPO_ROOT:-/root/repo}; var=$1; shift
for v in "$@"; do
  env $var=$v timeout 300 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-cold --no-reference-shaped 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$var=$v', 'ms/step', round(d['ms_per_step'],2), 'peel', round(d['stages_ms']['peel'],2), 'bool_map', round(d['stages_ms']['bool_map'],2))"
done
