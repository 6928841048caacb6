#!/bin/bash
# usage (on the GPU box): tools/peel_try.sh [VAR=value ...] — one short checked bench run and one statistics run of the peel
R=${GRAFT_REPO_ROOT:-/root/repo}
for kv in "$@"; do export "$kv"; done
timeout 300 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-cold --check 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$*', 'ms/step', round(d['ms_per_step'],2), 'peel', round(d['stages_ms']['peel'],2), 'bool_map', round(d['stages_ms']['bool_map'],2), d['checked'], 'rereads', d['stats']['peel_rereads'])
    else: print(l[:300])
"
if [ -z "$NO_STATS" ]; then C2A_PEEL_STATS=1 timeout 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-cold 2>&1 | grep "peel stats" | tail -4; fi
