#!/bin/bash
# usage (on the GPU box): tools/shape_try.sh LAYERSxWIDTH [...] [VAR=value ...] — peel time against graph shape / knobs
R=${GRAFT_REPO_ROOT:-/root/repo}
shapes=()
for a in "$@"; do case $a in *=*) export "$a";; *) shapes+=($a);; esac; done
for sh in "${shapes[@]}"; do
  L=${sh%x*}; W=${sh#*x}
  timeout 300 python $R/bench.py --steps 3 --warmup 1 --layers $L --layer-width $W --no-cpu-baseline --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-cold 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$sh', '$C2A_PEEL_WAVES', 'ms/step', round(d['ms_per_step'],2), 'peel', round(d['stages_ms']['peel'],2), 'bool_map', round(d['stages_ms']['bool_map'],2), 'levels', d['stats']['levels'], 'depth', d['stats']['max_depth'], 'waves', d['stats']['peel_waves'])
    else: print(l[:300])
"
done
