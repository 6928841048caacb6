#!/bin/bash
# usage (on the GPU box): tools/shape_check.sh LAYERSxWIDTH[@WAVES] ... — full-size parity check (checksums of sorted ids and
# of the emitted / bit-blasted circuits against the oracle) on other graph shapes than the headline one
R=${GRAFT_REPO_ROOT:-/root/repo}
for sh in "$@"; do
  w=${sh#*@}; s=${sh%@*}; [ "$w" = "$sh" ] && w=8
  L=${s%x*}; W=${s#*x}
  C2A_PEEL_WAVES=$w timeout 600 python $R/bench.py --steps 2 --warmup 1 --layers $L --layer-width $W --width 8 --no-cpu-baseline --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-cold --check 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$sh', 'peel', round(d['stages_ms']['peel'],2), 'depth', d['stats']['max_depth'], 'chunks', d['stats']['path_chunks'], 'waves', d['stats']['peel_waves'], '|', d['checked'])
    else: print(l[:300])
"
done
