#!/bin/bash
# usage (on the GPU box): tools/bench_stages.sh <label> [ENV=value ...] [-- bench.py args] — one short bench run, ms/step and the stage split
R=${GRAFT_REPO_ROOT:-/root/repo}; label=$1; shift
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done; [ "$1" == "--" ] && shift
env "${envs[@]}" python $R/bench.py --steps 10 --warmup 3 --no-cold --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-cpu-baseline --no-reference-shaped "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d.get('checked') or {}
        print('$label', 'ms/step', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['stages_ms'].items()}, 'checked', c.get('ok'))"
