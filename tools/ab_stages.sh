#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; rounds=$1; shift
for r in $(seq $rounds); do for v in "$@"; do cp $R/build_ab/$v.so $R/circom-2-arithc_amd/libc2a_hip.so; echo -n "[$v] "; timeout 300 python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-cold --check 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); s=d['stages_ms']; print('ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in s.items()}, d['checked'][:40])
"; done; done
