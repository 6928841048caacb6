#!/bin/bash
# usage (on the GPU box): tools/slices_ab.sh <variant> [<variant> ...] — k_boolify's SLICES (c2a_kernels.h): bool_map per gate type (tools/bool_by_op.py, width 32), the
# headline's boolify, and BASELINE's small configs, with build_ab/<variant>.so in place of the product library
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for v in "$@"; do cp build_ab/$v.so circom-2-arithc_amd/libc2a_hip.so; echo "== $v"
  timeout 600 python tools/bool_by_op.py 32 2>&1 | grep -v "^\[c2a" | cut -c1-150
  timeout 300 python tools/bool_ramp.py 2>&1 | grep -v "^\[c2a"
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cold --no-width64 --no-artefacts --no-prune --no-live-pmc --no-cpu-baseline --no-reference-shaped 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        for c in d['configs']: print(c['name'][:34], 'gpu steady', round(c['gpu_ms_steady'],3), 'bool_map', c['gpu_stages_ms']['bool_map'], 'hybrid', round(c['hybrid_ms'],3), 'load_circuit+boolify', round(c['hybrid_gpu_load_circuit_and_boolify_ms'],3), c['checked'])"
done
