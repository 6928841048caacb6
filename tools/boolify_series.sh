#!/bin/bash
# usage (on the GPU box): tools/boolify_series.sh [steps] — the duration of every k_boolify and k_peel launch of one bench run, in
# launch order with its start time: does the store-bound kernel drift (power / clocks) while the latency-bound one does not?
R=${GRAFT_REPO_ROOT:-/root/repo}; steps=${1:-60}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bs
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/bs -- python $R/bench.py --steps $steps --warmup 3 --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-cpu-baseline > /tmp/bs.log 2>&1
f=$(find /tmp/bs -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
b = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "k_boolify" in r["Kernel_Name"]]
p = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "k_peel<" in r["Kernel_Name"]]
print("k_boolify (start ms: us):", " ".join("%.0f:%.0f" % (s / 1e6, d / 1e3) for s, d in b))
print("k_peel us:", " ".join("%.0f" % (d / 1e3) for d in p))
PY
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -8
