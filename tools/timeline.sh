#!/bin/bash
# usage (on the GPU box): tools/timeline.sh <tag> — every dispatch of ONE steady-state step (build + boolify) in launch order:
# start offset, duration, gap to the end of the dispatch before it.  Says what the launch gaps and the memsets cost.
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-tl}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$tag
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-width64 --no-artefacts --no-prune --no-cold  > /tmp/tl_$tag.log 2>&1
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at k_producer; take the last complete one
starts = [i for i, r in enumerate(rows) if "k_producer" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = None; gaps = 0; busy = 0; fills = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else s - prev_end
    gaps += max(gap, 0); busy += e - s
    nm = r["Kernel_Name"].replace("c2a::", "").replace("void ", "")[:48]
    if "fillBuffer" in nm: fills += e - s
    print("%9.1f us  +%7.1f  gap %6.1f  %s  grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, nm, r.get("Grid_Size", r.get("Grid_Size_X", ""))))
    prev_end = max(e, prev_end or 0)
print("step %.1f us: busy %.1f, gaps %.1f, memsets %.1f, dispatches %d" % ((prev_end - t0) / 1e3, busy / 1e3, gaps / 1e3, fills / 1e3, b - a))
PY
