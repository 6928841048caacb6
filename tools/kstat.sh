#!/bin/bash
# usage (on the GPU box): tools/kstat.sh <tag> [env assignments...] — rocprofv3 kernel stats of one bench run, top rows
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python $R/bench.py --steps 2 --warmup 1 ${KSTAT_CHECK---check} --no-cpu-baseline --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-cold > /tmp/kt_$tag.log 2>&1
f=$(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1)
echo "== $tag $@"; grep '^{"metric' /tmp/kt_$tag.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],2), 'peel', round(d['stages_ms']['peel'],2), d.get('checked'))"
python3 - "$f" <<PY
import csv,sys
for i,r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i<3: print(r["Name"][:60], r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
mkdir -p $R/gpurun_out; cp $f $R/gpurun_out/kstats_$tag.csv
