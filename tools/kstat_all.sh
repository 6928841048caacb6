#!/bin/bash
# usage (on the GPU box): tools/kstat_all.sh <tag> [env assignments...] — rocprofv3 kernel stats of one short bench run, every kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-cold ${KSTAT_ARGS} > /tmp/kt_$tag.log 2>&1
f=$(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1)
echo "== $tag $@"; grep '^{"metric' /tmp/kt_$tag.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],2), {k: round(v,3) for k,v in d['stages_ms'].items()})"
python3 - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    calls=int(r["Calls"]); 
    print("%-70s calls %4d  avg_us %9.1f  per_step_us %9.1f" % (r["Name"][:70], calls, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3/4))
PY
mkdir -p $R/gpurun_out; cp $f $R/gpurun_out/kstats_$tag.csv
