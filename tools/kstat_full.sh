#!/bin/bash
# usage (on the GPU box): tools/kstat_full.sh <tag> [bench args...] — rocprofv3 kernel trace of one short bench run, every kernel's row
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kf_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kf_$tag -o $tag -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-cold --no-reference-shaped "$@" > /tmp/kf_$tag.log 2>&1
db=$(find /tmp/kf_$tag -name "*_results.db" | head -1)
mkdir -p $R/gpurun_out
python3 $R/tools/rocpd_summary.py $db > $R/gpurun_out/kfull_$tag.txt
grep '^{"metric' /tmp/kf_$tag.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['stages_ms'].items()})"
cut -c1-150 $R/gpurun_out/kfull_$tag.txt | head -50
