#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the default bench line and the rocprofv3 kernel summary of the same command,
# written to gpurun_out/ (copy the two files into profiles/ afterwards).
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=${1:-r01}
mkdir -p $R/gpurun_out
cd $R && timeout 900 python bench.py > $R/gpurun_out/${tag}_bench_default.json 2> $R/gpurun_out/${tag}_bench_default.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag -- python $R/bench.py --no-cpu-baseline --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-reference-shaped > /tmp/prof_$tag.log 2>&1
db=$(find /tmp/prof_$tag -name "*_results.db" | head -1)
python3 $R/tools/rocpd_summary.py $db | sed "s#/tmp/prof_$tag#rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline :#" > $R/gpurun_out/${tag}_kernel_stats_bench_default.txt
tail -c 600 $R/gpurun_out/${tag}_bench_default.json; echo; head -8 $R/gpurun_out/${tag}_kernel_stats_bench_default.txt
