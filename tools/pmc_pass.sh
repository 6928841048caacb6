#!/bin/bash
# usage (on the GPU box): tools/pmc_pass.sh <tag> <counter> [<counter> ...] — one rocprofv3 --pmc pass (kernel-trace only) of a
# short bench run; per-kernel means go to gpurun_out/pmc_<tag>_<first counter>.json
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
d=/tmp/pmc_${tag}_$1; rm -rf $d
timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $d -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-width64 --no-artefacts --no-prune --no-configs --no-live-pmc --no-reference-shaped --no-cold ${PMC_BENCH_ARGS} > $d.log 2>&1
mkdir -p $R/gpurun_out
python3 $R/tools/pmc_by_kernel.py $d > $R/gpurun_out/pmc_${tag}_$1.json
grep -c . $R/gpurun_out/pmc_${tag}_$1.json
