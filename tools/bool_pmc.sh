#!/bin/bash
# usage (on the GPU box): tools/bool_pmc.sh <counter> [<counter> ...] — tools/bool_alloc.py (k_boolify on buffers that landed in
# different places) under rocprofv3 --pmc: per k_boolify dispatch its duration and the counters
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
d=/tmp/bpmc_$1; rm -rf $d
timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $d -- python $R/tools/bool_alloc.py > $d.log 2>&1
grep "k_boolify ms" $d.log
python3 - $d <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_boolify" in r["Kernel_Name"]: dur[r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_boolify" in r["Kernel_Name"]: cnt[r["Dispatch_Id"]][r["Counter_Name"]] = cnt[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(cnt, key=lambda k: dur.get(k, (0, 0))[0]):
    print("%8.1f us  " % (dur.get(k, (0, 0))[1] / 1e3) + "  ".join("%s %.4g" % (c, v) for c, v in sorted(cnt[k].items())))
PY
