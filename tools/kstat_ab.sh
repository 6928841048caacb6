#!/bin/bash
# usage (on the GPU box): tools/kstat_ab.sh <rounds> <variant> [<variant> ...] — the prep / peel kernels of one bench run (rocprofv3 kernel trace, us) with
# build_ab/<variant>.so in place of the product library, the variants taken in turn on the same box
R=${GRAFT_REPO_ROOT:-/root/repo}; rounds=$1; shift
for r in $(seq $rounds); do for v in "$@"; do cp $R/build_ab/$v.so $R/circom-2-arithc_amd/libc2a_hip.so; KSTAT_CHECK= bash $R/tools/kstat.sh $v > /dev/null 2>&1; python3 - $R/gpurun_out/kstats_$v.csv $v <<PY
import csv,sys
rows={r["Name"]:r for r in csv.DictReader(open(sys.argv[1]))}
keys=["k_producer","k_relabel","k_deps","k_scan_stream<1, c2a::ScanFromU32","k_gstat","k_peel_sinks","k_peel_shallow","k_peel<","k_post_peel","k_root_bits","k_root_list","k_euler_next","k_rank_mark","k_rank_walk","k_rank_jump","k_pos_first","k_pos_bits","k_pos_rank","k_emit_rank","k_emit_split","k_boolify","k_clear"]
out=[]
for k in keys:
    for n,r in rows.items():
        if k in n: out.append("%s %.1f"%(k.split("<")[0] if k!="k_peel<" else "k_peel", float(r["AverageNs"])/1e3)); break
print(sys.argv[2], " | ".join(out))
PY
done; done
