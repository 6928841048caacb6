#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) rocpd sqlite result — `rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`
writes DIR/NAME_results.db — into the per-kernel table rocprofv3's CSV stats would hold.
usage: rocpd_summary.py results.db [> profiles/xyz.txt]"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"# {'kernel':<64} {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6} "
          f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>6} {'scratch':>7} {'max_grid':>10} {'wg':>5}")
    for r in rows:
        print(f"  {r[0][:64]:<64} {r[1]:>7} {r[2] / 1e3:>12.1f} {r[3] / 1e3:>10.2f} {r[4] / 1e3:>10.2f} {r[5] / 1e3:>10.2f} "
              f"{100.0 * r[2] / total:>6.2f} {r[6]:>5} {r[7]:>5} {r[8]:>5} {r[9]:>6} {r[10]:>7} {r[11]:>10} {r[12]:>5}")
    try:
        pmc = cur.execute("select name, count(*) from pmc_events group by name").fetchall()
        if pmc:
            print("# pmc events:", pmc)
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1])
