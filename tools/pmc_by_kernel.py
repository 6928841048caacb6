#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc run (counter_collection csv) per kernel: launches, mean counter value per launch."""
import csv, glob, sys, collections, json

def main(d):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0]
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[k].add(row["Dispatch_Id"])
    out = {}
    for k, cs in agg.items():
        n = len(launches[k])
        out[k] = {"launches": n, **{c: v / n for c, v in cs.items()}}
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main(sys.argv[1])
