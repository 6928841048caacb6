#!/usr/bin/env python3
"""usage (on the GPU box): tools/peel_series.py [runs] — k_peel / build time of every run on ONE loaded graph (the first is the
run after the node-record clear), checked against the oracle at the end: does the launch alternate with the run tag?"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
c2a = importlib.import_module("circom-2-arithc_amd")
bm = importlib.import_module("circom-2-arithc_amd.backend")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
name = sys.argv[2] if len(sys.argv) > 2 else "synthetic_10m"
fg = c2a.synth.config(name)
be = c2a.Backend(0)
be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
kp, bt = [], []
for _ in range(runs):
    be.build_circuit()
    t = be.timings()
    kp.append(t["k_peel"]); bt.append(t["build_total"])
print("k_peel ms:", " ".join("%.3f" % v for v in kp))
print("build  ms:", " ".join("%.3f" % v for v in bt))
print("odd runs %.3f even runs %.3f (first excluded)" % (np.mean(kp[2::2]), np.mean(kp[1::2])), "rereads", be.stats()["peel_rereads"])
if "--check" in sys.argv:
    from oracle import oracle as orc
    exp = orc.build_circuit(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes, mode=1)
    for nm, arr in (("sorted", exp.sorted), ("in0", exp.in0), ("in1", exp.in1), ("out", exp.out), ("op", exp.op)):
        assert be.checksum(nm) == bm.checksum_host(arr), nm
    print("== oracle")
