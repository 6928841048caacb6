#!/usr/bin/env python3
"""usage (on the GPU box): python tools/bool_ramp.py — does k_boolify run slower right behind the latency-bound build (a clock / power
state left low by 7 ms of one-wave workgroups) than right behind another k_boolify?  bool_map of: build + boolify; boolify again."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
c2a = importlib.import_module("circom-2-arithc_amd")
fg = c2a.synth.layered_dag(5000, 2000, seed=c2a.synth.SEED)
with c2a.Backend(0) as be:
    be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    for rep in range(4):
        be.build_circuit(); be.boolify(32); a = be.timings()["bool_map"]
        be.boolify(32); b = be.timings()["bool_map"]
        be.boolify(32); c = be.timings()["bool_map"]
        print(f"behind the build {a:.3f} ms | behind a boolify {b:.3f} | again {c:.3f}")
