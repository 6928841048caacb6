#!/usr/bin/env python3
"""usage (on the GPU box): python tools/shape_probe.py — which property of a layered graph the sort's time follows: gate ids permuted or not, window 1 or 64,
fresh constants, extra outputs (1000 x 2048 gates each), k_peel / peel / build ms of the third build."""
import importlib, sys, itertools
sys.path.insert(0, "/root/repo")
c2a = importlib.import_module("circom-2-arithc_amd")
S = c2a.synth
be = c2a.Backend(0)
for permute, window, cf, of in itertools.product((True, False), (1, 64), (0.0, 0.3), (0.0, 0.5)):
    fg = S.layered_dag(1000, 2048, n_in=20, n_const=4, window=window, seed=5, const_frac=cf, out_frac=of, permute=permute)
    be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    for _ in range(3): be.build_circuit()
    t, st = be.timings(), be.stats()
    print("permute", permute, "window", window, "const_frac", cf, "out_frac", of, "| k_peel", round(t["k_peel"], 3), "peel", round(t["peel"], 3), "build", round(t["build_total"], 3), "| roots", st["n_roots"], "depth", st["max_depth"], "rereads", st["peel_rereads"], "relays", st["n_relays"], flush=True)
