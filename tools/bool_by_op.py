#!/usr/bin/env python3
"""usage (on the GPU box): python tools/bool_by_op.py [width ...] — k_boolify's rate per gate type: a 1 000 x 1 000 layered graph of ONE arithmetic
op (scaled down for the big templates), bool_map ms of the third call, bytes written (13 per boolean gate) per second"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
c2a = importlib.import_module("circom-2-arithc_amd")
S = c2a.synth
widths = [int(a) for a in sys.argv[1:]] or [32, 64]
with c2a.Backend(0) as be:
    for w in widths:
        for op in ("AXor", "ABitOr", "AAdd", "AEq", "ALt", "AShiftR", "AMul", "ADiv", "AMod"):
            big = op in ("AMul", "ADiv", "AMod")
            fg = S.layered_dag(1000, 100 if big else 1000, n_in=64, n_const=4, window=8, mix=((op, 1),), seed=11)
            be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
            be.build_circuit()
            for _ in range(3):
                info = be.boolify(w)
            t = be.timings()
            gb = info.n_gates * 13 / 1e9
            print(f"width {w:2d} {op:8s} {fg.n:8d} gates -> {info.n_gates:11d} boolean ({info.n_gates // fg.n:6d} each) {gb:7.2f} GB | bool_map {t['bool_map']:7.3f} ms = {gb / t['bool_map']:5.2f} TB/s | prep {t['bool_prep']:.3f}", flush=True)
