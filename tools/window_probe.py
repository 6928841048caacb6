#!/usr/bin/env python3
"""usage (on the GPU box): python tools/window_probe.py [window [layers [width]]] — layered_dag with 20 inputs and the rh operand out of the last
`window` layers (1 = strict layers: DESIGN.md 4.2) through c2a_build_circuit twice: k_peel / peel / build ms and the record re-reads of the warm run.
Knobs of the launch come from the environment (C2A_PEEL_WAVES, C2A_PEEL_FIFOS, C2A_PEEL_RESERVE, C2A_PEEL_STATS)."""
import importlib, sys
sys.path.insert(0, "/root/repo")
c2a = importlib.import_module("circom-2-arithc_amd")
w = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
fg = c2a.synth.layered_dag(L, W, n_in=20, n_const=4, window=w, seed=5)
be = c2a.Backend(0)
be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
for _ in range(2): be.build_circuit()
print("window", w, L, "x", W, {k: round(v, 3) for k, v in be.timings().items() if k in ("k_peel", "peel", "build_total")}, be.stats()["peel_rereads"])
