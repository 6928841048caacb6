#!/usr/bin/env python3
"""What the stages around the peel would cost if gate ids carried the creation order (no relabel needed):
the headline graph with and without the id permutation, stage times from the library's own events.
(With permute=False the DFS order is the identity, so only prep / wires / emit / bool_prep compare.)
usage (on the GPU box): python tools/locality_ceiling.py [layers] [layer_width]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
c2a = importlib.import_module("circom-2-arithc_amd")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
for permute in (True, False):
    fg = c2a.synth.layered_dag(L, W, seed=c2a.synth.SEED, permute=permute)
    with c2a.Backend(0) as be:
        be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
        acc = {}
        for it in range(4):
            be.build_circuit(); be.boolify(32)
            if it:
                for k, v in be.timings().items():
                    acc[k] = acc.get(k, 0.0) + v / 3
        print(json.dumps({"permute": permute, "stages_ms": {k: round(v, 3) for k, v in acc.items()}}))
