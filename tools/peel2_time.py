#!/usr/bin/env python3
"""usage (on the GPU box): [C2A_PEEL2_DBG=1] [knobs...] python tools/peel2_time.py [layers width [reps]] — times the peel launch alone
(C2A_PEEL_STATS prints the kernel time from HIP events); with C2A_PEEL2_DBG the run stops after the launch (no results)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("C2A_PEEL_STATS", "1")
c2a = importlib.import_module("circom-2-arithc_amd")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
width = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
fg = c2a.synth.layered_dag(layers, width, seed=c2a.synth.SEED)
be = c2a.Backend(0)
be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
for _ in range(reps):
    try:
        be.topo_sort(fetch=False)
        print("peel stage ms", round(be.timings()["peel"], 3), flush=True)
    except c2a.BackendError as e:
        print("stopped:", str(e)[:80], flush=True)
