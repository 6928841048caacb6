#!/usr/bin/env python3
"""usage (on the GPU box): python tools/prune_time.py [layers width [bits [reps]]] — times c2a_boolify_prune alone on the
synthetic layered DAG (the first call includes its allocations); run it under rocprofv3 --kernel-trace --stats for the kernels."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
c2a = importlib.import_module("circom-2-arithc_amd")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
width = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 32
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
fg = c2a.synth.layered_dag(layers, width, seed=c2a.synth.SEED)
be = c2a.Backend(0)
be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
be.topo_sort(fetch=False); be.assign_wires(fetch=False); be.emit_gates(fetch=False)
be.boolify(bits)
for _ in range(reps):
    t0 = time.perf_counter()
    pi = be.boolify_prune()
    print("prune s", round(time.perf_counter() - t0, 4), pi["n_gates_before"], "->", pi["n_gates"], flush=True)
