import importlib, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
c2a = importlib.import_module("circom-2-arithc_amd")
for layers, width in ((10, 1000000), (5000, 2000)):
    fg = c2a.synth.layered_dag(layers, width, window=4)
    be = c2a.Backend(0)
    be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    be.build_circuit()
    rng = np.random.default_rng(1)
    vec = rng.integers(0, 2**32, (len(fg.input_nodes), 4), dtype=np.uint64)
    be.eval(vec, {}, width=32)
    t0 = time.perf_counter(); be.eval(vec, {}, width=32); dt = time.perf_counter() - t0
    print(layers, "x", width, "arithmetic eval (4 vectors, level lists included)", round(dt * 1e3, 2), "ms", flush=True)
    be.close()
