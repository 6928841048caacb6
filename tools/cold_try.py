import sys, importlib, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
c2a = importlib.import_module("circom-2-arithc_amd")
fg = c2a.synth.config("synthetic_10m")
for k in range(6):
    t0 = time.perf_counter(); be = c2a.Backend(0); t1 = time.perf_counter()
    be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes); t2 = time.perf_counter()
    be.build_circuit(); t3 = time.perf_counter(); tb = be.timings(); be.boolify(32); t4 = time.perf_counter(); tq = be.timings()
    print("first build", {k: round(v, 2) for k, v in tb.items() if k in ("prep", "peel", "order", "wires", "emit", "build_total", "k_peel")}, "first boolify", {k: round(v, 2) for k, v in tq.items() if k.startswith("bool")})
    be.build_circuit(); t5 = time.perf_counter(); be.boolify(32); t6 = time.perf_counter()
    print("create %.1f load %.1f build %.1f boolify %.1f | 2nd build %.1f boolify %.1f" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)), be.timings())
    be.close()
