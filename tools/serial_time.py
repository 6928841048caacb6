#!/usr/bin/env python3
"""usage (on the GPU box): python tools/serial_time.py [n ...] — what c2a_topo_sort_serial (the reference's DFS on ONE lane: the sort that
cannot fail, and what c2a_topo_sort falls back to when the dataflow launch gives up twice) costs on the headline shape, against the
dataflow launch and the CPU oracle on the same graph; results compared."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
c2a = importlib.import_module("circom-2-arithc_amd")
from oracle import oracle as orc
for n in [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000]:
    fg = c2a.synth.layered_dag(max(1, n // 2000), 2000)
    args = (fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    t0 = time.perf_counter(); exp = orc.build_circuit(*args, mode=1); t_cpu = time.perf_counter() - t0
    with c2a.Backend(0) as be:
        be.load_gates(*args)
        be.topo_sort(fetch=False)
        t0 = time.perf_counter(); got = be.topo_sort(); t_par = time.perf_counter() - t0
        assert np.array_equal(got, exp.sorted)
        t0 = time.perf_counter(); got = be.topo_sort(serial=True); t_ser = time.perf_counter() - t0
        assert np.array_equal(got, exp.sorted)
        # ... and as the fall-back of c2a_topo_sort (two launches given up first)
        be.debug_peel_abort(2)
        t0 = time.perf_counter(); got = be.topo_sort(); t_fb = time.perf_counter() - t0
        assert np.array_equal(got, exp.sorted)
    print(f"n {fg.n}: c2a_topo_sort (dataflow launch, D2H of the order included) {t_par * 1e3:.1f} ms | c2a_topo_sort_serial (one lane) {t_ser * 1e3:.0f} ms = {t_ser / fg.n * 1e9:.0f} ns per gate | "
          f"c2a_topo_sort through the fall-back (two launches given up, then the serial DFS + levels) {t_fb * 1e3:.0f} ms | CPU oracle build_circuit (flat) {t_cpu * 1e3:.0f} ms; all == oracle", flush=True)
