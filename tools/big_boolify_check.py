#!/usr/bin/env python3
"""usage (on the GPU box): python tools/big_boolify_check.py — c2a_boolify at the top of its range: a 100 x 100 matrix product at width 32 (2.98 G boolean
gates, 39 GB: totals, a slice against the oracle, every wire simulated) and a 130 x 130 one (6.5 G boolean gates: refused, boolean wire ids are u32)."""
import importlib, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
c2a = importlib.import_module("circom-2-arithc_amd")
from oracle import oracle as orc
for m in (100, 130):
    fg = c2a.synth.matmul(m)
    args = (fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    exp = orc.build_circuit(*args, mode=1)
    be = c2a.Backend(0)
    try:
        be.load_gates(*args)
        assert be.build_circuit() == exp.wire_count
        T = np.array([orc.template_size(o, 32)[0] for o in range(20)], dtype=np.int64)
        want = int(T[exp.op].sum())
        print("matmul", m, "gates", fg.n, "boolean gates wanted", want, flush=True)
        try:
            t0 = time.time(); info = be.boolify(32); dt = time.time() - t0
            print("  boolify ok:", info.n_gates, "gates", info.wire_count, "wires in", round(dt * 1e3, 1), "ms;", {k: round(v, 3) for k, v in be.timings().items() if k.startswith("bool")})
            assert info.n_gates == want
            sl, g0 = orc.boolify_range(exp, 32, fg.n // 2, 2000)
            got = be.bool_read(g0, len(sl.in0))
            for a, b in zip(got, (sl.in0, sl.in1, sl.out, sl.op)):
                np.testing.assert_array_equal(a, b)
            checked, bad = be.verify_boolify(seed=3)
            print("  slice == oracle; verify:", checked, bad)
        except Exception as e:
            print("  boolify refused:", type(e).__name__, str(e)[:300])
    finally:
        be.close()
