#!/usr/bin/env python3
"""usage (on the GPU box): python tools/fuzz_boolify.py [minutes [seed]] — random circuits of 1 .. 20 000 gates, random op mixes (every gate type but APow above
width 8) and widths 1 .. 64 through c2a_build_circuit + c2a_boolify: the whole boolean circuit element by element against the oracle's bit-blast (k_boolify picks
its own number of workgroups per chunk: c2a_kernels.h, SLICES), every third one also through the multi-shard context (the same device listed three times)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
c2a = importlib.import_module("circom-2-arithc_amd")
from oracle import oracle as orc  # noqa: E402  (the checker)
S = c2a.synth
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
be1, be3 = c2a.Backend(0), c2a.Backend([0, 0, 0])
t_end, it, gates, slow = time.time() + 60 * minutes, 0, 0, []
while time.time() < t_end:
    width = int(rng.choice([1, 2, 3, 5, 8, 16, 32, 64]))
    names = [n for n in S.OP_NAMES if n != "APow" or width <= 8]
    k = int(rng.integers(1, len(names) + 1))
    mix = tuple((str(n), int(rng.integers(1, 10))) for n in rng.choice(names, size=k, replace=False))
    big = sum(wt for n, wt in mix if n in ("AMul", "ADiv", "AIntDiv", "AMod", "APow")) / sum(wt for _, wt in mix)
    cap = 20_000 if big * width < 2 else (2_000 if width <= 32 else 400)
    layers, wd = int(rng.integers(1, 40)), int(rng.integers(1, 1 + cap // 40))
    fg = S.layered_dag(layers, wd, n_in=int(rng.integers(1, 20)), n_const=int(rng.integers(0, 5)), window=int(rng.integers(1, 6)), mix=mix, seed=int(rng.integers(1 << 30)))
    args = (fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    exp = orc.boolify(orc.build_circuit(*args, mode=1), width)
    for be in ((be1, be3) if it % 3 == 0 else (be1,)):
        be.load_gates(*args)
        be.build_circuit()
        info = be.boolify(width)
        assert info.n_gates == len(exp.in0) and info.wire_count == exp.wire_count, (mix, width, fg.n)
        for g, e, nm in zip(be.bool_read(), (exp.in0, exp.in1, exp.out, exp.op), ("in0", "in1", "out", "op")):
            assert np.array_equal(g, e), (nm, mix, width, fg.n, layers, wd)
    it += 1; gates += len(exp.in0)
be1.close(); be3.close()
print(f"{it} circuits, {gates} boolean gates == oracle in {minutes} min (widths 1..64, all gate types, single context and three shards)")
