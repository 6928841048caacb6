#!/usr/bin/env python3
"""usage (on the GPU box): python tools/fuzz_gpu.py [minutes] — random layered graphs of random shapes / op mixes / constant and
output densities (incl. duplicate-free adversarial sizes around the tile boundaries) through c2a_build_circuit on the MI355X, every
result array against the oracle; both numbering paths by turns; a fresh graph per iteration on ONE context (buffers reused)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
c2a = importlib.import_module("circom-2-arithc_amd")
bm = importlib.import_module("circom-2-arithc_amd.backend")
from oracle import oracle as orc
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
be = c2a.Backend(0)
t_end = time.time() + 60 * minutes
it = 0
while time.time() < t_end:
    layers = int(rng.choice([1, 2, 3, 7, 40, 200, 1000, 6000]))
    width = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 2048, 4097, 20000]))
    if layers * width > 3_000_000:
        continue
    cf, of = float(rng.choice([0, 0, 0.05, 0.3])), float(rng.choice([0, 0, 0.05, 0.5]))
    mix = [c2a.synth.MIX_BITWISE, c2a.synth.MIX_ALL, c2a.synth.MIX_SHA][int(rng.integers(3))]
    fg = c2a.synth.layered_dag(layers, width, n_in=int(rng.integers(1, 50)), n_const=int(rng.integers(0, 9)), window=int(rng.integers(1, 70)),
                               mix=mix, seed=int(rng.integers(1 << 30)), const_frac=cf, out_frac=of, permute=bool(rng.integers(2)))
    args = (fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    exp = orc.build_circuit(*args, mode=1)
    be.load_gates(*args)
    for rep in range(2):
        assert be.build_circuit() == exp.wire_count, (layers, width)
        for nm, arr in (("sorted", exp.sorted), ("in0", exp.in0), ("in1", exp.in1), ("out", exp.out), ("op", exp.op)):
            assert be.checksum(nm) == bm.checksum_host(arr), (nm, layers, width, cf, of, rep)
        nw1 = ((exp.node_wire.astype(np.uint64) + 1) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        assert be.checksum("node_wire1") == bm.checksum_host(nw1), ("node_wire", layers, width, cf, of, rep)
    if it % 7 == 0:
        np.testing.assert_array_equal(be.topo_sort(), exp.sorted)
        nw, wc = be.assign_wires()
        np.testing.assert_array_equal(nw, exp.node_wire)
    it += 1
print(f"{it} graphs == oracle ({minutes} min)")
