#!/usr/bin/env python3
"""usage (on the GPU box): python tools/node_perm_check.py [layers width] — the headline graph with its NODE ids permuted as well as
its gate ids: the relabelling by out-node order (c2a_kernels.h RELABELLING) then finds no locality at all.  Must stay exact
(checksums of sorted ids / emitted circuit / node -> wire against the oracle) and says what it costs."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
c2a = importlib.import_module("circom-2-arithc_amd")
bm = importlib.import_module("circom-2-arithc_amd.backend")
from oracle import oracle as orc
L = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
fg = c2a.synth.layered_dag(L, W, seed=c2a.synth.SEED)
for name, perm in (("node ids in creation order (the headline input)", None),
                   ("node ids permuted", np.random.default_rng(1).permutation(fg.n_nodes).astype(np.uint32))):
    lh, rh, out = (fg.lh, fg.rh, fg.out) if perm is None else (perm[fg.lh], perm[fg.rh], perm[fg.out])
    ins, outs = (fg.input_nodes, fg.output_nodes) if perm is None else (perm[fg.input_nodes], perm[fg.output_nodes])
    circ = orc.build_circuit(lh, rh, out, fg.op, fg.n_nodes, ins, outs, mode=1)
    with c2a.Backend(0) as be:
        be.load_gates(lh, rh, out, fg.op, fg.n_nodes, ins, outs)
        acc = {}
        for it in range(4):
            be.build_circuit(); be.boolify(32)
            if it:
                for k, v in be.timings().items():
                    acc[k] = acc.get(k, 0.0) + v / 3
        for nm, arr in (("sorted", circ.sorted), ("in0", circ.in0), ("in1", circ.in1), ("out", circ.out), ("op", circ.op)):
            assert be.checksum(nm) == bm.checksum_host(arr), nm
        nw1 = ((circ.node_wire.astype(np.uint64) + 1) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        assert be.checksum("node_wire1") == bm.checksum_host(nw1)
        print(f"{name}: == oracle | " + " ".join(f"{k} {v:.2f}" for k, v in acc.items()))
