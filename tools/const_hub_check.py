import importlib, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
c2a = importlib.import_module("circom-2-arithc_amd")
from oracle import oracle as orc
bm = importlib.import_module("circom-2-arithc_amd.backend")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
lib = sys.argv[3] if len(sys.argv) > 3 else None
fg = c2a.synth.layered_dag(max(2, n // 2000), 2000)
rng = np.random.default_rng(1)
m = rng.random(fg.n) < frac
rh = fg.rh.copy(); rh[m] = fg.const_nodes[0]
lh = fg.lh.copy(); m2 = rng.random(fg.n) < frac / 10; lh[m2] = fg.const_nodes[1]
args = (lh, rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
exp = orc.build_circuit(*args, mode=1)
for walk in (0, 1):
    import os
    if walk: os.environ["C2A_NUMBERING_WALK"] = "1"
    be = c2a.Backend(0, lib_path=lib)
    os.environ.pop("C2A_NUMBERING_WALK", None)
    be.load_gates(*args)
    for rep in range(3):
        assert be.build_circuit() == exp.wire_count
    for nm, arr in (("sorted", exp.sorted), ("in0", exp.in0), ("in1", exp.in1), ("out", exp.out)):
        assert be.checksum(nm) == bm.checksum_host(arr), nm
    t = be.timings()
    print("walk" if walk else "positional", "readers of one constant node:", int(m.sum()), "| build", round(t["build_total"], 3), "ms: wires", round(t["wires"], 3), "emit", round(t["emit"], 3), "prep", round(t["prep"], 3), "== oracle", flush=True)
    be.close()
