"""Driver entry points: build() compiles every native piece, smoke() runs the hot path once on cuda:0."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build() -> None:
    """hipcc --offload-arch=gfx950 for the product library (cross-compiles without a GPU), gcc for the CPU
    oracle (the checker; building it is not using it), then import the package."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "circom-2-arithc_amd", "csrc")])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    pkg = importlib.import_module("circom-2-arithc_amd")
    lib = pkg.load_library()          # dlopen + resolve every symbol of include/c2a.h (no GPU call)
    assert lib.c2a_version().decode().startswith("c2a")
    # /root/reference is Rust with un-vendored git dependencies: no oracle/_ref can be built.


def smoke() -> None:
    """One small invocation of the hot path on GPU 0 (sort + wire numbering + emission + boolify at w=32),
    checked bit-for-bit against the CPU oracle."""
    import numpy as np
    pkg = importlib.import_module("circom-2-arithc_amd")
    from oracle import oracle as orc
    fg = pkg.synth.layered_dag(64, 128, n_in=64, n_const=8, window=16, seed=1)
    with pkg.Backend(0) as be:
        assert "hip" in be.version, be.version
        be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
        sorted_ids = be.topo_sort()
        node_wire, wire_count = be.assign_wires()
        in0, in1, out, op = be.emit_gates()
        info = be.boolify(32)
        b = be.bool_read()
        exp = orc.build_circuit(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes, mode=1)
        eb = orc.boolify(exp, 32)
        assert np.array_equal(sorted_ids, exp.sorted), "sorted_gate_ids differ from the oracle"
        assert wire_count == exp.wire_count and np.array_equal(node_wire, exp.node_wire)
        assert np.array_equal(in0, exp.in0) and np.array_equal(in1, exp.in1) and np.array_equal(out, exp.out)
        assert np.array_equal(op, exp.op)
        assert info.n_gates == len(eb.in0) and info.wire_count == eb.wire_count
        assert all(np.array_equal(x, y) for x, y in zip(b, (eb.in0, eb.in1, eb.out, eb.op)))
        print(f"smoke ok: {fg.n} gates -> {info.n_gates} boolean gates, stats={be.stats()} timings={be.timings()}")


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
