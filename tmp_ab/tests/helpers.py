"""Test helpers: circuit simulation through the oracle's evaluators (semantics of
/root/reference/tests/integration.rs:94-115 taken mod 2^w) and fixture loading."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixtures():
    out = {}
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.json"))):
        with open(p) as f:
            fx = json.load(f)
        out[fx["name"]] = fx
    return out


def fixture_payload(fx, orc):
    g = fx["gates"]
    return dict(lh=np.array([x[1] for x in g], np.uint32), rh=np.array([x[2] for x in g], np.uint32),
                out=np.array([x[3] for x in g], np.uint32), op=np.array([orc.OP[x[0]] for x in g], np.uint8),
                n_nodes=fx["n_nodes"], input_nodes=np.array(fx["input_nodes"], np.uint32),
                output_nodes=np.array(fx["output_nodes"], np.uint32))


def simulate_arith(orc, in0, in1, out, op, wire_count, n_in, n_out, input_wires, const_wires, width=32):
    """input_wires / const_wires: {wire: value}.  Returns the value of every wire (uint64 array)."""
    circ = orc.ArithCircuit(sorted=np.empty(0, np.uint32), in0=in0, in1=in1, out=out, op=op,
                            node_wire=np.empty(0, np.uint32), wire_count=wire_count, n_in=n_in, n_out=n_out)
    wires = np.zeros((wire_count, 1), dtype=np.uint64)
    for w, v in list(input_wires.items()) + list(const_wires.items()):
        wires[w, 0] = v
    orc.eval_arith(circ, width, wires)
    return wires[:, 0]


def simulate_bool(orc, b_in0, b_in1, b_out, b_op, bool_wire_count, width, bit_wire_of, input_wires, const_wires):
    """bit_wire_of(W, bit) -> boolean wire.  input_wires/const_wires are ARITHMETIC {wire: value}.
    Returns a function value_of(W) reading the w result bits back."""
    bc = orc.BoolCircuit(in0=b_in0, in1=b_in1, out=b_out, op=b_op, wire_count=bool_wire_count, width=width,
                         n_in=0, n_out=0)
    wires = np.zeros(bool_wire_count, dtype=np.uint64)
    for W, v in list(input_wires.items()) + list(const_wires.items()):
        for b in range(width):
            wires[int(bit_wire_of(W, b))] = np.uint64(0xFFFFFFFFFFFFFFFF) if (int(v) >> b) & 1 else np.uint64(0)
    orc.eval_bool(bc, wires)

    def value_of(W):
        v = 0
        for b in range(width):
            v |= (int(wires[int(bit_wire_of(W, b))]) & 1) << b
        return v
    return value_of
