#!/usr/bin/env python3
"""Generates tests/golden/nonidentity/*.json — run from the repo root in the BUILD container:
`python tests/golden/make_nonidentity_fixtures.py`.

Every circuit the reference ships that the front-end's subset unrolls sorts to the IDENTITY (the gate list is in
dependency order unless a component's body is appended before the expressions that feed its inputs are evaluated).  The mains
under tests/golden/circuits/ are this repo's own text; they instantiate LIBRARY templates the reference ships —
circomlib/switcher.circom, circomlib/mux3.circom, circomlib-matrix/matMul.circom (+ matElemMul / matElemSum), which have no
main of their own — in exactly that order, so that topological_sort.rs has real work to do on circuits derived from Circom
text: 31 / 149 / 109 gates.  `include`s are resolved at generation time from /root/reference; the committed fixture is data
only (the call script the unroller makes, the flat gate list, what the literal oracle makes of it, IO vectors evaluated by the
oracle AND checked against what the circuit is for: a sorting network sorts, a multiplexer selects, a matrix product is numpy's)."""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import oracle as orc  # noqa: E402
import circom_subset  # noqa: E402
from make_ml_fixtures import evaluate, model_of  # noqa: E402

REF = "/root/reference/tests/circuits/machine-learning"
OUT = os.path.join(HERE, "nonidentity")
M32 = (1 << 32) - 1


def resolve(path, seen, order):
    path = os.path.normpath(path)
    if path in seen:
        return ""
    seen.add(path)
    order.append(path)
    out = []
    for line in open(path).read().split("\n"):
        m = re.match(r'\s*include\s+"([^"]+)"\s*;', line)
        if m:
            cand = os.path.join(os.path.dirname(path), m.group(1))
            if not os.path.exists(cand):
                cand = os.path.join(REF, m.group(1))
            out.append(resolve(cand, seen, order))
        else:
            out.append(line)
    return "\n".join(out)


def semantic(name, ins, outs):
    """what the circuit is FOR, mod 2^32 (inputs are kept small enough that nothing wraps unless it is meant to)"""
    if name == "switcherNet":
        xs = sorted(ins[f"0.in[{i}]"] for i in range(4))
        assert [outs[f"0.out[{i}]"] for i in range(4)] == xs and outs["0.lo"] == xs[0], (ins, outs)
    elif name == "mux3Select":
        x, y, t = ins["0.x"], ins["0.y"], [ins[f"0.t[{j}]"] for j in range(3)]
        im = sum((1 << j) for j in range(3) if t[j] > 10)
        iw = sum((1 << j) for j in range(3) if t[j] < 7)
        m_out, w0, w1 = (x * im + y) & M32, (y * iw + x) & M32, (x + iw) & M32
        assert outs["0.out"] == (m_out + w0) & M32 and outs["0.other"] == (w1 * m_out) & M32, (ins, outs)
    elif name == "matMulChain":
        n = 3
        A = np.array([[ins[f"0.a[{i}][{j}]"] for j in range(n)] for i in range(n)], dtype=object)
        B = np.array([[ins[f"0.b[{i}][{j}]"] for j in range(n)] for i in range(n)], dtype=object)
        Q = ((A + B).dot(B)).dot(A - B)
        for i in range(n):
            for j in range(n):
                assert outs[f"0.out[{i}][{j}]"] == int(Q[i][j]) & M32, (i, j)
        assert outs["0.tr"] == int(Q[0][0] + Q[n - 1][n - 1]) & M32


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20241008)
    made = []
    for name in ("switcherNet", "mux3Select", "matMulChain"):
        order = []
        text = resolve(os.path.join(HERE, "circuits", f"{name}.circom"), set(), order)
        d = circom_subset.unroll(text)
        m = model_of(d)
        circ = m.build_circuit()
        pay = m.flat_payload()
        in_names = list(circ.input_name_to_wire_index)
        cases = []
        for k in range(8):
            hi = 20 if k < 4 else 1000
            vals = {nm: int(v) for nm, v in zip(in_names, rng.integers(0, hi, len(in_names)))}
            outs = evaluate(circ, vals)
            semantic(name, vals, outs)
            cases.append({"inputs": vals, "outputs": outs})
        ident = circ.sorted_gate_ids == list(range(len(circ.sorted_gate_ids)))
        assert not ident, name
        fx = {"name": name, "source": f"tests/golden/circuits/{name}.circom (this repo's main) over the reference's library templates",
              "includes_resolved": [os.path.relpath(p, REF) for p in order[1:]],
              "derived_by": "tests/golden/make_nonidentity_fixtures.py (circom_frontend subset + oracle.CompilerModel)",
              "script": d["script"], "input_prefixes": d["input_prefixes"], "output_prefixes": d["output_prefixes"],
              "gates": [[orc.OP_NAMES[g.op], g.lh_in, g.rh_in, g.out] for g in m.gates],
              "n_nodes": pay["n_nodes"], "input_nodes": pay["input_nodes"].tolist(), "output_nodes": pay["output_nodes"].tolist(),
              "expect": {"wire_count": circ.wire_count, "sorted": circ.sorted_gate_ids, "sorted_is_identity": ident,
                         "emitted": [[g[0], g[1], g[2], g[3]] for g in circ.gates],
                         "input_name_to_wire_index": circ.input_name_to_wire_index,
                         "output_name_to_wire_index": circ.output_name_to_wire_index,
                         "constants": {k: [c.value, c.wire_index] for k, c in circ.constants.items()},
                         "io_cases": cases,
                         "io_cases_from": "oracle evaluator, each case asserted against what the circuit is for (sorted inputs / selected table entry / numpy matrix product)"}}
        with open(os.path.join(OUT, f"{name}.json"), "w") as fo:
            json.dump(fx, fo, indent=None, separators=(",", ":"))
        moved = sum(1 for i, g in enumerate(circ.sorted_gate_ids) if i != g)
        made.append((name, len(m.gates), f"{moved} of {len(m.gates)} positions differ from the list order"))
    print("fixtures:", made)


if __name__ == "__main__":
    main()
