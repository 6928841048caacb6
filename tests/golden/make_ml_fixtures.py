#!/usr/bin/env python3
"""Generates tests/golden/ml/*.json — run from the repo root in the BUILD container: `python tests/golden/make_ml_fixtures.py`.

The reference ships 42 larger circuits under tests/circuits/machine-learning/** (no test of its own drives them).  Every file
there that has a live `component main` and that the Circom subset of circom-2-arithc_amd/circom_frontend.py unrolls becomes a
fixture: the call script its unroller makes (`include`s resolved at generation time, from /root/reference), the flat gate list,
what the literal oracle (oracle.CompilerModel.build_circuit) makes of it, and input/output vectors evaluated by the oracle —
for ArgMax also checked against what an arg-max IS.  Every other file is listed in ml/_unsupported.json with the reason (no
main / main commented out upstream / the construct the subset does not read, cf. the reference's own README.md:25-40).
Data only: inputs and expected outputs, never source text.  The tests read the committed JSON; /root/reference is not needed."""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as orc  # noqa: E402
import circom_subset  # noqa: E402

REF = "/root/reference/tests/circuits/machine-learning"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ml")


def resolve(path, seen, order):
    """text of `path` with every `include "x";` replaced by the text of x (once)"""
    path = os.path.normpath(path)
    if path in seen:
        return ""
    seen.add(path)
    order.append(os.path.relpath(path, REF))
    out = []
    for line in open(path).read().split("\n"):
        m = re.match(r'\s*include\s+"([^"]+)"\s*;', line)
        out.append(resolve(os.path.join(os.path.dirname(path), m.group(1)), seen, order) if m else line)
    return "\n".join(out)


def model_of(d):
    m = orc.CompilerModel()
    for st in d["script"]:
        if st[0] == "signal":
            m.add_signal(st[1], st[2], st[3])
        elif st[0] == "gate":
            m.add_gate(orc.OP[st[1]], st[2], st[3], st[4])
        else:
            m.add_connection(st[1], st[2])
    for p in d["input_prefixes"]:
        m.add_inputs(m.get_signals(f"0.{p}"))
    for p in d["output_prefixes"]:
        m.add_outputs(m.get_signals(f"0.{p}"))
    return m


def evaluate(circ, inputs_by_name, width=32):
    """the literal circuit on named inputs (oracle evaluator: tests/integration.rs:94-115 mod 2^w) -> named outputs"""
    n = len(circ.gates)
    a = orc.ArithCircuit(sorted=np.empty(0, np.uint32), in0=np.array([g[0] for g in circ.gates], np.uint32),
                         in1=np.array([g[1] for g in circ.gates], np.uint32), out=np.array([g[2] for g in circ.gates], np.uint32),
                         op=np.array([orc.OP[g[3]] for g in circ.gates], np.uint8), node_wire=np.empty(0, np.uint32),
                         wire_count=circ.wire_count, n_in=len(circ.input_name_to_wire_index), n_out=len(circ.output_name_to_wire_index))
    wires = np.zeros((circ.wire_count, 1), np.uint64)
    for k, w in circ.input_name_to_wire_index.items():
        wires[w, 0] = inputs_by_name.get(k, 0)
    for c in circ.constants.values():
        wires[c.wire_index, 0] = int(c.value)
    if n:
        orc.eval_arith(a, width, wires)
    return {k: int(wires[w, 0]) for k, w in circ.output_name_to_wire_index.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    unsupported = {}
    made = []
    rng = np.random.default_rng(20241008)
    for root, _, files in sorted(os.walk(REF)):
        for f in sorted(files):
            if not f.endswith(".circom"):
                continue
            path = os.path.join(root, f)
            rel = os.path.relpath(path, REF)
            raw = open(path).read()
            live = re.search(r"^\s*component\s+main\b", raw, re.M)
            if not live:
                unsupported[rel] = ("main component commented out upstream" if re.search(r"//\s*component\s+main\b", raw)
                                    else "a library file: no main component")
                continue
            order = []
            text = resolve(path, set(), order)
            try:
                d = circom_subset.unroll(text)
            except circom_subset.ProgramError as e:
                why = str(e)
                if "\\" in re.sub(r"//.*", "", raw.split("component main")[0]) and "arsing" in why:
                    why += " (integer division `\\` inside an array dimension is not in the subset)"
                unsupported[rel] = "front-end: " + why
                continue
            m = model_of(d)
            circ = m.build_circuit()
            pay = m.flat_payload()
            name = os.path.splitext(f)[0]
            cases = []
            in_names = list(circ.input_name_to_wire_index)
            for k in range(6):
                vals = {nm: int(v) for nm, v in zip(in_names, rng.integers(0, 50 if k < 3 else 2 ** 16, len(in_names)))}
                cases.append({"inputs": vals, "outputs": evaluate(circ, vals)})
            if name == "ArgMax":          # what the circuit is FOR: the index of the maximum (the first one on a tie)
                for c in cases:
                    xs = [c["inputs"][f"0.in[{i}]"] for i in range(5)]
                    assert c["outputs"]["0.out"] == int(np.argmax(xs)), (xs, c["outputs"])
            fx = {"name": name, "source": "tests/circuits/machine-learning/" + rel, "includes_resolved": order[1:],
                  "derived_by": "tests/golden/make_ml_fixtures.py (circom_frontend subset + oracle.CompilerModel)",
                  "script": d["script"], "input_prefixes": d["input_prefixes"], "output_prefixes": d["output_prefixes"],
                  "gates": [[orc.OP_NAMES[g.op], g.lh_in, g.rh_in, g.out] for g in m.gates],
                  "n_nodes": pay["n_nodes"], "input_nodes": pay["input_nodes"].tolist(), "output_nodes": pay["output_nodes"].tolist(),
                  "expect": {"wire_count": circ.wire_count, "sorted": circ.sorted_gate_ids,
                             "sorted_is_identity": circ.sorted_gate_ids == list(range(len(circ.sorted_gate_ids))),
                             "emitted": [[g[0], g[1], g[2], g[3]] for g in circ.gates],
                             "input_name_to_wire_index": circ.input_name_to_wire_index,
                             "output_name_to_wire_index": circ.output_name_to_wire_index,
                             "constants": {k: [c.value, c.wire_index] for k, c in circ.constants.items()},
                             "io_cases": cases, "io_cases_from": "oracle evaluator (ArgMax: also asserted == numpy argmax)"}}
            with open(os.path.join(OUT, f"{name}.json"), "w") as fo:
                json.dump(fx, fo, indent=None, separators=(",", ":"))
            made.append((name, len(m.gates), fx["expect"]["sorted_is_identity"]))
    with open(os.path.join(OUT, "_unsupported.json"), "w") as fo:
        json.dump(unsupported, fo, indent=1, sort_keys=True)
    print("fixtures:", made)
    print("not unrolled:", len(unsupported))


if __name__ == "__main__":
    main()
