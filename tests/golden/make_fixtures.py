#!/usr/bin/env python3
"""Generates tests/golden/*.json — run from the repo root: `python tests/golden/make_fixtures.py`.

What a fixture is
-----------------
The reference cannot be built in this image (Rust + un-vendored git deps), so there is nothing to run to
produce vectors.  The flat gate list of every circuit is DERIVED MECHANICALLY from the circuit's .circom text
(read from /root/reference at generation time — this script runs in the build container only; the tests read the
committed JSON) by tests/golden/circom_subset.py: a Circom-subset parser plus a call-for-call restatement of the
reference's unroller (program.rs / process.rs / runtime.rs).  The hand traces of SURVEY.md Appendix A are kept as
"hand" expectations: machine derivation and hand trace agree on every circuit they share.
What the reference's own tests DO pin for the flat-gate-graph path is in
/root/reference/tests/integration.rs:
    :279-374  five functional input->output tables (addZero, infixOps, matElemMul, sum, xEqX)
    :393-415  constantSum: constants == {"0.const_signal_8_1": {value:"8", wire_index:0}}
    :417-441  directOutput: output map {"0.out":0}, constant "0.const_signal_42_1" at wire 0
Those expectations are copied below as DATA ("expect_*").  The flat gate list each circuit produces is
obtained by replaying — through the literal Python restatement of add_signal / add_gate / add_connection
(oracle.CompilerModel, src/compiler.rs:139-278) — the call sequence that src/process.rs performs for the
circuit (declarations in order: process.rs:53-101; `lhs <== a op b`: process.rs:461-475 then :266-269;
literals become named constant signals: process.rs:558-579; variables consume no signal ids:
runtime.rs:205-217).  (That description of the call sequence is what circom_subset.py implements.)
ArgMax(2) (the shipped input/circuit.circom, BASELINE config C1) is cross-checked against the table of SURVEY A.5.

Each JSON holds: the replayed Compiler state (signals, inputs, outputs, the call script), the flat payload
that crosses the C ABI, the reference-test expectations, and the hand-traced expectations.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import circom_subset  # noqa: E402

REF = "/root/reference"

OUT_DIR = os.path.dirname(os.path.abspath(__file__))


def replay(script):
    m = orc.CompilerModel()
    for step in script:
        kind = step[0]
        if kind == "signal":
            m.add_signal(step[1], step[2], step[3])
        elif kind == "gate":
            m.add_gate(orc.OP[step[1]], step[2], step[3], step[4])
        elif kind == "connect":
            m.add_connection(step[1], step[2])
        else:
            raise ValueError(kind)
    return m


def finish(name, script, input_prefixes, output_prefixes, expect, source):
    m = replay(script)
    for p in input_prefixes:                      # program.rs:57-60 (prefix filter)
        m.add_inputs(m.get_signals(f"0.{p}"))
    for p in output_prefixes:                     # program.rs:62-66
        m.add_outputs(m.get_signals(f"0.{p}"))
    try:
        pay = m.flat_payload()
    except orc.Inconsistency as e:                # the name-level checks of compiler.rs:323-383 already fail (prefixOps)
        fx = {"name": name, "source": source, "hand_traced": False,
              "derived_by": "tests/golden/circom_subset.py from the .circom text", "script": script,
              "input_prefixes": input_prefixes, "output_prefixes": output_prefixes,
              "gates": [[orc.OP_NAMES[g.op], g.lh_in, g.rh_in, g.out] for g in m.gates],
              "build_circuit_error": str(e), "expect": expect}
        with open(os.path.join(OUT_DIR, f"{name}.json"), "w") as f:
            json.dump(fx, f, indent=1)
        return fx
    fx = {
        "name": name,
        "source": source,
        "hand_traced": False,
        "derived_by": "tests/golden/circom_subset.py from the .circom text",
        "script": script,
        "input_prefixes": input_prefixes,
        "output_prefixes": output_prefixes,
        "gates": [[orc.OP_NAMES[g.op], g.lh_in, g.rh_in, g.out] for g in m.gates],
        "n_nodes": pay["n_nodes"],
        "input_names": pay["input_names"], "input_nodes": pay["input_nodes"].tolist(),
        "output_names": pay["output_names"], "output_nodes": pay["output_nodes"].tolist(),
        "constants": {k: [v[0], v[1]] for k, v in pay["constants"].items()},
        "expect": expect,
    }
    with open(os.path.join(OUT_DIR, f"{name}.json"), "w") as f:
        json.dump(fx, f, indent=1)
    return fx


def derive(rel_path, root=REF):
    with open(os.path.join(root, rel_path)) as f:
        return circom_subset.unroll(f.read())


def from_circom(name, rel_path, expect, root=REF):
    d = derive(rel_path, root)
    return finish(name, d["script"], d["input_prefixes"], d["output_prefixes"], expect,
                  rel_path if root == REF else os.path.join("tests/golden/circuits", os.path.basename(rel_path)))


def main():
    T = "tests/circuits/integration/"
    fx = from_circom("sum", T + "sum.circom", {
        "reference_test": "tests/integration.rs:365-372 (test_sum)",
        "io": {"inputs": {"0.a": 3, "0.b": 5}, "outputs": {"0.out": 8}},
        "hand": {"gates": [["AAdd", 1, 2, 5]], "wire_count": 3},
    })
    assert fx["gates"] == [["AAdd", 1, 2, 5]]
    fx = from_circom("addZero", T + "addZero.circom", {
        "reference_test": "tests/integration.rs:279-286 (test_add_zero)",
        "io": {"inputs": {"0.in": 42}, "outputs": {"0.out": 42}},
        "hand": {"gates": [["AAdd", 1, 3, 5]], "wire_count": 3,
                 "constants": {"0.const_signal_0_2": {"value": "0", "wire_index": 1}}},
    })
    assert fx["gates"] == [["AAdd", 1, 3, 5]]
    fx = from_circom("xEqX", T + "xEqX.circom", {
        "reference_test": "tests/integration.rs:375-382 (test_x_eq_x)",
        "io": {"inputs": {"0.x": 37}, "outputs": {"0.out": 1}},
        "hand": {"gates": [["AEq", 1, 1, 4]], "wire_count": 2},
    })
    assert fx["gates"] == [["AEq", 1, 1, 4]]
    fx = from_circom("matElemMul", T + "matElemMul.circom", {
        "reference_test": "tests/integration.rs:335-362 (test_matrix_element_multiplication)",
        "io": {"inputs": {f"0.{m}[{i}][{j}]": 2 for m in "ab" for i in range(2) for j in range(2)},
               "outputs": {f"0.out[{i}][{j}]": 4 for i in range(2) for j in range(2)}},
        "hand": {"gates": [["AMul", 1, 5, 14], ["AMul", 2, 6, 16], ["AMul", 3, 7, 18], ["AMul", 4, 8, 20]],
                 "wire_count": 12},
    })
    assert fx["gates"] == fx["expect"]["hand"]["gates"]
    from_circom("constantSum", T + "constantSum.circom", {
        "reference_test": "tests/integration.rs:393-415 (test_constant_sum)",
        "constants_exact": {"0.const_signal_8_1": {"value": "8", "wire_index": 0}},
    })
    from_circom("directOutput", T + "directOutput.circom", {
        "reference_test": "tests/integration.rs:417-441 (test_direct_output)",
        "outputs_exact": {"0.out": 0},
        "constants_len": 1,
        "constant_exact": {"0.const_signal_42_1": {"value": "42", "wire_index": 0}},
    })
    stmts = [("AMul", 2, 3), ("AIntDiv", 4, 3), ("AAdd", 3, 4), ("ASub", 4, 1), ("APow", 2, 4), ("AMod", 5, 3),
             ("AShiftL", 5, 1), ("AShiftR", 5, 1), ("ALEq", 2, 3), ("ALEq", 3, 3), ("ALEq", 4, 3), ("AGEq", 2, 3),
             ("AGEq", 3, 3), ("AGEq", 4, 3), ("ALt", 2, 3), ("ALt", 3, 3), ("ALt", 4, 3), ("AGt", 2, 3), ("AGt", 3, 3),
             ("AGt", 4, 3), ("AEq", 2, 3), ("AEq", 3, 3), ("ANeq", 2, 3), ("ANeq", 3, 3), ("ABoolOr", 0, 1),
             ("ABoolAnd", 0, 1), ("ABitOr", 1, 3), ("ABitAnd", 1, 3), ("AXor", 1, 3)]
    exp_out = {"mul_2_3": 6, "idiv_4_3": 1, "add_3_4": 7, "sub_4_1": 3, "pow_2_4": 16, "mod_5_3": 2, "shl_5_1": 10,
               "shr_5_1": 2, "leq_2_3": 1, "leq_3_3": 1, "leq_4_3": 0, "geq_2_3": 0, "geq_3_3": 1, "geq_4_3": 1,
               "lt_2_3": 1, "lt_3_3": 0, "lt_4_3": 0, "gt_2_3": 0, "gt_3_3": 0, "gt_4_3": 1, "eq_2_3": 0, "eq_3_3": 1,
               "neq_2_3": 1, "neq_3_3": 0, "or_0_1": 1, "and_0_1": 0, "bit_or_1_3": 3, "bit_and_1_3": 1,
               "bit_xor_1_3": 2}
    fx = from_circom("infixOps", T + "infixOps.circom", {
        "reference_test": "tests/integration.rs:289-332 (test_infix_ops)",
        "io": {"inputs": {f"0.x{i}": i for i in range(6)}, "outputs": {f"0.{k}": v for k, v in exp_out.items()}},
        "hand": {"gates": [[op, 1 + a, 1 + b, 37 + 2 * k] for k, (op, a, b) in enumerate(stmts)],
                 "wire_count": 6 + 29},
    })
    assert fx["gates"] == fx["expect"]["hand"]["gates"]

    # ---- ArgMax(2) = input/circuit.circom, BASELINE config C1: derived, and equal to the hand table of SURVEY A.5
    table = """AGt 49 49 50|ASub 49 49 23|AMul 23 50 25|AAdd 25 49 88|ASub 28 25 29|AAdd 29 49 31|ASub 52 52 38|
    AMul 38 50 40|AAdd 40 52 91|ASub 43 40 44|AAdd 44 52 46|AGt 89 88 90|ASub 89 88 63|AMul 63 90 65|AAdd 65 88 95|
    ASub 68 65 69|AAdd 69 89 71|ASub 93 91 78|AMul 78 90 80|AAdd 80 91 96|ASub 83 80 84|AAdd 84 93 86"""
    hand_gates = []
    for t in table.replace("\n", "").split("|"):
        op, a, b, o = t.split()
        hand_gates.append([op, int(a), int(b), int(o)])
    node_wire = {50: 2, 23: 3, 25: 4, 88: 5, 28: 6, 29: 7, 31: 8, 52: 9, 38: 10, 40: 11, 91: 12, 43: 13, 44: 14, 46: 15,
                 90: 16, 63: 17, 65: 18, 95: 19, 68: 20, 69: 21, 71: 22, 93: 23, 78: 24, 80: 25, 83: 26, 84: 27, 86: 28,
                 96: 29, 49: 0, 89: 1}
    fx = from_circom("argmax2", "input/circuit.circom", {
        "hand": {"gates": hand_gates, "wire_count": 30, "sorted_is_identity": True,
                 "node_wire": {str(k): v for k, v in node_wire.items()}},
        # ArgMax semantics (index of the maximum, first wins on ties) as functional vectors
        "io_cases": [{"inputs": {"0.in[0]": 2, "0.in[1]": 3}, "outputs": {"0.out": 1}},
                     {"inputs": {"0.in[0]": 7, "0.in[1]": 3}, "outputs": {"0.out": 0}},
                     {"inputs": {"0.in[0]": 4, "0.in[1]": 4}, "outputs": {"0.out": 0}}],
    })
    assert fx["gates"] == hand_gates and fx["input_nodes"] == [49, 89] and fx["output_nodes"] == [96]
    assert fx["constants"] == {"0.const_signal_0_11": [52, "0"], "Switcher.const_signal_0_22": [28, "0"],
                               "Switcher.const_signal_0_34": [43, "0"], "Switcher.const_signal_0_47": [68, "0"],
                               "Switcher.const_signal_0_59": [83, "0"], "0.const_signal_1_62": [93, "1"]}

    # ---- fixtures the reference ships without a (passing) test -------------------------------------------------------
    from_circom("arrayAssignment", T + "arrayAssignment.circom", {
        "reference_test": None,        # the file has no test upstream
        "io_cases": [{"inputs": {f"0.a_in[{i}][{j}]": 1 + 2 * i + j for i in range(2) for j in range(2)}, "outputs": {"0.out": 10}},
                     {"inputs": {f"0.a_in[{i}][{j}]": 7 for i in range(2) for j in range(2)}, "outputs": {"0.out": 28}}],
    })
    from_circom("mainTemplateArgument", T + "mainTemplateArgument.circom", {
        "reference_test": None,
        "io_cases": [{"inputs": {"0.in": 5}, "outputs": {"0.out": 105}}, {"inputs": {"0.in": 0}, "outputs": {"0.out": 100}}],
    })
    from_circom("underConstrained", T + "underConstrained.circom", {
        "reference_test": "tests/integration.rs:443-453 (#[ignore]: known bug, the output has no wire because no gate touches it)",
        "outputs_exact": {"0.x": 0},
    })
    # prefixOps: the input prefix filter "0.c" (program.rs:57-60) also captures 0.complementA/B/C and the constant
    # signals, so build_circuit reports Inconsistency (compiler.rs:363-383).  The message quoted upstream names node 10 —
    # the node 0.complementC holds BEFORE its connection re-issues the id (signal 9 -> node 10); with compiler.rs:257 as
    # it stands the check sees the merged node.
    d = derive(T + "prefixOps.circom")
    m = replay(d["script"])
    pre_merge = {}
    mm = orc.CompilerModel()
    for st in d["script"]:
        if st[0] == "signal":
            mm.add_signal(st[1], st[2], st[3])
            if st[2] == "0.complementC":
                pre_merge["0.complementC"] = max(mm.nodes)
    node_of = {sid: nid for nid, node in m.nodes.items() for sid in node.signals}
    clashes = {nm: node_of[sid] for sid, nm in [(s, n_[0] if isinstance(n_, tuple) else n_) for s, n_ in
                                                 [(sid, m.signals[sid].name) for sid in m.signals]]
               if nm.startswith("0.complement")}
    from_circom("prefixOps", T + "prefixOps.circom", {
        "reference_test": "tests/integration.rs:455-475 (#[ignore]: known bug)",
        "error": {"kind": "Inconsistency",
                  "reference_comment": "Node 10 used for both input 0.complementC and output 0.complementC",
                  "pre_merge_node_of_complementC": pre_merge["0.complementC"],
                  "messages_any_of": [f"Node {nid} used for both input {nm} and output {nm}" for nm, nid in sorted(clashes.items())]},
        "u32_max_constant": "0.const_signal_4294967295",
    })
    # indexOutOfBounds: the front-end itself fails (tests/integration.rs:376-391) — no flat list exists
    try:
        derive(T + "indexOutOfBounds.circom")
        raise AssertionError("indexOutOfBounds must fail")
    except circom_subset.ProgramError as e:
        with open(os.path.join(OUT_DIR, "indexOutOfBounds.json"), "w") as f:
            json.dump({"name": "indexOutOfBounds", "source": T + "indexOutOfBounds.circom", "hand_traced": False, "script": None,
                       "expect": {"reference_test": "tests/integration.rs:376-391 (test_out_of_bounds)",
                                  "compile_error": "Runtime error: Index out of bounds", "derived_error": str(e)}}, f, indent=1)
        assert str(e) == "Runtime error: Index out of bounds"
    # a circuit of this repo whose gate list is NOT in dependency order: the DFS returns a non-identity order
    fx = from_circom("nonIdentity", "nonIdentity.circom", {
        "reference_test": None,
        "hand": {"sorted": [3, 0, 2, 1, 4]},
        "io_cases": [{"inputs": {"0.x": 2, "0.z": 4}, "outputs": {"0.y": 25, "0.w": 36}},
                     {"inputs": {"0.x": 0, "0.z": 1}, "outputs": {"0.y": 0, "0.w": 1}}],
    }, root=os.path.join(OUT_DIR, "circuits"))
    print("wrote fixtures to", OUT_DIR)


if __name__ == "__main__":
    main()
