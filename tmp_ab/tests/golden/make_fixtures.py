#!/usr/bin/env python3
"""Generates tests/golden/*.json — run from the repo root: `python tests/golden/make_fixtures.py`.

What a fixture is
-----------------
The reference cannot be built in this image (Rust + un-vendored git deps), so there is nothing to run to
produce vectors.  What the reference's own tests DO pin for the flat-gate-graph path is in
/root/reference/tests/integration.rs:
    :279-374  five functional input->output tables (addZero, infixOps, matElemMul, sum, xEqX)
    :393-415  constantSum: constants == {"0.const_signal_8_1": {value:"8", wire_index:0}}
    :417-441  directOutput: output map {"0.out":0}, constant "0.const_signal_42_1" at wire 0
Those expectations are copied below as DATA ("expect_*").  The flat gate list each circuit produces is
obtained by replaying — through the literal Python restatement of add_signal / add_gate / add_connection
(oracle.CompilerModel, src/compiler.rs:139-278) — the call sequence that src/process.rs performs for the
circuit (declarations in order: process.rs:53-101; `lhs <== a op b`: process.rs:461-475 then :266-269;
literals become named constant signals: process.rs:558-579; variables consume no signal ids:
runtime.rs:205-217).  The call sequences are HAND-DERIVED — marked "hand_traced".
ArgMax(2) (the shipped input/circuit.circom, BASELINE config C1) is taken from the table of SURVEY A.5.

Each JSON holds: the replayed Compiler state (signals, inputs, outputs, the call script), the flat payload
that crosses the C ABI, the reference-test expectations, and the hand-traced expectations.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

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
    pay = m.flat_payload()
    fx = {
        "name": name,
        "source": source,
        "hand_traced": True,
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


def binop_script(n_in_signals, stmts, first_out_sid):
    """`out_k <== x_a op x_b` statements: random signal, gate, connect(random, out_k)."""
    script = []
    sid = first_out_sid + len(stmts)
    for k, (op, a, b) in enumerate(stmts):
        script.append(["signal", sid, f"0.random_{k}", None])        # process.rs:466-474
        script.append(["gate", op, a, b, sid])                       # process.rs:475
        script.append(["connect", sid, first_out_sid + k])           # process.rs:266-269
        sid += 1
    return script


def main():
    # ---- sum.circom -------------------------------------------------------------
    script = [["signal", 0, "0.a", None], ["signal", 1, "0.b", None], ["signal", 2, "0.out", None]]
    script += binop_script(2, [("AAdd", 0, 1)], 2)
    finish("sum", script, ["a", "b"], ["out"], {
        "reference_test": "tests/integration.rs:365-372 (test_sum)",
        "io": {"inputs": {"0.a": 3, "0.b": 5}, "outputs": {"0.out": 8}},
        "hand": {"gates": [["AAdd", 1, 2, 5]], "wire_count": 3},
    }, "tests/circuits/integration/sum.circom")

    # ---- addZero.circom ---------------------------------------------------------
    script = [["signal", 0, "0.in", None], ["signal", 1, "0.out", None],
              ["signal", 2, "0.const_signal_0", 0],                   # make_constant, process.rs:558-579
              ["signal", 3, "0.random_0", None], ["gate", "AAdd", 0, 2, 3], ["connect", 3, 1]]
    finish("addZero", script, ["in"], ["out"], {
        "reference_test": "tests/integration.rs:279-286 (test_add_zero)",
        "io": {"inputs": {"0.in": 42}, "outputs": {"0.out": 42}},
        "hand": {"gates": [["AAdd", 1, 3, 5]], "wire_count": 3,
                 "constants": {"0.const_signal_0_2": {"value": "0", "wire_index": 1}}},
    }, "tests/circuits/integration/addZero.circom")

    # ---- xEqX.circom ------------------------------------------------------------
    script = [["signal", 0, "0.x", None], ["signal", 1, "0.out", None]]
    script += binop_script(1, [("AEq", 0, 0)], 1)
    finish("xEqX", script, ["x"], ["out"], {
        "reference_test": "tests/integration.rs:375-382 (test_x_eq_x)",
        "io": {"inputs": {"0.x": 37}, "outputs": {"0.out": 1}},
        "hand": {"gates": [["AEq", 1, 1, 4]], "wire_count": 2},
    }, "tests/circuits/integration/xEqX.circom")

    # ---- matElemMul.circom (2,2) ------------------------------------------------
    script = []
    sid = 0
    for nm in ("a", "b", "out"):
        for i in range(2):
            for j in range(2):
                script.append(["signal", sid, f"0.{nm}[{i}][{j}]", None])
                sid += 1
    script += binop_script(8, [("AMul", k, 4 + k) for k in range(4)], 8)
    finish("matElemMul", script, ["a", "b"], ["out"], {
        "reference_test": "tests/integration.rs:335-362 (test_matrix_element_multiplication)",
        "io": {"inputs": {f"0.{m}[{i}][{j}]": 2 for m in "ab" for i in range(2) for j in range(2)},
               "outputs": {f"0.out[{i}][{j}]": 4 for i in range(2) for j in range(2)}},
        "hand": {"gates": [["AMul", 1, 5, 14], ["AMul", 2, 6, 16], ["AMul", 3, 7, 18], ["AMul", 4, 8, 20]],
                 "wire_count": 12},
    }, "tests/circuits/integration/matElemMul.circom")

    # ---- constantSum.circom: `out <== 3 + 5` folds to a variable (process.rs:445-458), then becomes the
    # constant signal const_signal_8 connected to out --------------------------------------------
    script = [["signal", 0, "0.out", None], ["signal", 1, "0.const_signal_8", 8], ["connect", 1, 0]]
    finish("constantSum", script, [], ["out"], {
        "reference_test": "tests/integration.rs:393-415 (test_constant_sum)",
        "constants_exact": {"0.const_signal_8_1": {"value": "8", "wire_index": 0}},
    }, "tests/circuits/integration/constantSum.circom")

    # ---- directOutput.circom: `out <== 42` -----------------------------------------------------
    script = [["signal", 0, "0.out", None], ["signal", 1, "0.const_signal_42", 42], ["connect", 1, 0]]
    finish("directOutput", script, [], ["out"], {
        "reference_test": "tests/integration.rs:417-441 (test_direct_output)",
        "outputs_exact": {"0.out": 0},
        "constants_len": 1,
        "constant_exact": {"0.const_signal_42_1": {"value": "42", "wire_index": 0}},
    }, "tests/circuits/integration/directOutput.circom")

    # ---- infixOps.circom: 6 inputs, 29 outputs, one gate each (19 of the 20 AGateTypes) -------
    outs = ["mul_2_3", "idiv_4_3", "add_3_4", "sub_4_1", "pow_2_4", "mod_5_3", "shl_5_1", "shr_5_1",
            "leq_2_3", "leq_3_3", "leq_4_3", "geq_2_3", "geq_3_3", "geq_4_3", "lt_2_3", "lt_3_3", "lt_4_3",
            "gt_2_3", "gt_3_3", "gt_4_3", "eq_2_3", "eq_3_3", "neq_2_3", "neq_3_3", "or_0_1", "and_0_1",
            "bit_or_1_3", "bit_and_1_3", "bit_xor_1_3"]
    stmts = [("AMul", 2, 3), ("AIntDiv", 4, 3), ("AAdd", 3, 4), ("ASub", 4, 1), ("APow", 2, 4), ("AMod", 5, 3),
             ("AShiftL", 5, 1), ("AShiftR", 5, 1), ("ALEq", 2, 3), ("ALEq", 3, 3), ("ALEq", 4, 3), ("AGEq", 2, 3),
             ("AGEq", 3, 3), ("AGEq", 4, 3), ("ALt", 2, 3), ("ALt", 3, 3), ("ALt", 4, 3), ("AGt", 2, 3), ("AGt", 3, 3),
             ("AGt", 4, 3), ("AEq", 2, 3), ("AEq", 3, 3), ("ANeq", 2, 3), ("ANeq", 3, 3), ("ABoolOr", 0, 1),
             ("ABoolAnd", 0, 1), ("ABitOr", 1, 3), ("ABitAnd", 1, 3), ("AXor", 1, 3)]
    script = [["signal", i, f"0.x{i}", None] for i in range(6)]
    script += [["signal", 6 + k, f"0.{nm}", None] for k, nm in enumerate(outs)]
    script += binop_script(6, stmts, 6)
    exp_out = {"mul_2_3": 6, "idiv_4_3": 1, "add_3_4": 7, "sub_4_1": 3, "pow_2_4": 16, "mod_5_3": 2, "shl_5_1": 10,
               "shr_5_1": 2, "leq_2_3": 1, "leq_3_3": 1, "leq_4_3": 0, "geq_2_3": 0, "geq_3_3": 1, "geq_4_3": 1,
               "lt_2_3": 1, "lt_3_3": 0, "lt_4_3": 0, "gt_2_3": 0, "gt_3_3": 0, "gt_4_3": 1, "eq_2_3": 0, "eq_3_3": 1,
               "neq_2_3": 1, "neq_3_3": 0, "or_0_1": 1, "and_0_1": 0, "bit_or_1_3": 3, "bit_and_1_3": 1,
               "bit_xor_1_3": 2}
    finish("infixOps", script, [f"x{i}" for i in range(6)], outs, {
        "reference_test": "tests/integration.rs:289-332 (test_infix_ops)",
        "io": {"inputs": {f"0.x{i}": i for i in range(6)}, "outputs": {f"0.{k}": v for k, v in exp_out.items()}},
        "hand": {"gates": [[op, 1 + a, 1 + b, 37 + 2 * k] for k, (op, a, b) in enumerate(stmts)],
                 "wire_count": 6 + 29},
    }, "tests/circuits/integration/infixOps.circom")

    # ---- ArgMax(2) = input/circuit.circom, BASELINE config C1 (not replayed) ---------
    table = """AGt 49 49 50|ASub 49 49 23|AMul 23 50 25|AAdd 25 49 88|ASub 28 25 29|AAdd 29 49 31|ASub 52 52 38|
    AMul 38 50 40|AAdd 40 52 91|ASub 43 40 44|AAdd 44 52 46|AGt 89 88 90|ASub 89 88 63|AMul 63 90 65|AAdd 65 88 95|
    ASub 68 65 69|AAdd 69 89 71|ASub 93 91 78|AMul 78 90 80|AAdd 80 91 96|ASub 83 80 84|AAdd 84 93 86"""
    gates = []
    for t in table.replace("\n", "").split("|"):
        op, a, b, o = t.split()
        gates.append([op, int(a), int(b), int(o)])
    node_wire = {50: 2, 23: 3, 25: 4, 88: 5, 28: 6, 29: 7, 31: 8, 52: 9, 38: 10, 40: 11, 91: 12, 43: 13, 44: 14, 46: 15,
                 90: 16, 63: 17, 65: 18, 95: 19, 68: 20, 69: 21, 71: 22, 93: 23, 78: 24, 80: 25, 83: 26, 84: 27, 86: 28,
                 96: 29, 49: 0, 89: 1}
    fx = {
        "name": "argmax2", "source": "input/circuit.circom (ArgMax(2)); SURVEY.md Appendix A.5", "hand_traced": True,
        "script": None, "gates": gates, "n_nodes": 97,
        "input_names": ["0.in[0]", "0.in[1]"], "input_nodes": [49, 89],
        "output_names": ["0.out"], "output_nodes": [96],
        "constants": {"0.const_signal_0_11": [52, "0"], "Switcher.const_signal_0_22": [28, "0"],
                      "Switcher.const_signal_0_34": [43, "0"], "Switcher.const_signal_0_47": [68, "0"],
                      "Switcher.const_signal_0_59": [83, "0"], "0.const_signal_1_62": [93, "1"]},
        "expect": {
            "hand": {"wire_count": 30, "sorted_is_identity": True, "node_wire": {str(k): v for k, v in node_wire.items()}},
            # ArgMax semantics (index of the maximum, first wins on ties) as functional vectors
            "io_cases": [{"inputs": {"0.in[0]": 2, "0.in[1]": 3}, "outputs": {"0.out": 1}},
                         {"inputs": {"0.in[0]": 7, "0.in[1]": 3}, "outputs": {"0.out": 0}},
                         {"inputs": {"0.in[0]": 4, "0.in[1]": 4}, "outputs": {"0.out": 0}}],
        },
    }
    with open(os.path.join(OUT_DIR, "argmax2.json"), "w") as f:
        json.dump(fx, f, indent=1)
    print("wrote fixtures to", OUT_DIR)


if __name__ == "__main__":
    main()
