"""Pins the CPU oracle (oracle/) against everything the reference's own tests hold for the hot path
(tests/integration.rs:279-475, copied as data into tests/golden/*.json).  The flat gate lists are derived from the
.circom texts by tests/golden/circom_subset.py (a restatement of the reference's unroller) and cross-checked against
the hand traces of SURVEY.md Appendix A.  CPU only."""
import numpy as np
import pytest

from helpers import fixture_payload, load_fixtures, simulate_arith
from golden.make_fixtures import replay

FX = load_fixtures()
BUILDABLE = [n for n in FX if FX[n].get("script") is not None and "build_circuit_error" not in FX[n]]


def _model(fx, orc):
    m = replay(fx["script"])
    for p in fx["input_prefixes"]:
        m.add_inputs(m.get_signals(f"0.{p}"))
    for p in fx["output_prefixes"]:
        m.add_outputs(m.get_signals(f"0.{p}"))
    return m


def _simulate(orc, circ, inputs):
    in0 = np.array([g[0] for g in circ.gates], np.uint32)
    in1 = np.array([g[1] for g in circ.gates], np.uint32)
    out = np.array([g[2] for g in circ.gates], np.uint32)
    op = np.array([orc.OP[g[3]] for g in circ.gates], np.uint8)
    return simulate_arith(orc, in0, in1, out, op, circ.wire_count, len(circ.input_name_to_wire_index),
                          len(circ.output_name_to_wire_index),
                          {circ.input_name_to_wire_index[k]: v for k, v in inputs.items()},
                          {c.wire_index: int(c.value) for c in circ.constants.values()})


@pytest.mark.parametrize("name", BUILDABLE)
def test_literal_restatement_matches_reference_expectations(name, orc):
    fx = FX[name]
    m = _model(fx, orc)
    circ = m.build_circuit()                                  # literal restatement of compiler.rs:321-494
    exp = fx["expect"]
    if "hand" in exp:
        if "gates" in exp["hand"]:
            assert [[orc.OP_NAMES[g.op], g.lh_in, g.rh_in, g.out] for g in m.gates] == exp["hand"]["gates"]
        if "wire_count" in exp["hand"]:
            assert circ.wire_count == exp["hand"]["wire_count"]
        if exp["hand"].get("sorted_is_identity"):
            assert circ.sorted_gate_ids == list(range(len(m.gates)))
        if "sorted" in exp["hand"]:                           # a NON-identity DFS order
            assert circ.sorted_gate_ids == exp["hand"]["sorted"] != list(range(len(m.gates)))
        for node, w in exp["hand"].get("node_wire", {}).items():
            assert circ.node_id_to_wire_id[int(node)] == w
        for k, v in exp["hand"].get("constants", {}).items():
            assert circ.constants[k] == orc.ConstantInfo(v["value"], v["wire_index"])
    for case in exp.get("io_cases", []):
        vals = _simulate(orc, circ, case["inputs"])
        for k, v in case["outputs"].items():
            assert int(vals[circ.output_name_to_wire_index[k]]) == v, (k, case)
    if "constants_exact" in exp:                              # integration.rs:407-414
        assert {k: {"value": c.value, "wire_index": c.wire_index} for k, c in circ.constants.items()} == \
            exp["constants_exact"]
        assert len(circ.constants) == 1
    if "outputs_exact" in exp:                                # integration.rs:431-440, :447-452
        assert circ.output_name_to_wire_index == exp["outputs_exact"]
        if "constants_len" in exp:
            assert len(circ.constants) == exp["constants_len"]
            (k, v), = exp["constant_exact"].items()
            assert circ.constants[k] == orc.ConstantInfo(v["value"], v["wire_index"])
    if "io" in exp:                                           # simulation_test, integration.rs:257-277
        vals = _simulate(orc, circ, exp["io"]["inputs"])
        for k, v in exp["io"]["outputs"].items():
            assert int(vals[circ.output_name_to_wire_index[k]]) == v, k
        assert len(circ.gates) == len(fx["gates"])


def test_prefix_ops_fixture_reproduces_the_known_inconsistency(orc):
    """tests/integration.rs:455-475 (#[ignore]d upstream): the input prefix "0.c" also captures 0.complementA/B/C, so
    build_circuit fails with Inconsistency (compiler.rs:363-383).  The message quoted upstream names node 10 — the node
    0.complementC holds before its connection re-issues the id; the derivation reproduces exactly that id."""
    fx = FX["prefixOps"]
    err = fx["expect"]["error"]
    assert err["pre_merge_node_of_complementC"] == 10
    assert err["reference_comment"] == "Node 10 used for both input 0.complementC and output 0.complementC"
    with pytest.raises(orc.Inconsistency) as ei:
        _model(fx, orc).build_circuit()
    assert ei.value.message in err["messages_any_of"]
    # ~x is (u32::MAX ^ x): the only source of a u32::MAX constant (process.rs:758-764)
    assert any(st[0] == "signal" and st[2] == err and st[3] == 0xFFFFFFFF for st in fx["script"]
               for err in [fx["expect"]["u32_max_constant"]])
    assert [g[0] for g in fx["gates"]] == ["ASub", "AEq", "AEq", "AEq", "AXor", "AXor", "AXor"]


def test_out_of_bounds_fixture_is_the_reference_error():
    fx = FX["indexOutOfBounds"]                               # tests/integration.rs:376-391
    assert fx["expect"]["compile_error"] == fx["expect"]["derived_error"] == "Runtime error: Index out of bounds"


def test_fixtures_are_machine_derived():
    """no fixture depends on a hand trace any more; where a hand trace exists it agrees"""
    for name, fx in FX.items():
        assert fx["hand_traced"] is False, name


@pytest.mark.parametrize("name", [n for n in FX if "n_nodes" in FX[n]])
@pytest.mark.parametrize("mode", [0, 1])
def test_c_restatement_matches_literal(name, mode, orc):
    fx = FX[name]
    p = fixture_payload(fx, orc)
    c = orc.build_circuit(p["lh"], p["rh"], p["out"], p["op"], p["n_nodes"], p["input_nodes"], p["output_nodes"],
                          mode=mode)
    if True:
        lit = _model(fx, orc).build_circuit()
        assert list(c.sorted) == lit.sorted_gate_ids
        assert c.wire_count == lit.wire_count
        for node, w in lit.node_id_to_wire_id.items():
            assert int(c.node_wire[node]) == w
        assert [(int(a), int(b), int(o), orc.OP_NAMES[k]) for a, b, o, k in zip(c.in0, c.in1, c.out, c.op)] == lit.gates


def test_literal_vs_c_on_random_graphs(orc):
    from conftest import random_gate_graph
    rng = np.random.default_rng(99)
    n_cyc = 0
    for _ in range(300):
        p = random_gate_graph(rng, int(rng.integers(1, 50)), p_dup_out=0.05, p_cycle=0.05)
        prod = {}
        for g, o in enumerate(p["out"].tolist()):
            prod[o] = g
        lh, rh = p["lh"].tolist(), p["rh"].tolist()

        def deps(g):
            d = []
            if lh[g] in prod:
                d.append(prod[lh[g]])
            if rh[g] in prod:
                d.append(prod[rh[g]])
            return d
        try:
            lit = orc.topological_sort_literal(len(lh), deps)
            err = None
        except orc.CyclicDependency as e:
            lit, err = None, str(e)
            n_cyc += 1
        for mode in (0, 1):
            try:
                c = orc.build_circuit(p["lh"], p["rh"], p["out"], p["op"], p["n_nodes"], [], [], mode=mode)
                assert err is None and list(c.sorted) == lit
            except orc.CyclicDependency as e:
                assert str(e) == err
    assert n_cyc > 5


def test_bit_blast_spec_is_functionally_correct(orc):
    """Every op template of the frozen spec against orc.eval_op on random + corner operands."""
    rng = np.random.default_rng(3)
    for w in (1, 2, 3, 5, 8, 16, 32, 64):
        mask = (1 << w) - 1
        for op in range(20):
            if op == orc.OP["APow"] and w > 8:
                continue
            circ = orc.ArithCircuit(sorted=np.array([0], np.uint32), in0=np.array([0], np.uint32),
                                    in1=np.array([1], np.uint32), out=np.array([2], np.uint32),
                                    op=np.array([op], np.uint8), node_wire=np.array([0, 1, 2], np.uint32),
                                    wire_count=3, n_in=2, n_out=1)
            bc = orc.boolify(circ, w)
            T, aux = orc.template_size(op, w)
            assert len(bc.in0) == T and bc.wire_count == 3 * w + aux
            a = [0, mask, 1, 0, mask, 5 & mask] + [int(x) & mask for x in rng.integers(0, 2 ** 63, 58)]
            b = [0, mask, 0, 1, 1, 0] + [int(x) & mask for x in rng.integers(0, 2 ** 63, 58)]
            for i in range(6, 22):
                b[i] = (i - 6) & mask                                    # small shift / exponent operands
            wires = np.zeros(bc.wire_count, np.uint64)
            for i in range(w):
                wires[i] = sum(((a[t] >> i) & 1) << t for t in range(64))
                wires[w + i] = sum(((b[t] >> i) & 1) << t for t in range(64))
            orc.eval_bool(bc, wires)
            for t in range(64):
                got = 0
                for i in range(w):
                    got |= ((int(wires[int(orc.bool_wire(circ, aux, w, 2, i))]) >> t) & 1) << i
                assert got == orc.eval_op(op, a[t], b[t], w), (orc.OP_NAMES[op], w, a[t], b[t])
