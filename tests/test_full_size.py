"""BASELINE.json's full-size configuration (10 M gates, depth 5 000, width 32) on the GPU: bit-exact parity of the
build_circuit outputs with the oracle through position-salted checksums computed on the device, size-independent
properties (permutation, topological validity, first-seen monotonicity), and slice-wise bit-exact parity of the
742 M-gate boolean circuit.  Also the Sha256 / Keccak shape stand-ins (configs[2], configs[3])."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check_properties(fg, sorted_ids, in0, in1, out, node_wire, wire_count):
    n = fg.n
    # permutation
    seen = np.zeros(n, dtype=bool)
    seen[sorted_ids] = True
    assert seen.all()
    # topological validity: every producer sits before its consumers
    pos = np.empty(n, dtype=np.int64)
    pos[sorted_ids] = np.arange(n)
    prod = np.full(fg.n_nodes, -1, dtype=np.int64)
    prod[fg.out] = np.arange(n)                      # distinct out nodes in the generator
    for side in (fg.lh, fg.rh):
        d = prod[side]
        m = d >= 0
        assert (pos[d[m]] < pos[np.nonzero(m)[0]]).all()
    # wire numbering: inputs first, outputs last, intermediates numbered in first-seen order along the walk
    n_in, n_out = len(fg.input_nodes), len(fg.output_nodes)
    np.testing.assert_array_equal(node_wire[fg.input_nodes], np.arange(n_in))
    np.testing.assert_array_equal(node_wire[fg.output_nodes], wire_count - n_out + np.arange(n_out))
    walk = np.stack([in0, in1, out], axis=1).reshape(-1)
    mid = (walk >= n_in) & (walk < wire_count - n_out)
    uniq, first_idx = np.unique(walk[mid], return_index=True)
    np.testing.assert_array_equal(uniq, np.arange(n_in, n_in + len(uniq)))
    assert (np.diff(first_idx) > 0).all()


def _check_checksums(be, exp):
    """sorted / in0 / in1 / out / op / node -> wire of the whole circuit against the oracle (position-salted checksums on the device)"""
    bm = importlib.import_module("circom-2-arithc_amd.backend")
    for name, arr in (("sorted", exp.sorted), ("in0", exp.in0), ("in1", exp.in1), ("out", exp.out), ("op", exp.op)):
        assert be.checksum(name) == bm.checksum_host(arr), name
    nw1 = ((exp.node_wire.astype(np.uint64) + 1) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    assert be.checksum("node_wire1") == bm.checksum_host(nw1), "node_wire"


@pytest.mark.parametrize("walk", [0, 1], ids=["positional", "walk"])
def test_reference_shaped_10m(orc, c2a, walk):
    """A 10 M-gate graph shaped like what the reference's unroller emits — a fresh named constant node at a tenth of the gates
    (src/process.rs:558-579, keys at src/compiler.rs:354-359), an output node at a twentieth: 1.5 M numbering events instead of the
    headline graph's 2 064 — through the staged calls AND the fused c2a_build_circuit, on both numbering paths
    (src/compiler.rs:423-449): every result array against the oracle."""
    from conftest import _Env
    fg = c2a.synth.config("reference_shaped_10m")
    assert fg.n == 10_000_000 and len(fg.const_nodes) > 900_000 and len(fg.output_nodes) > 450_000
    args = (fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    exp = orc.build_circuit(*args, mode=1)
    with _Env(C2A_NUMBERING_WALK=walk):
        be = c2a.Backend(0)
    try:
        be.load_gates(*args)
        # the fused call
        assert be.build_circuit() == exp.wire_count
        st = be.stats()
        assert st["numbering_path"] == 1 - walk, st
        if not walk:
            assert st["numbering_events"] == len(fg.const_nodes) - 64 + int(np.isin(fg.const_nodes[:64], np.concatenate([fg.lh, fg.rh])).sum()) \
                + len(fg.output_nodes), st
        _check_checksums(be, exp)
        # the staged calls
        sorted_ids = be.topo_sort()
        np.testing.assert_array_equal(sorted_ids, exp.sorted)
        node_wire, wire_count = be.assign_wires()
        assert wire_count == exp.wire_count
        np.testing.assert_array_equal(node_wire, exp.node_wire)
        in0, in1, out, op = be.emit_gates()
        for a, b in zip((in0, in1, out, op), (exp.in0, exp.in1, exp.out, exp.op)):
            np.testing.assert_array_equal(a, b)
        _check_checksums(be, exp)
        _check_properties(fg, sorted_ids, in0, in1, out, node_wire, wire_count)
        # and the boolean image of it: totals + one slice bit-exact, every wire simulated
        info = be.boolify(32)
        sl, g0 = orc.boolify_range(exp, 32, fg.n // 2, 20_000)
        assert sl.wire_count == info.wire_count
        for a, b in zip(be.bool_read(g0, len(sl.in0)), (sl.in0, sl.in1, sl.out, sl.op)):
            np.testing.assert_array_equal(a, b)
        checked, bad = be.verify_boolify(seed=5)
        assert checked == wire_count * 64 and bad == 0
    finally:
        be.close()


def test_synthetic_10m_full_size(hip_backend, orc, c2a):
    be = hip_backend
    backend_mod = importlib.import_module("circom-2-arithc_amd.backend")
    fg = c2a.synth.config("synthetic_10m")
    assert fg.n == 10_000_000
    be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    sorted_ids = be.topo_sort()
    node_wire, wire_count = be.assign_wires()
    in0, in1, out, op = be.emit_gates()
    st = be.stats()
    assert st["levels"] == 5000
    _check_properties(fg, sorted_ids, in0, in1, out, node_wire, wire_count)
    exp = orc.build_circuit(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes, mode=1)
    assert wire_count == exp.wire_count
    for name, arr in (("sorted", exp.sorted), ("in0", exp.in0), ("in1", exp.in1), ("out", exp.out), ("op", exp.op)):
        assert be.checksum(name) == backend_mod.checksum_host(arr), name
    np.testing.assert_array_equal(sorted_ids, exp.sorted)
    # boolean circuit: totals + three slices (head, middle, tail) bit-exact against the oracle
    info = be.boolify(32)
    T = np.array([orc.template_size(o, 32)[0] for o in range(20)], dtype=np.int64)
    assert info.n_gates == int(T[exp.op].sum())
    for first in (0, fg.n // 2 - 7, fg.n - 20_000):
        sl, g0 = orc.boolify_range(exp, 32, first, 20_000)
        assert sl.wire_count == info.wire_count
        got = be.bool_read(g0, len(sl.in0))
        for a, b in zip(got, (sl.in0, sl.in1, sl.out, sl.op)):
            np.testing.assert_array_equal(a, b)
    # the whole 742 M-gate boolean circuit simulated on the GPU against the arithmetic circuit: every arithmetic
    # wire x 64 vectors (the reference's simulation harness, tests/integration.rs:191-237, at full size)
    checked, bad = be.verify_boolify(seed=20241008)
    assert checked == wire_count * 64 and bad == 0
    # and the result is reproducible run to run
    c1 = be.checksum("bool_out")
    be.boolify(32)
    assert be.checksum("bool_out") == c1


def test_synthetic_10m_width_64(hip_backend, orc, c2a):
    """SURVEY §8(d) "widths 32 and 64": the 10 M-gate graph bit-blasted at --boolify-width 64 (1.5 G boolean gates, 19.5 GB):
    totals against the template table, three slices bit-exact against the oracle, and every wire of the boolean circuit
    simulated against the arithmetic one on the GPU."""
    be = hip_backend
    fg = c2a.synth.config("synthetic_10m")
    be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    wire_count = be.build_circuit()
    exp = orc.build_circuit(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes, mode=1)
    assert wire_count == exp.wire_count
    _check_checksums(be, exp)                        # (the fused call: sorted[] is written by the emission's split pass)
    info = be.boolify(64)
    T = np.array([orc.template_size(o, 64)[0] for o in range(20)], dtype=np.int64)
    assert info.n_gates == int(T[exp.op].sum())
    for first in (0, fg.n // 3 + 11, fg.n - 10_000):
        sl, g0 = orc.boolify_range(exp, 64, first, 10_000)
        assert sl.wire_count == info.wire_count
        got = be.bool_read(g0, len(sl.in0))
        for a, b in zip(got, (sl.in0, sl.in1, sl.out, sl.op)):
            np.testing.assert_array_equal(a, b)
    checked, bad = be.verify_boolify(seed=64)
    assert checked == wire_count * 64 and bad == 0


@pytest.mark.parametrize("name,width", [("sha256_standin", 32), ("keccak_standin", 64), ("poseidon2_standin", 32)])
def test_config_standins_bit_exact(hip_backend, orc, c2a, name, width):
    be = hip_backend
    fg = c2a.synth.config(name)
    be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    exp = orc.build_circuit(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes, mode=1)
    np.testing.assert_array_equal(be.topo_sort(), exp.sorted)
    nw, wc = be.assign_wires()
    assert wc == exp.wire_count
    np.testing.assert_array_equal(nw, exp.node_wire)
    in0, in1, out, op = be.emit_gates()
    for a, b in zip((in0, in1, out, op), (exp.in0, exp.in1, exp.out, exp.op)):
        np.testing.assert_array_equal(a, b)
    info = be.boolify(width)
    eb = orc.boolify(exp, width)
    assert info.n_gates == len(eb.in0) and info.wire_count == eb.wire_count
    for a, b in zip(be.bool_read(), (eb.in0, eb.in1, eb.out, eb.op)):
        np.testing.assert_array_equal(a, b)


def test_hub_10m(orc, c2a):
    """10 M gates with HUBS (synth.family("hub"): 43 757 produced nodes with 17-10^4 consumers, three with 10^5-10^6 — half of all
    edges; /root/reference/src/compiler.rs:408-421 allows any fan-out): the fused build against the oracle by checksums over all gates,
    the size-independent properties of the staged results, the relays counted, the reverse Kahn levels exact — and the sort NOT slower
    than twice the headline's (round 5 took a second for this graph: one consumer per memory round trip)."""
    fg = c2a.synth.family("hub", 10_000_000)
    args = (fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    exp = orc.build_circuit(*args, mode=1)
    with c2a.Backend(0) as be:
        be.load_gates(*args)
        for rep in range(2):
            assert be.build_circuit() == exp.wire_count
            _check_checksums(be, exp)
        st, t = be.stats(), be.timings()
        assert st["levels"] == 5000 and st["n_relays"] > 600_000 and st["numbering_path"] == 1
        assert t["k_peel"] < 20.0 and t["prep"] < 6.0, t          # ms (measured: 5.9 and 2.2)
        sorted_ids = be.topo_sort()
        node_wire, wire_count = be.assign_wires()
        in0, in1, out, op = be.emit_gates()
        np.testing.assert_array_equal(sorted_ids, exp.sorted)
        _check_properties(fg, sorted_ids, in0, in1, out, node_wire, wire_count)


def test_strict_layers_10m(orc, c2a):
    """10 M gates in STRICT layers (synth.family("strict"): both operands out of the layer right above, so every layer waits for all of
    the one below and a gate's other consumers are being worked on at the same time — 1.6 M record re-reads): the fused build against
    the oracle by checksums, the levels exact, and the sort within four times the headline's per level (it was 10 us per level — 50 ms —
    while every re-read did an atomic on the line all waiting waves poll: DESIGN.md 4.2, STRICT LAYERS; measured now: 11.3 ms)."""
    fg = c2a.synth.family("strict", 10_000_000)
    args = (fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    exp = orc.build_circuit(*args, mode=1)
    with c2a.Backend(0) as be:
        be.load_gates(*args)
        for rep in range(2):
            assert be.build_circuit() == exp.wire_count
            _check_checksums(be, exp)
        st, t = be.stats(), be.timings()
        assert st["levels"] == 5000 and st["numbering_path"] == 1 and st["peel_rereads"] > 100_000, st
        assert t["k_peel"] < 25.0, t          # ms
