"""Parity of the HIP path (through the C ABI) with the CPU oracle for the build_circuit numeric core:
sorted_gate_ids (src/topological_sort.rs:3-50), node->wire numbering (src/compiler.rs:388-449), emitted
gates (src/compiler.rs:451-464), and the cycle diagnostic.  Bit-exact: integer/index work."""
import numpy as np
import pytest

from conftest import random_gate_graph


def _compare(be, orc, p, check_serial=True):
    args = (p["lh"], p["rh"], p["out"], p["op"], p["n_nodes"], p["input_nodes"], p["output_nodes"])
    try:
        exp = orc.build_circuit(*args, mode=1)
        exp_err = None
    except (orc.CyclicDependency, orc.Inconsistency) as e:
        exp, exp_err = None, e
    be.load_gates(*args)
    if isinstance(exp_err, orc.Inconsistency):
        # the in/out clash is checked before the sort in the reference (compiler.rs:363-383); the ABI reports
        # it from c2a_assign_wires — the host mirror (compiler.py) checks names first, like the reference.
        try:
            be.topo_sort()
        except Exception:
            return "cyclic-and-inconsistent"
        with pytest.raises(Exception) as ei:
            be.assign_wires()
        assert "Inconsistency" in str(ei.value)
        return "inconsistent"
    if exp_err is not None:
        with pytest.raises(Exception) as ei:
            be.topo_sort()
        assert str(ei.value) == str(exp_err), (str(ei.value), str(exp_err))
        if check_serial:
            with pytest.raises(Exception) as ei2:
                be.topo_sort(serial=True)
            assert str(ei2.value) == str(exp_err)
        return "cyclic"
    got_sorted = be.topo_sort()
    np.testing.assert_array_equal(got_sorted, exp.sorted)
    if check_serial:
        np.testing.assert_array_equal(be.topo_sort(serial=True), exp.sorted)
        be.topo_sort()
    nw, wc = be.assign_wires()
    assert wc == exp.wire_count
    np.testing.assert_array_equal(nw, exp.node_wire)
    in0, in1, out, op = be.emit_gates()
    np.testing.assert_array_equal(in0, exp.in0)
    np.testing.assert_array_equal(in1, exp.in1)
    np.testing.assert_array_equal(out, exp.out)
    np.testing.assert_array_equal(op, exp.op)
    return "ok"


def test_random_small_graphs(backend, orc):
    rng = np.random.default_rng(20241008)
    seen = {"ok": 0, "cyclic": 0, "inconsistent": 0, "cyclic-and-inconsistent": 0}
    for trial in range(250):
        n = int(rng.integers(1, 60))
        p = random_gate_graph(rng, n, p_dup_out=0.1 if trial % 3 == 0 else 0.0, p_same=0.15,
                              p_cycle=0.08 if trial % 4 == 0 else 0.0)
        seen[_compare(backend, orc, p)] += 1
    assert seen["ok"] > 60 and seen["cyclic"] > 5, seen


def test_empty_and_single(backend, orc):
    e = np.empty(0, np.uint32)
    p = dict(lh=e, rh=e, out=e, op=np.empty(0, np.uint8), n_nodes=5, input_nodes=np.array([1, 2], np.uint32),
             output_nodes=np.array([4], np.uint32))
    assert _compare(backend, orc, p) == "ok"
    # zero gates, zero IO
    p = dict(lh=e, rh=e, out=e, op=np.empty(0, np.uint8), n_nodes=1, input_nodes=e, output_nodes=e)
    assert _compare(backend, orc, p) == "ok"
    # one gate reading one node twice (xEqX shape)
    p = dict(lh=np.array([1], np.uint32), rh=np.array([1], np.uint32), out=np.array([4], np.uint32),
             op=np.array([2], np.uint8), n_nodes=5, input_nodes=np.array([1], np.uint32),
             output_nodes=np.array([4], np.uint32))
    assert _compare(backend, orc, p) == "ok"


def test_self_loop_and_two_cycle(backend, orc):
    # gate 0 reads its own output
    p = dict(lh=np.array([3], np.uint32), rh=np.array([1], np.uint32), out=np.array([3], np.uint32),
             op=np.array([0], np.uint8), n_nodes=4, input_nodes=np.array([1], np.uint32),
             output_nodes=np.empty(0, np.uint32))
    assert _compare(backend, orc, p) == "cyclic"
    # a sink consumes a 2-cycle: the DFS enters the cycle through the peeled gate
    p = dict(lh=np.array([5, 6, 7], np.uint32), rh=np.array([1, 1, 5], np.uint32), out=np.array([6, 5, 8], np.uint32),
             op=np.zeros(3, np.uint8), n_nodes=9, input_nodes=np.array([1], np.uint32),
             output_nodes=np.array([8], np.uint32))
    assert _compare(backend, orc, p) == "cyclic"


def test_input_is_output_inconsistency(backend, orc):
    p = dict(lh=np.array([1], np.uint32), rh=np.array([2], np.uint32), out=np.array([3], np.uint32),
             op=np.array([0], np.uint8), n_nodes=4, input_nodes=np.array([1, 2], np.uint32),
             output_nodes=np.array([2], np.uint32))
    assert _compare(backend, orc, p) == "inconsistent"


def test_duplicate_io_nodes_last_wins(backend, orc):
    # two input names on one node / two output names on one node: later insert overwrites (compiler.rs:392-395,446-449)
    p = dict(lh=np.array([1, 3], np.uint32), rh=np.array([2, 1], np.uint32), out=np.array([3, 4], np.uint32),
             op=np.array([0, 7], np.uint8), n_nodes=5, input_nodes=np.array([1, 2, 1], np.uint32),
             output_nodes=np.array([4, 4], np.uint32))
    assert _compare(backend, orc, p) == "ok"


@pytest.mark.parametrize("layers,width,window", [(40, 25, 4), (300, 12, 64), (700, 3, 2)])
def test_layered_dags(backend, orc, c2a, layers, width, window):
    fg = c2a.synth.layered_dag(layers, width, n_in=32, n_const=4, window=window, mix=c2a.synth.MIX_ALL, seed=7 + layers)
    p = dict(lh=fg.lh, rh=fg.rh, out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes,
             output_nodes=fg.output_nodes)
    assert _compare(backend, orc, p, check_serial=(layers <= 300)) == "ok"
    st = backend.stats()
    assert st["levels"] >= layers


def test_peel_protocols_under_random_schedules(backend_peel, orc, c2a):
    """The ticket / hand-off / termination protocol of the dataflow launch with every wave of the launch alive at once and
    taking turns in a shuffled order (the emulator's concurrent launch, C2A_EMUL_SEED): the results do not depend on the
    interleaving.  Also cyclic inputs and duplicate writers."""
    be = backend_peel
    rng = np.random.default_rng(31337)
    seen = {"ok": 0, "cyclic": 0, "inconsistent": 0, "cyclic-and-inconsistent": 0}
    for trial in range(40):
        p = random_gate_graph(rng, int(rng.integers(1, 70)), p_dup_out=0.1 if trial % 3 == 0 else 0.0, p_same=0.15,
                              p_cycle=0.08 if trial % 4 == 0 else 0.0)
        seen[_compare(be, orc, p, check_serial=False)] += 1
    assert seen["ok"] > 10 and seen["cyclic"] > 0, seen
    for layers, width, window, seed in ((30, 40, 4, 1), (150, 6, 64, 2), (8, 200, 3, 3)):
        fg = c2a.synth.layered_dag(layers, width, n_in=16, n_const=3, window=window, mix=c2a.synth.MIX_ALL, seed=seed)
        p = dict(lh=fg.lh, rh=fg.rh, out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes, output_nodes=fg.output_nodes)
        assert _compare(be, orc, p, check_serial=False) == "ok"
        assert _compare(be, orc, p, check_serial=False) == "ok"       # the same buffers again (run tags)


def test_deep_chain_exercises_path_string_chunks(backend, orc):
    """A 9000-deep dependency chain with side branches: tree depth > 4096, so path strings span three chunks and
    comparisons go through the chunk links (cprev), including the chunk-boundary ancestor cases."""
    rng = np.random.default_rng(5)
    depth = 9000
    # chain gate k (ids permuted) : out = node 10+k, lh = node 10+k-1 ; side gates hang off random chain nodes
    n_side = 1500
    n = depth + n_side
    perm = rng.permutation(n)
    lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = np.empty(n, np.uint32)
    for k in range(depth):
        g = perm[k]
        out[g] = 10 + k
        lh[g] = 10 + k - 1 if k else 1
        rh[g] = 2 if rng.random() < 0.9 else (10 + int(rng.integers(0, k)) if k else 2)
    for s in range(n_side):
        g = perm[depth + s]
        out[g] = 10 + depth + s
        a = 10 + int(rng.integers(0, depth)); b = 10 + int(rng.integers(0, depth + s))
        lh[g], rh[g] = (a, b) if rng.random() < 0.5 else (b, a)
    p = dict(lh=lh, rh=rh, out=out, op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=10 + n + 1,
             input_nodes=np.array([1, 2], np.uint32), output_nodes=np.array([10 + depth - 1], np.uint32))
    assert _compare(backend, orc, p, check_serial=False) == "ok"
    assert backend.stats()["max_depth"] >= 4096


def test_handoff_arrays_any_number(backend_fifo, orc, c2a):
    """Hand-off of a second claimed producer (c2a_peel.h): entries travel through F ticketed arrays; a waiting wave is
    committed to one slot, so an entry pushed where nobody waits has to be picked up by a wave that sees the backlog.
    Forks are everywhere in a layered DAG; the result must not depend on F."""
    for seed in (1, 2, 3):
        fg = c2a.synth.layered_dag(12, 24, n_in=6, n_const=2, window=3, seed=c2a.synth.SEED + seed)
        p = dict(lh=fg.lh, rh=fg.rh, out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes, output_nodes=fg.output_nodes)
        assert _compare(backend_fifo, orc, p, check_serial=False) == "ok"


def test_wave_per_gate_kernel_small_graphs(backend_wave, orc):
    """Every peel variant (dataflow launch, wave-per-gate and lane-per-gate launch-per-level) on adversarial small graphs."""
    rng = np.random.default_rng(4242)
    seen = {"ok": 0, "cyclic": 0, "inconsistent": 0, "cyclic-and-inconsistent": 0}
    for trial in range(40):
        n = int(rng.integers(1, 48))
        p = random_gate_graph(rng, n, p_dup_out=0.1 if trial % 3 == 0 else 0.0, p_same=0.15,
                              p_cycle=0.08 if trial % 4 == 0 else 0.0)
        seen[_compare(backend_wave, orc, p, check_serial=False)] += 1
    assert seen["ok"] >= 8, seen


def test_wave_per_gate_kernel_high_fanout(backend_wave, orc):
    """One producer read by 150 consumers spread over several DFS roots: blocks of 64 candidate records, rounds of 4
    survivor strings, champion carried across blocks."""
    rng = np.random.default_rng(77)
    n_cons = 150
    chain = 40
    n = 1 + n_cons + chain
    perm = rng.permutation(n)
    lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = np.empty(n, np.uint32)
    hub = perm[0]
    lh[hub], rh[hub], out[hub] = 1, 2, 10                        # the hub gate: node 10
    for k in range(n_cons):                                       # consumers of the hub, some via lh some via rh
        g = perm[1 + k]
        other = 10 + 1 + int(rng.integers(0, k)) if k and rng.random() < 0.7 else 1
        lh[g], rh[g] = (10, other) if rng.random() < 0.5 else (other, 10)
        out[g] = 11 + k
    for k in range(chain):                                        # a chain on top so consumers sit at many depths
        g = perm[1 + n_cons + k]
        lh[g] = 11 + int(rng.integers(0, n_cons)) if k == 0 else 11 + n_cons + k - 1
        rh[g] = 11 + int(rng.integers(0, n_cons))
        out[g] = 11 + n_cons + k
    p = dict(lh=lh, rh=rh, out=out, op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=11 + n + 2,
             input_nodes=np.array([1, 2], np.uint32), output_nodes=np.array([11 + n_cons + chain - 1], np.uint32))
    assert _compare(backend_wave, orc, p, check_serial=False) == "ok"


@pytest.mark.parametrize("shape", ["in-tree", "wide-shallow", "two-chains", "fan-in-star"])
def test_dataflow_peel_shapes(backend, orc, shape):
    """Graph shapes that stress the hand-off and termination logic of the one-launch peel: a complete binary in-tree
    (every gate completes two producers: maximal queue traffic), a wide shallow graph (seeds >> waves), two long
    independent chains (almost every wave idle almost all the time), a star of consumers on one producer."""
    rng = np.random.default_rng(99)
    if shape == "in-tree":
        # gate k consumes the outputs of gates 2k+1 and 2k+2 (heap layout), leaves read inputs; ids permuted
        n = 2047
        perm = rng.permutation(n)
        lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = np.empty(n, np.uint32)
        for k in range(n):
            g = perm[k]
            out[g] = 100 + k
            a, b = 2 * k + 1, 2 * k + 2
            lh[g] = 100 + a if a < n else 1 + (k % 7)
            rh[g] = 100 + b if b < n else 10 + (k % 5)
        p = dict(lh=lh, rh=rh, out=out, op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=100 + n + 1,
                 input_nodes=np.arange(1, 16, dtype=np.uint32), output_nodes=np.array([100], np.uint32))
    elif shape == "wide-shallow":
        n = 6000
        lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = (1000 + np.arange(n)).astype(np.uint32)
        lh[:2000] = rng.integers(1, 50, 2000); rh[:2000] = rng.integers(1, 50, 2000)
        lh[2000:4000] = 1000 + rng.integers(0, 2000, 2000); rh[2000:4000] = 1000 + rng.integers(0, 2000, 2000)
        lh[4000:] = 1000 + rng.integers(2000, 4000, 2000); rh[4000:] = 1000 + rng.integers(0, 4000, 2000)
        perm = rng.permutation(n)
        p = dict(lh=lh[perm], rh=rh[perm], out=out[perm], op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=1000 + n + 1,
                 input_nodes=np.arange(1, 50, dtype=np.uint32), output_nodes=(1000 + np.arange(5990, 6000)).astype(np.uint32))
    elif shape == "two-chains":
        n = 3000
        lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = (10 + np.arange(n)).astype(np.uint32)
        for k in range(n):
            prev = k - 2                                   # chain A = even k, chain B = odd k
            lh[k] = 10 + prev if prev >= 0 else 1
            rh[k] = 2
        perm = rng.permutation(n)
        p = dict(lh=lh[perm], rh=rh[perm], out=out[perm], op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=10 + n + 1,
                 input_nodes=np.array([1, 2], np.uint32), output_nodes=np.array([10 + n - 1, 10 + n - 2], np.uint32))
    else:
        n = 1200
        lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = (10 + np.arange(n)).astype(np.uint32)
        lh[0], rh[0] = 1, 2                                # gate 0 feeds everybody
        for k in range(1, n):
            lh[k] = 10 if k % 3 else 10 + int(rng.integers(0, k))
            rh[k] = 10 + int(rng.integers(0, k))
        p = dict(lh=lh, rh=rh, out=out, op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=10 + n + 1,
                 input_nodes=np.array([1, 2], np.uint32), output_nodes=np.array([10 + n - 1], np.uint32))
    assert _compare(backend, orc, p, check_serial=False) == "ok"


@pytest.mark.parametrize("case", ["own-root", "both-edges-one-sink", "many-sinks", "mixed"])
def test_level_one_gates(backend, orc, case):
    """The gates whose consumers are all sinks are done by a kernel of their own (k_peel_shallow, level 1: the tournament is a minimum
    over (consumer id, edge label), won only by a consumer with a smaller id than the gate's own).  Its cases one by one:
    a gate all of whose consumers have larger ids (a DFS root itself), a sink that reads the same gate on both inputs
    (label 0 wins), a gate with more sink consumers than a wave has lanes, and all of it mixed with deeper gates."""
    rng = np.random.default_rng(5)
    if case == "own-root":
        # gate 0 writes node 10; gates 1..3 (sinks, larger ids) read it
        lh = np.array([1, 10, 10, 2], np.uint32); rh = np.array([2, 1, 10, 10], np.uint32); out = np.array([10, 11, 12, 13], np.uint32)
    elif case == "both-edges-one-sink":
        # gate 3 writes node 10; sink 0 reads it on both inputs, sink 1 on its right input only
        lh = np.array([10, 1, 2, 1], np.uint32); rh = np.array([10, 10, 1, 2], np.uint32); out = np.array([11, 12, 13, 10], np.uint32)
    elif case == "many-sinks":
        # gate 150 writes node 10; 200 sinks around it read it (ids below AND above its own), some on both inputs
        n = 201
        lh = np.full(n, 1, np.uint32); rh = np.full(n, 10, np.uint32); out = (20 + np.arange(n)).astype(np.uint32)
        lh[::7] = 10
        lh[150], rh[150], out[150] = 1, 2, 10
    else:
        # three layers: inputs -> 40 gates -> 40 gates -> 120 sinks, ids permuted
        n = 200
        lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = (100 + np.arange(n)).astype(np.uint32)
        lh[:40] = rng.integers(1, 9, 40); rh[:40] = rng.integers(1, 9, 40)
        lh[40:80] = 100 + rng.integers(0, 40, 40); rh[40:80] = 100 + rng.integers(0, 40, 40)
        lh[80:] = 100 + rng.integers(40, 80, 120); rh[80:] = 100 + rng.integers(0, 80, 120)
        perm = rng.permutation(n)
        lh, rh, out = lh[perm], rh[perm], out[perm]
    n = len(lh)
    p = dict(lh=lh, rh=rh, out=out, op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=int(out.max()) + 2,
             input_nodes=np.arange(1, 9, dtype=np.uint32), output_nodes=np.array([int(out.max())], np.uint32))
    assert _compare(backend, orc, p) == "ok"


@pytest.mark.parametrize("case", ["layered", "binary-tree", "chain-of-pairs", "fan"])
def test_shallow_passes(backend_shallow, orc, c2a, case):
    """k_peel_shallow does the first levels behind the sinks a whole level at once (a lane per gate, integer keys, one-line
    records) with 1, 2, 3, 16 or 48 passes in front of the dataflow launch: a layered random graph (the passes end in the
    middle of it, or swallow it whole); a complete binary tree under ONE sink (levels double inside one region: it overflows
    into the flat seed list); a 40-level ladder whose paths need 40 string bits with both labels; a gate with 300 consumers."""
    rng = np.random.default_rng(17)
    if case == "layered":
        fg = c2a.synth.layered_dag(30, 40, n_in=16, n_const=2, window=4, seed=21)
        p = dict(lh=fg.lh, rh=fg.rh, out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes, output_nodes=fg.output_nodes)
    elif case == "binary-tree":
        depth = 13                                            # gate k reads the outputs of gates 2k+1, 2k+2; gate 0 is the only sink
        n = (1 << depth) - 1
        ids = rng.permutation(n)                              # (gate ids permuted: the DFS order is not the creation order)
        lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = np.empty(n, np.uint32)
        for k in range(n):
            g = ids[k]
            out[g] = 100 + k
            if 2 * k + 2 < n:
                lh[g], rh[g] = 100 + 2 * k + 1, 100 + 2 * k + 2
            else:
                lh[g], rh[g] = 1, 2
        p = dict(lh=lh, rh=rh, out=out, op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=100 + n + 1,
                 input_nodes=np.array([1, 2], np.uint32), output_nodes=np.array([100], np.uint32))
    elif case == "chain-of-pairs":
        # level k holds two gates; each reads both gates of level k + 1 (left / right swapped for the second): every path bit
        # pattern of length <= 40 is in play, and every gate has two consumers whose strings share long prefixes
        L = 40
        n = 2 * L
        ids = rng.permutation(n)
        lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = np.empty(n, np.uint32)
        for k in range(L):
            for j in range(2):
                g = ids[2 * k + j]
                out[g] = 100 + 2 * k + j
                if k + 1 < L:
                    a, b = 100 + 2 * (k + 1), 100 + 2 * (k + 1) + 1
                    lh[g], rh[g] = (a, b) if j == 0 else (b, a)
                else:
                    lh[g], rh[g] = 1, 2
        p = dict(lh=lh, rh=rh, out=out, op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=100 + n + 1,
                 input_nodes=np.array([1, 2], np.uint32), output_nodes=np.array([100, 101], np.uint32))
    else:
        # one gate with 300 consumers of level 1 (each read by a sink of its own): more than the eight loaded together
        n = 1 + 300 + 300
        lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = (100 + np.arange(n)).astype(np.uint32)
        lh[0], rh[0] = 1, 2
        for k in range(300):
            lh[1 + k], rh[1 + k] = (100, 1) if k % 3 else (2, 100)
            lh[301 + k], rh[301 + k] = 100 + 1 + k, 100 + 1 + (k * 7) % 300
        perm = rng.permutation(n)
        inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
        lh, rh, out = lh[perm], rh[perm], out[perm]
        p = dict(lh=lh, rh=rh, out=out, op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=100 + n + 1,
                 input_nodes=np.array([1, 2], np.uint32), output_nodes=np.array([int(out.max())], np.uint32))
    assert _compare(backend_shallow, orc, p) == "ok"


@pytest.mark.gpu
@pytest.mark.parametrize("layers,width,window,seed", [(1500, 200, 8, 11), (300, 1000, 64, 12), (6000, 50, 2, 13), (60, 5000, 30, 14)])
def test_dataflow_peel_mid_size_full_compare(hip_backend, orc, c2a, layers, width, window, seed):
    """300 K-gate graphs of four aspect ratios through the shipped configuration (all 2 048 waves of the dataflow launch
    in play, real concurrency), every output array compared with the oracle element by element, three runs each
    (same buffers: stale data from the previous run must not leak into the next)."""
    fg = c2a.synth.layered_dag(layers, width, n_in=256, n_const=16, window=window, mix=c2a.synth.MIX_ALL, seed=seed)
    p = dict(lh=fg.lh, rh=fg.rh, out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes,
             output_nodes=fg.output_nodes)
    for _ in range(3):
        assert _compare(hip_backend, orc, p, check_serial=False) == "ok"
    assert hip_backend.stats()["levels"] >= layers


def test_error_precedence_inconsistency_before_cycle(backend):
    """compiler.rs checks the input/output clash (:363-383) before it sorts (:408): a payload that has both a dependency
    cycle and a node listed as input AND output must report Inconsistency through c2a_build_circuit."""
    lh = np.array([10, 11], np.uint32); rh = np.array([1, 2], np.uint32); out = np.array([11, 10], np.uint32)   # a 2-cycle
    op = np.zeros(2, np.uint8)
    backend.load_gates(lh, rh, out, op, 12, np.array([1, 2], np.uint32), np.array([2], np.uint32))
    with pytest.raises(Exception) as ei:
        backend.build_circuit()
    assert "Inconsistency" in str(ei.value)
    with pytest.raises(Exception) as ei:                      # the sort alone is topological_sort: it reports the cycle
        backend.topo_sort()
    assert "Cyclic dependency: detected at i=" in str(ei.value)


def test_connection_with_an_unknown_signal_merges_with_the_placeholder_node(c2a, orc):
    """compiler.rs:213-278 with a signal id that no node holds: the scan stays on `(0, &Node::new())`, the known node is
    re-issued under a fresh id and gates that reference node 0 are rewritten — the host mirrors and the oracle agree."""
    def drive(C):
        C.add_signal(0, "0.a", None); C.add_signal(1, "0.b", None); C.add_signal(2, "0.c", None)
        C.add_gate("AAdd", 0, 1, 2)
        C.add_connection(1, 77)                       # 77 was never declared
        return C
    import importlib
    host = drive(importlib.import_module("circom-2-arithc_amd.compiler").Compiler())
    lit = orc.CompilerModel()
    lit.add_signal(0, "0.a", None); lit.add_signal(1, "0.b", None); lit.add_signal(2, "0.c", None)
    lit.add_gate(orc.OP["AAdd"], 0, 1, 2)
    lit.add_connection(1, 77)
    assert [(x.lh_in, x.rh_in, x.out) for x in lit.gates] == [(a, b, c) for _, a, b, c in host.gates] == [(1, 4, 3)]
    assert sorted(lit.nodes) == sorted(host.nodes) == [1, 3, 4]


def test_serial_fallback_when_the_dataflow_launch_gives_up(backend, orc, c2a):
    """A sort that cannot give up (topological_sort.rs:3-21 always terminates on an acyclic graph): when the dataflow launch and
    its retry both end by their watchdog — simulated here by the c2a_debug_peel_abort hook, on the hardware too — c2a_topo_sort sorts
    with the serial DFS instead of failing, the numbering and the emission are the oracle's, and the reverse Kahn levels it
    derives let the level-parallel evaluator run."""
    fg = c2a.synth.layered_dag(25, 12, n_in=8, n_const=3, window=4, mix=tuple(m for m in c2a.synth.MIX_ALL if m[0] != "APow"), seed=77)
    p = dict(lh=fg.lh, rh=fg.rh, out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes, output_nodes=fg.output_nodes)
    be = backend
    be.debug_peel_abort(2)
    assert _compare(be, orc, p, check_serial=False) == "ok"          # first build: both launches "gave up" -> serial DFS
    st = be.stats()
    assert st["levels"] >= 25
    # the evaluator schedules by the levels of the fall-back
    nw, wc = be.assign_wires()
    in0, in1, out, op = be.emit_gates()
    rng = np.random.default_rng(3)
    ins = rng.integers(0, 2 ** 32, (len(fg.input_nodes), 8), dtype=np.uint64)
    consts = {int(nw[nd]): 5 + k for k, nd in enumerate(fg.const_nodes) if nw[nd] != 0xFFFFFFFF}
    vals = np.zeros((wc, 8), np.uint64)
    vals[:len(fg.input_nodes)] = ins
    for w, v in consts.items():
        vals[w] = v
    circ = orc.ArithCircuit(sorted=np.empty(0, np.uint32), in0=in0, in1=in1, out=out, op=op, node_wire=np.empty(0, np.uint32),
                            wire_count=wc, n_in=len(fg.input_nodes), n_out=len(fg.output_nodes))
    orc.eval_arith(circ, 32, vals)
    np.testing.assert_array_equal(be.eval(ins, consts, width=32), vals[wc - len(fg.output_nodes):])
    # the hook is spent: the next build takes the dataflow launch again, same results
    assert _compare(be, orc, p, check_serial=False) == "ok"
    # ONE launch given up: the retry on clean buffers gives the dataflow launch's own result
    be.debug_peel_abort(1)
    assert _compare(be, orc, p, check_serial=False) == "ok"
    # a cyclic graph through the fall-back still reports the reference's message
    be.debug_peel_abort(2)
    cyc = dict(lh=np.array([5, 6, 7], np.uint32), rh=np.array([1, 1, 5], np.uint32), out=np.array([6, 5, 8], np.uint32),
               op=np.zeros(3, np.uint8), n_nodes=9, input_nodes=np.array([1], np.uint32), output_nodes=np.array([8], np.uint32))
    assert _compare(be, orc, cyc, check_serial=False) == "cyclic"
    be.debug_peel_abort(0)


def test_node_ids_without_creation_order(backend, orc, c2a):
    """The relabelling by out-node order finds locality where node ids follow the creation order (compiler.rs:497-500); it must
    stay exact where they carry none: the same layered graphs with the NODE ids permuted as well as the gate ids."""
    for layers, width, seed in ((40, 25, 11), (120, 9, 12)):
        fg = c2a.synth.layered_dag(layers, width, n_in=16, n_const=3, window=8, mix=c2a.synth.MIX_ALL, seed=seed)
        rng = np.random.default_rng(seed)
        perm = rng.permutation(fg.n_nodes).astype(np.uint32)
        p = dict(lh=perm[fg.lh], rh=perm[fg.rh], out=perm[fg.out], op=fg.op, n_nodes=fg.n_nodes, input_nodes=perm[fg.input_nodes],
                 output_nodes=perm[fg.output_nodes])
        assert _compare(backend, orc, p) == "ok"


def test_build_numbers_wrap(backend, orc, c2a):
    """The node-table records are tagged with the number of the build that wrote them (24 bits) instead of being cleared per
    build; when the number wraps the table is cleared once.  Builds across the wrap give the same results (c2a_debug_set_build_no:
    the numbering goes on from two short of the wrap — on the hardware too)."""
    fg = c2a.synth.layered_dag(30, 40, n_in=16, n_const=3, window=6, mix=c2a.synth.MIX_BITWISE, seed=5)       # (1 200 gates: the XCD-aware sweeps too)
    p = dict(lh=fg.lh, rh=fg.rh, out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes, output_nodes=fg.output_nodes)
    be = backend
    be.load_gates(p["lh"], p["rh"], p["out"], p["op"], p["n_nodes"], p["input_nodes"], p["output_nodes"])
    be.debug_set_build_no((1 << 24) - 3)
    exp = orc.build_circuit(p["lh"], p["rh"], p["out"], p["op"], p["n_nodes"], p["input_nodes"], p["output_nodes"], mode=1)
    for _ in range(5):                                       # ... - 2, - 1, wrap -> 1, 2, 3
        np.testing.assert_array_equal(be.topo_sort(), exp.sorted)
    nw, wc = be.assign_wires()
    assert wc == exp.wire_count
    np.testing.assert_array_equal(nw, exp.node_wire)


def _check_fused(be, orc, bm, p, want_path=None):
    """c2a_build_circuit (sort + numbering + emission in one call: the sorted order is written by the emission's split pass
    when the positional numbering runs) against the oracle: sorted ids and node -> wire by checksum, the gates element-wise."""
    args = (p["lh"], p["rh"], p["out"], p["op"], p["n_nodes"], p["input_nodes"], p["output_nodes"])
    exp = orc.build_circuit(*args, mode=1)
    be.load_gates(*args)
    assert be.build_circuit() == exp.wire_count
    st = be.stats()
    if want_path is not None:
        assert st["numbering_path"] == want_path, st
    assert be.checksum("sorted") == bm.checksum_host(exp.sorted)
    nw1 = ((exp.node_wire.astype(np.uint64) + 1) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    assert be.checksum("node_wire1") == bm.checksum_host(nw1)
    in0, in1, out, op = be.emit_gates()
    np.testing.assert_array_equal(in0, exp.in0)
    np.testing.assert_array_equal(in1, exp.in1)
    np.testing.assert_array_equal(out, exp.out)
    np.testing.assert_array_equal(op, exp.op)
    return st


NUMBERING = [pytest.param(("emul", 0), id="emul"), pytest.param(("emul", 1), id="emul-walk"),
             pytest.param(("hip", 0), id="hip", marks=pytest.mark.gpu), pytest.param(("hip", 1), id="hip-walk", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("variant", NUMBERING)
def test_numbering_paths(variant, request, orc, c2a):
    """The positional numbering (wires and gates by formula from the sorted positions + the few events that shift them:
    IO-node outputs, constant-like nodes — src/compiler.rs:423-464) and the walk in sorted order give what the reference's walk
    gives: random graphs with constants read at both operands, inputs and outputs produced by gates, un-produced output nodes
    as operands; few events and many (constants at a tenth of the gates)."""
    import importlib
    from conftest import _Env
    bm = importlib.import_module("circom-2-arithc_amd.backend")
    kind, walk = variant
    with _Env(C2A_NUMBERING_WALK=walk):
        be = c2a.Backend(0, lib_path=request.getfixturevalue("emul_lib")) if kind == "emul" else c2a.Backend(0)
    rng = np.random.default_rng(77)
    seen = {0: 0, 1: 0}
    try:
        for trial in range(120):
            n = int(rng.integers(1, 90))
            p = random_gate_graph(rng, n, p_dup_out=0.0, p_same=0.2, p_cycle=0.0)
            try:
                orc.build_circuit(p["lh"], p["rh"], p["out"], p["op"], p["n_nodes"], p["input_nodes"], p["output_nodes"], mode=1)
            except (orc.CyclicDependency, orc.Inconsistency):
                continue
            st = _check_fused(be, orc, bm, p)
            assert st["numbering_path"] == (0 if walk else 1), st
            seen[st["numbering_path"]] += 1
            _compare(be, orc, p, check_serial=False)         # (the staged calls: positions from the sorted order that exists already)
        # layered graphs: 64 constants + `width` outputs = the events
        for layers, width, n_const in ((40, 20, 8), (30, 100, 64), (12, 3000, 64), (6, 5000, 3000)):
            fg = c2a.synth.layered_dag(layers, width, n_in=32, n_const=n_const, window=4, seed=5)
            p = dict(lh=fg.lh, rh=fg.rh, out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes, output_nodes=fg.output_nodes)
            st = _check_fused(be, orc, bm, p)
            assert st["numbering_path"] == (0 if walk else 1), st
            seen[st["numbering_path"]] += 1
        # reference-shaped graphs (src/process.rs:558-579: a named constant node per literal and context; an output signal per
        # template): a fresh constant at 10-30 % of the gates, an output node at 5-20 % — thousands of events
        for layers, width, cf, of in ((40, 20, 0.3, 0.2), (30, 100, 0.1, 0.05), (5, 2100, 0.1, 0.05), (120, 9, 0.25, 0.1)):
            fg = c2a.synth.layered_dag(layers, width, n_in=32, n_const=4, window=4, seed=9, const_frac=cf, out_frac=of)
            p = dict(lh=fg.lh, rh=fg.rh, out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes, output_nodes=fg.output_nodes)
            st = _check_fused(be, orc, bm, p)
            assert st["numbering_path"] == (0 if walk else 1), st
            assert walk or st["numbering_events"] >= int(0.8 * (cf + of) * fg.n), st
            _compare(be, orc, p, check_serial=False)
        assert seen[1 - walk] > 0 and seen[walk] == 0, seen
        # what the generators above do not make: a gate whose out node is an INPUT node, a constant read at both operands and
        # first seen as rh, an output node nobody produces read as an operand, an output named twice, an operand produced by a
        # gate that hands out no wire
        u32 = lambda *v: np.array(v, dtype=np.uint32)
        p = dict(lh=u32(1, 3, 5, 7, 8, 9, 6), rh=u32(3, 3, 6, 11, 4, 2, 7), out=u32(5, 6, 7, 8, 9, 10, 12),
                 op=np.array([0, 7, 9, 10, 0, 7, 19], dtype=np.uint8), n_nodes=14, input_nodes=u32(1, 2, 7), output_nodes=u32(9, 9, 11))
        for perm in (np.arange(7), np.array([6, 2, 4, 0, 5, 3, 1])):
            q = dict(p, lh=p["lh"][perm], rh=p["rh"][perm], out=p["out"][perm], op=p["op"][perm])
            st = _check_fused(be, orc, bm, q)
            assert st["numbering_path"] == (0 if walk else 1), st
            assert walk or st["numbering_events"] == 2 + 2, st      # gates 2, 4 + constants 3, 4 (counted by the positional numbering only)
            _compare(be, orc, q, check_serial=False)
    finally:
        be.close()


def test_load_gates_argument_errors(backend, orc, c2a):
    """c2a_load_gates validates its payload — node ids must address the node table, op bytes must be AGateType discriminants
    (src/a_gate_type.rs:8-27) — on the DEVICE behind the copy (k_validate; round 4: a host loop over all gates): the FIRST
    offending gate is named, node ids before the op byte of the same gate, the context stays unloaded, and a good payload loads
    right after."""
    fg = c2a.synth.layered_dag(12, 300, n_in=8, n_const=2, window=3, seed=3)       # 3 600 gates: several workgroups
    good = (fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    import importlib as _il
    BackendError = _il.import_module("circom-2-arithc_amd.backend").BackendError

    def bad(which, at, value):
        a = [np.array(x, copy=True) if isinstance(x, np.ndarray) else x for x in good]
        a[which][at] = value
        return a

    cases = [(bad(0, 2777, fg.n_nodes), "node id >= n_nodes at gate 2777"), (bad(1, 5, fg.n_nodes + 7), "node id >= n_nodes at gate 5"),
             (bad(2, 3599, 0xFFFFFFFF), "node id >= n_nodes at gate 3599"), (bad(3, 1234, 20), "unknown gate type at gate 1234"),
             (bad(3, 0, 255), "unknown gate type at gate 0")]
    # two offenders: the first one is reported; a bad op and a bad id in ONE gate: the id
    two = bad(3, 3000, 99); two[0][100] = fg.n_nodes
    cases.append((two, "node id >= n_nodes at gate 100"))
    both = bad(3, 40, 77); both[2][40] = fg.n_nodes + 1
    cases.append((both, "node id >= n_nodes at gate 40"))
    for args, msg in cases:
        with pytest.raises(BackendError, match=msg):
            backend.load_gates(*args)
        with pytest.raises(BackendError, match="no gates loaded"):
            backend.topo_sort()
    with pytest.raises(BackendError, match="input node id >= n_nodes"):
        backend.load_gates(*good[:5], np.array([fg.n_nodes], np.uint32), fg.output_nodes)
    with pytest.raises(BackendError, match="output node id >= n_nodes"):
        backend.load_gates(*good[:5], fg.input_nodes, np.array([1, fg.n_nodes + 3], np.uint32))
    backend.load_gates(*good)
    exp = orc.build_circuit(*good, mode=1)
    np.testing.assert_array_equal(backend.topo_sort(), exp.sorted)
    # "a node is both an input and an output" is found on the device at load time too and reported by build_circuit BEFORE the sort
    clash = list(good)
    clash[6] = np.concatenate([fg.output_nodes, fg.input_nodes[:1]])
    backend.load_gates(*clash)
    with pytest.raises(c2a.Inconsistency):
        backend.build_circuit()
