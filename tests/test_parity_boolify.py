"""Parity of the HIP boolify map (c2a_boolify, replacing boolify(&circuit, w) of src/main.rs:30-32) with the
oracle's procedural restatement of the frozen bit-blast spec — bit-exact SoA — plus functional
equivalence with the arithmetic circuit under the semantics of tests/integration.rs:94-115 (mod 2^w).
(The boolify crate itself is absent: gate-level parity with it is unpinned.)"""
import numpy as np
import pytest


def _load(be, fg):
    be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    be.build_circuit()


def _oracle(orc, fg):
    return orc.build_circuit(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes, mode=1)


def test_template_sizes_match_spec(backend, orc):
    for w in (1, 2, 3, 8, 31, 32, 64):
        for op in range(20):
            if op == orc.OP["APow"] and w > 32:
                continue
            assert backend.template_size(op, w) == orc.template_size(op, w), (orc.OP_NAMES[op], w)


@pytest.mark.parametrize("width", [1, 2, 7, 8, 32, 64])
def test_boolify_bit_exact(backend, orc, c2a, width):
    mix = c2a.synth.MIX_ALL if width <= 8 else tuple(m for m in c2a.synth.MIX_ALL if m[0] != "APow")
    fg = c2a.synth.layered_dag(12, 24, n_in=16, n_const=4, window=4, mix=mix, seed=100 + width)
    _load(backend, fg)
    info = backend.boolify(width)
    exp_c = _oracle(orc, fg)
    exp = orc.boolify(exp_c, width)
    assert info.n_gates == len(exp.in0)
    assert info.wire_count == exp.wire_count
    assert (info.n_in, info.n_out, info.m_wires) == (exp_c.n_in, exp_c.n_out, exp_c.wire_count - exp_c.n_out)
    in0, in1, out, op = backend.bool_read()
    np.testing.assert_array_equal(op, exp.op)
    np.testing.assert_array_equal(in0, exp.in0)
    np.testing.assert_array_equal(in1, exp.in1)
    np.testing.assert_array_equal(out, exp.out)
    # ranged read-back
    a = backend.bool_read(5, 11)
    np.testing.assert_array_equal(a[2], exp.out[5:16])


@pytest.mark.parametrize("width", [8, 32])
def test_boolean_circuit_computes_the_arithmetic_circuit(backend, orc, c2a, width):
    """64 random input vectors: every arithmetic wire's value == its w boolean wires."""
    mix = tuple(m for m in c2a.synth.MIX_ALL if m[0] != "APow") if width > 8 else c2a.synth.MIX_ALL
    fg = c2a.synth.layered_dag(10, 16, n_in=8, n_const=3, window=3, mix=mix, seed=width)
    _load(backend, fg)
    nw, wc = backend.assign_wires()
    in0, in1, out, op = backend.emit_gates()
    info = backend.boolify(width)
    b = backend.bool_read()
    rng = np.random.default_rng(width)
    T = 64
    mask = (1 << width) - 1
    free = [int(nw[n]) for n in list(fg.input_nodes) + list(fg.const_nodes) if nw[n] != 0xFFFFFFFF]
    circ = orc.ArithCircuit(sorted=np.empty(0, np.uint32), in0=in0, in1=in1, out=out, op=op,
                            node_wire=np.empty(0, np.uint32), wire_count=wc, n_in=len(fg.input_nodes),
                            n_out=len(fg.output_nodes))
    vals = np.zeros((wc, T), np.uint64)
    for W in free:
        vals[W] = rng.integers(0, 2 ** 63, T, dtype=np.uint64) & np.uint64(mask)
        vals[W, :4] = [0, mask, 1, mask >> 1]
    orc.eval_arith(circ, width, vals)
    bw = np.zeros(info.wire_count, np.uint64)
    for W in free:
        for bit in range(width):
            word = 0
            for t in range(T):
                word |= ((int(vals[W, t]) >> bit) & 1) << t
            bw[int(info.wire(W, bit))] = word
    bc = orc.BoolCircuit(in0=b[0], in1=b[1], out=b[2], op=b[3], wire_count=info.wire_count, width=width, n_in=0, n_out=0)
    orc.eval_bool(bc, bw)
    produced = np.unique(out)
    for W in produced.tolist():
        for t in (0, 1, 2, 3, 17, 63):
            got = 0
            for bit in range(width):
                got |= ((int(bw[int(info.wire(W, bit))]) >> t) & 1) << bit
            assert got == int(vals[W, t]), (W, t)


@pytest.mark.parametrize("width", [8, 32])
def test_evaluator_with_caller_inputs(backend, orc, c2a, width):
    """c2a_eval — the reference's simulation harness (tests/integration.rs:191-237: inputs in, outputs out) on the GPU: 64
    caller-supplied vectors through the arithmetic circuit and through its boolean image give the oracle's outputs."""
    mix = tuple(m for m in c2a.synth.MIX_ALL if m[0] != "APow") if width > 8 else c2a.synth.MIX_ALL
    fg = c2a.synth.layered_dag(12, 14, n_in=8, n_const=3, window=3, mix=mix, seed=100 + width)
    _load(backend, fg)
    nw, wc = backend.assign_wires()
    in0, in1, out, op = backend.emit_gates()
    rng = np.random.default_rng(width)
    mask = (1 << width) - 1
    n_in, n_out = len(fg.input_nodes), len(fg.output_nodes)
    T = 64
    vals = np.zeros((wc, T), np.uint64)
    ins = rng.integers(0, 2 ** 63, (n_in, T), dtype=np.uint64) & np.uint64(mask)
    ins[:, :4] = [0, mask, 1, mask >> 1]
    for i, nd in enumerate(fg.input_nodes):
        assert int(nw[nd]) == i                              # inputs are wires 0 .. n_in-1 in list order
        vals[i] = ins[i]
    consts = {}
    for k, nd in enumerate(fg.const_nodes):
        if nw[nd] != 0xFFFFFFFF:
            consts[int(nw[nd])] = (0x9E3779B97F4A7C15 * (k + 1)) & mask
            vals[int(nw[nd])] = consts[int(nw[nd])]
    circ = orc.ArithCircuit(sorted=np.empty(0, np.uint32), in0=in0, in1=in1, out=out, op=op, node_wire=np.empty(0, np.uint32),
                            wire_count=wc, n_in=n_in, n_out=n_out)
    orc.eval_arith(circ, width, vals)
    want = vals[wc - n_out:]
    got = backend.eval(ins, consts, width=width)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(backend.eval(ins[:, :5], consts, width=width), want[:, :5])     # fewer vectors
    backend.boolify(width)
    np.testing.assert_array_equal(backend.eval(ins, consts, boolean=True), want)
    with pytest.raises(c2a.BackendError):
        backend.eval(np.zeros((n_in, 65), np.uint64), consts, width=width)
    with pytest.raises(c2a.BackendError):
        backend.eval(ins, {wc + 5: 1}, width=width)


@pytest.mark.parametrize("width", [3, 8, 32])
def test_prune_pass(backend, orc, c2a, width):
    """c2a_boolify_prune (optional: constant folding + dead-gate removal, SURVEY C.2's second half of `boolify`): gate for
    gate the oracle's sequential twin, fewer gates than the per-gate map, and the same outputs on 64 vectors (c2a_eval)."""
    mix = tuple(m for m in c2a.synth.MIX_ALL if m[0] != "APow") if width > 8 else c2a.synth.MIX_ALL
    fg = c2a.synth.layered_dag(9, 15, n_in=8, n_const=3, window=3, mix=mix, seed=400 + width)
    _load(backend, fg)
    nw, wc = backend.assign_wires()
    backend.emit_gates()
    arith = _oracle(orc, fg)
    info = backend.boolify(width)
    eb = orc.boolify(arith, width)
    want, wcnt = orc.prune_bool(eb, int(info.wire(wc - len(fg.output_nodes))))
    pi = backend.boolify_prune()
    assert {k: pi[k] for k in wcnt} == wcnt
    assert pi["n_gates"] - 2 + pi["n_folded"] + pi["n_dead"] == info.n_gates and pi["n_gates"] < info.n_gates
    got = backend.pruned_read()
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    # same function: 64 vectors through the arithmetic circuit, the per-gate map and the pruned circuit
    rng = np.random.default_rng(width)
    mask = (1 << width) - 1
    ins = rng.integers(0, 2 ** 63, (len(fg.input_nodes), 64), dtype=np.uint64) & np.uint64(mask)
    ins[:, :4] = [0, mask, 1, mask >> 1]
    consts = {int(nw[nd]): (0x9E3779B97F4A7C15 * (k + 1)) & mask for k, nd in enumerate(fg.const_nodes) if nw[nd] != 0xFFFFFFFF}
    ref = backend.eval(ins, consts, width=width)
    np.testing.assert_array_equal(backend.eval(ins, consts, boolean=True), ref)
    np.testing.assert_array_equal(backend.eval(ins, consts, pruned=True), ref)
    # a new boolify invalidates the pruned circuit
    backend.boolify(width)
    with pytest.raises(c2a.BackendError):
        backend.pruned_read(0, 1)


@pytest.mark.parametrize("kind", ["emul", pytest.param("hip", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("slices", [1, 3, 7, 64])
def test_boolify_slices(request, orc, c2a, kind, slices):
    """k_boolify with several workgroups per chunk of 256 arithmetic gates (c2a_kernels.h, SLICES: what a circuit of a few hundred
    multipliers gets — the Poseidon-shaped config went from 0.20 to 0.015 ms): any number of slices, the same boolean circuit — whole
    groups of 16 gates per slice, the unaligned ends with the first; more slices than groups leaves some with nothing"""
    from conftest import _Env
    mix = tuple(m for m in c2a.synth.MIX_ALL if m[0] != "APow")
    fg = c2a.synth.layered_dag(9, 37, n_in=8, n_const=3, window=3, mix=mix, seed=31 + slices)
    with _Env(C2A_BOOL_SLICES=slices):
        be = c2a.Backend(0, lib_path=request.getfixturevalue("emul_lib")) if kind == "emul" else c2a.Backend(0)
    try:
        for width in (5, 16):
            _load(be, fg)
            info = be.boolify(width)
            exp = orc.boolify(_oracle(orc, fg), width)
            assert info.n_gates == len(exp.in0) and info.wire_count == exp.wire_count
            for g, e in zip(be.bool_read(), (exp.in0, exp.in1, exp.out, exp.op)):
                np.testing.assert_array_equal(g, e)
    finally:
        be.close()


def test_boolify_refuses_a_circuit_whose_boolean_wire_ids_do_not_fit_u32(backend, c2a):
    """c2a_boolify knows the boolean circuit's size from the gate types alone (no allocation, no kernel): 10 000 APow gates at width 64
    need more than 2^32 boolean wires — the Bristol fashion's wire ids, like the `boolify` crate's usize on a 32-bit id space, stop there:
    C2A_ERR_OVERFLOW with its message, and the context stays usable (measured on the hardware with a 130 x 130 matrix product at
    width 32: 6.5 G boolean gates refused; 100 x 100 — 2.98 G gates, 39 GB — bit-blast in 9.1 ms and verified: tools/big_boolify_check.py)"""
    fg = c2a.synth.layered_dag(10, 1000, n_in=8, n_const=2, window=3, mix=(("APow", 1),), seed=77)
    _load(backend, fg)
    with pytest.raises(OverflowError, match="exceed u32"):
        backend.boolify(64)
    info = backend.boolify(1)                 # (the same context, a width that fits)
    assert info.n_gates > 0 and info.width == 1


@pytest.mark.parametrize("width", [8, 32, 64])
def test_boolify_of_a_circuit_the_host_built(backend, orc, c2a, width):
    """c2a_load_circuit: the second half of the path alone — `boolify(&circuit, width)` (src/main.rs:30-32) on a BristolCircuit the
    host built itself (here: the oracle's build_circuit, standing in for the reference's own src/compiler.rs:321-494).  Same boolean
    circuit as after c2a_build_circuit on the gate graph, bit for bit; the calls that need the gate graph refuse."""
    mix = tuple(m for m in c2a.synth.MIX_ALL if m[0] != "APow")
    fg = c2a.synth.layered_dag(14, 30, n_in=16, n_const=4, window=4, mix=mix, seed=300 + width)
    circ = _oracle(orc, fg)
    backend.load_circuit(circ.in0, circ.in1, circ.out, circ.op, circ.wire_count, circ.n_in, circ.n_out)
    info = backend.boolify(width)
    exp = orc.boolify(circ, width)
    assert info.n_gates == len(exp.in0) and info.wire_count == exp.wire_count
    assert (info.n_in, info.n_out, info.m_wires) == (circ.n_in, circ.n_out, circ.wire_count - circ.n_out)
    for a, b in zip(backend.bool_read(), (exp.in0, exp.in1, exp.out, exp.op)):
        np.testing.assert_array_equal(a, b)
    # in chunks too, and as text
    info2 = backend.boolify_plan(width)
    assert info2.n_gates == info.n_gates
    q0, chunk = backend.boolify_chunk(5, 40)
    sl, g0 = orc.boolify_range(circ, width, 5, 40)
    assert q0 == g0
    for a, b in zip(chunk, (sl.in0, sl.in1, sl.out, sl.op)):
        np.testing.assert_array_equal(a, b)
    e_in0, e_in1, e_out, e_op = backend.emit_gates()                       # (the circuit as it was handed over)
    np.testing.assert_array_equal(e_out, circ.out)
    for call in (backend.topo_sort, backend.build_circuit, backend.assign_wires, lambda: backend.topo_sort(serial=True)):
        with pytest.raises(c2a.BackendError):
            call()
    # argument errors: a wire beyond wire_count, an unknown gate type
    bad = circ.in1.copy(); bad[7] = circ.wire_count
    with pytest.raises(c2a.BackendError, match="wire id >= wire_count at gate 7"):
        backend.load_circuit(circ.in0, bad, circ.out, circ.op, circ.wire_count, circ.n_in, circ.n_out)
    bop = circ.op.copy(); bop[3] = 20
    with pytest.raises(c2a.BackendError, match="unknown gate type at gate 3"):
        backend.load_circuit(circ.in0, circ.in1, circ.out, bop, circ.wire_count, circ.n_in, circ.n_out)
    # ... and the gate graph can be loaded into the same context afterwards
    _load(backend, fg)
    assert backend.boolify(width).n_gates == info.n_gates


def test_boolify_empty_circuit(backend):
    e = np.empty(0, np.uint32)
    backend.load_gates(e, e, e, np.empty(0, np.uint8), 4, [1], [2])
    assert backend.build_circuit() == 2
    info = backend.boolify(16)
    assert (info.n_gates, info.wire_count, info.aux_total) == (0, 32, 0)
    # the evaluator, the verifier and the prune pass on a circuit without gates: the output is a wire nothing drives (0)
    assert backend.eval(np.array([[5, 9]], np.uint64), {}, width=16).tolist() == [[0, 0]]
    assert backend.eval(np.array([[5, 9]], np.uint64), {}, boolean=True).tolist() == [[0, 0]]
    assert backend.verify_boolify(3) == (2 * 64, 0)
    pi = backend.boolify_prune()
    assert (pi["n_gates"], pi["n_folded"], pi["n_dead"]) == (2, 0, 0)          # just the two constant gates
    assert backend.eval(np.array([[5, 9]], np.uint64), {}, pruned=True).tolist() == [[0, 0]]


def test_call_order_is_enforced(backend, c2a):
    e = np.empty(0, np.uint32)
    backend.load_gates(e, e, e, np.empty(0, np.uint8), 2, [], [])
    with pytest.raises(c2a.BackendError):
        backend.boolify(8)
    with pytest.raises(c2a.BackendError):
        backend.emit_gates()
    backend.build_circuit()
    with pytest.raises(c2a.BackendError):
        backend.boolify(0)
    with pytest.raises(c2a.BackendError):
        backend.boolify(65)


def test_evaluator_template_chunks_and_more_gates_than_waves(backend, c2a):
    """The evaluator runs a template on one wave, 64 boolean gates at a time, values forwarded between the lanes of a chunk and
    through memory between chunks.  One level of 40 independent gates of every template length — among them the dividers,
    whose last chunk holds a single gate (8 193 = 128 x 64 + 1) — on more gates than the emulated launch has waves, so that
    waves run a second template after a long first one: every wire must agree with the arithmetic circuit."""
    ops = [o for o in range(20) if o != 11]
    n = 40
    lh = np.array([1 + (k % 3) for k in range(n)], np.uint32)
    rh = np.array([2 + (k % 5) for k in range(n)], np.uint32)
    out = (10 + np.arange(n)).astype(np.uint32)
    op = np.array([ops[(7 * k) % len(ops)] for k in range(n)], np.uint8)
    backend.load_gates(lh, rh, out, op, 10 + n + 1, list(range(1, 8)), list(out[-3:]))
    backend.build_circuit()
    backend.boolify(32)
    for seed in (1, 2, 3):
        checked, bad = backend.verify_boolify(seed=seed)
        assert checked == backend.wire_count * 64 and bad == 0


@pytest.mark.parametrize("width", [1, 8, 32, 64])
def test_gpu_verifier_agrees_and_detects_faults(backend, c2a, width):
    """c2a_verify_boolify = the reference's simulation harness (tests/integration.rs:191-237) as kernels: every
    arithmetic wire x 64 vectors against its boolean wires.  It must report 0 on the real circuit and > 0 as soon
    as one boolean gate is corrupted."""
    mix = c2a.synth.MIX_ALL if width <= 8 else tuple(m for m in c2a.synth.MIX_ALL if m[0] != "APow")
    # (under the emulation a divider at width 64 is 33 000 boolean gates walked by fibres: a third of the gates there, all of them on the hardware)
    layers = 5 if width >= 32 and "emulation" in backend.version else 14
    fg = c2a.synth.layered_dag(layers, 18, n_in=8, n_const=3, window=3, mix=mix, seed=900 + width)
    _load(backend, fg)
    backend.boolify(width)
    checked, bad = backend.verify_boolify(seed=42)
    assert checked == backend.wire_count * 64 and bad == 0
    # fault detection on the linear-size templates (in a multiplier / divider / power template a single flipped
    # gate is often masked for all 64 vectors, so those are not a fair detector test)
    fg = c2a.synth.layered_dag(14, 18, n_in=8, n_const=3, window=3, mix=c2a.synth.MIX_BITWISE, seed=950 + width)
    _load(backend, fg)
    backend.boolify(width)
    assert backend.verify_boolify(seed=42)[1] == 0
    in0, in1, out, op = backend.bool_read()
    # corrupt single gates (XOR <-> AND) spread over the circuit; a few may sit in logic that no vector excites,
    # but most must be caught
    cand = np.nonzero(op < 2)[0]
    picks = cand[np.linspace(0, len(cand) - 1, num=min(12, len(cand)), dtype=int)]
    caught = 0
    for k in picks.tolist():
        backend.debug_patch_bool_op(k, 1 - int(op[k]))
        caught += backend.verify_boolify(seed=42)[1] > 0
        backend.debug_patch_bool_op(k, int(op[k]))
    assert caught >= (len(picks) + 1) // 2, (caught, len(picks))
    assert backend.verify_boolify(seed=42)[1] == 0


@pytest.mark.parametrize("width", [3, 32])
def test_chunked_boolify_equals_the_full_result(backend, orc, c2a, width):
    """c2a_boolify_plan + c2a_boolify_chunk (streaming emission / sharding by sorted-position range): every chunk is
    bit-identical to the matching slice of the full circuit, whatever the chunk boundaries."""
    mix = tuple(m for m in c2a.synth.MIX_ALL if m[0] != "APow")
    fg = c2a.synth.layered_dag(11, 23, n_in=8, n_const=3, window=3, mix=mix, seed=77 + width)
    _load(backend, fg)
    exp = orc.boolify(_oracle(orc, fg), width)
    info = backend.boolify_plan(width)
    assert info.n_gates == len(exp.in0) and info.wire_count == exp.wire_count
    n = fg.n
    cuts = [0, 1, 2, 7, n // 3, n // 3 + 1, n - 5, n]
    total = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        q0, got = backend.boolify_chunk(a, b - a)
        assert q0 == total
        for x, y in zip(got, (exp.in0, exp.in1, exp.out, exp.op)):
            np.testing.assert_array_equal(x, y[q0:q0 + len(x)])
        total += len(got[0])
    assert total == info.n_gates
    q0, got = backend.boolify_chunk(5, 0)                  # empty chunk
    assert len(got[0]) == 0
    with pytest.raises(c2a.BackendError):
        backend.boolify_chunk(n - 1, 2)
    # the full map still works after chunking, and chunking needs a plan
    assert backend.boolify(width).n_gates == info.n_gates
    backend.build_circuit()
    with pytest.raises(c2a.BackendError):
        backend.boolify_chunk(0, 1)


MULTI = [pytest.param(("emul", [0, 1]), id="emul-2dev"), pytest.param(("emul", [0, 1, 1]), id="emul-3shards"),
         pytest.param(("emul-nopeer", [0, 1, 1]), id="emul-3shards-no-peer-access"),      # (the devices cannot map each other: the pieces are gathered)
         pytest.param(("hip", [0, 0]), id="hip-2shards", marks=pytest.mark.gpu),
         pytest.param(("hip", [0, 0, 0, 0, 0]), id="hip-5shards", marks=pytest.mark.gpu),
         # real peers (hipSetDevice switching, cross-device hipMemcpyPeerAsync, per-device allocation): run wherever the box
         # has that many GPUs, skipped on the 1-GPU test boxes
         pytest.param(("hip", [0, 1]), id="hip-2gpus", marks=pytest.mark.gpu),
         pytest.param(("hip", [1, 0, 1]), id="hip-2gpus-3shards", marks=pytest.mark.gpu),
         pytest.param(("hip", [0, 1, 2, 3]), id="hip-4gpus", marks=pytest.mark.gpu),
         pytest.param(("hip", list(range(8))), id="hip-8gpus", marks=pytest.mark.gpu)]


@pytest.fixture(params=MULTI)
def multi_backend(request, c2a):
    """A multi-device context (c2a_create(n_devices, device_ids)).  A box with one GPU lists it several
    times: every listed id gets its own stream, buffers and shard, so the peer-copy / shard / gather code runs for real."""
    kind, ids = request.param
    if kind == "hip" and max(ids) >= c2a.visible_devices():
        pytest.skip(f"needs {max(ids) + 1} GPUs, this box has {c2a.visible_devices()}")
    if kind == "emul-nopeer":
        from conftest import _Env
        with _Env(HIPEMU_NO_PEER=1):
            be = c2a.Backend(ids, lib_path=request.getfixturevalue("emul_lib"))
    else:
        be = c2a.Backend(ids, lib_path=request.getfixturevalue("emul_lib")) if kind == "emul" else c2a.Backend(ids)
    be._test_shards = len(ids)
    yield be
    be.close()


@pytest.mark.parametrize("width", [3, 32])
def test_multi_device_boolify_equals_the_single_device_result(multi_backend, orc, c2a, width):
    """boolify cut by sorted-position range over the devices of the context: the gathered SoA, ranged reads across shard
    boundaries and the (additive) checksums are those of the whole circuit."""
    be = multi_backend
    if width == 32 and "emulation" in be.version and be._test_shards > 2:
        pytest.skip("the emulator takes 45 s for this circuit at width 32: the two-device context covers it there, every shard count runs on the hardware")
    mix = tuple(m for m in c2a.synth.MIX_ALL if m[0] != "APow")
    # (the emulator walks a width-32 divider's 8 193 gates by fibres: a smaller circuit there, this one on the hardware)
    layers, wd = (6, 14) if width == 32 and "emulation" in be.version else (14, 23)
    fg = c2a.synth.layered_dag(layers, wd, n_in=16, n_const=4, window=4, mix=mix, seed=7 + width)
    for rerun in range(2):                                  # same buffers twice
        _load(be, fg)
        info = be.boolify(width)
        exp = orc.boolify(_oracle(orc, fg), width)
        assert info.n_gates == len(exp.in0) and info.wire_count == exp.wire_count
        got = be.bool_read()
        for g, e in zip(got, (exp.in0, exp.in1, exp.out, exp.op)):
            np.testing.assert_array_equal(g, e)
        n = info.n_gates
        for first, count in ((0, 1), (n // 3 - 2, 9), (n // 2 - 5, n // 3), (n - 4, 4)):
            part = be.bool_read(first, count)
            np.testing.assert_array_equal(part[0], exp.in0[first:first + count])
            np.testing.assert_array_equal(part[3], exp.op[first:first + count])
        backend_mod = __import__("importlib").import_module("circom-2-arithc_amd.backend")
        for name, arr in (("bool_in0", exp.in0), ("bool_in1", exp.in1), ("bool_out", exp.out), ("bool_op", exp.op)):
            assert be.checksum(name) == backend_mod.checksum_host(arr), name
    # the verifier of a multi-device context: every device checks the gates it holds (a wave per arithmetic gate: 64 vectors
    # through its boolean gates out of a private scratch, every wire they name must be the gate's own) — nothing is gathered;
    # a flipped op anywhere — first gate, a gate of every shard, last gate — is caught where it lies
    checked, bad = be.verify_boolify(1)
    assert bad == 0 and checked == fg.n * 64
    assert be.stats()["verifier"] == 2                        # (the per-device local check, not the whole-circuit simulation: c2a.h c2a_stats)
    if width <= 8 or multi_backend.version.find("emulation") < 0:      # (the emulator takes seconds per pass at width 32: once is enough there)
        ops = be.bool_read()[3]
        rng = np.random.default_rng(width)
        caught = tried = 0
        for k in sorted(set(rng.integers(0, n, size=8).tolist()) | {0, n - 1}):       # (gates of every shard)
            if ops[k] == 2:
                continue                                      # (INV -> anything changes the arity: not a fair fault)
            be.debug_patch_bool_op(k, 1 - int(ops[k]))
            caught += be.verify_boolify(1)[1] > 0
            tried += 1
            be.debug_patch_bool_op(k, int(ops[k]))
        # (a flip inside a multiplier / divider template, or in zero-filled logic, is often masked for all 64 vectors: cf.
        # test_gpu_verifier_agrees_and_detects_faults, which uses the linear-size templates for its detection rate)
        assert tried >= 4 and caught >= 1, (caught, tried)
        assert be.verify_boolify(1)[1] == 0
    # the evaluator of the boolean image and the prune pass simulate the circuit level by level across all its gates, on the primary
    # device: they read every device's piece where it lies (peer access; gathered on the primary only where the devices cannot map
    # each other) and give the single-device answers
    rng = np.random.default_rng(11)
    vec = rng.integers(0, 2 ** 63, (len(fg.input_nodes), 5), dtype=np.uint64) & np.uint64((1 << width) - 1)
    want_vals = be.eval(vec, {}, width=width)
    np.testing.assert_array_equal(be.eval(vec, {}, width=width, boolean=True), want_vals)
    pi = be.boolify_prune()
    np.testing.assert_array_equal(be.eval(vec, {}, width=width, pruned=True), want_vals)
    want, wcnt = orc.prune_bool(exp, int(info.wire(_oracle(orc, fg).wire_count - len(fg.output_nodes))))
    assert {k: pi[k] for k in wcnt} == wcnt
    for g, e in zip(be.pruned_read(), want):
        np.testing.assert_array_equal(g, e)
    lines = orc.bristol_text_of(exp).encode().split(b"\n")[4:]
    assert be.format_bristol(1, n // 3, 50) == b"\n".join(lines[n // 3:n // 3 + 50]) + b"\n"
    # circuit.txt of a circuit spread over several devices: streamed chunk by chunk, the text of the single-device writer
    import io
    bristol = __import__("importlib").import_module("circom-2-arithc_amd.bristol")
    info_c = bristol.CircuitInfo({f"in{i}": i for i in range(info.n_in)}, {}, {f"out{i}": i for i in range(info.n_out)})
    lazy = bristol.BristolCircuit(wire_count=info.wire_count, info=info_c, in0=np.empty(0, np.uint32), in1=np.empty(0, np.uint32),
                                  out=np.empty(0, np.uint32), op=np.empty(0, np.uint8), op_names=bristol.BOOL_OP_NAMES if hasattr(bristol, "BOOL_OP_NAMES") else ["XOR", "AND", "INV"],
                                  io_widths=([width] * info.n_in, [width] * info.n_out), unary_ops=(2,), gates_on_device=info.n_gates)
    buf = io.BytesIO()
    lazy.write_bristol_gpu(buf, be, chunk_gates=4096)
    eb = orc.boolify(_oracle(orc, fg), width)
    assert buf.getvalue() == orc.bristol_text_of(eb).encode()


def test_verifier_refuses_stale_level_data(backend, c2a):
    """c2a_topo_sort_serial leaves no reverse Kahn levels behind: the verifier must say so instead of scheduling by the
    level data of an earlier circuit."""
    fg = c2a.synth.layered_dag(6, 10, n_in=6, n_const=2, window=2, seed=3)
    _load(backend, fg)
    backend.boolify(8)
    assert backend.verify_boolify(5)[1] == 0
    backend.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    backend.topo_sort(serial=True)
    backend.assign_wires(); backend.emit_gates(); backend.boolify(8)
    with pytest.raises(c2a.BackendError):
        backend.verify_boolify(5)
    with pytest.raises(c2a.BackendError):                   # and a chunk needs a plan made for THIS circuit
        backend.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
        backend.build_circuit()
        backend.boolify_chunk(0, 4)


def test_level_parallel_passes_refuse_duplicate_writers(backend, c2a):
    """Two gates writing one node (the reference keeps the last, compiler.rs:403-406): the non-producer writer carries no
    dependency edge, so the level-parallel passes would race it against the wire's readers.  c2a_eval, c2a_verify_boolify and
    c2a_boolify_prune say so instead of returning a schedule-dependent result; sort, numbering, emission and boolify still work."""
    # g0: n2 = n0 + n1, g1: n2 = n0 * n1 (the producer: last writer), g2: n3 = n2 + n0
    lh = np.array([10, 10, 12], np.uint32); rh = np.array([11, 11, 10], np.uint32); out = np.array([12, 12, 13], np.uint32)
    backend.load_gates(lh, rh, out, np.array([0, 7, 0], np.uint8), 14, np.array([10, 11], np.uint32), np.array([13], np.uint32))
    backend.build_circuit()
    backend.boolify(8)
    for call in (lambda: backend.eval(np.array([5, 7], np.uint64), {}, width=8), lambda: backend.eval(np.array([5, 7], np.uint64), {}, boolean=True),
                 lambda: backend.verify_boolify(seed=1), lambda: backend.boolify_prune()):
        with pytest.raises(c2a.BackendError) as ei:
            call()
        assert "two gates write one node" in str(ei.value)
