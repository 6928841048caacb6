"""HUBS — produced nodes that many gates read — and the graph families beyond synth.layered_dag.

The reference's deps closure allows any fan-out and any distance (/root/reference/src/compiler.rs:408-421); its DFS walks
them all the same (src/topological_sort.rs:42-44).  Here a gate with more than 8 consumers gets a tree of RELAYS
(circom-2-arithc_amd/csrc/c2a_peel.h HUBS AND RELAYS): virtual gates that compare its consumers eight at a time.  Every
result must stay bit-identical to the oracle's — sorted ids, node -> wire, emitted gates —, the reverse Kahn levels must not
count the relays, and the relay count is a function of the fan-outs alone."""
import importlib
import os

import numpy as np
import pytest

from conftest import BACKENDS, _Env  # noqa: F401
from test_parity_build_circuit import _compare

RELAY_FAN, HUB_MIN = 8, 8


def relay_count(N):
    """mirror of c2a_peel.h relay_count (the test's own arithmetic, not the library's)"""
    c = -(-N // RELAY_FAN)
    t = c
    while c > HUB_MIN:
        c = -(-c // RELAY_FAN)
        t += c
    return t


def expected_relays(lh, rh, out):
    """sum over the produced nodes with more than HUB_MIN consumers (one edge per consumer: lh == rh counts once)"""
    prod = {}
    for g, o in enumerate(out.tolist()):
        prod[o] = g                                              # (last writer wins, compiler.rs:403-406)
    cnt = {}
    for a, b in zip(lh.tolist(), rh.tolist()):
        for nd in ({a, b}):
            if nd in prod:
                cnt[nd] = cnt.get(nd, 0) + 1
    return sum(relay_count(c) for c in cnt.values() if c > HUB_MIN)


def _payload(fg):
    return dict(lh=fg.lh, rh=fg.rh, out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes, output_nodes=fg.output_nodes)


def _star(rng, n_cons, chain, hub_gate_first, deep_tail=0):
    """one hub (node 10) read by n_cons gates — over lh or rh, some of them reading each other as well —, a chain on top so
    that the consumers sit at many depths, optionally a long chain BELOW the hub (the hub's own producers: what waits for it)"""
    n = 1 + n_cons + chain + deep_tail
    perm = rng.permutation(n)
    if hub_gate_first:                                           # the hub gets the smallest gate id: it is its own DFS root
        j = int(np.where(perm == 0)[0][0]); perm[0], perm[j] = perm[j], perm[0]
    lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = np.empty(n, np.uint32)
    base = 10 + deep_tail                                        # node of the hub
    for k in range(deep_tail):                                   # producers below the hub: a chain ending in it
        g = perm[1 + n_cons + chain + k]
        lh[g] = 10 + k - 1 if k else 1
        rh[g] = 2
        out[g] = 10 + k
    hub = perm[0]
    lh[hub], rh[hub], out[hub] = (base - 1 if deep_tail else 1), 2, base
    for k in range(n_cons):
        g = perm[1 + k]
        other = base + 1 + int(rng.integers(0, k)) if k and rng.random() < 0.7 else 1
        r = rng.random()
        lh[g], rh[g] = (base, other) if r < 0.45 else ((other, base) if r < 0.9 else (base, base))
        out[g] = base + 1 + k
    for k in range(chain):
        g = perm[1 + n_cons + k]
        lh[g] = base + 1 + int(rng.integers(0, n_cons)) if k == 0 else base + 1 + n_cons + k - 1
        rh[g] = base + 1 + int(rng.integers(0, n_cons))
        out[g] = base + 1 + n_cons + k
    return dict(lh=lh, rh=rh, out=out, op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=base + n + 3,
                input_nodes=np.array([1, 2], np.uint32), output_nodes=np.array([base + n_cons + chain], np.uint32))


@pytest.mark.parametrize("n_cons,own_root", [(9, False), (17, True), (64, False), (65, True), (700, False), (5000, False)])
def test_one_hub_relay_levels(backend, orc, n_cons, own_root):
    """9 consumers: two relays; 64: eight, one level; 65: a second level; 5 000: four (625 + 79 + 10 + 2).  own_root: the hub has
    the smallest gate id — the DFS starts there (topological_sort.rs:11-13), every relay chain ends in a hub that takes none."""
    if n_cons > 1000 and "emul" in backend.version:
        n_cons = 1100                                            # (the emulation runs every wave of the launch as a fiber: keep it short)
    rng = np.random.default_rng(1000 + n_cons)
    p = _star(rng, n_cons, chain=30, hub_gate_first=own_root)
    assert _compare(backend, orc, p, check_serial=False) == "ok"
    st = backend.stats()
    assert st["n_relays"] == expected_relays(p["lh"], p["rh"], p["out"]) > 0


def test_hub_below_a_deep_chain_takes_the_deep_build(backend, orc):
    """A hub whose consumers hang off a 4 200-deep chain: path strings past one chunk (3 782 bits) AND relays — the launch's DEEP
    build compares relay records through the chunk links like any other record."""
    rng = np.random.default_rng(31)
    depth, n_cons = 4200, 90
    n = 1 + depth + n_cons
    perm = rng.permutation(n)
    j = int(np.where(perm == 0)[0][0]); perm[1], perm[j] = perm[j], perm[1]      # the chain's sink has gate id 0: the DFS descends the whole chain from it
    lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = np.empty(n, np.uint32)
    hub = perm[0]
    lh[hub], rh[hub], out[hub] = 1, 2, 10
    for k in range(depth):                                       # the chain: gate k reads gate k + 1 (the sink is k = 0); a few read the hub too
        g = perm[1 + k]
        out[g] = 11 + k
        lh[g] = 11 + k + 1 if k + 1 < depth else 10
        rh[g] = 10 if rng.random() < 0.01 else 2
    for k in range(n_cons):                                      # consumers of the hub that feed the chain at random depths
        g = perm[1 + depth + k]
        out[g] = 11 + depth + k
        lh[g], rh[g] = (10, 1) if k % 2 else (2, 10)
    feed = rng.choice(depth, size=n_cons, replace=False)
    for k, at in enumerate(feed):
        rh[perm[1 + at]] = 11 + depth + k
    p = dict(lh=lh, rh=rh, out=out, op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=11 + n + 2,
             input_nodes=np.array([1, 2], np.uint32), output_nodes=np.array([11], np.uint32))
    assert _compare(backend, orc, p, check_serial=False) == "ok"
    st = backend.stats()
    assert st["path_chunks"] >= 2 and st["n_relays"] > 0


def test_hub_with_a_long_tail_below(backend, orc):
    """everything upstream of the hub waits for it: a 300-gate chain of producers below a hub of 600 consumers"""
    rng = np.random.default_rng(5)
    p = _star(rng, 600, chain=25, hub_gate_first=False, deep_tail=300)
    assert _compare(backend, orc, p, check_serial=False) == "ok"


def test_cycle_through_a_hub_is_reported_like_the_reference(backend, orc):
    """a hub inside a dependency cycle: the peel leaves the cycle (and the relays above it) behind, the serial DFS formats
    topological_sort.rs:34-38's message"""
    rng = np.random.default_rng(8)
    p = _star(rng, 80, chain=10, hub_gate_first=False)
    hub_gate = int(np.where(p["out"] == 10)[0][0])
    p["lh"][hub_gate] = 10 + 1 + 80 + 9                          # the hub now reads the end of the chain above its own consumers
    assert _compare(backend, orc, p, check_serial=True) == "cyclic"


def test_hot_producers_take_their_slots_a_wave_at_a_time(backend, orc):
    """k_deps hands a HOT producer's consumers their list slots one range per wave (c2a_kernels.h HOT PRODUCERS) from its
    16 384th consumer on; with the threshold lowered (c2a_debug_hot_every) the small stars here take that path: three hubs in
    one graph — two hot ones and one below the threshold — both edges, lh == rh consumers."""
    rng = np.random.default_rng(12)
    backend.debug_hot_every(32)
    sizes = (700, 300, 20)
    n = 3 + sum(sizes) + 40
    perm = rng.permutation(n)
    lh = np.empty(n, np.uint32); rh = np.empty(n, np.uint32); out = np.empty(n, np.uint32)
    for h in range(3):
        g = perm[h]
        lh[g], rh[g], out[g] = 1, 2, 10 + h
    k = 3
    for h, sz in enumerate(sizes):
        for _ in range(sz):
            g = perm[k]
            other = 10 + int(rng.integers(0, 3)) if rng.random() < 0.3 else (13 + int(rng.integers(0, k - 3)) if k > 3 and rng.random() < 0.5 else 2)
            r = rng.random()
            lh[g], rh[g] = (10 + h, other) if r < 0.45 else ((other, 10 + h) if r < 0.9 else (10 + h, 10 + h))
            out[g] = 13 + k - 3
            k += 1
    for j in range(40):
        g = perm[k]
        lh[g] = 13 + int(rng.integers(0, k - 3)); rh[g] = 13 + int(rng.integers(0, k - 3)); out[g] = 13 + k - 3
        k += 1
    p = dict(lh=lh, rh=rh, out=out, op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=13 + n + 2,
             input_nodes=np.array([1, 2], np.uint32), output_nodes=np.array([13 + n - 4], np.uint32))
    try:
        assert _compare(backend, orc, p, check_serial=False) == "ok"
        assert backend.stats()["n_relays"] == expected_relays(lh, rh, out)
    finally:
        backend.debug_hot_every(1 << 14)


def _hubby_random(rng, n):
    """random DAG in which half of the operands come from a handful of popular nodes: hubs of every size up to n / 3, relays of
    mixed fill, consumers over both edges"""
    K = n + 4
    lh = np.empty(n, np.int64); rh = np.empty(n, np.int64)
    popular = rng.integers(0, max(1, n // 3), size=4)
    for g in range(n):
        ops = []
        for _ in range(2):
            if g and rng.random() < 0.5:
                pick = int(popular[rng.integers(4)])
                ops.append(4 + (pick if pick < g else int(rng.integers(0, g))))
            elif g and rng.random() < 0.8:
                ops.append(4 + int(rng.integers(0, g)))
            else:
                ops.append(int(rng.integers(0, 4)))
        lh[g], rh[g] = ops
    out = 4 + np.arange(n)
    perm = rng.permutation(n)
    return dict(lh=(lh[perm] + 1).astype(np.uint32), rh=(rh[perm] + 1).astype(np.uint32), out=(out[perm] + 1).astype(np.uint32),
                op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=K + 2, input_nodes=np.arange(1, 4, dtype=np.uint32),
                output_nodes=np.array([K], np.uint32))


def test_random_graphs_with_hubs(backend, orc):
    rng = np.random.default_rng(77)
    relays = 0
    for trial in range(12):
        p = _hubby_random(rng, int(rng.integers(40, 400)))
        assert _compare(backend, orc, p, check_serial=False) == "ok"
        assert backend.stats()["n_relays"] == expected_relays(p["lh"], p["rh"], p["out"])
        relays += backend.stats()["n_relays"]
    assert relays > 50


FAMILIES = ["hub", "hub_mild", "window_all", "forest", "const_hub", "strict"]


@pytest.mark.parametrize("name", FAMILIES)
def test_families_small(backend, orc, c2a, name):
    """the generators of synth.family at a size the emulation finishes: every array against the oracle"""
    fg = c2a.synth.family(name, 24_000, seed=c2a.synth.SEED + 3)
    assert _compare(backend, orc, _payload(fg), check_serial=False) == "ok"
    st = backend.stats()
    assert st["n_relays"] == expected_relays(fg.lh, fg.rh, fg.out)
    if name.startswith("hub"):
        assert st["n_relays"] > 100
    assert st["levels"] == fg.layers if name != "forest" else st["levels"] > 5      # (relays do not count as levels)


def _sha_block(c2a):
    comp = importlib.import_module("circom-2-arithc_amd.compiler")
    text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "circuits", "sha256Block.circom")).read()
    C = comp.Compiler.from_circom(text, backend=None)
    inputs, outputs, _ = C._io_maps()
    lh, rh, out, op = C._flat()
    return lh, rh, out, op, C.node_count + 1, [nd for _, nd in inputs], [nd for _, nd in outputs]


@pytest.mark.parametrize("shape,permute", [("chain", False), ("tree", False), ("tree", True)])
def test_tiled_sha256_small(backend, orc, c2a, shape, permute):
    """five / seven copies of the REAL SHA-256 block's flat list (3 448 gates each, as the unroller emits them) wired by numpy
    id offsets into a chain / a Merkle tree: sort, numbering and emission against the oracle"""
    fg = c2a.synth.tile_block(*_sha_block(c2a), copies=5 if shape == "chain" else 7, shape=shape, permute=permute)
    assert fg.n == 3448 * (5 if shape == "chain" else 7)
    assert _compare(backend, orc, _payload(fg), check_serial=False) == "ok"
    exp_levels = backend.stats()["levels"]
    assert exp_levels > (2000 if shape == "chain" else 1000)     # the chain is five blocks deep, the tree three


@pytest.mark.parametrize("shape", ["matmul", "butterfly"])
def test_wide_and_shallow_small(backend, orc, c2a, shape):
    """wide and shallow graphs — a matrix product's reduction chains, a butterfly — pile entries up in the hand-off arrays with nobody in
    line for them: what calls the launch's parked waves in (c2a_peel.h, THE RESERVE; under the emulation four waves are parked and 24
    waiting entries are the mark): every array against the oracle"""
    fg = c2a.synth.matmul(9) if shape == "matmul" else c2a.synth.butterfly(6, 9)
    assert _compare(backend, orc, _payload(fg), check_serial=False) == "ok"
    assert backend.stats()["levels"] == (9 if shape == "matmul" else 9)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["matmul", "butterfly", "one_layer"])
def test_wide_and_shallow_1m(shape, orc, c2a):
    """the same at size: a 100 x 100 matrix product (1.99 M gates, 10 000 reduction chains), a butterfly of 16 x 2^16, ONE layer of 2 M gates —
    short of waves, not of latency: the reserve comes in (matmul 170^3: 7.4 -> 5.0 ms).  Every array against the oracle, twice."""
    fg = {"matmul": lambda: c2a.synth.matmul(100), "butterfly": lambda: c2a.synth.butterfly(16, 16),
          "one_layer": lambda: c2a.synth.layered_dag(1, 2_000_000)}[shape]()
    with c2a.Backend(0) as be:
        assert _compare(be, orc, _payload(fg), check_serial=False) == "ok"
        bm = importlib.import_module("circom-2-arithc_amd.backend")
        exp = orc.build_circuit(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes, mode=1)
        for rep in range(2):
            assert be.build_circuit() == exp.wire_count
            for nm, arr in (("sorted", exp.sorted), ("in0", exp.in0), ("in1", exp.in1), ("out", exp.out), ("op", exp.op)):
                assert be.checksum(nm) == bm.checksum_host(arr), (nm, rep)
        assert be.stats()["peel_waves"] >= 2 * 8      # (eight per CU and the reserve)


# ---- at size, on the hardware: every family at >= 1 M gates, both numbering paths, every array element-wise
@pytest.mark.gpu
@pytest.mark.parametrize("walk", [False, True], ids=["positional", "walk"])
@pytest.mark.parametrize("name", FAMILIES + ["sha_chain", "sha_tree"])
def test_families_1m(name, walk, orc, c2a):
    if name.startswith("sha_"):
        blk = _sha_block(c2a)
        # (a chain of 290 blocks is 165 000 levels deep: 0.2 s of dataflow launch — the tree has the same gates in 5 000)
        fg = c2a.synth.tile_block(*blk, copies=290 if name == "sha_tree" else 60, shape=name[4:])
    else:
        fg = c2a.synth.family(name, 1_000_000)
    with _Env(**({"C2A_NUMBERING_WALK": "1"} if walk else {})):
        be = c2a.Backend(0)
    try:
        assert _compare(be, orc, _payload(fg), check_serial=False) == "ok"
        st = be.stats()
        assert st["numbering_path"] == (0 if walk else 1)
        if name.startswith("hub"):
            assert st["n_relays"] > 10_000
        if name == "hub":
            assert st["n_relays"] > 30_000                       # (the three mega hubs alone: ~ 9 000 relays)
        # the fused call on the same loaded graph: same arrays (checksums)
        bm = importlib.import_module("circom-2-arithc_amd.backend")
        exp = orc.build_circuit(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes, mode=1)
        for rep in range(2):
            assert be.build_circuit() == exp.wire_count
            for nm, arr in (("sorted", exp.sorted), ("in0", exp.in0), ("in1", exp.in1), ("out", exp.out), ("op", exp.op)):
                assert be.checksum(nm) == bm.checksum_host(arr), (nm, rep)
    finally:
        be.close()


@pytest.mark.gpu
def test_fuzz_slice_on_the_hardware(hip_backend):
    """a minute of tools/fuzz_gpu.py (random graphs of every family, random shapes and hub populations, up to 1 M gates, every
    result array of the fused call against the oracle, the staged calls on every seventh) as part of the suite"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    fuzz = importlib.import_module("tools.fuzz_gpu")
    n, per = fuzz.run(minutes=1.0, seed=20241008, max_gates=1_000_000, be=hip_backend)
    assert n >= 20 and len(per) >= 4, (n, per)


def test_hub_with_duplicate_writers(backend, orc):
    """two gates write the hub's node (the reference keeps the LAST writer as its producer, src/compiler.rs:403-406): the general
    numbering path, the identity relabelling — and the relay tree hangs under the later gate"""
    rng = np.random.default_rng(21)
    p = _star(rng, 120, chain=12, hub_gate_first=False)
    n = len(p["lh"])
    extra = dict(lh=np.array([1, 2], np.uint32), rh=np.array([2, 1], np.uint32), out=np.array([10, 10 + 1 + 5], np.uint32),
                 op=np.array([0, 10], np.uint8))                    # a second writer of the hub's node, and one of a consumer's node
    for k in ("lh", "rh", "out", "op"):
        p[k] = np.concatenate([p[k], extra[k]])
    assert _compare(backend, orc, p, check_serial=True) == "ok"
    st = backend.stats()
    assert st["numbering_path"] == 0 and st["n_relays"] == expected_relays(p["lh"], p["rh"], p["out"]) > 0


def test_constant_read_by_many_gates(backend, orc, c2a):
    """A NAMED CONSTANT that a whole template context reads — the reference's unroller makes one node per literal and context
    (src/process.rs:558-579; keys at src/compiler.rs:354-359): `0`, `1` — is one un-produced node with 10^4-10^6 readers: it gets
    its wire where the walk first sees it (src/compiler.rs:431-441), and "first" is a minimum over all its readers on ONE word.
    Both numbering paths against the oracle; on the hardware the numbering must not take the 12 ms it took before the look-before-
    atomic (tools/const_hub_check.py: 10 M gates, 10^6 readers)."""
    from conftest import _Env
    on_gpu = "hip" in backend.version
    n = 1_000_000 if on_gpu else 24_000
    fg = c2a.synth.layered_dag(n // 2000, 2000, seed=c2a.synth.SEED + 9)
    rng = np.random.default_rng(2)
    rh, lh = fg.rh.copy(), fg.lh.copy()
    rh[rng.random(fg.n) < 0.2] = fg.const_nodes[0]
    lh[rng.random(fg.n) < 0.02] = fg.const_nodes[1]
    p = dict(lh=lh, rh=rh, out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes, output_nodes=fg.output_nodes)
    assert _compare(backend, orc, p, check_serial=False) == "ok"
    assert backend.stats()["numbering_path"] == 1
    if on_gpu:
        backend.build_circuit()
        assert backend.timings()["wires"] < 1.0, backend.timings()           # ms
    with _Env(C2A_NUMBERING_WALK=1):
        be = c2a.Backend(0) if on_gpu else c2a.Backend(0, lib_path=os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul", "libc2a_emul.so"))
    try:
        assert _compare(be, orc, p, check_serial=False) == "ok"
        assert be.stats()["numbering_path"] == 0
    finally:
        be.close()
