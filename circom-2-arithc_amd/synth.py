"""Seeded synthetic flat gate lists (the boundary payload of DESIGN.md §3).

The reference's front-end is quadratic (src/compiler.rs:185-195, :219-226, :260-270), so large gate
graphs can only be injected at the flat-gate boundary.  This module freezes the generator that
SURVEY.md §8(d) specifies for BASELINE.json's configs: splitmix64 counter PRNG, layered fan-in-2 DAG,
gate ids permuted (otherwise the DFS order is the identity), sparse node ids that mimic
the gaps left by `add_connection` (src/compiler.rs:257).

Pure numpy; no GPU, no oracle.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

SEED = 20241008
_GOLD = np.uint64(0x9E3779B97F4A7C15)

# AGateType discriminants (src/a_gate_type.rs:8-27)
OP_NAMES = [
    "AAdd", "ADiv", "AEq", "AGEq", "AGt", "ALEq", "ALt", "AMul", "ANeq", "ASub", "AXor", "APow",
    "AIntDiv", "AMod", "AShiftL", "AShiftR", "ABoolOr", "ABoolAnd", "ABitOr", "ABitAnd",
]
OP = {n: i for i, n in enumerate(OP_NAMES)}

# op mix of the headline config: 40 % AXor, 20 % ABitAnd, 10 % ABitOr, 20 % AAdd,
# 5 % ASub, 5 % {AEq, ALt}
MIX_BITWISE = (("AXor", 40), ("ABitAnd", 20), ("ABitOr", 10), ("AAdd", 20), ("ASub", 5), ("AEq", 3), ("ALt", 2))
MIX_POSEIDON = (("AAdd", 50), ("AMul", 50))
MIX_SHA = (("AXor", 40), ("ABitAnd", 25), ("ABitOr", 5), ("AAdd", 25), ("AShiftR", 5))
MIX_ALL = tuple((n, 1) for n in OP_NAMES)


def splitmix64(seed: int, stream: int, count: int) -> np.ndarray:
    """count outputs of splitmix64 started at state seed + stream*2^40 (counter form, vectorised)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) + np.uint64(stream) * np.uint64(1 << 40)
        z = base + (np.arange(1, count + 1, dtype=np.uint64) * _GOLD)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


@dataclass
class FlatGates:
    """Flat gate SoA with raw (sparse) node ids + canonical IO node lists."""
    lh: np.ndarray
    rh: np.ndarray
    out: np.ndarray
    op: np.ndarray
    n_nodes: int
    input_nodes: np.ndarray
    output_nodes: np.ndarray
    const_nodes: np.ndarray
    layers: int
    layer_width: int

    @property
    def n(self) -> int:
        return int(self.lh.shape[0])


def layered_dag(layers: int, layer_width: int, n_in: int = 4096, n_const: int = 64, window: int = 64,
                mix: Sequence[Tuple[str, int]] = MIX_BITWISE, seed: int = SEED, permute: bool = True,
                sparse_ids: bool = True, const_frac: float = 0.0, out_frac: float = 0.0) -> FlatGates:
    """n = layers*layer_width gates; gate (k, j): lh uniform over layer k-1's outputs (layer 0: the
    primary inputs) => depth == layers; rh uniform over inputs, constants and the outputs of the previous
    `window` layers; one distinct out node per gate; last layer's out nodes are the circuit outputs.

    REFERENCE-SHAPED extras (both 0 in the headline config, whose graph they leave bit-identical): the reference's
    unroller makes one named constant node per literal and template context (src/process.rs:558-579, keys at
    src/compiler.rs:354-359) — in the committed fixtures 5-30 % of the gates read an un-produced non-input node — and
    names an output signal per template.  `const_frac` of the gates read a FRESH constant node as rh, created right
    before their own out node (node ids follow the creation order, src/compiler.rs:497-500); `out_frac` of the gates
    (besides the last layer) write a circuit output node."""
    L, Wd = int(layers), int(layer_width)
    n = L * Wd
    k = np.repeat(np.arange(L, dtype=np.int64), Wd)              # layer of each gate (generation order)
    r_lh = splitmix64(seed, 1, n)
    r_rh = splitmix64(seed, 2, n)
    r_op = splitmix64(seed, 3, n)

    in_base, const_base, gate_base = 0, n_in, n_in + n_const     # logical node numbering
    # lh
    # logical node order: inputs, shared constants, then per gate (generation order) [its fresh constant,] its out node
    fresh = np.zeros(n, dtype=bool)
    if const_frac > 0:
        fresh = (splitmix64(seed, 6, n) % np.uint64(1 << 20)).astype(np.int64) < int(const_frac * (1 << 20))
    n_fresh = int(fresh.sum())
    out_of = gate_base + np.arange(n, dtype=np.int64) + np.cumsum(fresh)      # logical id of gate i's out node
    lh_prev = out_of[np.maximum((k - 1) * Wd + (r_lh % np.uint64(Wd)).astype(np.int64), 0)]
    lh_in = in_base + (r_lh % np.uint64(n_in)).astype(np.int64)
    lh_log = np.where(k == 0, lh_in, lh_prev)
    # rh: pool = inputs + constants + outputs of layers [max(0,k-window), k)
    back = np.minimum(k, window)
    pool = n_in + n_const + back * Wd
    idx = (r_rh % pool.astype(np.uint64)).astype(np.int64)
    rel = idx - (n_in + n_const)                                 # >=0 -> a gate output in the window
    lay = k - 1 - rel // Wd
    rh_gate = out_of[np.clip(lay * Wd + rel % Wd, 0, n - 1)]
    rh_log = np.where(idx < n_in + n_const, idx, rh_gate)
    rh_log = np.where(fresh, out_of - 1, rh_log)
    out_log = out_of
    # ops
    names = [m[0] for m in mix]
    wts = np.array([m[1] for m in mix], dtype=np.int64)
    cum = np.cumsum(wts)
    pick = np.searchsorted(cum, (r_op % np.uint64(cum[-1])).astype(np.int64), side="right")
    op = np.array([OP[nm] for nm in names], dtype=np.uint8)[pick]

    n_log = n_in + n_const + n + n_fresh
    if sparse_ids:
        gaps = (splitmix64(seed, 4, n_log) & np.uint64(1)).astype(np.int64)
        node_id = 1 + np.arange(n_log, dtype=np.int64) + np.cumsum(gaps)
    else:
        node_id = 1 + np.arange(n_log, dtype=np.int64)
    n_nodes = int(node_id[-1]) + 1
    assert n_nodes < 2 ** 32

    lh_id = node_id[lh_log].astype(np.uint32)
    rh_id = node_id[rh_log].astype(np.uint32)
    out_id = node_id[out_log].astype(np.uint32)
    is_out = k == L - 1
    if out_frac > 0:
        is_out = is_out | ((splitmix64(seed, 7, n) % np.uint64(1 << 20)).astype(np.int64) < int(out_frac * (1 << 20)))
    output_nodes = node_id[out_log[is_out]].astype(np.uint32)
    const_nodes = np.concatenate([node_id[const_base:const_base + n_const], node_id[out_log[fresh] - 1]]).astype(np.uint32)
    if permute:
        perm = np.argsort(splitmix64(seed, 5, n), kind="stable")  # new gate id g <- generation index perm[g]
        lh_id, rh_id, out_id, op = lh_id[perm], rh_id[perm], out_id[perm], op[perm]
    return FlatGates(lh=lh_id, rh=rh_id, out=out_id, op=np.ascontiguousarray(op), n_nodes=n_nodes,
                     input_nodes=node_id[in_base:in_base + n_in].astype(np.uint32),
                     output_nodes=output_nodes, const_nodes=const_nodes,
                     layers=L, layer_width=Wd)


# BASELINE.json configs as shape stand-ins (the real circuits are not in the reference tree
# and cannot pass its front-end; these reproduce n / depth / op mix only).
CONFIGS: Dict[str, dict] = {
    "poseidon2_standin": dict(layers=30, layer_width=10, n_in=2, n_const=16, window=4, mix=MIX_POSEIDON),
    "sha256_standin": dict(layers=300, layer_width=100, n_in=512, n_const=64, window=16, mix=MIX_SHA),
    "keccak_standin": dict(layers=600, layer_width=250, n_in=1088, n_const=64, window=8, mix=MIX_SHA),
    "synthetic_10m": dict(layers=5000, layer_width=2000, n_in=4096, n_const=64, window=64, mix=MIX_BITWISE),
    # the same graph shape with what a circuit from the reference's own unroller carries: a fresh constant node at a tenth of the
    # gates, an output node at a twentieth (the headline graph has 64 constants and 2 000 outputs in 10 M gates)
    "reference_shaped_10m": dict(layers=5000, layer_width=2000, n_in=4096, n_const=64, window=64, mix=MIX_BITWISE,
                                 const_frac=0.10, out_frac=0.05),
}


def config(name: str, seed: int = SEED, scale: Optional[float] = None) -> FlatGates:
    kw = dict(CONFIGS[name])
    if scale is not None:
        kw["layers"] = max(1, int(kw["layers"] * scale))
    return layered_dag(seed=seed, **kw)


# ------------------------------------------------------------------------------------------------------------------------
# Families beyond layered_dag (every large graph used to come from that one generator — Poisson(2) fan-out,
# a 64-layer window, no trees).  The reference's deps closure (src/compiler.rs:408-421) allows ANY fan-out and ANY distance:
# a broadcast selector or a scale factor feeds a whole layer (tests/circuits/machine-learning/ in the reference), a reduction
# is an in-tree, and a real circuit is a tiling of sub-circuits.  All of them: seeded, pure numpy, gate ids permuted unless
# said otherwise, sparse node ids in creation order like layered_dag.
# ------------------------------------------------------------------------------------------------------------------------
def _finish(lh_log, rh_log, out_log, op, n_log, n_in, const_log, out_mask, seed, permute, sparse_ids, layers, width):
    """logical node numbers (creation order) -> sparse raw ids, gate ids permuted: the tail of layered_dag, shared"""
    n = int(lh_log.shape[0])
    if sparse_ids:
        gaps = (splitmix64(seed, 4, n_log) & np.uint64(1)).astype(np.int64)
        node_id = 1 + np.arange(n_log, dtype=np.int64) + np.cumsum(gaps)
    else:
        node_id = 1 + np.arange(n_log, dtype=np.int64)
    n_nodes = int(node_id[-1]) + 1
    assert n_nodes < 2 ** 32
    lh_id, rh_id, out_id = node_id[lh_log].astype(np.uint32), node_id[rh_log].astype(np.uint32), node_id[out_log].astype(np.uint32)
    output_nodes = node_id[out_log[out_mask]].astype(np.uint32)
    if permute:
        perm = np.argsort(splitmix64(seed, 5, n), kind="stable")
        lh_id, rh_id, out_id, op = lh_id[perm], rh_id[perm], out_id[perm], op[perm]
    return FlatGates(lh=lh_id, rh=rh_id, out=out_id, op=np.ascontiguousarray(op), n_nodes=n_nodes,
                     input_nodes=node_id[:n_in].astype(np.uint32), output_nodes=output_nodes,
                     const_nodes=node_id[const_log].astype(np.uint32), layers=int(layers), layer_width=int(width))


def _pick_ops(mix, r_op):
    names = [m[0] for m in mix]
    cum = np.cumsum(np.array([m[1] for m in mix], dtype=np.int64))
    pick = np.searchsorted(cum, (r_op % np.uint64(cum[-1])).astype(np.int64), side="right")
    return np.array([OP[nm] for nm in names], dtype=np.uint8)[pick]


def hub_dag(layers: int, layer_width: int, n_in: int = 4096, n_const: int = 64, window: int = 64,
            mix: Sequence[Tuple[str, int]] = MIX_BITWISE, seed: int = SEED, permute: bool = True, sparse_ids: bool = True,
            small_frac: float = 0.01, small_lo: float = 17.0, small_alpha: float = 2.2, small_cap: float = 900.0,
            big: int = 300, big_lo: float = 1e3, big_hi: float = 1e4, mega: Sequence[float] = (0.1, 0.03, 0.01),
            p_hub: float = 0.45) -> FlatGates:
    """layered_dag's skeleton (lh from the previous layer => depth == layers) with HUBS: produced nodes that many gates read.
    `small_frac` of the gates are small hubs (Pareto weights from `small_lo`, capped), `big` gates are big ones (log-uniform
    weights big_lo..big_hi), and every entry f of `mega` is one gate in the first layers that a fraction f of ALL gates reads
    (0.1 of 10 M gates: a fan-out of 10^6).  A gate's rh goes to a mega hub with probability f each, else with probability
    `p_hub` to a small / big hub of ANY earlier layer in proportion to the weights (consumers at many depths), else where
    layered_dag would send it.  With fan-in 2 the edges are 2 n: "1 % of the nodes with a fan-out of 10^2-10^4" does not fit —
    1 % of the nodes with 17-900, 300 (per 10 M) with 10^3-10^4 and three with 10^5-10^6 does."""
    L, Wd = int(layers), int(layer_width)
    n = L * Wd
    k = np.repeat(np.arange(L, dtype=np.int64), Wd)
    r_lh, r_rh, r_op = splitmix64(seed, 1, n), splitmix64(seed, 2, n), splitmix64(seed, 3, n)
    gate_base = n_in + n_const
    out_of = gate_base + np.arange(n, dtype=np.int64)
    lh_prev = out_of[np.maximum((k - 1) * Wd + (r_lh % np.uint64(Wd)).astype(np.int64), 0)]
    lh_log = np.where(k == 0, (r_lh % np.uint64(n_in)).astype(np.int64), lh_prev)
    back = np.minimum(k, window)
    pool = n_in + n_const + back * Wd
    idx = (r_rh % pool.astype(np.uint64)).astype(np.int64)
    rel = idx - (n_in + n_const)
    lay = k - 1 - rel // Wd
    rh_log = np.where(idx < n_in + n_const, idx, out_of[np.clip(lay * Wd + rel % Wd, 0, n - 1)])
    # ---- the hubs: generation indices (below the last layer), weights
    u = lambda stream, cnt: (splitmix64(seed, stream, cnt) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    n_small = int(small_frac * n)
    n_big = max(0, int(round(big * n / 1e7)))
    hub_idx = np.unique((splitmix64(seed, 8, n_small + n_big) % np.uint64(max(1, n - Wd))).astype(np.int64))
    H = len(hub_idx)
    w = np.minimum(small_lo * (1.0 - u(9, H)) ** (-1.0 / (small_alpha - 1.0 + 1e-9)), small_cap)
    is_big = np.zeros(H, dtype=bool)
    if n_big and H:
        is_big[(splitmix64(seed, 10, n_big) % np.uint64(H)).astype(np.int64)] = True
        w = np.where(is_big, big_lo * (big_hi / big_lo) ** u(11, H), w)
    if H:
        cumw = np.cumsum(w)
        r_sel, r_pick = u(12, n), u(13, n)
        cnt = np.searchsorted(hub_idx, k * Wd, side="left")                 # hubs in strictly earlier layers
        take = (r_sel < p_hub) & (cnt > 0)
        tot = np.where(cnt > 0, cumw[np.maximum(cnt, 1) - 1], 1.0)
        pick = np.minimum(np.searchsorted(cumw, r_pick * tot, side="right"), np.maximum(cnt, 1) - 1)
        rh_log = np.where(take, out_of[hub_idx[pick]], rh_log)
    # ---- the mega hubs: one gate each in the first layers, read by a fraction f of the gates of every later layer
    r_m = u(14, n)
    lo = 0.0
    for j, f in enumerate(mega):
        g_m = int(splitmix64(seed, 15 + j, 1)[0] % np.uint64(max(1, min(n - Wd, max(Wd, n // 100)))))
        hit = (r_m >= lo) & (r_m < lo + f) & (k > g_m // Wd)
        rh_log = np.where(hit, out_of[g_m], rh_log)
        lo += f
    op = _pick_ops(mix, r_op)
    return _finish(lh_log, rh_log, out_of, op, n_in + n_const + n, n_in, np.arange(n_in, n_in + n_const), k == L - 1, seed, permute,
                   sparse_ids, L, Wd)


def reduction_forest(n: int, width: int = 2000, n_in: int = 4096, n_const: int = 64, mix: Sequence[Tuple[str, int]] = MIX_BITWISE,
                     seed: int = SEED, permute: bool = True, sparse_ids: bool = True, p_merge: float = 0.4,
                     p_chain: float = 0.3) -> FlatGates:
    """In-trees: every gate's out node is read by AT MOST ONE gate (fan-out 1; no claim ticket is ever needed, and every gate
    with two produced operands hands one of them off).  Built in rounds over a pool of about `width` unread outputs: a fraction
    `p_merge` of the pool is paired up (a gate reads two of them), `p_chain` is extended (a gate reads one of them and an
    input), the rest waits; fresh leaves (both operands inputs) keep the pool at `width`.  Subtrees of very different depths
    meet at a merge; what is still unread at the end are the roots of the forest = the circuit outputs."""
    rng = np.random.default_rng(seed)
    lh_parts, rh_parts = [], []
    pool = np.empty(0, dtype=np.int64)          # generation indices of gates whose out node nobody reads yet
    g, rounds = 0, 0
    gate_base = n_in + n_const
    while g < n:
        rounds += 1
        P = len(pool)
        pool = pool[rng.permutation(P)]
        a = min(int(p_merge * P) // 2 * 2, 2 * (n - g))
        merges = a // 2
        b = min(int(p_chain * P), P - a, n - g - merges)
        leaves = min(max(width - (P - merges), 1 if P == 0 else 0), n - g - merges - b)
        if merges + b + leaves == 0:                                      # (a pool too small for the fractions: extend one chain, so that every round makes a gate)
            b = 1 if P > a else 0
            leaves = 1 - b
        m_l, m_r = pool[:merges], pool[merges:a]
        c_p = pool[a:a + b]
        ext = rng.integers(0, n_in + n_const, size=b + 2 * leaves)
        side = rng.integers(0, 2, size=b).astype(bool)                    # which operand of a chain gate is the produced one
        lh_parts += [gate_base + m_l, np.where(side, gate_base + c_p, ext[:b]), ext[b:b + leaves]]
        rh_parts += [gate_base + m_r, np.where(side, ext[:b], gate_base + c_p), ext[b + leaves:]]
        made = merges + b + leaves
        pool = np.concatenate([pool[a + b:], np.arange(g, g + made, dtype=np.int64)])
        g += made
    lh_log, rh_log = np.concatenate(lh_parts).astype(np.int64), np.concatenate(rh_parts).astype(np.int64)
    assert len(lh_log) == n
    out_of = gate_base + np.arange(n, dtype=np.int64)
    roots = np.zeros(n, dtype=bool)
    roots[pool] = True
    op = _pick_ops(mix, splitmix64(seed, 3, n))
    return _finish(lh_log, rh_log, out_of, op, gate_base + n, n_in, np.arange(n_in, n_in + n_const), roots, seed, permute, sparse_ids,
                   rounds, width)


def tile_block(lh, rh, out, op, n_nodes, in_nodes, out_nodes, copies: int, shape: str = "chain", seed: int = SEED,
               permute: bool = False) -> FlatGates:
    """`copies` instances of ONE real circuit's flat gate list (e.g. the SHA-256 compression of tests/golden/circuits/), wired
    into a chain or a binary (Merkle) tree by numpy id offsets — no parser run per copy.  in_nodes / out_nodes: the block's
    input and output nodes in their canonical order, len(in_nodes) == 2 * len(out_nodes).  chain: the first half of copy c's
    inputs are copy c - 1's outputs; tree (heap order, leaves last, i.e. created first in node-id terms): the two halves of a
    copy's inputs are its two children's outputs.  Copy ids are laid out so that node ids follow the creation order (a child
    before its parent), like the reference's counter (src/compiler.rs:497-500).  Gate ids stay in the unroller's order unless
    `permute`."""
    lh, rh, out = (np.asarray(x, dtype=np.int64) for x in (lh, rh, out))
    in_nodes, out_nodes = np.asarray(in_nodes, dtype=np.int64), np.asarray(out_nodes, dtype=np.int64)
    nb, half = len(lh), len(out_nodes)
    assert len(in_nodes) == 2 * half and copies >= 1
    N = int(n_nodes)
    # creation slot of copy c (its node-id offset is slot * N): chain: c; tree: children are created before their parent
    if shape == "chain":
        slot = np.arange(copies, dtype=np.int64)
        src = [(c - 1, None) if c else (None, None) for c in range(copies)]
    else:
        slot = (copies - 1 - np.arange(copies, dtype=np.int64))             # heap index 0 = the root: created last
        src = [(2 * c + 1 if 2 * c + 1 < copies else None, 2 * c + 2 if 2 * c + 2 < copies else None) for c in range(copies)]
    off = slot * N
    LH = (lh[None, :] + off[:, None])
    RH = (rh[None, :] + off[:, None])
    OUT = (out[None, :] + off[:, None])
    # an input node of copy c that is fed by another copy's output: substitute wherever it is read
    remap_from, remap_to = [], []
    fed = np.zeros((copies, 2 * half), dtype=bool)
    for c, (s0, s1) in enumerate(src):
        for h, s in enumerate((s0, s1)):
            if s is None:
                continue
            remap_from.append(in_nodes[h * half:(h + 1) * half] + off[c])
            remap_to.append(out_nodes + off[s])
            fed[c, h * half:(h + 1) * half] = True
    if remap_from:
        rf, rt = np.concatenate(remap_from), np.concatenate(remap_to)
        order = np.argsort(rf)
        rf, rt = rf[order], rt[order]
        def sub(A):
            flat = A.reshape(-1)
            pos = np.minimum(np.searchsorted(rf, flat), len(rf) - 1)
            hit = rf[pos] == flat
            return np.where(hit, rt[pos], flat).reshape(A.shape)
        LH, RH = sub(LH), sub(RH)
    # gate order: copy by copy in creation order (what an unroller walking the tree bottom-up would append)
    by_slot = np.argsort(slot)
    lh_f, rh_f, out_f = LH[by_slot].reshape(-1), RH[by_slot].reshape(-1), OUT[by_slot].reshape(-1)
    op_f = np.tile(np.asarray(op, dtype=np.uint8), copies)
    inputs = np.concatenate([(in_nodes + off[c])[~fed[c]] for c in by_slot])
    final = 0 if shape != "chain" else copies - 1
    outputs = out_nodes + off[final]
    if permute:
        perm = np.argsort(splitmix64(seed, 5, nb * copies), kind="stable")
        lh_f, rh_f, out_f, op_f = lh_f[perm], rh_f[perm], out_f[perm], op_f[perm]
    return FlatGates(lh=lh_f.astype(np.uint32), rh=rh_f.astype(np.uint32), out=out_f.astype(np.uint32), op=np.ascontiguousarray(op_f),
                     n_nodes=N * copies, input_nodes=inputs.astype(np.uint32), output_nodes=outputs.astype(np.uint32),
                     const_nodes=np.empty(0, np.uint32), layers=copies, layer_width=nb)


# the new families at the sizes the GPU suite (1 M) and tools/stress.sh (10 M) run them at: name -> callable(scale)
# ---- shapes at the edges of what layered_dag makes (tools/extreme_probe.py, tests/test_hubs.py: wide and shallow graphs are short of waves, not of latency)
def butterfly(log_w, layers, seed=3):
    """layer k, gate j: lh = (k-1, j), rh = (k-1, j ^ 2^((k-1) % log_w)): strict layers with structure (an FFT's data flow)"""
    W = 1 << log_w
    n = layers * W
    n_in = W
    out = (1 + n_in + np.arange(n, dtype=np.int64)).astype(np.uint32)
    k = np.repeat(np.arange(layers, dtype=np.int64), W)
    j = np.tile(np.arange(W, dtype=np.int64), layers)
    prev = lambda jj: np.where(k == 0, 1 + jj, 1 + n_in + (k - 1) * W + jj)
    lh = prev(j).astype(np.uint32)
    rh = prev(j ^ (1 << ((np.maximum(k, 1) - 1) % log_w))).astype(np.uint32)
    op = np.full(n, OP["AAdd"], dtype=np.uint8)
    perm = np.argsort(splitmix64(seed, 5, n), kind="stable")
    outs = out[(layers - 1) * W:]
    return FlatGates(lh=lh[perm], rh=rh[perm], out=out[perm], op=op, n_nodes=int(1 + n_in + n), input_nodes=(1 + np.arange(n_in)).astype(np.uint32),
                       output_nodes=outs, const_nodes=np.zeros(0, np.uint32), layers=layers, layer_width=W)


def matmul(m, seed=3):
    """C = A x B for m x m matrices as the reference's unroller would emit it: m^3 AMul gates (A[i,k] and B[k,j] are INPUT nodes with m readers each)
    and, per output, a chain of m - 1 AAdd gates — m^2 reduction chains of equal length, every chain step with a leaf beside it"""
    n_in = 2 * m * m
    a = lambda i, k: 1 + i * m + k
    b = lambda k, j: 1 + m * m + k * m + j
    i, j, k = np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij")
    i, j, k = i.reshape(-1), j.reshape(-1), k.reshape(-1)
    n_mul = m ** 3
    mul_out = 1 + n_in + np.arange(n_mul, dtype=np.int64)                   # product (i, j, k)
    add_out = 1 + n_in + n_mul + np.arange(m * m * (m - 1), dtype=np.int64)   # partial sum (i, j, k), k = 1 .. m-1
    ai, aj, ak = np.meshgrid(np.arange(m), np.arange(m), np.arange(1, m), indexing="ij")
    ai, aj, ak = ai.reshape(-1), aj.reshape(-1), ak.reshape(-1)
    prod_of = lambda ii, jj, kk: mul_out[(ii * m + jj) * m + kk]
    sum_of = lambda ii, jj, kk: add_out[(ii * m + jj) * (m - 1) + kk - 1]
    add_lh = np.where(ak == 1, prod_of(ai, aj, 0), sum_of(ai, aj, np.maximum(ak - 1, 1)))
    add_rh = prod_of(ai, aj, ak)
    lh = np.concatenate([a(i, k), add_lh]).astype(np.uint32)
    rh = np.concatenate([b(k, j), add_rh]).astype(np.uint32)
    out = np.concatenate([mul_out, add_out]).astype(np.uint32)
    op = np.concatenate([np.full(n_mul, OP["AMul"]), np.full(len(add_out), OP["AAdd"])]).astype(np.uint8)
    n = len(out)
    perm = np.argsort(splitmix64(seed, 5, n), kind="stable")
    oi, oj = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    outs = sum_of(oi.reshape(-1), oj.reshape(-1), m - 1).astype(np.uint32)
    return FlatGates(lh=lh[perm], rh=rh[perm], out=out[perm], op=op[perm], n_nodes=int(1 + n_in + n), input_nodes=(1 + np.arange(n_in)).astype(np.uint32),
                       output_nodes=outs, const_nodes=np.zeros(0, np.uint32), layers=m, layer_width=m * m)



def family(name: str, n_target: int = 1_000_000, seed: int = SEED) -> FlatGates:
    Wd = 2000
    L = max(8, n_target // Wd)
    if name == "hub":
        return hub_dag(L, Wd, seed=seed)
    if name == "hub_mild":            # hubs of 17-900 only (no big, no mega ones): the fan-outs a broadcast inside one layer makes
        return hub_dag(L, Wd, seed=seed, big=0, mega=())
    if name == "window_all":          # rh uniform over ALL earlier layers: no locality at any distance
        return layered_dag(L, Wd, window=L, seed=seed)
    if name == "forest":
        return reduction_forest(L * Wd, width=Wd, seed=seed)
    if name == "strict":              # STRICT LAYERS: both operands out of the layer right above (20 inputs, 4 constants: ~1 % of the rh), so every
        return layered_dag(L, Wd, n_in=20, n_const=4, window=1, seed=seed)      # layer waits for ALL of the one below it — a wave per gate, all in step
    if name == "const_hub":           # two named constants that a tenth / a hundredth of ALL gates read (one node per literal and template context: process.rs:558-579)
        return shared_constants(layered_dag(L, Wd, seed=seed), (0.1, 0.01), seed)
    raise KeyError(name)


def shared_constants(fg: FlatGates, fracs: Sequence[float], seed: int = SEED) -> FlatGates:
    """fg with the rh (every second time: lh) operand of a fraction fracs[k] of its gates replaced by its k-th constant node: un-produced
    nodes with 10^3-10^6 readers, as a literal used all over a template context is"""
    lh, rh = fg.lh.copy(), fg.rh.copy()
    for k, f in enumerate(fracs):
        hit = (splitmix64(seed, 40 + k, fg.n) >> np.uint64(11)).astype(np.float64) / float(1 << 53) < f
        (rh if k % 2 == 0 else lh)[hit] = fg.const_nodes[k % len(fg.const_nodes)]
    return FlatGates(lh=lh, rh=rh, out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes, output_nodes=fg.output_nodes,
                     const_nodes=fg.const_nodes, layers=fg.layers, layer_width=fg.layer_width)
