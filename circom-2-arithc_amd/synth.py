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
