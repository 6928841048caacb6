"""BASELINE.json configs[2] and configs[3] as REAL circuits with an external known answer.

`tests/golden/circuits/sha256Block.circom` (the SHA-256 compression of one block, 32-bit words, --boolify-width 32) and
`tests/golden/circuits/keccakF1600.circom` (the Keccak-f[1600] permutation, 64-bit lanes, --boolify-width 64) are this repo's
own texts in the subset the reference's front-end supports (README.md:14-40) — the circomlib originals are bit level and do
not pass it.  Each goes .circom text -> the unroller's calls (circom_frontend.py restating src/process.rs) -> the
`Compiler` mirror -> the C ABI -> sort, numbering, emission (against the oracle: src/topological_sort.rs:3-50,
src/compiler.rs:321-494) -> c2a_boolify -> c2a_boolify_prune, and the ARITHMETIC circuit, its BOOLEAN image and the PRUNED
image are evaluated (c2a_eval: the reference's harness, tests/integration.rs:191-237) on eight messages each against
`hashlib.sha256` / `hashlib.sha3_256` — a published function, not this repo's evaluator or its frozen bit-blast templates
checking themselves (tests/integration.rs:94-115 defines the gate semantics the templates must have).

Gate counts (next to BASELINE.json's estimates for the bit-level circomlib originals: ~30 K and ~150 K arithmetic gates): the
word-level texts unroll to 3 448 arithmetic gates (SHA-256, 1 392 named constant nodes) and 5 112 (Keccak-f, 1 392), i.e.
0.85 M boolean gates at width 32 and 1.9 M at width 64 — the boolean sizes are asserted below."""
import hashlib
import importlib
import os
import struct

import numpy as np
import pytest

from conftest import BACKENDS  # noqa: F401  (the `backend` fixture is parametrised over them)

CIRCUITS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "circuits")
KECCAK_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
             0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
             0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
             0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
             0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]


def _messages(max_len):
    rng = np.random.default_rng(20241008)
    fixed = [b"", b"abc", b"The quick brown fox jumps over the lazy dog"[:max_len], bytes(range(max_len))]
    return fixed + [rng.integers(0, 256, int(L), dtype=np.uint8).tobytes() for L in rng.integers(1, max_len + 1, 4)]


def keccak_f_numpy(lanes):
    """Keccak-f[1600] on 25 uint64 lanes (FIPS 202 section 3.2), the reference the permutation itself is compared with."""
    a = [int(x) for x in lanes]
    M = (1 << 64) - 1
    rot = lambda v, r: ((v << r) | (v >> (64 - r))) & M if r else v
    for rc in KECCAK_RC:
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x + 4) % 5] ^ rot(c[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        b = [0] * 25
        b[0] = a[0]
        x, y = 1, 0
        for t in range(24):
            ny = (2 * x + 3 * y) % 5
            b[y + 5 * ny] = rot(a[x + 5 * y], ((t + 1) * (t + 2) // 2) % 64)
            x, y = y, ny
        a = [b[i] ^ ((~b[(i % 5 + 1) % 5 + 5 * (i // 5)]) & M & b[(i % 5 + 2) % 5 + 5 * (i // 5)]) for i in range(25)]
        a[0] ^= rc
    return a


def _sha_case(msgs):
    """inputs by name + expected outputs by name: the padded block in, the digest out"""
    ins, outs = [], []
    for m in msgs:
        blk = m + b"\x80" + b"\0" * (55 - len(m)) + struct.pack(">Q", 8 * len(m))
        ins.append({f"0.in[{i}]": w for i, w in enumerate(struct.unpack(">16I", blk))})
        outs.append({f"0.out[{i}]": w for i, w in enumerate(struct.unpack(">8I", hashlib.sha256(m).digest()))})
    return ins, outs


def _keccak_case(msgs):
    ins, outs = [], []
    for m in msgs:
        blk = bytearray(m) + bytearray(136 - len(m))
        blk[len(m)] ^= 0x06
        blk[135] ^= 0x80
        lanes = list(struct.unpack("<17Q", bytes(blk))) + [0] * 8
        d = {f"0.in[{i}]": w for i, w in enumerate(lanes)}
        d.update({f"0.rc[{i}]": w for i, w in enumerate(KECCAK_RC)})
        d["0.ones"] = (1 << 64) - 1
        ins.append(d)
        full = keccak_f_numpy(lanes)
        assert struct.pack("<4Q", *full[:4]) == hashlib.sha3_256(m).digest()          # the numpy permutation is FIPS 202's
        outs.append({f"0.out[{i}]": w for i, w in enumerate(full)})
    return ins, outs


CASES = [("sha256Block", 32, 55, _sha_case, 3448, 16, 8), ("keccakF1600", 64, 135, _keccak_case, 5112, 50, 25)]


# ---- BASELINE.json configs[1..3] at their STATED sizes: the same templates with a block-count parameter
def _sha_multi_case(msgs, n_blocks):
    ins, outs = [], []
    for m in msgs:
        padded = m + b"\x80" + b"\0" * ((55 - len(m)) % 64) + struct.pack(">Q", 8 * len(m))
        assert len(padded) == 64 * n_blocks, (len(m), len(padded))
        ins.append({f"0.in[{i}]": w for i, w in enumerate(struct.unpack(f">{16 * n_blocks}I", padded))})
        outs.append({f"0.out[{i}]": w for i, w in enumerate(struct.unpack(">8I", hashlib.sha256(m).digest()))})
    return ins, outs


def _sha3_multi_case(msgs, n_blocks):
    ins, outs = [], []
    for m in msgs:
        blk = bytearray(m) + bytearray(136 * n_blocks - len(m))
        assert 136 * (n_blocks - 1) <= len(m) < 136 * n_blocks
        blk[len(m)] ^= 0x06
        blk[-1] ^= 0x80
        d = {f"0.in[{i}]": w for i, w in enumerate(struct.unpack(f"<{17 * n_blocks}Q", bytes(blk)))}
        d.update({f"0.rc[{i}]": w for i, w in enumerate(KECCAK_RC)})
        d["0.ones"] = (1 << 64) - 1
        ins.append(d)
        outs.append({f"0.out[{i}]": w for i, w in enumerate(struct.unpack("<4Q", hashlib.sha3_256(m).digest()))})
    return ins, outs


def poseidon_like_numpy(a, b):
    """tests/golden/circuits/poseidonLike.circom in plain integers mod 2^32 (the constants recomputed, not read from the text)"""
    M = 0xFFFFFFFF
    rc, x = [], 20241008
    for _ in range(36):
        x = (x * 1664525 + 1013904223) & M
        rc.append(x)
    mds = [[5, 7, 3], [3, 5, 7], [7, 3, 5]]
    st = [0, a & M, b & M]
    p5 = lambda v: (pow(v, 5, 1 << 32))
    for r in range(12):
        st = [(st[j] + rc[3 * r + j]) & M for j in range(3)]
        st = [p5(v) for v in st] if (r < 4 or r >= 8) else [p5(st[0]), st[1], st[2]]
        st = [sum(mds[i][j] * st[j] for j in range(3)) & M for i in range(3)]
    return st[0]


def _poseidon_case(msgs, _n):
    rng = np.random.default_rng(5)
    pairs = [(0, 0), (1, 2), (0xFFFFFFFF, 0xFFFFFFFF)] + [tuple(int(v) for v in rng.integers(0, 1 << 32, 2)) for _ in range(5)]
    return ([{"0.in[0]": a, "0.in[1]": b} for a, b in pairs], [{"0.out": poseidon_like_numpy(a, b)} for a, b in pairs])


def _lens(lo, hi):
    rng = np.random.default_rng(20241008)
    return [bytes(rng.integers(0, 256, int(L), dtype=np.uint8)) for L in [lo, hi] + list(rng.integers(lo, hi + 1, 6))]


# name, main as written, (blocks under the emulation, blocks on the GPU), width, messages(n_blocks), case maker, gates per block + fixed, n_in(n_blocks), n_out
SIZED = [
    ("poseidonLike", None, (1, 1), 32, lambda nb: [None] * 8, _poseidon_case, lambda nb: 301, lambda nb: 2, 1),
    ("sha256", "Sha256(9)", (2, 9), 32, lambda nb: _lens(64 * (nb - 1), 64 * nb - 9), _sha_multi_case, lambda nb: 3448 * nb, lambda nb: 16 * nb, 8),
    ("sha3_256", "Sha3_256(29)", (2, 29), 64, lambda nb: _lens(136 * (nb - 1), 136 * nb - 1), _sha3_multi_case, lambda nb: 5112 * nb + 8 + 17 * (nb - 1), lambda nb: 17 * nb + 25, 4),
]


@pytest.mark.parametrize("name,main,blocks,width,msgs_of,make,n_gates_of,n_in_of,n_out", SIZED, ids=[c[0] for c in SIZED])
def test_configs_at_their_stated_sizes(name, main, blocks, width, msgs_of, make, n_gates_of, n_in_of, n_out, backend, orc):
    """BASELINE.json configs[1] (Poseidon-shaped, 301 gates, AMul chains), configs[2] (SHA-256 over NINE blocks, 31 032 gates, width
    32) and configs[3] (the SHA3-256 sponge over 29 rate blocks, 148 732 gates, width 64) as real circuits from Circom text: sort /
    numbering / emission against the oracle's hash-map faithful build_circuit, and the ARITHMETIC circuit, its BOOLEAN image and the
    PRUNED image evaluated (c2a_eval) against hashlib / numpy on eight inputs.  Under the host emulation the block count is 2 (the
    same text with a smaller loop bound) and only the arithmetic circuit is evaluated; the stated sizes run on the hardware."""
    on_gpu = "hip" in backend.version
    nb = blocks[1] if on_gpu else blocks[0]
    comp_mod = importlib.import_module("circom-2-arithc_amd.compiler")
    text = open(os.path.join(CIRCUITS, f"{name}.circom")).read()
    if main is not None:
        assert text.count(main) == 1
        text = text.replace(main, main.split("(")[0] + f"({nb})")
    C = comp_mod.Compiler.from_circom(text, backend=backend)
    assert len(C.gates) == n_gates_of(nb)
    circ = C.build_circuit()
    inputs, outputs, _ = C._io_maps()
    lh, rh, out, op = C._flat()
    exp = orc.build_circuit(lh, rh, out, op, C.node_count + 1, np.array([nd for _, nd in inputs], np.uint32),
                            np.array([nd for _, nd in outputs], np.uint32), mode=0)
    assert circ.wire_count == exp.wire_count
    np.testing.assert_array_equal(circ.sorted_gate_ids, exp.sorted)
    assert int((exp.sorted != np.arange(len(lh))).sum()) > len(lh) // 10              # NOT the identity
    for a, b in zip((circ.in0, circ.in1, circ.out, circ.op), (exp.in0, exp.in1, exp.out, exp.op)):
        np.testing.assert_array_equal(a, b)
    iw, ow = circ.info.input_name_to_wire_index, circ.info.output_name_to_wire_index
    assert len(iw) == n_in_of(nb) and len(ow) == n_out
    ins_by_name, outs_by_name = make(msgs_of(nb), nb)
    T = len(ins_by_name)
    ins = np.zeros((len(iw), T), np.uint64)
    want = np.zeros((n_out, T), np.uint64)
    for t, (i_, o_) in enumerate(zip(ins_by_name, outs_by_name)):
        for k, v in i_.items():
            ins[iw[k], t] = v
        for k, v in o_.items():
            want[ow[k] - (circ.wire_count - n_out), t] = v
    cst = {c.wire_index: int(c.value) for c in circ.info.constants.values()}
    np.testing.assert_array_equal(backend.eval(ins, cst, width=width), want)
    if not on_gpu and name != "poseidonLike":
        return
    bi = backend.boolify(width)
    Tsz = np.array([orc.template_size(o, width)[0] for o in range(20)], dtype=np.int64)
    assert bi.n_gates == int(Tsz[exp.op].sum())
    np.testing.assert_array_equal(backend.eval(ins, cst, width=width, boolean=True), want)
    sl, g0 = orc.boolify_range(exp, width, len(lh) // 2, 100)
    for a, b in zip(backend.bool_read(g0, len(sl.in0)), (sl.in0, sl.in1, sl.out, sl.op)):
        np.testing.assert_array_equal(a, b)
    if not on_gpu:
        return
    checked, bad = backend.verify_boolify(seed=5)
    assert checked == circ.wire_count * 64 and bad == 0
    pi = backend.boolify_prune()
    assert 0 < pi["n_gates"] <= bi.n_gates + 2
    np.testing.assert_array_equal(backend.eval(ins, cst, width=width, pruned=True), want)


@pytest.mark.parametrize("name,width,max_len,make,n_gates,n_in,n_out", CASES, ids=[c[0] for c in CASES])
def test_hash_circuit_known_answers(name, width, max_len, make, n_gates, n_in, n_out, backend, orc):
    comp_mod = importlib.import_module("circom-2-arithc_amd.compiler")
    text = open(os.path.join(CIRCUITS, f"{name}.circom")).read()
    C = comp_mod.Compiler.from_circom(text, backend=backend)
    assert len(C.gates) == n_gates
    circ = C.build_circuit()
    # ---- the sort / numbering / emission against the oracle on the same flat payload (src/compiler.rs:321-494)
    inputs, outputs, _ = C._io_maps()
    lh, rh, out, op = C._flat()
    exp = orc.build_circuit(lh, rh, out, op, C.node_count + 1, np.array([nd for _, nd in inputs], np.uint32),
                            np.array([nd for _, nd in outputs], np.uint32), mode=0)        # the hash-map faithful variant
    assert circ.wire_count == exp.wire_count
    np.testing.assert_array_equal(circ.sorted_gate_ids, exp.sorted)
    assert int((exp.sorted != np.arange(n_gates)).sum()) > 500                            # NOT the identity: the DFS has work to do
    for a, b in zip((circ.in0, circ.in1, circ.out, circ.op), (exp.in0, exp.in1, exp.out, exp.op)):
        np.testing.assert_array_equal(a, b)
    assert backend.stats()["numbering_path"] == 1 and backend.stats()["numbering_events"] > 1000      # a constant-heavy REAL circuit
    # ---- known answers: arithmetic circuit, boolean image, pruned image
    iw, ow = circ.info.input_name_to_wire_index, circ.info.output_name_to_wire_index
    assert len(iw) == n_in and len(ow) == n_out
    msgs = _messages(max_len)
    assert len(msgs) == 8
    ins_by_name, outs_by_name = make(msgs)
    ins = np.zeros((n_in, len(msgs)), np.uint64)
    want = np.zeros((n_out, len(msgs)), np.uint64)
    for t, (i_, o_) in enumerate(zip(ins_by_name, outs_by_name)):
        for k, v in i_.items():
            ins[iw[k], t] = v
        for k, v in o_.items():
            want[ow[k] - (circ.wire_count - n_out), t] = v
    cst = {c.wire_index: int(c.value) for c in circ.info.constants.values()}
    np.testing.assert_array_equal(backend.eval(ins, cst, width=width), want)
    bi = backend.boolify(width)
    T = np.array([orc.template_size(o, width)[0] for o in range(20)], dtype=np.int64)
    assert bi.n_gates == int(T[exp.op].sum())
    on_gpu = "hip" in backend.version       # (under the host emulation a pass over 1-2 M boolean gates takes 20-50 s: one boolean image — SHA-256's — is enough there)
    if on_gpu or width == 32:
        np.testing.assert_array_equal(backend.eval(ins, cst, width=width, boolean=True), want)
    if on_gpu:
        checked, bad = backend.verify_boolify(seed=3)
        assert checked == circ.wire_count * 64 and bad == 0
    # the boolean circuit itself, bit for bit against the oracle's procedural bit-blast (a slice in the middle)
    sl, g0 = orc.boolify_range(exp, width, n_gates // 2, 200)
    for a, b in zip(backend.bool_read(g0, len(sl.in0)), (sl.in0, sl.in1, sl.out, sl.op)):
        np.testing.assert_array_equal(a, b)
    if not on_gpu:
        return
    pi = backend.boolify_prune()
    # (every gate of a hash feeds its outputs and the constants are wires whose values the pass does not know: nothing to fold or
    # drop here — at most the two constant wires' gates are added; what matters is that the image still computes the hash)
    assert 0 < pi["n_gates"] <= bi.n_gates + 2
    np.testing.assert_array_equal(backend.eval(ins, cst, width=width, pruned=True), want)
