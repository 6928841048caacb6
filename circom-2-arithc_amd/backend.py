"""ctypes binding of include/c2a.h (the drop-in boundary).  numpy in, numpy out.

The product path loads ``libc2a_hip.so`` (hand-written HIP, gfx950) that sits next to this file and
fails loudly when it is missing or no GPU is visible — there is no CPU fallback here.  ``lib_path`` exists
so that the CPU test-suite can point the same binding at tests/emul/libc2a_emul.so (the kernels compiled
against a host emulation header); nothing in the package itself ever passes it.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# src/a_gate_type.rs:8-27 — declaration order == discriminant == the u8 that crosses the ABI;
# the strum Display string is the Bristol op name (src/compiler.rs:462)
OP_NAMES = [
    "AAdd", "ADiv", "AEq", "AGEq", "AGt", "ALEq", "ALt", "AMul", "ANeq", "ASub", "AXor", "APow",
    "AIntDiv", "AMod", "AShiftL", "AShiftR", "ABoolOr", "ABoolAnd", "ABitOr", "ABitAnd",
]
OP = {name: i for i, name in enumerate(OP_NAMES)}
BOOL_OP_NAMES = ["XOR", "AND", "INV"]
NO_WIRE = 0xFFFFFFFF

C2A_OK, C2A_ERR_CYCLIC, C2A_ERR_INCONSISTENCY, C2A_ERR_OVERFLOW = 0, 1, 2, 3


class BackendError(RuntimeError):
    """Argument / state / HIP failure (negative status of include/c2a.h)."""


class CircuitError(Exception):
    """Mirror of CircuitError (src/compiler.rs:550-575); str() == the thiserror Display."""


class CyclicDependency(CircuitError):     # src/compiler.rs:570-571
    def __init__(self, i: int):
        self.index = int(i)
        self.message = f"detected at i={int(i)}"       # src/topological_sort.rs:36
        super().__init__(f"Cyclic dependency: {self.message}")


class Inconsistency(CircuitError):        # src/compiler.rs:572-573
    def __init__(self, message: str):
        self.message = message
        super().__init__(f"Inconsistency: {message}")


class _BoolInfo(ctypes.Structure):
    _fields_ = [("n_gates", ctypes.c_uint64), ("wire_count", ctypes.c_uint64), ("aux_total", ctypes.c_uint64),
                ("width", ctypes.c_uint32), ("n_in", ctypes.c_uint32), ("n_out", ctypes.c_uint32),
                ("m_wires", ctypes.c_uint32)]


class _PruneInfo(ctypes.Structure):
    _fields_ = [("n_gates", ctypes.c_uint64), ("n_gates_before", ctypes.c_uint64), ("n_folded", ctypes.c_uint64),
                ("n_dead", ctypes.c_uint64), ("wire_count", ctypes.c_uint64), ("zero_wire", ctypes.c_uint32), ("one_wire", ctypes.c_uint32)]


class _Timings(ctypes.Structure):
    _fields_ = [(k, ctypes.c_float) for k in ("prep", "peel", "order", "wires", "emit", "bool_prep", "bool_map",
                                              "build_total", "boolify_total", "k_peel")]


class _Stats(ctypes.Structure):
    _fields_ = [("n_gates", ctypes.c_uint64), ("n_edges", ctypes.c_uint64), ("levels", ctypes.c_uint32),
                ("max_depth", ctypes.c_uint32), ("n_roots", ctypes.c_uint32), ("n_splitters", ctypes.c_uint32),
                ("level_launches", ctypes.c_uint32), ("peel_waves", ctypes.c_uint32),
                ("path_chunks", ctypes.c_uint32), ("peel_rereads", ctypes.c_uint32),
                ("numbering_events", ctypes.c_uint32), ("numbering_path", ctypes.c_uint32),
                ("n_relays", ctypes.c_uint32), ("verifier", ctypes.c_uint32)]


@dataclass
class BoolInfo:
    n_gates: int
    wire_count: int
    aux_total: int
    width: int
    n_in: int
    n_out: int
    m_wires: int

    def wire(self, W, bit=0):
        """Boolean wire of (arithmetic wire W, bit) — layout of DESIGN.md §5.1."""
        W = np.asarray(W, dtype=np.int64)
        M, w = self.m_wires, self.width
        return np.where(W < M, W * w + bit, M * w + self.aux_total + (W - M) * w + bit)


ABI_VERSION = 7          # == C2A_ABI_VERSION of include/c2a.h this binding was written against

_EXPORTS = ["c2a_abi_version", "c2a_visible_devices", "c2a_create", "c2a_device_count", "c2a_format_bristol", "c2a_destroy", "c2a_last_error", "c2a_version", "c2a_load_gates", "c2a_load_circuit", "c2a_topo_sort",
            "c2a_topo_sort_serial", "c2a_assign_wires", "c2a_emit_gates", "c2a_build_circuit", "c2a_boolify",
            "c2a_bool_read", "c2a_template_size", "c2a_checksum", "c2a_get_timings", "c2a_get_stats", "c2a_verify_boolify",
            "c2a_debug_patch_bool_op", "c2a_debug_peel_abort", "c2a_debug_set_build_no", "c2a_debug_hot_every", "c2a_boolify_plan", "c2a_boolify_chunk", "c2a_boolify_shard_range", "c2a_eval", "c2a_boolify_prune", "c2a_pruned_read"]


def library_path() -> str:
    return os.path.join(_HERE, "libc2a_hip.so")


_libs = {}


def load_library(lib_path: Optional[str] = None):
    path = lib_path or library_path()
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise BackendError(
            f"{path} not found: build it with `make -C circom-2-arithc_amd/csrc` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback.")
    L = ctypes.CDLL(path)
    if not hasattr(L, "c2a_abi_version") or L.c2a_abi_version() != ABI_VERSION:
        got = L.c2a_abi_version() if hasattr(L, "c2a_abi_version") else "none (pre-versioning build)"
        raise BackendError(f"{path}: C ABI version {got}, this binding needs {ABI_VERSION} — rebuild the library "
                           "(make -C circom-2-arithc_amd/csrc)")
    L.c2a_visible_devices.restype = ctypes.c_int
    vp, u32p, u8p, u64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint8), \
        ctypes.POINTER(ctypes.c_uint64)
    L.c2a_create.restype = ctypes.c_int
    L.c2a_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(vp)]
    L.c2a_format_bristol.restype = ctypes.c_int
    L.c2a_format_bristol.argtypes = [vp, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64, u64p]
    L.c2a_device_count.restype = ctypes.c_int
    L.c2a_device_count.argtypes = [vp]
    L.c2a_destroy.restype = None
    L.c2a_destroy.argtypes = [vp]
    L.c2a_last_error.restype = ctypes.c_char_p
    L.c2a_last_error.argtypes = [vp]
    L.c2a_version.restype = ctypes.c_char_p
    L.c2a_load_circuit.restype = ctypes.c_int
    L.c2a_load_circuit.argtypes = [vp, ctypes.c_uint64, u32p, u32p, u32p, u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    L.c2a_load_gates.restype = ctypes.c_int
    L.c2a_load_gates.argtypes = [vp, ctypes.c_uint64, u32p, u32p, u32p, u8p, ctypes.c_uint32, ctypes.c_uint32, u32p,
                                 ctypes.c_uint32, u32p]
    L.c2a_topo_sort.restype = ctypes.c_int
    L.c2a_topo_sort.argtypes = [vp, u32p, u64p]
    L.c2a_topo_sort_serial.restype = ctypes.c_int
    L.c2a_topo_sort_serial.argtypes = [vp, u32p, u64p]
    L.c2a_assign_wires.restype = ctypes.c_int
    L.c2a_assign_wires.argtypes = [vp, u32p, u32p]
    L.c2a_emit_gates.restype = ctypes.c_int
    L.c2a_emit_gates.argtypes = [vp, u32p, u32p, u32p, u8p]
    L.c2a_build_circuit.restype = ctypes.c_int
    L.c2a_build_circuit.argtypes = [vp, u64p, u32p]
    L.c2a_boolify.restype = ctypes.c_int
    L.c2a_boolify.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(_BoolInfo)]
    L.c2a_boolify_plan.restype = ctypes.c_int
    L.c2a_boolify_plan.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(_BoolInfo)]
    L.c2a_boolify_chunk.restype = ctypes.c_int
    L.c2a_boolify_chunk.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, u32p, u32p, u32p, u8p, u64p, u64p]
    L.c2a_boolify_shard_range.restype = ctypes.c_int
    L.c2a_boolify_shard_range.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, u64p, u64p]
    L.c2a_bool_read.restype = ctypes.c_int
    L.c2a_bool_read.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, u32p, u32p, u32p, u8p]
    L.c2a_template_size.restype = ctypes.c_int
    L.c2a_template_size.argtypes = [ctypes.c_uint32, ctypes.c_uint32, u64p, u64p]
    L.c2a_checksum.restype = ctypes.c_int
    L.c2a_checksum.argtypes = [vp, ctypes.c_int, u64p]
    L.c2a_verify_boolify.restype = ctypes.c_int
    L.c2a_verify_boolify.argtypes = [vp, ctypes.c_uint64, u64p, u64p]
    L.c2a_boolify_prune.restype = ctypes.c_int
    L.c2a_boolify_prune.argtypes = [vp, ctypes.POINTER(_PruneInfo)]
    L.c2a_pruned_read.restype = ctypes.c_int
    L.c2a_pruned_read.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, u32p, u32p, u32p, u8p]
    L.c2a_eval.restype = ctypes.c_int
    L.c2a_eval.argtypes = [vp, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, u64p, ctypes.c_uint32, u32p, u64p, u64p]
    L.c2a_debug_patch_bool_op.restype = ctypes.c_int
    L.c2a_debug_patch_bool_op.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint8]
    for name in ("c2a_debug_peel_abort", "c2a_debug_set_build_no", "c2a_debug_hot_every"):
        getattr(L, name).restype = ctypes.c_int
        getattr(L, name).argtypes = [vp, ctypes.c_uint32]
    L.c2a_get_timings.restype = ctypes.c_int
    L.c2a_get_timings.argtypes = [vp, ctypes.POINTER(_Timings)]
    L.c2a_get_stats.restype = ctypes.c_int
    L.c2a_get_stats.argtypes = [vp, ctypes.POINTER(_Stats)]
    _libs[path] = L
    return L


def visible_devices(lib_path: Optional[str] = None) -> int:
    """HIP devices c2a_create can be given (0 without a GPU)."""
    return int(load_library(lib_path).c2a_visible_devices())


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a: Optional[np.ndarray], ct):
    if a is None:
        return ctypes.POINTER(ct)()
    return a.ctypes.data_as(ctypes.POINTER(ct))


CHECKSUM_STREAMS = {"sorted": 0, "in0": 1, "in1": 2, "out": 3, "op": 4, "bool_in0": 5, "bool_in1": 6, "bool_out": 7,
                    "bool_op": 8, "node_wire1": 9}


class Backend:
    """One c2a context.  `device` is one HIP device id or a list of them: the first is the primary (sort, numbering,
    emission), boolify is cut by sorted-position range over all of them.  Call order mirrors build_circuit + boolify:
    load_gates -> topo_sort -> assign_wires -> emit_gates -> boolify (or build_circuit for the first three)."""

    def __init__(self, device=0, lib_path: Optional[str] = None):
        self._lib = load_library(lib_path)
        self._ctx = ctypes.c_void_p()
        ids = [int(device)] if np.isscalar(device) else [int(d) for d in device]
        self.devices = ids
        arr = (ctypes.c_int * len(ids))(*ids)
        rc = self._lib.c2a_create(len(ids), arr, ctypes.byref(self._ctx))
        if rc != C2A_OK:
            raise BackendError(f"c2a_create(device={device}) failed with status {rc}: no usable HIP device "
                               "(this back end has no CPU fallback)")
        self.n = 0
        self.n_nodes = 0
        self._n_in = self._n_out = 0
        self.wire_count = None

    # -- lifecycle -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.c2a_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def version(self) -> str:
        return self._lib.c2a_version().decode()

    def _check(self, rc: int, cycle_at: int = 0):
        if rc == C2A_OK:
            return
        msg = self._lib.c2a_last_error(self._ctx).decode()
        if rc == C2A_ERR_CYCLIC:
            raise CyclicDependency(cycle_at)
        if rc == C2A_ERR_INCONSISTENCY:
            raise Inconsistency(msg.split("Inconsistency: ", 1)[-1])
        if rc == C2A_ERR_OVERFLOW:
            raise OverflowError(msg)
        raise BackendError(f"c2a status {rc}: {msg}")

    # -- the ABI ------------------------------------------------------------------------------
    def load_gates(self, lh, rh, out, op, n_nodes: int, input_nodes, output_nodes):
        lh, rh, out, op = _c(lh, np.uint32), _c(rh, np.uint32), _c(out, np.uint32), _c(op, np.uint8)
        inn, outn = _c(input_nodes, np.uint32), _c(output_nodes, np.uint32)
        if not (len(lh) == len(rh) == len(out) == len(op)):
            raise ValueError("gate arrays must have equal length")
        rc = self._lib.c2a_load_gates(self._ctx, len(lh), _p(lh, ctypes.c_uint32), _p(rh, ctypes.c_uint32),
                                      _p(out, ctypes.c_uint32), _p(op, ctypes.c_uint8), int(n_nodes), len(inn),
                                      _p(inn, ctypes.c_uint32), len(outn), _p(outn, ctypes.c_uint32))
        self._check(rc)
        self.n, self.n_nodes = len(lh), int(n_nodes)
        self._n_in, self._n_out = len(inn), len(outn)
        self.wire_count = None

    def load_circuit(self, in0, in1, out, op, wire_count: int, n_in: int, n_out: int):
        """an arithmetic circuit the host built itself (the reference's own build_circuit), for boolify alone: main.rs:30-32"""
        in0, in1, out, op = _c(in0, np.uint32), _c(in1, np.uint32), _c(out, np.uint32), _c(op, np.uint8)
        if not (len(in0) == len(in1) == len(out) == len(op)):
            raise ValueError("gate arrays must have equal length")
        self._check(self._lib.c2a_load_circuit(self._ctx, len(in0), _p(in0, ctypes.c_uint32), _p(in1, ctypes.c_uint32), _p(out, ctypes.c_uint32),
                                               _p(op, ctypes.c_uint8), int(wire_count), int(n_in), int(n_out)))
        self.n, self.n_nodes = len(in0), 0
        self._n_in, self._n_out = int(n_in), int(n_out)
        self.wire_count = int(wire_count)

    def topo_sort(self, fetch: bool = True, serial: bool = False) -> Optional[np.ndarray]:
        """sorted_gate_ids == topological_sort (src/topological_sort.rs:3-21)."""
        res = np.empty(self.n, dtype=np.uint32) if fetch else None
        cyc = ctypes.c_uint64(0)
        fn = self._lib.c2a_topo_sort_serial if serial else self._lib.c2a_topo_sort
        rc = fn(self._ctx, _p(res, ctypes.c_uint32), ctypes.byref(cyc))
        self._check(rc, cyc.value)
        return res

    def assign_wires(self, fetch: bool = True) -> Tuple[Optional[np.ndarray], int]:
        res = np.empty(self.n_nodes, dtype=np.uint32) if fetch else None
        wc = ctypes.c_uint32(0)
        rc = self._lib.c2a_assign_wires(self._ctx, _p(res, ctypes.c_uint32), ctypes.byref(wc))
        self._check(rc)
        self.wire_count = wc.value
        return res, wc.value

    def emit_gates(self, fetch: bool = True):
        if fetch:
            in0, in1, out = (np.empty(self.n, dtype=np.uint32) for _ in range(3))
            op = np.empty(self.n, dtype=np.uint8)
        else:
            in0 = in1 = out = op = None
        rc = self._lib.c2a_emit_gates(self._ctx, _p(in0, ctypes.c_uint32), _p(in1, ctypes.c_uint32),
                                      _p(out, ctypes.c_uint32), _p(op, ctypes.c_uint8))
        self._check(rc)
        return in0, in1, out, op

    def build_circuit(self) -> int:
        """topo_sort + assign_wires + emit_gates, results left in HBM. Returns wire_count."""
        cyc, wc = ctypes.c_uint64(0), ctypes.c_uint32(0)
        rc = self._lib.c2a_build_circuit(self._ctx, ctypes.byref(cyc), ctypes.byref(wc))
        self._check(rc, cyc.value)
        self.wire_count = wc.value
        return wc.value

    def boolify(self, width: int) -> BoolInfo:
        info = _BoolInfo()
        rc = self._lib.c2a_boolify(self._ctx, int(width), ctypes.byref(info))
        self._check(rc)
        self.bool_info = BoolInfo(int(info.n_gates), int(info.wire_count), int(info.aux_total), int(info.width),
                                  int(info.n_in), int(info.n_out), int(info.m_wires))
        return self.bool_info

    def boolify_plan(self, width: int) -> BoolInfo:
        """Sizes / offsets / wire layout only (no boolean gates are produced)."""
        info = _BoolInfo()
        self._check(self._lib.c2a_boolify_plan(self._ctx, int(width), ctypes.byref(info)))
        self.bool_info = BoolInfo(int(info.n_gates), int(info.wire_count), int(info.aux_total), int(info.width),
                                  int(info.n_in), int(info.n_out), int(info.m_wires))
        return self.bool_info

    def boolify_chunk(self, first_gate: int, n_gates: int, fetch: bool = True):
        """Boolean gates of the arithmetic gates at sorted positions [first_gate, first_gate + n_gates).
        Returns (first boolean gate index, (in0, in1, out, op))."""
        q0, cnt = ctypes.c_uint64(0), ctypes.c_uint64(0)
        if not fetch:
            self._check(self._lib.c2a_boolify_chunk(self._ctx, int(first_gate), int(n_gates), None, None, None, None,
                                                    ctypes.byref(q0), ctypes.byref(cnt)))
            return q0.value, cnt.value
        # size the host arrays from the plan: run once without copies to learn the count
        self._check(self._lib.c2a_boolify_chunk(self._ctx, int(first_gate), int(n_gates), None, None, None, None,
                                                ctypes.byref(q0), ctypes.byref(cnt)))
        in0, in1, out = (np.empty(cnt.value, dtype=np.uint32) for _ in range(3))
        op = np.empty(cnt.value, dtype=np.uint8)
        self._check(self._lib.c2a_boolify_chunk(self._ctx, int(first_gate), int(n_gates), _p(in0, ctypes.c_uint32),
                                                _p(in1, ctypes.c_uint32), _p(out, ctypes.c_uint32), _p(op, ctypes.c_uint8),
                                                ctypes.byref(q0), ctypes.byref(cnt)))
        return q0.value, (in0, in1, out, op)

    def boolify_shard_range(self, k: int, n_shards: int) -> Tuple[int, int]:
        """Sorted positions (first_gate, n_gates) of shard k of n_shards ranges with equal boolean-gate counts (after a plan)."""
        first, cnt = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._lib.c2a_boolify_shard_range(self._ctx, int(k), int(n_shards), ctypes.byref(first), ctypes.byref(cnt)))
        return first.value, cnt.value

    def bool_read(self, first: int = 0, count: Optional[int] = None):
        if count is None:
            count = self.bool_info.n_gates - first
        in0, in1, out = (np.empty(count, dtype=np.uint32) for _ in range(3))
        op = np.empty(count, dtype=np.uint8)
        rc = self._lib.c2a_bool_read(self._ctx, int(first), int(count), _p(in0, ctypes.c_uint32),
                                     _p(in1, ctypes.c_uint32), _p(out, ctypes.c_uint32), _p(op, ctypes.c_uint8))
        self._check(rc)
        return in0, in1, out, op

    def format_bristol(self, which: int, first: int, count: int) -> bytes:
        """The gate lines of circuit.txt for gates [first, first + count), printed by the GPU (c2a_format_bristol):
        which = 0 arithmetic circuit, 1 boolean circuit, 2 the last boolify chunk."""
        need = ctypes.c_uint64(0)
        self._check(self._lib.c2a_format_bristol(self._ctx, int(which), int(first), int(count), None, 0, ctypes.byref(need)))
        buf = ctypes.create_string_buffer(max(1, need.value))
        got = ctypes.c_uint64(0)
        self._check(self._lib.c2a_format_bristol(self._ctx, int(which), int(first), int(count), buf, need.value, ctypes.byref(got)))
        return buf.raw[:got.value]

    def template_size(self, op: int, width: int) -> Tuple[int, int]:
        g, a = ctypes.c_uint64(0), ctypes.c_uint64(0)
        rc = self._lib.c2a_template_size(int(op), int(width), ctypes.byref(g), ctypes.byref(a))
        if rc:
            raise BackendError(f"c2a_template_size status {rc}")
        return g.value, a.value

    def checksum(self, stream: str) -> int:
        v = ctypes.c_uint64(0)
        self._check(self._lib.c2a_checksum(self._ctx, CHECKSUM_STREAMS[stream], ctypes.byref(v)))
        return v.value

    def verify_boolify(self, seed: int = 1) -> Tuple[int, int]:
        """GPU simulation of the arithmetic circuit and its boolean image on 64 seeded vectors; returns
        (number of (wire, vector) pairs compared, number that differ)."""
        chk, bad = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._lib.c2a_verify_boolify(self._ctx, int(seed), ctypes.byref(chk), ctypes.byref(bad)))
        return chk.value, bad.value

    def boolify_prune(self) -> dict:
        """The optional prune pass over the circuit of boolify() (constant folding + dead-gate removal); returns its counts."""
        info = _PruneInfo()
        self._check(self._lib.c2a_boolify_prune(self._ctx, ctypes.byref(info)))
        self.prune_info = {k: int(getattr(info, k)) for k, _ in _PruneInfo._fields_}
        return self.prune_info

    def pruned_read(self, first: int = 0, count: Optional[int] = None):
        if count is None:
            count = self.prune_info["n_gates"] - first
        in0, in1, out = (np.empty(count, dtype=np.uint32) for _ in range(3))
        op = np.empty(count, dtype=np.uint8)
        self._check(self._lib.c2a_pruned_read(self._ctx, int(first), int(count), _p(in0, ctypes.c_uint32), _p(in1, ctypes.c_uint32),
                                              _p(out, ctypes.c_uint32), _p(op, ctypes.c_uint8)))
        return in0, in1, out, op

    def eval(self, inputs, constants=None, width: int = 32, boolean: bool = False, pruned: bool = False) -> np.ndarray:
        """Run the circuit on caller-supplied values on the GPU (c2a_eval: the reference's simulation harness,
        tests/integration.rs:191-237).  inputs: array [n_in] or [n_in, T] (T <= 64 vectors), in the order of load_gates'
        input list; constants: {arithmetic wire: value}; returns [n_out, T] uint64.  boolean=True evaluates the circuit of
        boolify() (same values in, same values out: bit-slicing happens on the device)."""
        a = np.ascontiguousarray(np.asarray(inputs, dtype=np.uint64))
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        T = a.shape[1] if a.size else 1
        cst = constants or {}
        cw = np.ascontiguousarray(np.fromiter(cst.keys(), dtype=np.uint32, count=len(cst)))
        cv = np.ascontiguousarray(np.fromiter((int(v) & 0xFFFFFFFFFFFFFFFF for v in cst.values()), dtype=np.uint64, count=len(cst)))
        n_out = ctypes.c_uint32(0)
        out = np.zeros((self._n_out, T), dtype=np.uint64)
        self._check(self._lib.c2a_eval(self._ctx, 2 if pruned else (1 if boolean else 0), int(width), int(T), _p(a, ctypes.c_uint64) if a.size else None,
                                       len(cst), _p(cw, ctypes.c_uint32) if len(cst) else None,
                                       _p(cv, ctypes.c_uint64) if len(cst) else None, _p(out, ctypes.c_uint64) if out.size else None))
        return out

    def debug_peel_abort(self, launches: int):
        """tests: the next `launches` dataflow launches count as given up (retry, then the serial DFS)"""
        self._check(self._lib.c2a_debug_peel_abort(self._ctx, int(launches)))

    def debug_set_build_no(self, build_no: int):
        """tests: continue the build numbers (tags of the node-table records) from here; after load_gates"""
        self._check(self._lib.c2a_debug_set_build_no(self._ctx, int(build_no)))

    def debug_hot_every(self, ticket: int):
        """tests: the consumer ticket from which a producer counts as hot (power of two)"""
        self._check(self._lib.c2a_debug_hot_every(self._ctx, int(ticket)))

    def debug_patch_bool_op(self, index: int, new_op: int):
        """Fault injection for the verifier's tests."""
        self._check(self._lib.c2a_debug_patch_bool_op(self._ctx, int(index), int(new_op)))

    def timings(self) -> dict:
        t = _Timings()
        self._check(self._lib.c2a_get_timings(self._ctx, ctypes.byref(t)))
        return {k: float(getattr(t, k)) for k, _ in _Timings._fields_}

    def stats(self) -> dict:
        s = _Stats()
        self._check(self._lib.c2a_get_stats(self._ctx, ctypes.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in _Stats._fields_}


def checksum_host(a: np.ndarray) -> int:
    """Host-side twin of the GPU checksum kernels (k_checksum_u32/u8): sum_i mix64((i<<32) ^ (i>>32) ^ v*GOLD)."""
    a = np.ascontiguousarray(a)
    v = a.astype(np.uint64)
    i = np.arange(len(v), dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (i << np.uint64(32)) ^ (i >> np.uint64(32)) ^ (v * np.uint64(0x9E3779B97F4A7C15))
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xff51afd7ed558ccd)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xc4ceb9fe1a85ec53)
        x ^= x >> np.uint64(33)
        return int(np.add.reduce(x, dtype=np.uint64))
