"""CPU oracle for the flat-gate-graph stage of circom-2-arithc — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (circom-2-arithc_amd/) never does.

Two layers:

* pure-Python *literal* restatements (recursion kept, dicts kept) of
    - src/topological_sort.rs:3-50            -> topological_sort_literal
    - src/compiler.rs:139-278 (gate-graph builder) -> CompilerModel.add_signal/add_gate/add_connection
    - src/compiler.rs:321-494 (build_circuit) -> CompilerModel.build_circuit
  used on small cases to pin the C restatement and to replay the reference's integration fixtures;
* ctypes bindings to oracle/c2a_oracle.c (same algorithms, explicit stack, fast) used for the
  seeded random parity tests and as bench.py's cpu_baseline.

Parity status: the reference cannot be built here (Rust, un-vendored git deps) — pinned against
tests/integration.rs expectations + SURVEY.md Appendix A hand traces (tests/golden/).
boolify / write_bristol sources are absent (third-party crates) => PARITY UNPINNED for those; the
bit-blast follows the frozen spec of DESIGN.md §5.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libc2a_oracle.so")

NONE = 0xFFFFFFFF

# src/a_gate_type.rs:8-27, declaration order == discriminant
OP_NAMES = [
    "AAdd", "ADiv", "AEq", "AGEq", "AGt", "ALEq", "ALt", "AMul", "ANeq", "ASub", "AXor", "APow",
    "AIntDiv", "AMod", "AShiftL", "AShiftR", "ABoolOr", "ABoolAnd", "ABitOr", "ABitAnd",
]
OP = {name: i for i, name in enumerate(OP_NAMES)}
BOOL_OP_NAMES = ["XOR", "AND", "INV"]

ORC_OK, ORC_CYCLIC, ORC_INCONSISTENCY, ORC_OVERFLOW, ORC_ARG, ORC_NOMEM = range(6)


class CircuitError(Exception):
    """Mirror of CircuitError (src/compiler.rs:550-575); str() matches the thiserror Display."""


class CyclicDependency(CircuitError):
    def __init__(self, message: str):
        self.message = message
        super().__init__(f"Cyclic dependency: {message}")


class Inconsistency(CircuitError):
    def __init__(self, message: str):
        self.message = message
        super().__init__(f"Inconsistency: {message}")


# ---------------------------------------------------------------------------------------------
# literal restatement of src/topological_sort.rs
# ---------------------------------------------------------------------------------------------
def topological_sort_literal(length: int, get_deps: Callable[[int], List[int]]) -> List[int]:
    """src/topological_sort.rs:3-21 (recursive, like the reference)."""
    sorted_: List[int] = []
    visiting = [False] * length
    visited = [False] * length
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, length * 2 + 1000))
    try:
        for i in range(length):
            _topological_sort_visit(i, visiting, visited, get_deps, sorted_)
    finally:
        sys.setrecursionlimit(old)
    assert len(sorted_) == length, "Topological sort did not return all elements"
    return sorted_


def _topological_sort_visit(i, visiting, visited, get_deps, sorted_):
    """src/topological_sort.rs:23-50."""
    if visited[i]:
        return
    if visiting[i]:
        raise CyclicDependency(f"detected at i={i}")
    visiting[i] = True
    for j in get_deps(i):
        _topological_sort_visit(j, visiting, visited, get_deps, sorted_)
    sorted_.append(i)
    visited[i] = True


# ---------------------------------------------------------------------------------------------
# literal restatement of the gate-graph builder + build_circuit (src/compiler.rs)
# ---------------------------------------------------------------------------------------------
@dataclass
class Signal:          # compiler.rs:16-20
    name: str
    value: Optional[int]


@dataclass
class Node:            # compiler.rs:32-36
    is_const: bool = False
    is_out: bool = False
    signals: List[int] = field(default_factory=list)


@dataclass
class ArithmeticGate:  # compiler.rs:85-90
    op: int
    lh_in: int
    rh_in: int
    out: int


@dataclass
class ConstantInfo:    # bristol-circuit ConstantInfo as used at compiler.rs:471-474
    value: str
    wire_index: int


@dataclass
class BristolCircuit:  # bristol-circuit types as used at compiler.rs:478-493
    wire_count: int
    input_name_to_wire_index: Dict[str, int]
    constants: Dict[str, ConstantInfo]
    output_name_to_wire_index: Dict[str, int]
    gates: List[Tuple[int, int, int, str]]          # (in0, in1, out, op name)
    sorted_gate_ids: List[int]
    node_id_to_wire_id: Dict[int, int]
    io_widths: Optional[Tuple[List[int], List[int]]] = None


class CompilerModel:
    """Python mirror of `Compiler` (src/compiler.rs:107-115) with the mutators the unroller calls."""

    def __init__(self):
        self.node_count = 0
        self.inputs: Dict[int, str] = {}
        self.outputs: Dict[int, str] = {}
        self.signals: Dict[int, Signal] = {}
        self.nodes: Dict[int, Node] = {}
        self.gates: List[ArithmeticGate] = []

    def _get_node_id(self) -> int:                      # compiler.rs:497-500
        self.node_count += 1
        return self.node_count

    def add_inputs(self, inputs: Dict[int, str]):       # compiler.rs:130-132
        self.inputs.update(inputs)

    def add_outputs(self, outputs: Dict[int, str]):     # compiler.rs:134-136
        self.outputs.update(outputs)

    def add_signal(self, sid: int, name: str, value: Optional[int] = None):   # compiler.rs:139-161
        if sid in self.signals:
            raise CircuitError("Signal already declared")
        self.signals[sid] = Signal(name, value)
        node = Node(is_const=value is not None, is_out=False, signals=[sid])
        self.nodes[self._get_node_id()] = node

    def get_signals(self, flt: str) -> Dict[int, str]:  # compiler.rs:163-171
        return {sid: s.name for sid, s in self.signals.items() if s.name.startswith(flt)}

    def _node_of(self, sid: int) -> int:
        found = 0
        for nid, node in self.nodes.items():
            if sid in node.signals:
                found = nid
        return found

    def add_gate(self, op: int, lhs: int, rhs: int, out: int):                 # compiler.rs:174-209
        ids = [self._node_of(lhs), self._node_of(rhs), self._node_of(out)]
        self.nodes[ids[2]].is_out = True
        self.gates.append(ArithmeticGate(op, ids[0], ids[1], ids[2]))

    def add_connection(self, a: int, b: int):                                  # compiler.rs:213-278
        na, nb = self._node_of(a), self._node_of(b)
        if na == nb:
            return
        # (:215-228: a signal found in no node leaves the scan on its placeholder `(0, &Node::new())`)
        node_a = self.nodes[na] if na in self.nodes else Node(is_const=False, is_out=False, signals=[])
        node_b = self.nodes[nb] if nb in self.nodes else Node(is_const=False, is_out=False, signals=[])
        if node_a.is_out and node_b.is_out:
            raise CircuitError("Cannot merge output nodes")
        if node_a.is_const and node_b.is_const:
            raise CircuitError("Cannot merge constant nodes")
        merged = Node(is_const=node_a.is_const or node_b.is_const, is_out=node_a.is_out or node_b.is_out,
                      signals=list(node_a.signals) + list(node_b.signals))
        mid = self._get_node_id()
        for g in self.gates:
            if g.lh_in in (na, nb):
                g.lh_in = mid
            if g.rh_in in (na, nb):
                g.rh_in = mid
            if g.out in (na, nb):
                g.out = mid
        self.nodes.pop(na, None)                                                # HashMap::remove (:273-274)
        self.nodes.pop(nb, None)
        self.nodes[mid] = merged

    def generate_circuit_report(self, value_type: str = "sint") -> dict:          # compiler.rs:287-319, :502-531
        input_nodes, output_nodes = [], []
        for nid, node in self.nodes.items():                                      # :291-297
            (output_nodes if node.is_out else input_nodes).append(nid)
        output_nodes = [nid for nid in output_nodes                               # :300-304
                        if all(g.lh_in != nid and g.rh_in != nid for g in self.gates)]
        input_nodes.sort(); output_nodes.sort()                                   # :307-308

        def reports(ids):                                                         # :503-531
            out = []
            for nid in ids:
                names, value = [], None
                for sid in self.nodes[nid].signals:
                    sg = self.signals[sid]
                    if "random_" not in sg.name:                                  # :519
                        names.append(sg.name)
                    if sg.value is not None:                                      # :522-524
                        value = sg.value
                out.append({"id": nid, "names": names, "value": value})
            return out
        return {"inputs": reports(input_nodes), "outputs": reports(output_nodes), "value_type": value_type}

    # -- name maps (compiler.rs:323-383), canonical order of DESIGN.md §3 -------------------------
    def io_maps(self):
        """Returns (inputs [(name,node)], outputs [(name,node)], constants {key:(node,value)}).

        The reference fills std HashMaps and later iterates them (order undefined);
        the canonical order is ascending signal id of the named IO signal.
        """
        sig_node = {sid: nid for nid, node in self.nodes.items() for sid in node.signals}
        inputs: List[Tuple[str, int]] = []
        outputs: List[Tuple[str, int]] = []
        constants: Dict[str, Tuple[int, str]] = {}
        seen_in, seen_out = set(), set()
        for sid in sorted(sig_node):
            nid = sig_node[sid]
            if sid in self.inputs:
                name = self.inputs[sid]
                if name in seen_in:
                    raise Inconsistency(f"Duplicate input {name}")         # compiler.rs:335-339
                seen_in.add(name)
                inputs.append((name, nid))
            if sid in self.outputs:
                name = self.outputs[sid]
                if name in seen_out:
                    raise Inconsistency(f"Duplicate output {name}")        # compiler.rs:345-349
                seen_out.add(name)
                outputs.append((name, nid))
            sig = self.signals[sid]
            if sig.value is not None:
                constants[f"{sig.name}_{sid}"] = (nid, str(sig.value))       # compiler.rs:354-359
        node_to_input = {nid: name for name, nid in inputs}
        for name, nid in outputs:                                            # compiler.rs:363-383
            if nid in node_to_input:
                raise Inconsistency(
                    f"Node {nid} used for both input {node_to_input[nid]} and output {name}")
        return inputs, outputs, constants

    def build_circuit(self) -> BristolCircuit:
        """Literal restatement of src/compiler.rs:321-494 (dicts for HashMaps, recursive DFS)."""
        inputs, outputs, constants = self.io_maps()
        node_id_to_wire_id: Dict[int, int] = {}
        next_wire_id = 0
        for _, nid in inputs:                                                # :392-395
            node_id_to_wire_id[nid] = next_wire_id
            next_wire_id += 1
        node_id_to_required_gate: Dict[int, int] = {}
        for gid, gate in enumerate(self.gates):                              # :403-406
            node_id_to_required_gate[gate.out] = gid

        def get_deps(gid: int) -> List[int]:                                 # :408-421
            gate = self.gates[gid]
            deps = []
            if gate.lh_in in node_id_to_required_gate:
                deps.append(node_id_to_required_gate[gate.lh_in])
            if gate.rh_in in node_id_to_required_gate:
                deps.append(node_id_to_required_gate[gate.rh_in])
            return deps

        sorted_gate_ids = topological_sort_literal(len(self.gates), get_deps)
        output_node_ids = {nid for _, nid in outputs}                        # :423
        for gid in sorted_gate_ids:                                          # :427-443
            gate = self.gates[gid]
            for nid in (gate.lh_in, gate.rh_in, gate.out):
                if nid in output_node_ids:
                    continue
                if nid in node_id_to_wire_id:
                    continue
                node_id_to_wire_id[nid] = next_wire_id
                next_wire_id += 1
        for _, nid in outputs:                                               # :446-449
            node_id_to_wire_id[nid] = next_wire_id
            next_wire_id += 1
        new_gates = []
        for gid in sorted_gate_ids:                                          # :453-464
            gate = self.gates[gid]
            new_gates.append((node_id_to_wire_id[gate.lh_in], node_id_to_wire_id[gate.rh_in],
                              node_id_to_wire_id[gate.out], OP_NAMES[gate.op]))
        consts = {name: ConstantInfo(value, node_id_to_wire_id[nid])         # :466-476 (KeyError == Rust panic)
                  for name, (nid, value) in constants.items()}
        return BristolCircuit(
            wire_count=next_wire_id,
            input_name_to_wire_index={n: node_id_to_wire_id[nid] for n, nid in inputs},
            constants=consts,
            output_name_to_wire_index={n: node_id_to_wire_id[nid] for n, nid in outputs},
            gates=new_gates, sorted_gate_ids=sorted_gate_ids, node_id_to_wire_id=node_id_to_wire_id)

    # -- boundary payload ----------------------------------------------------------------------
    def flat_payload(self):
        """The flat SoA payload that crosses the C-ABI: raw node ids, canonical IO."""
        inputs, outputs, constants = self.io_maps()
        n = len(self.gates)
        lh = np.fromiter((g.lh_in for g in self.gates), dtype=np.uint32, count=n)
        rh = np.fromiter((g.rh_in for g in self.gates), dtype=np.uint32, count=n)
        out = np.fromiter((g.out for g in self.gates), dtype=np.uint32, count=n)
        op = np.fromiter((g.op for g in self.gates), dtype=np.uint8, count=n)
        return dict(lh=lh, rh=rh, out=out, op=op, n_nodes=self.node_count + 1,
                    input_nodes=np.array([nid for _, nid in inputs], dtype=np.uint32),
                    output_nodes=np.array([nid for _, nid in outputs], dtype=np.uint32),
                    input_names=[nm for nm, _ in inputs], output_names=[nm for nm, _ in outputs],
                    constants=constants)


# ---------------------------------------------------------------------------------------------
# ctypes bindings to c2a_oracle.c
# ---------------------------------------------------------------------------------------------
class _OrcCircuit(ctypes.Structure):
    _fields_ = [("n", ctypes.c_uint64), ("n_nodes", ctypes.c_uint32), ("n_in", ctypes.c_uint32),
                ("n_out", ctypes.c_uint32), ("wire_count", ctypes.c_uint32),
                ("sorted", ctypes.POINTER(ctypes.c_uint32)), ("in0", ctypes.POINTER(ctypes.c_uint32)),
                ("in1", ctypes.POINTER(ctypes.c_uint32)), ("out", ctypes.POINTER(ctypes.c_uint32)),
                ("op", ctypes.POINTER(ctypes.c_uint8)), ("node_wire", ctypes.POINTER(ctypes.c_uint32))]


class _OrcBool(ctypes.Structure):
    _fields_ = [("n_gates", ctypes.c_uint64), ("wire_count", ctypes.c_uint64), ("width", ctypes.c_uint32),
                ("n_in", ctypes.c_uint32), ("n_out", ctypes.c_uint32),
                ("in0", ctypes.POINTER(ctypes.c_uint32)), ("in1", ctypes.POINTER(ctypes.c_uint32)),
                ("out", ctypes.POINTER(ctypes.c_uint32)), ("op", ctypes.POINTER(ctypes.c_uint8))]


_lib = None


def build_lib(force: bool = False) -> str:
    src = os.path.join(_HERE, "c2a_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libc2a_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build_lib()
        L = ctypes.CDLL(_LIB_PATH)
        u32p, u8p, u64p = (ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint8),
                           ctypes.POINTER(ctypes.c_uint64))
        L.orc_build_circuit.restype = ctypes.c_int
        L.orc_build_circuit.argtypes = [ctypes.c_uint64, u32p, u32p, u32p, u8p, ctypes.c_uint32, ctypes.c_uint32,
                                        u32p, ctypes.c_uint32, u32p, ctypes.c_int,
                                        ctypes.POINTER(ctypes.POINTER(_OrcCircuit)), u64p]
        L.orc_free_circuit.argtypes = [ctypes.POINTER(_OrcCircuit)]
        L.orc_free_circuit.restype = None
        L.orc_topo_sort_deps.restype = ctypes.c_int
        L.orc_topo_sort_deps.argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_int64),
                                         ctypes.POINTER(ctypes.c_int64), u32p, u64p]
        L.orc_boolify.restype = ctypes.c_int
        L.orc_boolify.argtypes = [ctypes.POINTER(_OrcCircuit), ctypes.c_uint32,
                                  ctypes.POINTER(ctypes.POINTER(_OrcBool))]
        L.orc_boolify_range.restype = ctypes.c_int
        L.orc_boolify_range.argtypes = [ctypes.POINTER(_OrcCircuit), ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64,
                                        ctypes.POINTER(ctypes.POINTER(_OrcBool)), u64p]
        L.orc_free_bool.argtypes = [ctypes.POINTER(_OrcBool)]
        L.orc_free_bool.restype = None
        L.orc_template_size.restype = ctypes.c_int
        L.orc_template_size.argtypes = [ctypes.c_int, ctypes.c_uint32, u64p, u64p]
        L.orc_eval_op.restype = ctypes.c_uint64
        L.orc_eval_op.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32]
        L.orc_eval_arith.restype = ctypes.c_int
        L.orc_eval_arith.argtypes = [ctypes.c_uint64, u32p, u32p, u32p, u8p, ctypes.c_uint32, ctypes.c_uint64, u64p]
        L.orc_eval_bool.restype = ctypes.c_int
        L.orc_eval_bool.argtypes = [ctypes.c_uint64, u32p, u32p, u32p, u8p, u64p]
        L.orc_fnv1a.restype = ctypes.c_uint64
        L.orc_fnv1a.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        _lib = L
    return _lib


def _p(a: np.ndarray, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


@dataclass
class ArithCircuit:
    """Numeric result of build_circuit (sorted order, dense wire ids)."""
    sorted: np.ndarray
    in0: np.ndarray
    in1: np.ndarray
    out: np.ndarray
    op: np.ndarray
    node_wire: np.ndarray
    wire_count: int
    n_in: int
    n_out: int


@dataclass
class BoolCircuit:
    in0: np.ndarray
    in1: np.ndarray
    out: np.ndarray
    op: np.ndarray
    wire_count: int
    width: int
    n_in: int
    n_out: int


def topo_sort_deps(dep0, dep1) -> np.ndarray:
    """DFS post-order for explicit deps (-1 = none) — C restatement of topological_sort.rs."""
    d0, d1 = _c(dep0, np.int64), _c(dep1, np.int64)
    n = len(d0)
    out = np.empty(max(n, 1), dtype=np.uint32)
    cyc = ctypes.c_uint64(0)
    rc = lib().orc_topo_sort_deps(n, _p(d0, ctypes.c_int64), _p(d1, ctypes.c_int64), _p(out, ctypes.c_uint32),
                                  ctypes.byref(cyc))
    if rc == ORC_CYCLIC:
        raise CyclicDependency(f"detected at i={cyc.value}")
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    return out[:n]


def build_circuit(lh, rh, out, op, n_nodes, input_nodes, output_nodes, mode: int = 1,
                  keep_handle: bool = False):
    """C restatement of compiler.rs:385-493 on the flat payload. mode 0 = faithful (hash maps), 1 = flat."""
    lh, rh, out, op = _c(lh, np.uint32), _c(rh, np.uint32), _c(out, np.uint32), _c(op, np.uint8)
    inn, outn = _c(input_nodes, np.uint32), _c(output_nodes, np.uint32)
    n = len(lh)
    res = ctypes.POINTER(_OrcCircuit)()
    cyc = ctypes.c_uint64(0)
    rc = lib().orc_build_circuit(n, _p(lh, ctypes.c_uint32), _p(rh, ctypes.c_uint32), _p(out, ctypes.c_uint32),
                                 _p(op, ctypes.c_uint8), int(n_nodes), len(inn), _p(inn, ctypes.c_uint32),
                                 len(outn), _p(outn, ctypes.c_uint32), mode, ctypes.byref(res), ctypes.byref(cyc))
    if rc == ORC_CYCLIC:
        raise CyclicDependency(f"detected at i={cyc.value}")
    if rc == ORC_INCONSISTENCY:
        raise Inconsistency("node used for both input and output")
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    c = res.contents

    def arr(ptr, count, dt):
        if count == 0:
            return np.empty(0, dtype=dt)
        return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dt, copy=True)

    circ = ArithCircuit(sorted=arr(c.sorted, n, np.uint32), in0=arr(c.in0, n, np.uint32),
                        in1=arr(c.in1, n, np.uint32), out=arr(c.out, n, np.uint32), op=arr(c.op, n, np.uint8),
                        node_wire=arr(c.node_wire, int(n_nodes), np.uint32), wire_count=int(c.wire_count),
                        n_in=len(inn), n_out=len(outn))
    if keep_handle:
        return circ, res
    lib().orc_free_circuit(res)
    return circ


def free_circuit(handle):
    lib().orc_free_circuit(handle)


def boolify_handle(handle, width: int, copy: bool = True):
    res = ctypes.POINTER(_OrcBool)()
    rc = lib().orc_boolify(handle, width, ctypes.byref(res))
    if rc == ORC_OVERFLOW:
        raise OverflowError("boolean wire ids exceed u32")
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    b = res.contents
    g = int(b.n_gates)

    def arr(ptr, dt):
        if g == 0:
            return np.empty(0, dtype=dt)
        a = np.ctypeslib.as_array(ptr, shape=(g,))
        return a.astype(dt, copy=True) if copy else a

    bc = BoolCircuit(in0=arr(b.in0, np.uint32), in1=arr(b.in1, np.uint32), out=arr(b.out, np.uint32),
                     op=arr(b.op, np.uint8), wire_count=int(b.wire_count), width=width, n_in=int(b.n_in),
                     n_out=int(b.n_out))
    if copy:
        lib().orc_free_bool(res)
        return bc
    return bc, res


def _as_handle(circ: ArithCircuit):
    c = _OrcCircuit()
    keep = [_c(circ.sorted, np.uint32), _c(circ.in0, np.uint32), _c(circ.in1, np.uint32), _c(circ.out, np.uint32),
            _c(circ.op, np.uint8), _c(circ.node_wire, np.uint32)]
    c.n = len(keep[1])
    c.n_nodes = len(keep[5])
    c.n_in, c.n_out, c.wire_count = circ.n_in, circ.n_out, circ.wire_count
    c.sorted, c.in0, c.in1, c.out = (_p(keep[0], ctypes.c_uint32), _p(keep[1], ctypes.c_uint32),
                                     _p(keep[2], ctypes.c_uint32), _p(keep[3], ctypes.c_uint32))
    c.op, c.node_wire = _p(keep[4], ctypes.c_uint8), _p(keep[5], ctypes.c_uint32)
    return c, keep


def boolify_range(circ: ArithCircuit, width: int, first: int, count: int):
    """Boolean gates of the arithmetic gates at sorted positions [first, first+count) with the global numbering;
    returns (BoolCircuit slice, global index of its first boolean gate)."""
    c, keep = _as_handle(circ)
    res = ctypes.POINTER(_OrcBool)()
    g0 = ctypes.c_uint64(0)
    rc = lib().orc_boolify_range(ctypes.pointer(c), width, first, count, ctypes.byref(res), ctypes.byref(g0))
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    b = res.contents
    g = int(b.n_gates)

    def arr(ptr, dt):
        return np.ctypeslib.as_array(ptr, shape=(g,)).astype(dt, copy=True) if g else np.empty(0, dtype=dt)

    bc = BoolCircuit(in0=arr(b.in0, np.uint32), in1=arr(b.in1, np.uint32), out=arr(b.out, np.uint32),
                     op=arr(b.op, np.uint8), wire_count=int(b.wire_count), width=width, n_in=int(b.n_in),
                     n_out=int(b.n_out))
    lib().orc_free_bool(res)
    return bc, g0.value


def boolify(circ: ArithCircuit, width: int) -> BoolCircuit:
    """Frozen bit-blast spec applied to an arithmetic circuit in sorted order."""
    c = _OrcCircuit()
    keep = [_c(circ.sorted, np.uint32), _c(circ.in0, np.uint32), _c(circ.in1, np.uint32), _c(circ.out, np.uint32),
            _c(circ.op, np.uint8), _c(circ.node_wire, np.uint32)]
    c.n = len(keep[1])
    c.n_nodes = len(keep[5])
    c.n_in, c.n_out, c.wire_count = circ.n_in, circ.n_out, circ.wire_count
    c.sorted, c.in0, c.in1, c.out = (_p(keep[0], ctypes.c_uint32), _p(keep[1], ctypes.c_uint32),
                                     _p(keep[2], ctypes.c_uint32), _p(keep[3], ctypes.c_uint32))
    c.op, c.node_wire = _p(keep[4], ctypes.c_uint8), _p(keep[5], ctypes.c_uint32)
    return boolify_handle(ctypes.pointer(c), width)


def template_size(op: int, width: int) -> Tuple[int, int]:
    g, a = ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = lib().orc_template_size(op, width, ctypes.byref(g), ctypes.byref(a))
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    return g.value, a.value


def bool_wire(circ: ArithCircuit, aux_total: int, width: int, W, bit=0):
    """Boolean wire id of (arithmetic wire W, bit) — layout of DESIGN.md §5.1."""
    W = np.asarray(W, dtype=np.int64)
    M = circ.wire_count - circ.n_out
    return np.where(W < M, W * width + bit, M * width + aux_total + (W - M) * width + bit)


def eval_op(op: int, a: int, b: int, width: int) -> int:
    return int(lib().orc_eval_op(op, a, b, width))


def eval_arith(circ: ArithCircuit, width: int, wires: np.ndarray) -> np.ndarray:
    """wires: uint64 [wire_count, T] with inputs/constants filled; evaluated in place and returned."""
    assert wires.dtype == np.uint64 and wires.flags.c_contiguous and wires.shape[0] == circ.wire_count
    in0, in1, out, op = _c(circ.in0, np.uint32), _c(circ.in1, np.uint32), _c(circ.out, np.uint32), _c(circ.op, np.uint8)
    lib().orc_eval_arith(len(in0), _p(in0, ctypes.c_uint32), _p(in1, ctypes.c_uint32), _p(out, ctypes.c_uint32),
                         _p(op, ctypes.c_uint8), width, wires.shape[1], _p(wires, ctypes.c_uint64))
    return wires


def eval_bool(bc: BoolCircuit, wires: np.ndarray) -> np.ndarray:
    """wires: uint64 [bool wire_count], 64 test vectors per word, inputs/constants filled."""
    assert wires.dtype == np.uint64 and wires.shape[0] == bc.wire_count
    in0, in1, out, op = _c(bc.in0, np.uint32), _c(bc.in1, np.uint32), _c(bc.out, np.uint32), _c(bc.op, np.uint8)
    lib().orc_eval_bool(len(in0), _p(in0, ctypes.c_uint32), _p(in1, ctypes.c_uint32), _p(out, ctypes.c_uint32),
                        _p(op, ctypes.c_uint8), _p(wires, ctypes.c_uint64))
    return wires


def fnv1a(a: np.ndarray, seed: int = 0) -> int:
    a = np.ascontiguousarray(a)
    return int(lib().orc_fnv1a(a.ctypes.data, a.nbytes, seed))


# ---------------------------------------------------------------------------------------------
# artefact writers (src/main.rs:34-35 circuit.txt, :43-44 circuit_info.json) — the checker's own, written per gate, one
# `Gate{inputs, outputs, op}` at a time (compiler.rs:456-463), so that the product's GPU formatter (c2a_format_bristol) and
# its host writers are compared with something that is not product code.
# PARITY UNPINNED: the `bristol-circuit` crate (Cargo.toml:20) is not in the reference tree and no reference test reads
# the text; the layout is Bristol fashion as SURVEY.md Appendix C.2 recollects it — "{ngates} {nwires}", "{n_in} {widths}",
# "{n_out} {widths}", a blank line, then "{n_inputs} {n_outputs} {inputs...} {outputs...} {op}" per gate; width 1 per named
# input / output when io_widths is None (compiler.rs:492).
# ---------------------------------------------------------------------------------------------
def bristol_text(gates, wire_count: int, n_in: int, n_out: int, io_widths=None) -> str:
    """`gates`: iterable of (inputs list, outputs list, op name)."""
    gates = list(gates)
    iw, ow = io_widths if io_widths is not None else ([1] * n_in, [1] * n_out)
    out = [f"{len(gates)} {wire_count}", " ".join(str(v) for v in [len(iw)] + list(iw)),
           " ".join(str(v) for v in [len(ow)] + list(ow)), ""]
    for ins, outs, op in gates:
        out.append(" ".join([str(len(ins)), str(len(outs))] + [str(v) for v in ins] + [str(v) for v in outs] + [op]))
    return "\n".join(out) + "\n"


def bristol_text_of(circ, io_widths=None) -> str:
    """circuit.txt of a literal BristolCircuit (CompilerModel.build_circuit), an ArithCircuit or a BoolCircuit."""
    if isinstance(circ, BristolCircuit):
        gates = (([a, b], [o], op) for a, b, o, op in circ.gates)
        return bristol_text(gates, circ.wire_count, len(circ.input_name_to_wire_index), len(circ.output_name_to_wire_index),
                            circ.io_widths)
    if isinstance(circ, BoolCircuit):
        def gen():
            for a, b, o, p in zip(circ.in0.tolist(), circ.in1.tolist(), circ.out.tolist(), circ.op.tolist()):
                yield ([a] if p == 2 else [a, b]), [o], BOOL_OP_NAMES[p]          # INV has one input
        return bristol_text(gen(), circ.wire_count, circ.n_in, circ.n_out, ([circ.width] * circ.n_in, [circ.width] * circ.n_out))
    gates = (([a, b], [o], OP_NAMES[p]) for a, b, o, p in zip(circ.in0.tolist(), circ.in1.tolist(), circ.out.tolist(), circ.op.tolist()))
    return bristol_text(gates, circ.wire_count, circ.n_in, circ.n_out, io_widths)


def circuit_info_json(circ: BristolCircuit, bool_width: Optional[int] = None, bool_wire_of=None) -> str:
    """circuit_info.json (serde_json::to_string_pretty of CircuitInfo, src/main.rs:43-44; keys sorted: the reference's map
    order is undefined).  For the boolified circuit pass the wire map of DESIGN.md §5.1 (first bit of each wire)."""
    import json
    m = (lambda w: int(bool_wire_of(w))) if bool_wire_of is not None else (lambda w: int(w))
    return json.dumps({
        "input_name_to_wire_index": {k: m(v) for k, v in sorted(circ.input_name_to_wire_index.items())},
        "constants": {k: {"value": c.value, "wire_index": m(c.wire_index)} for k, c in sorted(circ.constants.items())},
        "output_name_to_wire_index": {k: m(v) for k, v in sorted(circ.output_name_to_wire_index.items())},
    }, indent=2)


# ---------------------------------------------------------------------------------------------
# optional prune pass over a bit-blasted circuit (constant folding + dead-gate removal): the checker's sequential twin of
# c2a_boolify_prune.  The boolean circuit is in topological order, so one forward sweep over the gates folds, one backward
# sweep marks what the outputs depend on.  (What the real `boolify` crate does here is unknown — SURVEY C.2 — so this, like
# the bit-blast itself, is a frozen rule set written twice, not a pinned parity.)
# ---------------------------------------------------------------------------------------------
def prune_bool(bc: BoolCircuit, out_base: int):
    """bc: BoolCircuit with n_out / width set; out_base: first boolean wire of the circuit outputs.
    Returns (in0, in1, out, op) of the pruned circuit (gates 0, 1 make the constant wires) and a dict of counts."""
    G = len(bc.op)
    zero_wire, one_wire = bc.wire_count, bc.wire_count + 1
    out_end = out_base + bc.n_out * bc.width
    rep = {}                                   # wire -> 0 / 1 / other wire + 2 (absent: itself)

    def R(w):
        return rep.get(w, w + 2)
    kept = []                                  # (in0, in1, out, op) or None
    folded = 0
    in0, in1, out, op = bc.in0.tolist(), bc.in1.tolist(), bc.out.tolist(), bc.op.tolist()
    for k in range(G):
        o, a = op[k], R(in0[k])
        b = a if o == 2 else R(in1[k])
        r, nop, na, nb = None, o, a, b
        if o == 0:
            if a == b: r = 0
            elif a == 0: r = b
            elif b == 0: r = a
            elif a == 1: nop, na, nb = 2, b, b
            elif b == 1: nop, nb = 2, a
        elif o == 1:
            if a == b: r = a
            elif a == 0 or b == 0: r = 0
            elif a == 1: r = b
            elif b == 1: r = a
        else:
            if a <= 1: r = 1 - a
        is_output = out_base <= out[k] < out_end
        if r is not None and not is_output:
            rep[out[k]] = r
            kept.append(None)
            folded += 1
        else:
            if r is not None:
                nop, na, nb = o, a, b
            wa = zero_wire if na == 0 else (one_wire if na == 1 else na - 2)
            wb = zero_wire if nb == 0 else (one_wire if nb == 1 else nb - 2)
            kept.append((wa, wa if nop == 2 else wb, out[k], nop))
            rep[out[k]] = out[k] + 2
    need = set()
    dead = 0
    live = [False] * G
    for k in range(G - 1, -1, -1):
        g = kept[k]
        if g is None:
            continue
        if out_base <= g[2] < out_end or g[2] in need:
            live[k] = True
            need.add(g[0])
            if g[3] != 2:
                need.add(g[1])
        else:
            dead += 1
    res = [(0, 0, zero_wire, 0), (zero_wire, zero_wire, one_wire, 2)] + [kept[k] for k in range(G) if live[k]]
    arr = np.array(res, dtype=np.int64).reshape(-1, 4)
    return ((arr[:, 0].astype(np.uint32), arr[:, 1].astype(np.uint32), arr[:, 2].astype(np.uint32), arr[:, 3].astype(np.uint8)),
            {"n_gates": len(res), "n_gates_before": G, "n_folded": folded, "n_dead": dead, "wire_count": bc.wire_count + 2,
             "zero_wire": zero_wire, "one_wire": one_wire})
