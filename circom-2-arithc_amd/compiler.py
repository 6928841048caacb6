"""Host mirror of the reference's `Compiler` (src/compiler.rs:107-115) for the flat-gate-graph path.

Same method names, argument meaning and errors as the reference, so code written against
`circom_2_arithc::compiler::Compiler` reads the same here:

    add_signal / add_gate / add_connection / add_inputs / add_outputs / get_signals   (src/compiler.rs:130-278)
    build_circuit() -> BristolCircuit                                                  (src/compiler.rs:321-494)
    boolify(circuit, width) -> BristolCircuit                                          (src/main.rs:30-32)

What differs is where the work happens.  The gate-graph builder keeps a signal->node index and a
node-forwarding table instead of the reference's O(nodes)+O(gates) scans per call
(src/compiler.rs:185-195, :219-226, :260-270); `build_circuit` does the string work on the host
(src/compiler.rs:323-383) and ships the flat gate SoA through the C ABI (include/c2a.h) where the
topological sort, wire numbering, gate emission and bit-blast run as HIP kernels on the MI355X.
There is no CPU implementation of those steps in this package.

Canonical ordering: the reference iterates std HashMaps for the input / output wire order
(src/compiler.rs:392-395, :446-449) — not deterministic run to run.  Here inputs and outputs are ordered
by ascending signal id of the named IO signal.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from .backend import (BOOL_OP_NAMES, NO_WIRE, OP, OP_NAMES, Backend, CircuitError, CyclicDependency,  # noqa: F401
                      Inconsistency)
from .bristol import BristolCircuit, CircuitInfo, ConstantInfo


class CannotMergeOutputNodes(CircuitError):        # compiler.rs:554-555
    def __init__(self):
        super().__init__("Cannot merge output nodes")


class CannotMergeConstantNodes(CircuitError):      # compiler.rs:552-553
    def __init__(self):
        super().__init__("Cannot merge constant nodes")


class SignalAlreadyDeclared(CircuitError):         # compiler.rs:564-565
    def __init__(self):
        super().__init__("Signal already declared")


@dataclass
class _Node:
    is_const: bool
    is_out: bool
    signals: List[int]


class Compiler:
    def __init__(self, backend: Optional[Backend] = None, device: int = 0):
        self._backend = backend
        self._device = device
        self.node_count = 0
        self.inputs: Dict[int, str] = {}
        self.outputs: Dict[int, str] = {}
        self.signals: Dict[int, Tuple[str, Optional[int]]] = {}
        self.nodes: Dict[int, _Node] = {}
        self._sig_node: Dict[int, int] = {}
        self._fwd: Dict[int, int] = {}              # merged node id -> the node that replaced it
        self._g_op: List[int] = []
        self._g_lh: List[int] = []
        self._g_rh: List[int] = []
        self._g_out: List[int] = []

    # -- reference API ---------------------------------------------------------------------------
    def add_inputs(self, inputs: Dict[int, str]) -> None:           # compiler.rs:130-132
        self.inputs.update(inputs)

    def add_outputs(self, outputs: Dict[int, str]) -> None:         # compiler.rs:134-136
        self.outputs.update(outputs)

    def _get_node_id(self) -> int:                                  # compiler.rs:497-500
        self.node_count += 1
        return self.node_count

    def add_signal(self, id: int, name: str, value: Optional[int] = None) -> None:   # compiler.rs:139-161
        if id in self.signals:
            raise SignalAlreadyDeclared()
        self.signals[id] = (name, value)
        nid = self._get_node_id()
        self.nodes[nid] = _Node(value is not None, False, [id])
        self._sig_node[id] = nid

    def get_signals(self, filter: str) -> Dict[int, str]:           # compiler.rs:163-171
        return {sid: nm for sid, (nm, _) in self.signals.items() if nm.startswith(filter)}

    def add_gate(self, gate_type, lhs_signal_id: int, rhs_signal_id: int, output_signal_id: int) -> None:
        """compiler.rs:174-209.  gate_type: AGateType name or discriminant.  An unknown signal maps to node 0
        like the reference's zero-initialised scan result."""
        op = OP[gate_type] if isinstance(gate_type, str) else int(gate_type)
        n0 = self._sig_node.get(lhs_signal_id, 0)
        n1 = self._sig_node.get(rhs_signal_id, 0)
        n2 = self._sig_node.get(output_signal_id, 0)
        self.nodes[n2].is_out = True                                 # KeyError == the reference's unwrap() panic
        self._g_op.append(op)
        self._g_lh.append(n0)
        self._g_rh.append(n1)
        self._g_out.append(n2)

    def add_connection(self, a: int, b: int) -> None:               # compiler.rs:213-278
        na, nb = self._sig_node.get(a, 0), self._sig_node.get(b, 0)
        if na == nb:
            return
        # an unknown signal id leaves the reference's scan on its placeholder `(0, &Node::new())` (:215-228): the merge then
        # goes through with an empty node 0 (node ids start at 1, so 0 is never a real node)
        node_a = self.nodes[na] if na else _Node(False, False, [])
        node_b = self.nodes[nb] if nb else _Node(False, False, [])
        if node_a.is_out and node_b.is_out:
            raise CannotMergeOutputNodes()
        if node_a.is_const and node_b.is_const:
            raise CannotMergeConstantNodes()
        merged = _Node(node_a.is_const or node_b.is_const, node_a.is_out or node_b.is_out,
                       node_a.signals + node_b.signals)
        mid = self._get_node_id()
        # the reference rewrites every gate here (:260-270); we forward lazily and resolve in _flat()
        for old in (na, nb):
            if old:
                self._fwd[old] = mid
                del self.nodes[old]
            else:
                # gates that reference the placeholder id 0 are rewritten NOW, like the reference does (:260-270): a gate
                # added later with an unknown signal must keep its 0
                for arr in (self._g_lh, self._g_rh, self._g_out):
                    for k, v in enumerate(arr):
                        if v == 0:
                            arr[k] = mid
        self.nodes[mid] = merged
        for sid in merged.signals:
            self._sig_node[sid] = mid

    # -- flat payload ----------------------------------------------------------------------------
    def _resolve(self, nid: int) -> int:
        root = nid
        while root in self._fwd:
            root = self._fwd[root]
        while nid in self._fwd:                                      # path compression
            nxt = self._fwd[nid]
            self._fwd[nid] = root
            nid = nxt
        return root

    @property
    def gates(self) -> List[Tuple[str, int, int, int]]:
        """Vec<ArithmeticGate> as (op name, lh_in, rh_in, out) with merges applied (compiler.rs:113)."""
        return [(OP_NAMES[o], self._resolve(a), self._resolve(b), self._resolve(c))
                for o, a, b, c in zip(self._g_op, self._g_lh, self._g_rh, self._g_out)]

    def _flat(self):
        n = len(self._g_op)
        lh = np.fromiter((self._resolve(x) for x in self._g_lh), dtype=np.uint32, count=n)
        rh = np.fromiter((self._resolve(x) for x in self._g_rh), dtype=np.uint32, count=n)
        out = np.fromiter((self._resolve(x) for x in self._g_out), dtype=np.uint32, count=n)
        op = np.asarray(self._g_op, dtype=np.uint8)
        return lh, rh, out, op

    def _io_maps(self):
        """compiler.rs:323-383 in canonical order: ([(input name, node)], [(output name, node)],
        {constant key: (node, value string)})."""
        inputs: List[Tuple[str, int]] = []
        outputs: List[Tuple[str, int]] = []
        constants: Dict[str, Tuple[int, str]] = {}
        seen_in, seen_out = set(), set()
        for sid in sorted(self._sig_node):
            nid = self._sig_node[sid]
            if sid in self.inputs:
                name = self.inputs[sid]
                if name in seen_in:
                    raise Inconsistency(f"Duplicate input {name}")              # :335-339
                seen_in.add(name)
                inputs.append((name, nid))
            if sid in self.outputs:
                name = self.outputs[sid]
                if name in seen_out:
                    raise Inconsistency(f"Duplicate output {name}")             # :345-349
                seen_out.add(name)
                outputs.append((name, nid))
            name, value = self.signals[sid]
            if value is not None:
                constants[f"{name}_{sid}"] = (nid, str(value))                    # :354-359
        node_to_input = {nid: name for name, nid in inputs}
        for name, nid in outputs:                                                 # :363-383
            if nid in node_to_input:
                raise Inconsistency(f"Node {nid} used for both input {node_to_input[nid]} and output {name}")
        return inputs, outputs, constants

    def backend(self) -> Backend:
        if self._backend is None:
            self._backend = Backend(self._device)
        return self._backend

    def build_circuit(self) -> BristolCircuit:
        """compiler.rs:321-494.  Raises CyclicDependency / Inconsistency like the reference returns them."""
        inputs, outputs, constants = self._io_maps()
        lh, rh, out, op = self._flat()
        be = self.backend()
        be.load_gates(lh, rh, out, op, self.node_count + 1, [nid for _, nid in inputs], [nid for _, nid in outputs])
        sorted_ids = be.topo_sort()                                               # :408-421
        node_wire, wire_count = be.assign_wires()                                 # :388-449
        in0, in1, o, g_op = be.emit_gates()                                       # :451-464
        consts: Dict[str, ConstantInfo] = {}
        for key, (nid, value) in constants.items():                               # :466-476
            w = int(node_wire[nid])
            if w == NO_WIRE:
                raise KeyError(f"constant node {nid} has no wire")                # the reference panics (HashMap index)
            consts[key] = ConstantInfo(value, w)
        info = CircuitInfo(
            input_name_to_wire_index={nm: int(node_wire[nid]) for nm, nid in inputs},
            constants=consts,
            output_name_to_wire_index={nm: int(node_wire[nid]) for nm, nid in outputs})
        self._last = dict(n_in=len(inputs), n_out=len(outputs))
        return BristolCircuit(wire_count=wire_count, info=info, in0=in0, in1=in1, out=o, op=g_op, op_names=OP_NAMES,
                              io_widths=None, sorted_gate_ids=sorted_ids)

    def boolify(self, circuit: BristolCircuit, width: int, fetch: bool = True) -> BristolCircuit:
        """boolify(&circuit, width) (src/main.rs:30-32) of the circuit just built by build_circuit(); frozen
        bit-blast spec of DESIGN.md §5.  With fetch=False the boolean SoA stays in HBM (arrays are empty)."""
        be = self.backend()
        bi = be.boolify(width)
        if fetch:
            in0, in1, out, op = be.bool_read()
        else:
            in0 = in1 = out = np.empty(0, np.uint32)
            op = np.empty(0, np.uint8)
        ci = circuit.info
        info = CircuitInfo(
            input_name_to_wire_index={k: int(bi.wire(v)) for k, v in ci.input_name_to_wire_index.items()},
            constants={k: ConstantInfo(c.value, int(bi.wire(c.wire_index))) for k, c in ci.constants.items()},
            output_name_to_wire_index={k: int(bi.wire(v)) for k, v in ci.output_name_to_wire_index.items()})
        return BristolCircuit(wire_count=bi.wire_count, info=info, in0=in0, in1=in1, out=out, op=op,
                              op_names=BOOL_OP_NAMES, io_widths=([width] * bi.n_in, [width] * bi.n_out),
                              unary_ops=(2,), gates_on_device=None if fetch else int(bi.n_gates))

    @classmethod
    def from_circom(cls, text: str, backend: Optional[Backend] = None, device: int = 0) -> "Compiler":
        """program.rs::compile for the supported Circom subset: unroll the templates (circom_frontend.py restates
        src/process.rs call for call) into add_signal / add_gate / add_connection, then IO discovery by name prefix
        (program.rs:57-66).  Raises circom_frontend.ProgramError with the reference's Display strings."""
        from .circom_frontend import unroll
        d = unroll(text)
        c = cls(backend=backend, device=device)
        for st in d["script"]:
            if st[0] == "signal":
                c.add_signal(st[1], st[2], st[3])
            elif st[0] == "gate":
                c.add_gate(st[1], st[2], st[3], st[4])
            else:
                c.add_connection(st[1], st[2])
        for p in d["input_prefixes"]:
            c.add_inputs(c.get_signals(f"0.{p}"))
        for p in d["output_prefixes"]:
            c.add_outputs(c.get_signals(f"0.{p}"))
        return c

    # -- report.json (src/main.rs:22, :46-47) ------------------------------------------------------
    def generate_circuit_report(self, value_type: str = "sint") -> dict:
        """compiler.rs:287-319 + :502-531: nodes split into inputs (not the output of any gate) and outputs (gate outputs
        that no gate reads), each sorted by node id; per node the names of its signals (those containing "random_" are
        dropped, :519) and the value of its last valued signal.  The reference scans all gates per output node
        (:300-304); a set of the nodes some gate reads does the same in one pass."""
        read = set()
        for a, b in zip(self._g_lh, self._g_rh):
            read.add(self._resolve(a)); read.add(self._resolve(b))
        input_nodes = sorted(nid for nid, nd in self.nodes.items() if not nd.is_out)
        output_nodes = sorted(nid for nid, nd in self.nodes.items() if nd.is_out and nid not in read)

        def reports(ids):
            out = []
            for nid in ids:
                names, value = [], None
                for sid in self.nodes[nid].signals:
                    nm, v = self.signals[sid]
                    if "random_" not in nm:
                        names.append(nm)
                    if v is not None:
                        value = v
                out.append({"id": nid, "names": names, "value": value})
            return out
        return {"inputs": reports(input_nodes), "outputs": reports(output_nodes), "value_type": value_type}

    def report_json(self, value_type: str = "sint") -> str:
        """serde_json::to_string_pretty(&report) (src/main.rs:46-47)"""
        import json
        return json.dumps(self.generate_circuit_report(value_type), indent=2)
