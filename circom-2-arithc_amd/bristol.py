"""BristolCircuit / CircuitInfo containers and the artefact writers of src/main.rs:34-47.

The `bristol-circuit` crate (github.com/voltrevo/bristol-circuit rev 2a8b001, Cargo.toml:20) is absent from
the reference tree.  Struct and field names below are the ones visible at the reference's call sites
(src/compiler.rs:456-463, :471-474, :478-493; src/main.rs:35,44); the TEXT format of `write_bristol` is not
pinned by any reference test — what is written here is Bristol-fashion as described in SURVEY.md Appendix C.2
(unverified recollection) and is flagged as such in DESIGN.md.

Gates are kept as SoA numpy arrays (what the GPU hands back), not as a list of Gate structs.
JSON maps are written with sorted keys (the reference iterates std HashMaps: order undefined).
"""
from __future__ import annotations

import io
import json
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class ConstantInfo:                 # compiler.rs:471-474
    value: str
    wire_index: int

    def to_json(self):
        return {"value": self.value, "wire_index": self.wire_index}


@dataclass
class CircuitInfo:                  # compiler.rs:480-490
    input_name_to_wire_index: Dict[str, int] = field(default_factory=dict)
    constants: Dict[str, ConstantInfo] = field(default_factory=dict)
    output_name_to_wire_index: Dict[str, int] = field(default_factory=dict)

    def to_json(self) -> dict:
        return {
            "input_name_to_wire_index": dict(sorted(self.input_name_to_wire_index.items())),
            "constants": {k: v.to_json() for k, v in sorted(self.constants.items())},
            "output_name_to_wire_index": dict(sorted(self.output_name_to_wire_index.items())),
        }


@dataclass
class BristolCircuit:               # compiler.rs:478-493
    wire_count: int
    info: CircuitInfo
    in0: np.ndarray
    in1: np.ndarray
    out: np.ndarray
    op: np.ndarray                  # u8 discriminants; names in op_names
    op_names: Sequence[str]
    io_widths: Optional[Tuple[List[int], List[int]]] = None     # None for arithmetic circuits (:492)
    unary_ops: Sequence[int] = ()    # ops printed with one input (INV)
    sorted_gate_ids: Optional[np.ndarray] = None
    gates_on_device: Optional[int] = None   # set when the SoA was left in HBM (fetch=False): the gate count

    @property
    def n_gates(self) -> int:
        return int(self.op.shape[0])

    @property
    def n_gates_total(self) -> int:
        return int(self.gates_on_device) if self.gates_on_device is not None else self.n_gates

    def gates(self) -> Iterator[Tuple[List[int], List[int], str]]:
        """Gate{inputs, outputs, op} triples (compiler.rs:456-463)."""
        un = set(int(u) for u in self.unary_ops)
        for a, b, o, p in zip(self.in0.tolist(), self.in1.tolist(), self.out.tolist(), self.op.tolist()):
            yield ([a] if p in un else [a, b]), [o], self.op_names[p]

    # -- writers -------------------------------------------------------------------------------
    def write_bristol(self, w) -> None:
        """circuit.txt (src/main.rs:34-35).  Header: '{ngates} {nwires}', '{n_in} {widths...}',
        '{n_out} {widths...}', blank line; then '{nin} {nout} {ins...} {outs...} {op}' per gate."""
        n_in = len(self.info.input_name_to_wire_index)
        n_out = len(self.info.output_name_to_wire_index)
        iw, ow = self.io_widths if self.io_widths is not None else ([1] * n_in, [1] * n_out)
        text = isinstance(w, io.TextIOBase)
        def emit(s: str):
            w.write(s if text else s.encode())
        emit(f"{self.n_gates} {self.wire_count}\n")
        emit(" ".join([str(len(iw))] + [str(x) for x in iw]) + "\n")
        emit(" ".join([str(len(ow))] + [str(x) for x in ow]) + "\n\n")
        un = np.isin(self.op, np.asarray(list(self.unary_ops), dtype=self.op.dtype)) if len(self.unary_ops) else None
        names = np.asarray(self.op_names, dtype=object)
        step = 1 << 18
        for s in range(0, self.n_gates, step):
            e = min(self.n_gates, s + step)
            a, b, o = self.in0[s:e].tolist(), self.in1[s:e].tolist(), self.out[s:e].tolist()
            nm = names[self.op[s:e]].tolist()
            if un is None:
                lines = [f"2 1 {x} {y} {z} {k}" for x, y, z, k in zip(a, b, o, nm)]
            else:
                u = un[s:e].tolist()
                lines = [(f"1 1 {x} {z} {k}" if uu else f"2 1 {x} {y} {z} {k}") for x, y, z, k, uu in zip(a, b, o, nm, u)]
            emit("\n".join(lines) + "\n")

    def header(self) -> str:
        """the three header lines + the blank line of circuit.txt"""
        n_in = len(self.info.input_name_to_wire_index)
        n_out = len(self.info.output_name_to_wire_index)
        iw, ow = self.io_widths if self.io_widths is not None else ([1] * n_in, [1] * n_out)
        return (f"{self.n_gates_total} {self.wire_count}\n" + " ".join([str(len(iw))] + [str(x) for x in iw]) + "\n" +
                " ".join([str(len(ow))] + [str(x) for x in ow]) + "\n\n")

    def write_bristol_gpu(self, w, backend, chunk_gates: int = 1 << 24) -> int:
        """circuit.txt with the gate lines printed on the GPU and streamed chunk by chunk (the boolean circuit of the
        10 M-gate config is ~27 GB of text: Python string formatting would take hours).  `backend` must still hold the
        circuit this object describes.  Returns the bytes written."""
        which = 1 if self.io_widths is not None else 0
        total = self.n_gates_total
        head = self.header().encode()
        w.write(head)
        nbytes = len(head)
        for first in range(0, total, chunk_gates):
            part = backend.format_bristol(which, first, min(chunk_gates, total - first))
            w.write(part)
            nbytes += len(part)
        return nbytes

    def info_json(self) -> str:
        """circuit_info.json (src/main.rs:43-44), serde_json::to_string_pretty layout with sorted keys."""
        return json.dumps(self.info.to_json(), indent=2)


def read_bristol(text: str):
    """Parse the text written by write_bristol back into (n_gates, wire_count, iw, ow, gate tuples)."""
    lines = text.split("\n")
    ng, nw = (int(x) for x in lines[0].split())
    iw = [int(x) for x in lines[1].split()][1:]
    ow = [int(x) for x in lines[2].split()][1:]
    gates = []
    for ln in lines[4:]:
        if not ln:
            continue
        t = ln.split()
        nin, nout = int(t[0]), int(t[1])
        gates.append(([int(x) for x in t[2:2 + nin]], [int(x) for x in t[2 + nin:2 + nin + nout]], t[-1]))
    assert len(gates) == ng
    return ng, nw, iw, ow, gates
