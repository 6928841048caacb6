"""circom-2-arithc_amd — MI355X-native back end for the flat-gate-graph stage of circom-2-arithc.

The directory name carries the reference's hyphens, so import it with
``importlib.import_module("circom-2-arithc_amd")``.

Layers:
  * ``csrc/``        hand-written HIP for gfx950 + the C ABI of ``include/c2a.h`` -> ``libc2a_hip.so``
  * ``backend.py``   thin ctypes binding of that ABI (numpy in / numpy out, nothing else)
  * ``compiler.py``  host mirror of the reference's ``Compiler`` (src/compiler.rs): same method names,
                     same errors; ``build_circuit`` keeps the string work on the host and sends the flat
                     gate SoA through the C ABI
  * ``bristol.py``   BristolCircuit / CircuitInfo containers and the three artefact writers of
                     src/main.rs:34-47
  * ``synth.py``     seeded synthetic flat gate lists (BASELINE.json configs)

There is NO CPU fallback: if ``libc2a_hip.so`` is missing or no GPU is visible, the calls raise.
"""
from .backend import (Backend, BackendError, BoolInfo, CircuitError, CyclicDependency, Inconsistency,  # noqa: F401
                      OP, OP_NAMES, BOOL_OP_NAMES, library_path, load_library, visible_devices)
from . import synth  # noqa: F401

__all__ = ["Backend", "BackendError", "BoolInfo", "CircuitError", "CyclicDependency", "Inconsistency", "OP",
           "OP_NAMES", "BOOL_OP_NAMES", "library_path", "load_library", "visible_devices", "synth"]
