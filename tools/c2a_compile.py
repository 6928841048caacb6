#!/usr/bin/env python3
"""Command-line driver mirroring src/main.rs:15-50 (cli.rs:16-53) on the MI355X back end:

    tools/c2a_compile.py -i input/circuit.circom -o output/ [--boolify-width W] [-v sint|sfloat] [--device D]

compile (Circom subset, circom_frontend.py) -> report -> build_circuit (HIP) -> boolify (HIP) -> circuit.txt (gate lines
printed on the GPU), circuit_info.json, report.json.  No CPU fallback: fails when libc2a_hip.so or a GPU is missing."""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None, lib_path=None):
    ap = argparse.ArgumentParser(description="Arithmetic Circuits Compiler (MI355X back end)")
    ap.add_argument("-i", "--input", default="./input/circuit.circom")             # cli.rs:23-28
    ap.add_argument("-o", "--output", default="./output/")                         # cli.rs:30-38
    ap.add_argument("-v", "--value-type", default="sint", choices=["sint", "sfloat"])
    ap.add_argument("--boolify-width", type=int, default=None)                     # cli.rs:47-52
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    pkg = importlib.import_module("circom-2-arithc_amd")
    comp_mod = importlib.import_module("circom-2-arithc_amd.compiler")
    fe = importlib.import_module("circom-2-arithc_amd.circom_frontend")
    with open(args.input) as f:
        text = f.read()
    be = pkg.Backend(args.device, lib_path=lib_path)
    try:
        compiler = comp_mod.Compiler.from_circom(text, backend=be)                 # main.rs:21
        report = compiler.report_json(args.value_type)                             # main.rs:22
        os.makedirs(args.output, exist_ok=True)                                    # main.rs:24-26
        circuit = compiler.build_circuit()                                         # main.rs:28
        if args.boolify_width is not None:
            circuit = compiler.boolify(circuit, args.boolify_width, fetch=False)   # main.rs:30-32
        with open(os.path.join(args.output, "circuit.txt"), "wb") as f:            # main.rs:34-35
            circuit.write_bristol_gpu(f, be)
        with open(os.path.join(args.output, "circuit_info.json"), "w") as f:       # main.rs:43-44
            f.write(circuit.info_json())
        with open(os.path.join(args.output, "report.json"), "w") as f:             # main.rs:46-47
            f.write(report)
    except (fe.ProgramError, pkg.CircuitError) as e:
        print(f"Error: {e}", file=sys.stderr)
        return 1
    finally:
        be.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
