#!/usr/bin/env python3
"""usage (on the GPU box): python tools/bool_alloc.py — does what k_boolify takes depend on WHERE its 9.6 GB of output landed?
Several contexts in one process (each allocates its own buffers), some of them with other allocations made and freed in
between; k_boolify's HIP-event time per context."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
c2a = importlib.import_module("circom-2-arithc_amd")
fg = c2a.synth.layered_dag(5000, 2000, seed=c2a.synth.SEED)
VARIANTS = os.environ.get("BOOL_ALLOC_ENVS", "C2A_BOOL_CHUNK=256").split(";")      # e.g. "C2A_BOOL_CHUNK=256;C2A_BOOL_CHUNK=128"
def one(tag):
  for copies in VARIANTS:
    for kv in copies.split(","):
        k, v = kv.split("="); os.environ[k] = v
    with c2a.Backend(0) as be:
        be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
        ts = []
        for _ in range(4):
            be.build_circuit(); be.boolify(32); ts.append(be.timings()["bool_map"])
        print(f"{tag} [{copies}]: k_boolify ms " + " ".join(f"{t:.3f}" for t in ts), flush=True)
one("context 1 (fresh process)")
one("context 2 (after context 1 was destroyed)")
hog = [torch.empty(3 << 30, dtype=torch.uint8, device="cuda") for _ in range(8)]        # 24 GB held while the context allocates
one("context 3 (24 GB of other allocations alive)")
del hog[::2]; torch.cuda.empty_cache()
one("context 4 (half of them freed: holes)")
del hog; torch.cuda.empty_cache()
one("context 5 (all freed)")
be = c2a.Backend(0)
be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
be.build_circuit(); be.boolify(64); be.build_circuit(); be.boolify(32)
ts = []
for _ in range(4):
    be.build_circuit(); be.boolify(32); ts.append(be.timings()["bool_map"])
print("context 6, width 32 in buffers sized by a width-64 call: k_boolify ms " + " ".join(f"{t:.3f}" for t in ts))
