#!/usr/bin/env python3
"""usage: python tools/family_check.py [--n GATES] [--reps R] [--emul] [--walk] [--sha-tiles] FAMILY ...

The graph families beyond layered_dag (synth.family: hub, hub_mild, window_all, forest, const_hub, strict; sha_chain / sha_tree = tilings of the
REAL SHA-256 block's flat list) through c2a_build_circuit, every result array against the oracle (checksums of sorted ids, the
emitted circuit and node -> wire), and what the build's stages took: one line per family.  --emul: the host-emulation build of
the library (CPU boxes, small sizes); --walk: the general wire numbering as well."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
c2a = importlib.import_module("circom-2-arithc_amd")
bm = importlib.import_module("circom-2-arithc_amd.backend")
from oracle import oracle as orc  # noqa: E402  (the checker)


def sha_block():
    """the flat gate list of tests/golden/circuits/sha256Block.circom, through the front-end (one parser run, 3 448 gates)"""
    comp = importlib.import_module("circom-2-arithc_amd.compiler")
    text = open(os.path.join(ROOT, "tests", "golden", "circuits", "sha256Block.circom")).read()
    C = comp.Compiler.from_circom(text, backend=None)
    inputs, outputs, _ = C._io_maps()
    lh, rh, out, op = C._flat()
    return lh, rh, out, op, C.node_count + 1, [nd for _, nd in inputs], [nd for _, nd in outputs]


def make(name, n, seed):
    if name.startswith("sha_"):
        blk = sha_block()
        copies = max(1, n // len(blk[0]))
        return c2a.synth.tile_block(*blk, copies=copies, shape="chain" if name == "sha_chain" else "tree", seed=seed,
                                    permute=name.endswith("_perm"))
    return c2a.synth.family(name, n, seed=seed)


def fanout(fg):
    cnt = np.bincount(np.concatenate([fg.lh, fg.rh[fg.rh != fg.lh]]), minlength=fg.n_nodes)
    fo = cnt[fg.out]
    return int(fo.max()), int((fo > 16).sum()), int(fo[fo > 16].sum())


def check(be, fg, exp, tag):
    assert be.build_circuit() == exp.wire_count, (tag, "wire_count")
    for nm, arr in (("sorted", exp.sorted), ("in0", exp.in0), ("in1", exp.in1), ("out", exp.out), ("op", exp.op)):
        assert be.checksum(nm) == bm.checksum_host(arr), (tag, nm)
    nw1 = ((exp.node_wire.astype(np.uint64) + 1) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    assert be.checksum("node_wire1") == bm.checksum_host(nw1), (tag, "node_wire")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("families", nargs="+")
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--seed", type=int, default=c2a.synth.SEED)
    ap.add_argument("--emul", action="store_true")
    ap.add_argument("--walk", action="store_true")
    a = ap.parse_args()
    lib = os.path.join(ROOT, "tests", "emul", "libc2a_emul.so") if a.emul else None
    fails = 0
    for name in a.families:
        fg = make(name, a.n, a.seed)
        args = (fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
        t0 = time.time()
        exp = orc.build_circuit(*args, mode=1)
        t_cpu = time.time() - t0
        mx, hubs, hub_edges = fanout(fg)
        paths = [("positional", {})] + ([("walk", {"C2A_NUMBERING_WALK": "1"})] if a.walk else [])
        for pname, env in paths:
            os.environ.update(env)
            be = c2a.Backend(0, lib_path=lib)
            for k in env:
                os.environ.pop(k)
            try:
                be.load_gates(*args)
                best = None
                for rep in range(a.reps):
                    t0 = time.time()
                    check(be, fg, exp, (name, pname, rep))
                    wall = (time.time() - t0) * 1e3
                    t = be.timings()
                    if best is None or t["build_total"] < best["build_total"]:
                        best = dict(t, wall=wall)
                st = be.stats()
                print(f"{name:12s} n {fg.n:9d} max fan-out {mx:8d} hubs(>16) {hubs:7d} hub edges {hub_edges:9d} | {pname:10s} == oracle x{a.reps} | "
                      f"build {best['build_total']:8.3f} ms: prep {best['prep']:.3f} peel {best['peel']:.3f} (k_peel {best['k_peel']:.3f}) order {best['order']:.3f} "
                      f"wires {best['wires']:.3f} emit {best['emit']:.3f} | levels {st['levels']} depth {st['max_depth']} chunks {st['path_chunks']} roots {st['n_roots']} "
                      f"rereads {st['peel_rereads']} path {st['numbering_path']} | cpu oracle {t_cpu * 1e3:.0f} ms", flush=True)
            except Exception as e:  # noqa: BLE001  (report and go on with the next family)
                fails += 1
                print(f"{name:12s} n {fg.n:9d} max fan-out {mx:8d} | {pname:10s} FAIL: {type(e).__name__}: {str(e)[:300]}", flush=True)
            finally:
                be.close()
    print(f"failures: {fails}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
