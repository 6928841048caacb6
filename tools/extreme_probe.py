#!/usr/bin/env python3
"""usage (on the GPU box): python tools/extreme_probe.py [case ...] — shapes at the edges of what the generators make (one layer of 10 M gates,
two layers, a butterfly, lh == rh everywhere, every gate a circuit output, a pure chain), each through c2a_build_circuit against the oracle,
with the stage times: a cliff shows as a stage far above its share in the headline's build."""
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

S = c2a.synth


butterfly, matmul = S.butterfly, S.matmul


def same_operand(fg):
    return S.FlatGates(lh=fg.lh, rh=fg.lh.copy(), out=fg.out, op=fg.op, n_nodes=fg.n_nodes, input_nodes=fg.input_nodes, output_nodes=fg.output_nodes,
                       const_nodes=fg.const_nodes, layers=fg.layers, layer_width=fg.layer_width)


CASES = {
    "one_layer_10m": lambda: S.layered_dag(1, 10_000_000),
    "one_layer_10m_4_inputs": lambda: S.layered_dag(1, 10_000_000, n_in=4, n_const=1),
    "two_layers_5m": lambda: S.layered_dag(2, 5_000_000, n_in=64, n_const=2, window=1),
    "ten_layers_1m": lambda: S.layered_dag(10, 1_000_000, n_in=64, n_const=2, window=1),
    "butterfly_20x2^19": lambda: butterfly(19, 20),
    "butterfly_2000x2^11": lambda: butterfly(11, 2000),
    "matmul_170": lambda: matmul(170),
    "lh_eq_rh_10m": lambda: same_operand(S.layered_dag(5000, 2000)),
    "all_outputs_10m": lambda: S.layered_dag(5000, 2000, out_frac=1.0),
    "all_fresh_constants_10m": lambda: S.layered_dag(5000, 2000, const_frac=1.0),
    "narrow_20x100k": lambda: S.layered_dag(100_000, 20, n_in=8, n_const=2, window=4),
    "chain_300k": lambda: S.layered_dag(300_000, 1, n_in=2, n_const=1, window=1),
}


def main():
    names = sys.argv[1:] or list(CASES)
    be = c2a.Backend(0)
    fails = 0
    for name in names:
        fg = CASES[name]()
        args = (fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
        t0 = time.time()
        exp = orc.build_circuit(*args, mode=1)
        t_cpu = (time.time() - t0) * 1e3
        try:
            be.load_gates(*args)
            best = None
            for rep in range(3):
                assert be.build_circuit() == exp.wire_count, "wire_count"
                for nm, arr in (("sorted", exp.sorted), ("in0", exp.in0), ("in1", exp.in1), ("out", exp.out), ("op", exp.op)):
                    assert be.checksum(nm) == bm.checksum_host(arr), nm
                nw1 = ((exp.node_wire.astype(np.uint64) + 1) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
                assert be.checksum("node_wire1") == bm.checksum_host(nw1), "node_wire"
                t = be.timings()
                if best is None or t["build_total"] < best["build_total"]:
                    best = t
            w = 32 if name != "matmul_170" else 4          # (4.9 M multipliers at width 32 would be 14 G boolean gates)
            t0 = time.perf_counter()
            info = be.boolify(w)
            wall_first = (time.perf_counter() - t0) * 1e3
            tb_first = be.timings()
            info = be.boolify(w)
            tb = be.timings()
            st = be.stats()
            print(f"{name:26s} n {fg.n:9d} == oracle x3 | build {best['build_total']:9.3f} ms: prep {best['prep']:.3f} peel {best['peel']:.3f} (k_peel {best['k_peel']:.3f}) "
                  f"order {best['order']:.3f} wires {best['wires']:.3f} emit {best['emit']:.3f} | boolify {tb['boolify_total']:.3f} ms = prep {tb['bool_prep']:.3f} + map {tb['bool_map']:.3f} (first call on this graph: {tb_first['boolify_total']:.1f} ms of GPU time, {wall_first:.1f} ms on the host's clock; {info.n_gates} gates) | levels {st['levels']} "
                  f"depth {st['max_depth']} chunks {st['path_chunks']} roots {st['n_roots']} rereads {st['peel_rereads']} relays {st['n_relays']} path {st['numbering_path']} | "
                  f"cpu oracle {t_cpu:.0f} ms", flush=True)
        except Exception as e:  # noqa: BLE001
            fails += 1
            print(f"{name:26s} n {fg.n:9d} FAIL: {type(e).__name__}: {str(e)[:300]}", flush=True)
    be.close()
    print(f"failures: {fails}")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
