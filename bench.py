#!/usr/bin/env python3
"""bench.py — gates/sec through (topo-sort + wire numbering + gate emission + boolify) on the synthetic
10 M-gate DAG of BASELINE.json (configs[4]), device-resident in / device-resident out.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--width 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N ...

A "step" = one pass of the hot path over one gate graph already resident in HBM: c2a_build_circuit (sort,
numbering, emission) + c2a_boolify(width).  N > 1: the sort is a chain of ~5 000 dependent level steps and does
not shard, so every rank processes its OWN 10 M-gate graph (independent circuits, different
seeds) with no data-path collective: weak scaling; value = all gates processed / max-over-ranks time.

One JSON line on stdout (rank 0).  `roofline` is for the dominant kernel (k_boolify): algorithmic bytes per
launch = 13 B read per arithmetic gate + 13 B written per boolean gate, divided by the kernel's
launch duration measured with HIP events on the library's stream.  `cpu_baseline` = the CPU oracle ("port":
the reference is Rust and cannot be built here) timed on a bounded sample of the same generator, 1 core (the
reference path is single-threaded).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def timed_region(warm_step, step, steps, warmup, dist=None, torch=None, device=None):
    """W untimed warm-up steps, then EXACTLY `steps` timed steps bracketed by barrier + device sync on both sides;
    returns the MAX over ranks of the elapsed seconds.  (Every c2a call ends with a hipStreamSynchronize, so the
    host clock brackets device work.)  dist/torch are None for a single process."""
    def sync_all():
        if dist is not None:
            dist.barrier()
            if device == "cuda":
                torch.cuda.synchronize()

    for _ in range(warmup):
        warm_step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def whole_job_rate(world, units_per_rank, steps, elapsed_max):
    """value = units processed by ALL ranks / max-over-ranks time (weak scaling: one graph per rank)."""
    return world * units_per_rank * steps / elapsed_max


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=32, help="--boolify-width")
    ap.add_argument("--layers", type=int, default=5000)
    ap.add_argument("--layer-width", type=int, default=2000)
    ap.add_argument("--cpu-sample-layers", type=int, default=1000, help="layers of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--check", action="store_true", help="verify the GPU result against the oracle (sample-sized run)")
    ap.add_argument("--mode", choices=["replicas", "shard"], default="replicas",
                    help="N>1: 'replicas' = one independent graph per rank (weak scaling, default); 'shard' = ONE graph, "
                         "sort replicated on every rank, boolify sharded by sorted-position range (strong scaling, DESIGN.md §7)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    dist = None
    torch = None
    # one rank per GPU under torch.distributed.run; a 1-rank launch takes the same path (RANK is set by the launcher)
    if world > 1 or ("RANK" in os.environ and os.environ.get("C2A_BENCH_PLAIN") is None and "TORCHELASTIC_RUN_ID" in os.environ):
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    c2a = importlib.import_module("circom-2-arithc_amd")
    synth = c2a.synth

    t0 = time.time()
    shard = args.mode == "shard" and world > 1
    fg = synth.layered_dag(args.layers, args.layer_width, seed=synth.SEED + (0 if shard else rank))
    gen_s = time.time() - t0
    be = c2a.Backend(local_rank)
    t0 = time.time()
    be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    h2d_s = time.time() - t0

    n_all = fg.n
    my_lo, my_hi = (rank * n_all) // world, ((rank + 1) * n_all) // world

    def step():
        be.build_circuit()
        if shard:      # every rank holds the whole sorted circuit, so it can place its own range without any exchange
            info = be.boolify_plan(args.width)
            be.boolify_chunk(my_lo, my_hi - my_lo, fetch=False)
            return info
        return be.boolify(args.width)

    stage_acc = {}
    last = {}

    def timed_step():
        last["info"] = step()
        for k, v in be.timings().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v

    elapsed = timed_region(step, timed_step, args.steps, args.warmup, dist, torch, "cuda" if dist is not None else None)
    info = last.get("info")

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    n = fg.n
    steps = max(1, args.steps)
    stages = {k: v / steps for k, v in stage_acc.items()}
    ms_per_step = elapsed * 1e3 / steps
    value = whole_job_rate(1 if shard else world, n, steps, elapsed)
    algo_bytes = 13.0 * n + 13.0 * info.n_gates                  # per k_boolify launch
    bool_ms = stages.get("bool_map", 0.0)
    achieved = algo_bytes / (bool_ms * 1e-3) / 1e9 if bool_ms > 0 else 0.0
    stats = be.stats()
    # HBM bytes of the dominant kernel from the PMC passes committed under profiles/ (bench.py cannot run
    # rocprofv3 on itself): only quoted when the workload is the one those passes measured
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_bytes.json")) as f:
            pmc = json.load(f)
        if pmc["workload"] == {"n_gates": n, "width": args.width}:
            traffic = pmc["kernels"]["c2a::k_boolify"]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass

    cpu = None
    if args.cpu_sample_layers > 0:
        from oracle import oracle as orc
        sfg = synth.layered_dag(min(args.cpu_sample_layers, args.layers), args.layer_width, seed=synth.SEED)
        orc.lib()
        t0 = time.perf_counter()
        circ, handle = orc.build_circuit(sfg.lh, sfg.rh, sfg.out, sfg.op, sfg.n_nodes, sfg.input_nodes,
                                         sfg.output_nodes, mode=0, keep_handle=True)
        t1 = time.perf_counter()
        bc, bh = orc.boolify_handle(handle, args.width, copy=False)
        t2 = time.perf_counter()
        ng = len(bc.in0)
        orc.lib().orc_free_bool(bh)
        tf0 = time.perf_counter()
        orc.build_circuit(sfg.lh, sfg.rh, sfg.out, sfg.op, sfg.n_nodes, sfg.input_nodes, sfg.output_nodes, mode=1)
        tf1 = time.perf_counter()
        orc.free_circuit(handle)
        cpu = {"value": sfg.n / (t2 - t0), "unit": "gates/s", "cores": 1, "kind": "port",
               "sample": f"same generator, first {sfg.layers} layers x {sfg.layer_width} = {sfg.n} gates, width {args.width}: "
                         f"structure-faithful build_circuit (hash maps, per-visit Vec) {t1 - t0:.2f}s + bit-blast of {ng} "
                         f"boolean gates {t2 - t1:.2f}s; flat-array build_circuit variant {tf1 - tf0:.2f}s",
               "host_cores_available": os.cpu_count()}

    checked = None
    if args.check:
        # full-size parity of the build_circuit outputs against the oracle, by position-salted checksums
        from oracle import oracle as orc
        backend_mod = importlib.import_module("circom-2-arithc_amd.backend")
        exp = orc.build_circuit(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes, mode=1)
        for name, arr in (("sorted", exp.sorted), ("in0", exp.in0), ("in1", exp.in1), ("out", exp.out), ("op", exp.op)):
            assert be.checksum(name) == backend_mod.checksum_host(arr), f"{name} differs from the oracle"
        checked = "sorted/in0/in1/out/op checksums == oracle at full size"

    line = {
        "metric": "gates/sec (topo-sort + boolify), 10M-gate DAG",
        "value": value, "unit": "gates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"synthetic layered DAG, {args.layers} layers x {args.layer_width} = {n} gates/GPU, fan-in 2, "
                               f"gate ids permuted, sparse node ids, seed {synth.SEED}(+rank), --boolify-width {args.width}",
                   "n_gates_per_gpu": n, "boolean_gates_per_gpu": info.n_gates, "boolify_width": args.width,
                   "levels": stats["levels"], "dfs_tree_depth": stats["max_depth"],
                   "parallelism": ("1 graph, sort replicated, boolify sharded by sorted-position range" if shard else
                                   "1 graph per GPU (replicated pipeline, no collective)") if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "kernel": "k_boolify", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": "profiles/r01_pmc_hbm_bytes.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)" if traffic else None,
                     "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": bool_ms},
        "cpu_baseline": cpu,
        "stages_ms": stages,
        "whole_job_algorithmic_GBps": (30.0 * n + algo_bytes) / (ms_per_step * 1e-3) / 1e9,
        "setup_s": {"generate": gen_s, "h2d_and_alloc": h2d_s},
        "stats": stats,
        "checked": checked,
    }
    print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
