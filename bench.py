#!/usr/bin/env python3
"""bench.py — gates/sec through (topo-sort + wire numbering + gate emission + boolify) on the synthetic
10 M-gate DAG of BASELINE.json (configs[4]), device-resident in / device-resident out.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--width 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N ...

A "step" = one pass of the hot path over ONE gate graph already resident in HBM: c2a_build_circuit (sort,
numbering, emission) + c2a_boolify(width).

N > 1 (default `--mode shard`): still ONE 10 M-gate graph — BASELINE's metric.  The sort is a chain of ~5 000
dependent steps and does not shard, so every rank sorts the same graph and bit-blasts only its own
sorted-position range (no data-path collective: each rank holds the whole sorted circuit).  `"scaling": "strong"`,
value = n / max-over-ranks time; the line says what bounds it (`config.strong_scaling_bound`).  `--mode replicas` is
the explicitly named throughput mode: N independent graphs, one per GPU, `"scaling": "weak"`.

One JSON line on stdout (rank 0):
  * `roofline` describes the WHOLE timed step: achieved = (30 n + sum_g (13 + 13 T(op_g, w))) bytes
    / ms_per_step, against the 8 TB/s HBM spec; `roofline.kernels` has one entry per dominant kernel (the dataflow
    peel, the boolify map) with its own algorithmic bytes, its launch time from HIP events on the library's stream,
    and its HBM traffic from the rocprofv3 PMC passes committed under profiles/;
  * `cpu_baseline`: the CPU oracle ("port": the reference is Rust and cannot be built here), 1 core, on the SAME
    10 M-gate input, nothing scaled: the structure-faithful build_circuit (hash maps, per-visit Vec — what BASELINE.md §2
    defines as THE baseline) on the whole graph + the bit-blast of ALL sorted gates (in chunks, so that the 9.6 GB boolean
    output never sits in host memory at once); `flat_array` is the second row of BASELINE.md §2 (dense ids, no hashing);
  * `width64`: the same step at --boolify-width 64 (SURVEY §8(d) "widths 32 and 64").
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

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (~6.3 TB/s achievable)
PMC_PROFILE = os.path.join(ROOT, "profiles", "r06_pmc_hbm_bytes.json")


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


def whole_job_rate(graphs, gates_per_graph, steps, elapsed_max):
    """value = gates of all graphs processed by the job / max-over-ranks time."""
    return graphs * gates_per_graph * steps / elapsed_max


def shard_step(be, width, rank, world, fetch=False):
    """One step of shard mode on one rank: the whole build_circuit (replicated), then only this rank's range of the
    boolify map — the rank-th of `world` ranges of sorted positions holding equal numbers of BOOLEAN gates (the library's
    own cut, c2a_boolify_shard_range: every rank holds the whole plan, so no exchange).  Returns (plan info, chunk, (lo, n))."""
    be.build_circuit()
    info = be.boolify_plan(width)
    lo, cnt = be.boolify_shard_range(rank, world)
    chunk = be.boolify_chunk(lo, cnt, fetch=fetch)
    return info, chunk, (lo, cnt)


def load_pmc(n, width):
    try:
        with open(PMC_PROFILE) as f:
            pmc = json.load(f)
        if pmc["workload"] == {"n_gates": n, "width": width}:
            return pmc["kernels"]
    except (OSError, KeyError, ValueError):
        pass
    return {}


def live_pmc(args, timeout_s=150):
    """HBM traffic of THIS command on THIS box: two child runs of the same step under `rocprofv3 --kernel-trace --pmc` (FETCH_SIZE, then
    WRITE_SIZE: the TCC block cannot hold both in one pass), summarised per kernel under the byte model calibrated in
    profiles/r06_pmc_calibration.txt (every read request is a 128-byte line: read bytes = 2 x FETCH_SIZE; WRITE_SIZE exact).  Returns the
    same {kernel: {...}} table as profiles/r06_pmc_hbm_bytes.json, or None (no rocprofv3 on this box, a pass failed or timed out) — the
    line then quotes the committed profile and says so."""
    import re
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    # (this run may itself be a child of rocprofv3 — the driver profiling the bench, tools/kstat.sh —: no profiler inside a profiler)
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    import csv
    import glob
    child_steps = 2
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="c2a_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--pmc-child", "--steps", str(child_steps), "--warmup", "1", "--layers", str(args.layers), "--layer-width", str(args.layer_width), "--width", str(args.width)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
        except (subprocess.TimeoutExpired, OSError):
            shutil.rmtree(d, ignore_errors=True)
            return None
        agg, launches = {}, {}
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if row["Counter_Name"] != counter:
                        continue
                    k = re.sub(r"<.*$", "", re.sub(r"^void ", "", row["Kernel_Name"].split("(")[0]))
                    agg[k] = agg.get(k, 0.0) + float(row["Counter_Value"])
                    launches.setdefault(k, set()).add(row["Dispatch_Id"])
        shutil.rmtree(d, ignore_errors=True)
        if r.returncode != 0 or "c2a::k_boolify" not in agg:
            return None
        for k in agg:
            if k.startswith("c2a::"):
                vals.setdefault(k, {})[counter] = agg[k] / len(launches[k])
                vals[k]["launches"] = len(launches[k])
    base = vals["c2a::k_boolify"]["launches"] or 1              # one k_boolify per step
    out = {}
    for k, v in vals.items():
        f, w = v.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0, v.get("WRITE_SIZE", 0.0) * 1024.0
        out[k] = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes_per_launch": f + w, "launches_per_step": v["launches"] / base}
    return out


def oracle_circuit(fg):
    """The CPU oracle's build_circuit of the benchmark input (flat-array variant): the checker of `checked`, and the first
    half of `cpu_baseline`.  Returns (circuit, handle, seconds)."""
    from oracle import oracle as orc
    orc.lib()
    t0 = time.perf_counter()
    circ, handle = orc.build_circuit(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes, mode=1,
                                     keep_handle=True)
    return circ, handle, time.perf_counter() - t0


def check_against_oracle(be, backend_mod, circ, width, shard=None, slice_gates=20_000):
    """Bit-for-bit check of what the timed steps left in HBM: position-salted checksums of the sorted ids and
    the emitted circuit over ALL gates, plus a bounded slice of the boolean circuit (the rank's own range in shard mode)
    element by element against the oracle's bit-blast.  Raises on any difference; returns a description."""
    from oracle import oracle as orc
    for name, arr in (("sorted", circ.sorted), ("in0", circ.in0), ("in1", circ.in1), ("out", circ.out), ("op", circ.op)):
        assert be.checksum(name) == backend_mod.checksum_host(arr), f"{name} differs from the oracle"
    # the node -> wire map (device side: wire + 1, 0 = no wire; oracle: 0xFFFFFFFF = no wire)
    nw1 = ((circ.node_wire.astype(np.uint64) + 1) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    assert be.checksum("node_wire1") == backend_mod.checksum_host(nw1), "node -> wire map differs from the oracle"
    n = len(circ.sorted)
    if shard is None:
        first, cnt = max(0, n // 2 - slice_gates // 2), min(slice_gates, n)
        sl, g0 = orc.boolify_range(circ, width, first, cnt)
        got = be.bool_read(g0, len(sl.in0))
    else:
        lo, ln = shard
        first, cnt = lo, min(slice_gates, ln)
        sl, g0 = orc.boolify_range(circ, width, first, cnt)
        q0, got = be.boolify_chunk(first, cnt)
        assert q0 == g0, "first boolean gate of the shard differs from the oracle"
    for a, b in zip(got, (sl.in0, sl.in1, sl.out, sl.op)):
        assert np.array_equal(a, b), "boolean gates differ from the oracle"
    msg = (f"sorted/in0/in1/out/op/node->wire checksums == oracle over all {n} gates; boolean gates of sorted positions "
           f"[{first}, {first + cnt}) == oracle element by element ({len(sl.in0)} gates)")
    if shard is None:
        # ... and ALL of the boolean circuit functionally: both circuits simulated on 64 vectors on the GPU, every arithmetic
        # wire compared with its boolean wires (c2a_verify_boolify)
        pairs, bad = be.verify_boolify(seed=20241008)
        assert bad == 0, f"{bad} (wire, vector) pairs of the boolean circuit differ from the arithmetic circuit"
        msg += f"; every wire of the boolean circuit == the arithmetic circuit on 64 vectors ({pairs} pairs, c2a_verify_boolify)"
    return msg


def cpu_baseline(fg, width, chunk_gates, circ, handle, t_flat):
    """CPU oracle on the host cores of this box, 1 thread, the whole benchmark input: the structure-faithful
    build_circuit (hash maps, per-visit Vec: THE baseline) and the flat-array one (second row), and the bit-blast of every sorted
    gate — chunk by chunk, each chunk's boolean gates freed before the next (the whole output is 9.6 GB)."""
    from oracle import oracle as orc
    t0 = time.perf_counter()
    c2, h2 = orc.build_circuit(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes, mode=0, keep_handle=True)
    t_faithful = time.perf_counter() - t0
    same = bool(np.array_equal(c2.sorted, circ.sorted) and np.array_equal(c2.in0, circ.in0) and np.array_equal(c2.out, circ.out))
    orc.free_circuit(h2)
    del c2
    t_bool, n_bool = 0.0, 0
    for first in range(0, fg.n, chunk_gates):
        cnt = min(chunk_gates, fg.n - first)
        t0 = time.perf_counter()
        bslice, _ = orc.boolify_range(circ, width, first, cnt)
        t_bool += time.perf_counter() - t0
        n_bool += len(bslice.in0)
        del bslice
    return {"value": fg.n / (t_faithful + t_bool), "unit": "gates/s", "cores": 1, "kind": "port",
            "sample": f"the SAME {fg.n}-gate input as the GPU run, nothing scaled: structure-faithful build_circuit (hash maps, per-visit Vec) "
                      f"{t_faithful:.2f}s + bit-blast of all {fg.n} sorted gates ({n_bool} boolean gates, in chunks of {chunk_gates}) {t_bool:.2f}s",
            "flat_array": {"value": fg.n / (t_flat + t_bool), "unit": "gates/s",
                           "sample": f"flat-array build_circuit (dense ids, no hashing) {t_flat:.2f}s + the same bit-blast {t_bool:.2f}s"},
            "variants_agree": same, "boolean_gates": n_bool, "host_cores_available": os.cpu_count()}


def small_configs(c2a, new_backend, reps=10):
    """BASELINE.json configs[0..3] as REAL circuits, the way the reference runs them — build_circuit + boolify ONCE per process
    (src/main.rs:28-32) on 10^1 .. 10^5 gates —: the shipped input/circuit.circom (ArgMax(2); its flat list is the committed
    fixture tests/golden/argmax2.json), a Poseidon-shaped permutation over Z/2^32, SHA-256 over nine blocks (width 32) and the
    SHA3-256 sponge over 29 blocks (width 64), the last three unrolled here from tests/golden/circuits/*.circom.  Per circuit: the
    GPU step with the circuit resident (steady, mean of `reps`), the first load + build + boolify on a fresh context (cold; the
    context's creation apart), the CPU oracle on one core (structure-faithful build_circuit + bit-blast), and the GPU's arrays —
    sorted ids, emitted gates, every boolean gate — against the oracle's.  Where the GPU path starts to pay is read off this
    table ("When to dispatch")."""
    from oracle import oracle as orc
    comp = importlib.import_module("circom-2-arithc_amd.compiler")
    gold = os.path.join(ROOT, "tests", "golden")
    out = []

    def payload_of(name):
        if name == "argmax2":
            fx = json.load(open(os.path.join(gold, "argmax2.json")))
            g = fx["gates"]
            return (np.array([x[1] for x in g], np.uint32), np.array([x[2] for x in g], np.uint32), np.array([x[3] for x in g], np.uint32),
                    np.array([orc.OP[x[0]] for x in g], np.uint8), fx["n_nodes"], np.array(fx["input_nodes"], np.uint32),
                    np.array(fx["output_nodes"], np.uint32)), 0.0
        t0 = time.perf_counter()
        C = comp.Compiler.from_circom(open(os.path.join(gold, "circuits", name + ".circom")).read(), backend=None)
        inputs, outputs, _ = C._io_maps()
        lh, rh, o, op = C._flat()
        return (lh, rh, o, op, C.node_count + 1, np.array([nd for _, nd in inputs], np.uint32), np.array([nd for _, nd in outputs], np.uint32)), time.perf_counter() - t0

    for name, label, width in (("argmax2", "configs[0]: input/circuit.circom = ArgMax(2)", 32), ("poseidonLike", "configs[1]: Poseidon-shaped permutation over Z/2^32", 32),
                               ("sha256", "configs[2]: SHA-256 over 9 blocks", 32), ("sha3_256", "configs[3]: SHA3-256 sponge over 29 rate blocks", 64)):
        args, t_unroll = payload_of(name)
        n = len(args[0])
        # CPU: the oracle's faithful variant + the whole bit-blast, best of 3
        cpu = []
        for _ in range(3):
            t0 = time.perf_counter()
            circ, h = orc.build_circuit(*args, mode=0, keep_handle=True)
            t1 = time.perf_counter()
            eb = orc.boolify(circ, width)
            t2 = time.perf_counter()
            cpu.append((t2 - t0, t1 - t0))
            orc.free_circuit(h)
        cpu_ms, cpu_build_ms = min(cpu)[0] * 1e3, min(cpu)[1] * 1e3
        # GPU cold: a fresh context
        t0 = time.perf_counter()
        be = new_backend()
        t1 = time.perf_counter()
        be.load_gates(*args)
        be.build_circuit()
        be.boolify(width)
        t2 = time.perf_counter()
        # steady
        for _ in range(2):
            be.build_circuit(); be.boolify(width)
        t3 = time.perf_counter()
        acc = {}
        for _ in range(reps):
            be.build_circuit()
            bi = be.boolify(width)
            for k, v in be.timings().items():
                acc[k] = acc.get(k, 0.0) + v / reps
        steady_ms = (time.perf_counter() - t3) * 1e3 / reps
        # check: every array
        ok = bool(np.array_equal(be.topo_sort(), circ.sorted))
        be.build_circuit()
        in0, in1, o_, op_ = be.emit_gates()
        ok = ok and all(np.array_equal(a, b) for a, b in zip((in0, in1, o_, op_), (circ.in0, circ.in1, circ.out, circ.op)))
        bi = be.boolify(width)
        ok = ok and bi.n_gates == len(eb.in0) and all(np.array_equal(a, b) for a, b in zip(be.bool_read(), (eb.in0, eb.in1, eb.out, eb.op)))
        st = be.stats()
        # the HYBRID a maintainer can choose for a deep and narrow circuit (c2a_load_circuit): the reference's own build_circuit on
        # the CPU, only boolify(&circuit, w) on the GPU — the emitted circuit's way over PCIe included
        hyb = []
        for _ in range(3):
            t4 = time.perf_counter()
            be.load_circuit(circ.in0, circ.in1, circ.out, circ.op, circ.wire_count, circ.n_in, circ.n_out)
            bh = be.boolify(width)
            hyb.append((time.perf_counter() - t4) * 1e3)
        ok = ok and bh.n_gates == len(eb.in0) and all(np.array_equal(a, b) for a, b in zip(be.bool_read(), (eb.in0, eb.in1, eb.out, eb.op)))
        be.close()
        out.append({"name": label, "n_gates": n, "width": width, "boolean_gates": int(bi.n_gates), "levels": st["levels"],
                    "gpu_ms_steady": steady_ms, "gpu_stages_ms": {k: round(v, 4) for k, v in acc.items()},
                    "gpu_ms_cold": (t2 - t1) * 1e3, "gpu_context_create_ms": (t1 - t0) * 1e3,
                    "cpu_ms": cpu_ms, "cpu_build_circuit_ms": cpu_build_ms, "cpu_cores": 1,
                    "hybrid_ms": cpu_build_ms + min(hyb), "hybrid_gpu_load_circuit_and_boolify_ms": min(hyb),
                    "gates_per_level": n / max(1, st["levels"]),
                    "gpu_over_cpu_steady": cpu_ms / steady_ms, "gpu_over_cpu_cold": cpu_ms / ((t2 - t1) * 1e3),
                    "unroll_s": t_unroll, "checked": ok})
        assert ok, f"{label}: the GPU's arrays differ from the oracle's"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=32, help="--boolify-width")
    ap.add_argument("--layers", type=int, default=5000)
    ap.add_argument("--layer-width", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the CPU baseline (the oracle on the whole input, 1 host core: ~5 s structure-faithful build_circuit + ~4 s "
                         "bit-blast of all gates at the default size) and, unless --check, the oracle check")
    ap.add_argument("--cpu-bool-chunk", type=int, default=1_000_000, help="sorted gates per chunk of the CPU bit-blast (all gates are timed)")
    ap.add_argument("--no-width64", action="store_true", help="skip the extra --boolify-width 64 step")
    ap.add_argument("--no-artefacts", action="store_true", help="skip the circuit.txt formatting measurement")
    ap.add_argument("--check", action="store_true", help="verify the GPU result against the oracle even when the CPU baseline is skipped")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold single-shot measurement")
    ap.add_argument("--no-prune", action="store_true", help="skip the optional prune pass report")
    ap.add_argument("--no-reference-shaped", action="store_true", help="skip the step on the reference-shaped graph (constants at a tenth of the gates)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not measure the step's HBM traffic in two child runs under rocprofv3 --pmc (~1 min); quote profiles/ instead")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)      # (the child of live_pmc: load, warm-up + steps, nothing printed)
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE's configs[0..3] as real circuits (GPU steady / cold vs CPU per circuit; ~1.5 min, most of it unrolling the 148 K-gate sponge)")
    ap.add_argument("--mode", choices=["both", "shard", "replicas"], default="both",
                    help="N>1: 'shard' = ONE graph, sort replicated on every rank, boolify sharded by sorted-position range (strong "
                         "scaling, BASELINE's metric: the line's `value`); 'replicas' = N independent graphs, one per GPU (throughput, "
                         "weak); 'both' (default) = the shard region, then the replicas region in the same process, reported as "
                         "`aggregate_replicas` beside `value`")
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
    # TESTS ONLY (tests/test_multi_rank_gloo.py): the path of the emulated library — the N-rank flow of this very function then
    # runs on CPUs (gloo, no torch.cuda), so that what the driver launches on 2-8 GPUs has been executed before it gets there
    test_lib = os.environ.get("C2A_BENCH_TEST_LIB")
    # one rank per GPU under torch.distributed.run; a 1-rank launch takes the same path (RANK is set by the launcher)
    if world > 1 or ("RANK" in os.environ and os.environ.get("C2A_BENCH_PLAIN") is None and "TORCHELASTIC_RUN_ID" in os.environ):
        import torch
        import torch.distributed as dist
        if test_lib:
            dist.init_process_group("gloo")
        else:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    sync_device = "cuda" if (dist is not None and not test_lib) else None

    c2a = importlib.import_module("circom-2-arithc_amd")
    synth = c2a.synth

    def new_backend():
        return c2a.Backend(0, lib_path=test_lib) if test_lib else c2a.Backend(local_rank)

    shard = args.mode in ("shard", "both") and world > 1
    replicas = args.mode == "replicas" and world > 1
    both = args.mode == "both" and world > 1
    t0 = time.time()
    fg = synth.layered_dag(args.layers, args.layer_width, seed=synth.SEED + (rank if replicas else 0))
    gen_s = time.time() - t0
    # ---- cold single shot: what one call of the reference's main.rs:28-32 costs from nothing — a fresh context, workspace
    # allocation, the 130 MB payload over PCIe, the node-record clear (overlapped with the copy), ONE build_circuit + boolify
    cold = None
    if world == 1 and not args.no_cold and not args.pmc_child:        # (first thing on the device: nothing of this process is resident yet)
        t0 = time.perf_counter()
        be2 = new_backend()
        t1 = time.perf_counter()
        be2.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
        t2 = time.perf_counter()
        be2.build_circuit()
        t3 = time.perf_counter()
        be2.boolify(args.width)
        t4 = time.perf_counter()
        cold = {"ms": (t4 - t0) * 1e3, "create_ms": (t1 - t0) * 1e3, "alloc_h2d_clear_ms": (t2 - t1) * 1e3,
                "build_circuit_ms": (t3 - t2) * 1e3, "boolify_ms": (t4 - t3) * 1e3,
                "gates_per_s": fg.n / (t4 - t0), "gates_per_s_resident": fg.n / (t4 - t2),
                "note": "first and only run on a fresh context (host clock, PCIe and hipMalloc included); `value` is the steady-state "
                        "rate with the input resident"}
        be2.close()

    be = new_backend()
    t0 = time.time()
    be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    h2d_s = time.time() - t0

    n = fg.n
    last = {}
    if args.pmc_child:
        for _ in range(args.warmup + args.steps):
            be.build_circuit()
            be.boolify(args.width)
        be.close()
        return

    def step():
        if shard:      # every rank holds the whole sorted circuit, so it can place its own range without any exchange
            info_, _, last["range"] = shard_step(be, args.width, rank, world)
            return info_
        be.build_circuit()
        return be.boolify(args.width)

    stage_acc = {}

    def timed_step():
        last["info"] = step()
        for k, v in be.timings().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v

    elapsed = timed_region(step, timed_step, args.steps, args.warmup, dist, torch, sync_device)
    info = last.get("info")

    steps = max(1, args.steps)
    stages = {k: v / steps for k, v in stage_acc.items()}
    stats = be.stats()                                  # (of the timed region's own graph: the replicas region below loads another)
    # ---- check (every rank: its own results against the oracle) — BASELINE.md §2: "outputs compared bit-for-bit"
    backend_mod = importlib.import_module("circom-2-arithc_amd.backend")
    want_oracle = args.check or not args.no_cpu_baseline
    circ = handle = None
    t_oracle_build = 0.0
    checked = None
    if want_oracle and not replicas or (replicas and args.check):
        circ, handle, t_oracle_build = oracle_circuit(fg)
        checked = check_against_oracle(be, backend_mod, circ, args.width, shard=last.get("range") if shard else None)
    # ---- N > 1, mode both: the SAME launch also answers the throughput question — every rank now takes an independent graph of
    # its own (seed + rank), the whole pipeline per GPU, no exchange: N graphs / max-over-ranks time (weak scaling).  north_star's
    # ">= 6x aggregate at 8 GPUs" can only be met here: the one-graph number above is bounded by the replicated sort
    agg = None
    if both:
        fgr = synth.layered_dag(args.layers, args.layer_width, seed=synth.SEED + rank)
        be.load_gates(fgr.lh, fgr.rh, fgr.out, fgr.op, fgr.n_nodes, fgr.input_nodes, fgr.output_nodes)

        def rep_step():
            be.build_circuit()
            be.boolify(args.width)

        el_r = timed_region(rep_step, rep_step, args.steps, args.warmup, dist, torch, sync_device)
        rchecked = None
        if want_oracle:
            rc_, rh_, _ = oracle_circuit(fgr)
            rchecked = check_against_oracle(be, backend_mod, rc_, args.width, slice_gates=2_000)
            from oracle import oracle as orc
            orc.free_circuit(rh_)
            del rc_
        agg = {"value": whole_job_rate(world, fgr.n, max(1, args.steps), el_r), "unit": "gates/s", "scaling": "weak",
               "ms_per_step": el_r * 1e3 / max(1, args.steps), "graphs": world,
               "workload": f"{world} INDEPENDENT graphs of the headline shape, one per GPU (seed {synth.SEED} + rank), the whole pipeline per GPU, no collective",
               "checked": rchecked}
    per_rank = None
    if dist is not None:
        mine = {"rank": rank, "stages_ms": stages, "checked": checked, "shard": last.get("range"), "replica_checked": agg["checked"] if agg else None}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
    if rank != 0:
        if handle is not None:
            from oracle import oracle as orc
            orc.free_circuit(handle)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed * 1e3 / steps
    value = whole_job_rate(world if replicas else 1, n, steps, elapsed)
    # ---- roofline of the whole step: 30 B per gate for sort + numbering + emission, 13 B read per
    # arithmetic gate + 13 B written per boolean gate for the map
    sort_bytes = 30.0 * n
    bool_bytes = 13.0 * n + 13.0 * info.n_gates
    step_bytes = sort_bytes + bool_bytes
    achieved = step_bytes / (ms_per_step * 1e-3) / 1e9
    pmc_live = None
    if world == 1 and not args.no_live_pmc and not test_lib:
        pmc_live = live_pmc(args)
    pmc = pmc_live or load_pmc(n, args.width)

    def kernel_entry(name, pmc_key, algo_bytes, ms):
        e = {"kernel": name, "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": ms,
             "achieved": (algo_bytes / (ms * 1e-3) / 1e9) if ms > 0 else 0.0, "unit": "GB/s"}
        e["frac"] = e["achieved"] / HBM_PEAK_GBS
        k = pmc.get(pmc_key)
        e["traffic"] = k["hbm_bytes_per_launch"] if k else None
        if k:
            for extra in ("fetch_bytes", "write_bytes", "ea_read_requests", "ea_write_requests", "atomics", "sq_wait_frac"):
                if extra in k:
                    e[extra] = k[extra]
        return e

    # what each entry's kernel_ms is: the k_peel launch ALONE (its own event pair); every launch of the step that is neither
    # k_peel nor k_boolify ("around the peel": producer map, relabelling, deps, consumer lists, sinks + whole-level passes,
    # Euler tour + list ranking, wire numbering, emission, the boolify plan) as stage time minus those two; k_boolify alone.
    # Algorithmic bytes: the sort stage's 30 B/gate are booked on the launches around the peel, where the
    # payload is read and sorted ids / emitted gates are written — k_peel itself moves node records only (its algorithmic
    # share is the 8 B/gate of deps it consumes: reported, structurally ~0 of the roofline).
    k_peel_ms = stages.get("k_peel", 0.0)
    around_ms = (stages.get("prep", 0.0) + (stages.get("peel", 0.0) - k_peel_ms) + stages.get("order", 0.0) + stages.get("wires", 0.0) +
                 stages.get("emit", 0.0) + stages.get("bool_prep", 0.0))
    kernels = [kernel_entry("k_peel (the dataflow launch alone: DFS-tree parents of the gates the whole-level passes leave)", "c2a::k_peel", 8.0 * n, k_peel_ms),
               kernel_entry("k_boolify (bit-blast map)", "c2a::k_boolify", bool_bytes, stages.get("bool_map", 0.0))]
    around = {"kernel": "around the peel: every other launch of the step (producer map, relabelling, deps, consumer lists, sinks and whole-level "
                        "passes, Euler tour + list ranking, wire numbering, emission, boolify plan)",
              "algorithmic_bytes_per_step": sort_bytes, "kernel_ms": around_ms,
              "achieved": (sort_bytes / (around_ms * 1e-3) / 1e9) if around_ms > 0 else 0.0, "unit": "GB/s"}
    around["frac"] = around["achieved"] / HBM_PEAK_GBS
    if pmc:
        t = sum(k["hbm_bytes_per_launch"] * k.get("launches_per_step", 1) for name, k in pmc.items() if name not in ("c2a::k_peel", "c2a::k_boolify"))
        around["traffic"] = t
        around["traffic_over_algorithmic"] = t / sort_bytes
    else:
        around["traffic"] = None
    kernels.append(around)
    total_traffic = None
    if pmc:
        total_traffic = sum(k["hbm_bytes_per_launch"] * k.get("launches_per_step", 1) for k in pmc.values())

    cpu = None
    if not args.no_cpu_baseline and not replicas:
        cpu = cpu_baseline(fg, args.width, args.cpu_bool_chunk, circ, handle, t_oracle_build)
    if handle is not None:
        from oracle import oracle as orc
        orc.free_circuit(handle)
    if per_rank is not None and checked is not None:
        assert all(r["checked"] for r in per_rank), "a rank did not check its results"
        checked = f"every one of the {world} ranks: " + checked

    width64 = None
    if not args.no_width64 and args.width != 64 and world == 1:
        be.build_circuit(); be.boolify(64)                     # warm-up (templates, buffers)
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            be.build_circuit()
            i64 = be.boolify(64)
        dt = (time.perf_counter() - t0) / reps
        t64 = be.timings()
        width64 = {"ms_per_step": dt * 1e3, "value": n / dt, "unit": "gates/s", "boolean_gates": i64.n_gates,
                   "bool_map_ms": t64.get("bool_map"), "roofline_frac": (30.0 * n + 13.0 * n + 13.0 * i64.n_gates) / dt / 1e9 / HBM_PEAK_GBS}

    # ---- artefact emission (outside the timed region): the gate lines of circuit.txt printed on the GPU and copied to
    # the host — all of the arithmetic circuit, a bounded slice of the boolean one (the whole text is ~27 GB)
    artefacts = None
    if world == 1 and not args.no_artefacts:
        be.build_circuit()
        bi = be.boolify(args.width)
        t0 = time.perf_counter()
        nbytes = 0
        for first in range(0, n, 1 << 23):
            nbytes += len(be.format_bristol(0, first, min(1 << 23, n - first)))
        t_arith = time.perf_counter() - t0
        cnt = min(1 << 24, bi.n_gates)
        t0 = time.perf_counter()
        bbytes = len(be.format_bristol(1, 0, cnt))
        t_bool = time.perf_counter() - t0
        artefacts = {"circuit_txt_arithmetic": {"gates": n, "bytes": nbytes, "seconds": t_arith},
                     "circuit_txt_boolean_slice": {"gates": cnt, "bytes": bbytes, "seconds": t_bool,
                                                   "whole_circuit_estimate": {"bytes": bbytes * bi.n_gates / max(1, cnt),
                                                                              "seconds": t_bool * bi.n_gates / max(1, cnt)}},
                     "note": "c2a_format_bristol: lengths + scan + print on the GPU, one D2H copy per chunk (PCIe-bound); file writing not included"}

    # ---- the optional prune pass (outside the timed region and NOT the metric's path: c2a_boolify_prune — constant folding and
    # dead-gate removal over the per-gate map, what the absent `boolify` crate is believed to do): how much smaller
    # the circuit gets, and that it still computes the same outputs (64 random vectors through c2a_eval, both circuits)
    prune = None
    if world == 1 and not args.no_prune:
        be.build_circuit()
        be.boolify(args.width)
        t0 = time.perf_counter()
        pi = be.boolify_prune()
        t_prune = time.perf_counter() - t0
        rng = np.random.default_rng(7)
        mask = (1 << args.width) - 1
        vec = rng.integers(0, 2 ** 63, (len(fg.input_nodes), 64), dtype=np.uint64) & np.uint64(mask)
        same = bool(np.array_equal(be.eval(vec, {}, boolean=True), be.eval(vec, {}, pruned=True)))
        assert same, "the pruned circuit computes something else"
        prune = {"gates_before": pi["n_gates_before"], "gates_after": pi["n_gates"], "folded": pi["n_folded"], "dead": pi["n_dead"],
                 "kept_fraction": pi["n_gates"] / max(1, pi["n_gates_before"]), "seconds": t_prune,
                 "outputs_equal_on_64_vectors": same}

    # ---- the same step on a graph shaped like what the reference's own unroller emits (outside the timed region; the headline
    # graph has 64 constant nodes and 2 000 outputs in 10 M gates): a fresh named constant node at a tenth of the gates
    # (src/process.rs:558-579), an output node at a twentieth — 1.5 M events of the wire numbering (src/compiler.rs:431-438)
    # instead of 2 064 — checked against the oracle like the headline
    ref_shaped = None
    if world == 1 and not args.no_reference_shaped:
        fr = synth.layered_dag(args.layers, args.layer_width, const_frac=0.10, out_frac=0.05)
        be.load_gates(fr.lh, fr.rh, fr.out, fr.op, fr.n_nodes, fr.input_nodes, fr.output_nodes)
        be.build_circuit(); be.boolify(args.width)
        acc = {}
        t0 = time.perf_counter()
        for _ in range(steps):
            be.build_circuit()
            ri = be.boolify(args.width)
            for k, v in be.timings().items():
                acc[k] = acc.get(k, 0.0) + v
        dt = (time.perf_counter() - t0) / steps
        rchecked = None
        if want_oracle:
            from oracle import oracle as orc
            rc, rh_ = orc.build_circuit(fr.lh, fr.rh, fr.out, fr.op, fr.n_nodes, fr.input_nodes, fr.output_nodes, mode=1, keep_handle=True)
            rchecked = check_against_oracle(be, backend_mod, rc, args.width)
            orc.free_circuit(rh_)
            del rc
        rst = be.stats()
        ref_shaped = {"workload": f"the headline generator with const_frac 0.10 / out_frac 0.05: {fr.n} gates, {len(fr.const_nodes)} constant nodes, "
                                  f"{len(fr.output_nodes)} output nodes, {len(fr.input_nodes)} inputs",
                      "ms_per_step": dt * 1e3, "value": fr.n / dt, "unit": "gates/s", "vs_headline_ms": dt * 1e3 / ms_per_step,
                      "boolean_gates": ri.n_gates, "stages_ms": {k: v / steps for k, v in acc.items()},
                      "numbering_path": rst["numbering_path"], "numbering_events": rst["numbering_events"], "checked": rchecked}

    configs = None
    if world == 1 and not args.no_configs and not test_lib:
        configs = small_configs(c2a, new_backend)

    sort_ms = stages.get("build_total", 0.0)
    bool_ms = stages.get("boolify_total", 0.0) if not shard else ms_per_step - sort_ms
    # what strong scaling can reach at all: the sort is replicated, only the boolify part B divides by N (Amdahl)
    b1 = bool_ms * world if shard else bool_ms             # boolify of the whole circuit on one GPU
    amdahl = {n_: (sort_ms + b1) / (sort_ms + b1 / n_) for n_ in (1, 2, 4, 8)}
    line = {
        "metric": "gates/sec (topo-sort + boolify), 10M-gate DAG",
        "value": value, "unit": "gates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if replicas else "strong", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": (f"{world} INDEPENDENT synthetic layered DAGs, one per GPU (throughput mode, NOT BASELINE's one-graph metric), each "
                                if replicas else "ONE synthetic layered DAG, ") +
                               f"{args.layers} layers x {args.layer_width} = {n} gates, fan-in 2, gate ids permuted, sparse node ids, "
                               f"seed {synth.SEED}{'(+rank)' if replicas else ''}, --boolify-width {args.width}",
                   "n_gates": n, "boolean_gates": info.n_gates, "boolify_width": args.width,
                   "levels": stats["levels"], "dfs_tree_depth": stats["max_depth"],
                   "parallelism": ("1 graph: sort + numbering + emission replicated on every rank, boolify sharded by sorted-position range, no collective"
                                   if shard else "1 graph per GPU (replicated pipeline, no collective)" if replicas else "single GPU"),
                   "strong_scaling_bound": {"speedup_at_n_gpus": amdahl, "sort_ms": sort_ms, "boolify_ms_one_gpu": b1,
                                            "note": f"the sort does not shard (a chain of {stats['levels']} dependent levels, DESIGN.md §7): "
                                                    "speed-up over 1 GPU <= (sort + B) / (sort + B / N); north_star's >= 6x at 8 GPUs is out of reach "
                                                    "by construction, read the curve against this bound"}},
        "roofline": {"bound": "hbm", "scope": "whole timed step (sort + numbering + emission + boolify)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_step": step_bytes, "traffic": total_traffic,
                     "traffic_source": "MEASURED in this run: two child runs of the same step under rocprofv3 --kernel-trace --pmc (FETCH_SIZE; WRITE_SIZE), byte model calibrated in profiles/r06_pmc_calibration.txt (reads = 2 x FETCH_SIZE = 128 B per request in every access pattern, WRITE_SIZE exact)" if pmc_live else "profiles/r06_pmc_hbm_bytes.json — a QUOTED figure from the committed rocprofv3 --pmc passes of this command under the byte model calibrated in profiles/r06_pmc_calibration.txt (reads = 128 B x TCC_EA0_RDREQ in every access pattern, WRITE_SIZE exact), not counters of this run" if pmc else None,
                     "kernels": kernels,
                     "note": "the step is bound by the dependent-step latency of the exact DFS order (k_peel), not by bytes: its algorithmic traffic is 0.3 GB"},
        "aggregate_replicas": agg,
        "cpu_baseline": cpu,
        "width64": width64,
        "reference_shaped": ref_shaped,
        "configs": configs,
        "artefacts": artefacts,
        "stages_ms": stages,
        "per_rank": per_rank,
        "cold": cold,
        "prune": prune,
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
