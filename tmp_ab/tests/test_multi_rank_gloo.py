"""The N > 1 path of bench.py on CPU: world_size-2 gloo.  bench.py's multi-GPU mode is "one independent gate graph
per rank, no data-path collective" (DESIGN.md §7), so what has to be right is (a) the timed region: barrier on
both sides and MAX over ranks, (b) the whole-job rate, (c) per-rank seeds giving different graphs."""
import importlib
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    c2a = importlib.import_module("circom-2-arithc_amd")
    fg = c2a.synth.layered_dag(6, 10, n_in=4, n_const=2, window=2, seed=c2a.synth.SEED + rank)
    calls = {"warm": 0, "timed": 0}

    def warm():
        calls["warm"] += 1

    def step():
        calls["timed"] += 1
        time.sleep(0.05 * (rank + 1))            # rank 1 is the slow one

    elapsed = bench.timed_region(warm, step, steps=3, warmup=2, dist=dist, torch=torch, device=None)
    q.put((rank, elapsed, calls["warm"], calls["timed"], int(fg.lh.sum()), fg.n))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, e0, w0, t0, s0, n0), (r1, e1, w1, t1, s1, n1) = res
    assert (w0, t0, w1, t1) == (2, 3, 2, 3)                 # W untimed + exactly K timed steps on every rank
    assert abs(e0 - e1) < 1e-9                              # both ranks hold the MAX
    assert e0 >= 3 * 0.1 * 0.95                             # ... which is the slow rank's time
    assert s0 != s1 and n0 == n1                            # different graphs, same size (weak scaling)
    import bench
    assert bench.whole_job_rate(2, 1000, 3, 2.0) == 3000.0
