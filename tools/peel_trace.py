#!/usr/bin/env python3
"""usage (on the GPU box): python tools/peel_trace.py [layers width [window [n_in]]] — where the dataflow launch spends its critical path.
Runs the statistics build once with C2A_PEEL_TRACE (every gate notes when its step started, how it came to its wave — chain step,
popped from a hand-off array, seed — and when its record was stored), then walks the dependency graph: for every gate the consumer
whose record was stored LAST is its critical consumer; from the gate that finished last back to a sink along critical consumers =
the critical path of the launch, split by how its gates came."""
import importlib
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
width = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
window = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n_in = int(sys.argv[4]) if len(sys.argv) > 4 else 4096      # (few inputs: nearly every rh is a gate of the window)
d = tempfile.mkdtemp()
os.environ["C2A_PEEL_STATS"] = "1"
os.environ["C2A_PEEL_TRACE"] = d
c2a = importlib.import_module("circom-2-arithc_amd")
if os.environ.get("PEEL_TRACE_SHA"):       # PEEL_TRACE_SHA=chain:8 / tree:290 — a tiling of the real SHA-256 block instead (tools/family_check.py)
    from tools.family_check import sha_block
    shape, copies = os.environ["PEEL_TRACE_SHA"].split(":")
    fg = c2a.synth.tile_block(*sha_block(), copies=int(copies), shape=shape, seed=c2a.synth.SEED, permute=False)
else:
    fg = c2a.synth.layered_dag(layers, width, window=window, n_in=n_in, n_const=64 if n_in >= 64 else 4, seed=c2a.synth.SEED)
be = c2a.Backend(0)
be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
be.topo_sort(fetch=False)
be.topo_sort(fetch=False)                    # (the second run is the warm one; its trace overwrites the first)
n = fg.n
# (the launch works in RANK space — gates relabelled by out-node order, c2a_kernels.h RELABELLING — and so do its trace arrays:
# bring the graph into the same numbering)
order = np.argsort(fg.out, kind="stable")
fg.lh, fg.rh, fg.out = fg.lh[order], fg.rh[order], fg.out[order]
tr = np.fromfile(os.path.join(d, "peel_trace.bin"), dtype=np.uint64).reshape(n, 3)
meta = np.fromfile(os.path.join(d, "peel_meta.bin"), dtype=np.uint32).reshape(n, 4)
t_start = (tr[:, 0] >> np.uint64(2)).astype(np.int64)
came = (tr[:, 0] & np.uint64(3)).astype(np.int64)
t_done = tr[:, 1].astype(np.int64)
level = (meta[:, 3] >> 1).astype(np.int64)
traced = t_done > 0                                   # (sinks are done by the sinks pass: no trace)
t0 = t_start[traced].min()
t_start = np.where(traced, t_start - t0, 0)
t_done = np.where(traced, t_done - t0, 0)
prod = np.full(fg.n_nodes, -1, dtype=np.int64)
prod[fg.out] = np.arange(n)
d0 = prod[fg.lh]
d1 = prod[fg.rh]
d1 = np.where(d1 == d0, -1, d1)
cons = np.concatenate([np.nonzero(d0 >= 0)[0], np.nonzero(d1 >= 0)[0]])
dep = np.concatenate([d0[d0 >= 0], d1[d1 >= 0]])
maxdone = np.zeros(n, dtype=np.int64)
np.maximum.at(maxdone, dep, t_done[cons])
crit = np.full(n, -1, dtype=np.int64)
hit = t_done[cons] == maxdone[dep]
crit[dep[hit]] = cons[hit]
us = 0.01                                             # 100 MHz clock
print(f"launch: {t_done.max() * us:.1f} us traced, {int(traced.sum())} gates with a step, levels {level.max() + 1}")
names = {0: "chain step", 1: "popped (handed off)", 2: "seed"}
for k in (0, 1, 2):
    m = traced & (came == k)
    if not m.any():
        continue
    delay = (t_done[m] - maxdone[m]) * us
    waitrec = (maxdone[m] > t_start[m]).mean()
    print(f"  {names[k]:20s}: {int(m.sum()):9d} gates | own record stored {delay.mean():6.2f} us after the critical consumer's (median {np.median(delay):5.2f}) | "
          f"step {((t_done[m] - t_start[m]) * us).mean():5.2f} us | critical consumer still running at step start: {100 * waitrec:4.1f} %")
# the critical path
g = int(np.argmax(t_done))
tot = {0: 0.0, 1: 0.0, 2: 0.0}
stp = {0: 0.0, 1: 0.0, 2: 0.0}
gap = {0: 0.0, 1: 0.0, 2: 0.0}
cnt = {0: 0, 1: 0, 2: 0}
hops = 0
same_wave = 0
durs = []
while g >= 0 and traced[g]:
    k = int(came[g])
    tot[k] += (t_done[g] - maxdone[g]) * us
    stp[k] += (t_done[g] - t_start[g]) * us
    gap[k] += (t_start[g] - maxdone[g]) * us
    durs.append((t_done[g] - t_start[g]) * us)
    cnt[k] += 1
    hops += 1
    g = int(crit[g])
print(f"critical path: {hops} gates, {sum(tot.values()):.1f} us")
for k in (0, 1, 2):
    if cnt[k]:
        print(f"  {names[k]:20s}: {cnt[k]:6d} gates, {tot[k]:9.1f} us ({tot[k] / cnt[k]:5.2f} us each = own step {stp[k] / cnt[k]:5.2f} + "
              f"{gap[k] / cnt[k]:5.2f} between the critical consumer's record and the start of the step)")
# what the steps of the critical path are made of, against all chain steps
w3 = tr[:, 2]
ph = [((w3 >> np.uint64(12 * i)) & np.uint64(0xFFF)).astype(np.int64) * us for i in range(4)]
pushed = ((w3 >> np.uint64(48)) & np.uint64(1)).astype(bool)
take = ((w3 >> np.uint64(49)) & np.uint64(7)).astype(np.int64)
cold = ((w3 >> np.uint64(52)) & np.uint64(1)).astype(bool)
path = []
g = int(np.argmax(t_done))
while g >= 0 and traced[g]:
    path.append(g)
    g = int(crit[g])
path = np.array(path)
chain_all = np.nonzero(traced & (came == 0))[0]
for label, idx in (("critical path", path[came[path] == 0]), ("all chain steps", chain_all)):
    print(f"  {label:16s}: top wait {ph[0][idx].mean():.2f}  issue {ph[1][idx].mean():.2f}  tournament {ph[2][idx].mean():.2f}  stores {ph[3][idx].mean():.2f} us | "
          f"pushes a second producer {100 * pushed[idx].mean():4.1f} %  records loaded ahead {take[idx].mean():.2f}  reads the list itself {100 * cold[idx].mean():4.1f} %")
for label, m in (("steps that push", pushed), ("steps that do not", ~pushed)):
    idx = chain_all[m[chain_all]]
    print(f"  {label:16s}: top wait {ph[0][idx].mean():.2f}  issue {ph[1][idx].mean():.2f}  tournament {ph[2][idx].mean():.2f}  stores {ph[3][idx].mean():.2f} us ({len(idx)} steps)")
for k in range(5):
    for pu in (False, True):
        idx = chain_all[(take[chain_all] == k) & (pushed[chain_all] == pu) & ~cold[chain_all]]
        if len(idx):
            print(f"  {k} records ahead, {'push   ' if pu else 'no push'}: issue {ph[1][idx].mean():.2f} tournament {ph[2][idx].mean():.2f} stores {ph[3][idx].mean():.2f} top wait {ph[0][idx].mean():.2f} us ({len(idx)} steps)")
durs = np.array(durs)
print("  step durations on the critical path (us): p10 %.2f p50 %.2f p90 %.2f p99 %.2f" % tuple(np.percentile(durs, [10, 50, 90, 99])))
alld = ((t_done - t_start) * us)[traced & (came == 0)]
print("  step durations of all chain steps   (us): p10 %.2f p50 %.2f p90 %.2f p99 %.2f" % tuple(np.percentile(alld, [10, 50, 90, 99])))
