# headroom experiment: K independent 10M-gate graphs on ONE GPU, one context + host thread each
import sys, time, threading, importlib
sys.path.insert(0, '.')
c2a = importlib.import_module('circom-2-arithc_amd')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fgs = [c2a.synth.layered_dag(5000, 2000, seed=20241008 + i) for i in range(K)]
bes = []
for fg in fgs:
    be = c2a.Backend(0)
    be.load_gates(fg.lh, fg.rh, fg.out, fg.op, fg.n_nodes, fg.input_nodes, fg.output_nodes)
    bes.append(be)
def step(be):
    be.build_circuit(); be.boolify(32)
for be in bes: step(be)          # warm-up
for rep in range(3):
    t0 = time.perf_counter()
    th = [threading.Thread(target=step, args=(be,)) for be in bes]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"K={K}: {dt*1e3:.1f} ms for {K} graphs -> {K*10e6/dt/1e6:.0f} M gates/s aggregate")
