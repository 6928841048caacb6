"""Shared fixtures.

Two ways to reach the C ABI of include/c2a.h:
  * "hip"  — the product library (circom-2-arithc_amd/libc2a_hip.so, hand-written HIP on a real MI355X);
             every test using it is marked `gpu`;
  * "emul" — tests/emul/libc2a_emul.so: the SAME kernel/runtime sources compiled with g++ against a host
             emulation header.  Test infrastructure only; lets the CPU suite exercise the kernel logic.
Parity tests are parametrised over both, so `-m "not gpu"` runs them under emulation and `-m gpu` runs
them on the hardware.
"""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMUL_DIR = os.path.join(ROOT, "tests", "emul")
EMUL_LIB = os.path.join(EMUL_DIR, "libc2a_emul.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _pkg():
    return importlib.import_module("circom-2-arithc_amd")


@pytest.fixture(scope="session")
def c2a():
    return _pkg()


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def emul_lib():
    srcs = [os.path.join(ROOT, "circom-2-arithc_amd", "csrc", f) for f in
            ("c2a_api.hip", "c2a_kernels.h", "c2a_peel.h", "c2a_wave.h", "c2a_templates.h", "c2a_platform.h")] + [
        os.path.join(EMUL_DIR, "hip_emul.h"), os.path.join(ROOT, "include", "c2a.h")]
    if (not os.path.exists(EMUL_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(EMUL_LIB) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", EMUL_DIR])
    return EMUL_LIB


BACKENDS = [pytest.param("emul", id="emul"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


class _Env:
    """Temporarily set the library's tuning knobs (read once in c2a_create)."""

    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(params=BACKENDS)
def backend(request, c2a):
    """Emulated build (kernel sources compiled for the host) or the product library on a real GPU."""
    if request.param == "emul":
        be = c2a.Backend(0, lib_path=request.getfixturevalue("emul_lib"))
    else:
        be = c2a.Backend(0)
        assert "hip" in be.version
    yield be
    be.close()


def _variant(kind, mode):
    marks = [pytest.mark.gpu] if kind == "hip" else []
    return pytest.param((kind, mode), id=f"{kind}-{mode}", marks=marks)


# (library build, waves per CU of the dataflow peel: the default, a starved launch, an oversubscribed one)
WAVE_BACKENDS = [_variant("emul", "w8"), _variant("hip", "w8"), _variant("hip", "w1"), _variant("hip", "w32")]


FIFO_BACKENDS = [_variant("emul", "f1"), _variant("emul", "f2"), _variant("emul", "f64"), _variant("hip", "f1"), _variant("hip", "f4"),
                 _variant("hip", "r8")]       # (r8: 8 reserve waves per CU, parked until a backlog builds up in the hand-off arrays — whatever the size of the graph)


@pytest.fixture(params=FIFO_BACKENDS)
def backend_fifo(request, c2a):
    """The dataflow peel with 1 .. 64 hand-off arrays: every pushed entry must be picked up wherever the waves wait."""
    kind, mode = request.param
    with _Env(**({"C2A_PEEL_RESERVE": int(mode[1:])} if mode[0] == "r" else {"C2A_PEEL_FIFOS": int(mode[1:])})):
        be = c2a.Backend(0, lib_path=request.getfixturevalue("emul_lib")) if kind == "emul" else c2a.Backend(0)
    yield be
    be.close()


@pytest.fixture(params=WAVE_BACKENDS)
def backend_wave(request, c2a):
    """The dataflow peel at several launch sizes: termination must not depend on how many waves are resident."""
    kind, mode = request.param
    with _Env(C2A_PEEL_WAVES=int(mode[1:])):
        be = c2a.Backend(0, lib_path=request.getfixturevalue("emul_lib")) if kind == "emul" else c2a.Backend(0)
    yield be
    be.close()


# (the dataflow launch under emulation: every wave is alive at once and C2A_EMUL_SEED shuffles the schedule per pass)
PEEL_BACKENDS = [_variant("emul", "s1"), _variant("emul", "s2"), _variant("emul", "s3"), _variant("emul", "s4"), _variant("emul", "s5"),
                 _variant("hip", "s0")]       # (hip: the protocol on the hardware — memory ordering and visibility are not the emulator's)


@pytest.fixture(params=PEEL_BACKENDS)
def backend_peel(request, c2a):
    """The dataflow peel under a seeded random interleaving of its waves (emulation)."""
    kind, mode = request.param
    seed = mode[1:]
    be = c2a.Backend(0, lib_path=request.getfixturevalue("emul_lib")) if kind == "emul" else c2a.Backend(0)
    old = os.environ.get("C2A_EMUL_SEED")
    if int(seed):
        os.environ["C2A_EMUL_SEED"] = seed
    yield be
    be.close()
    if old is None:
        os.environ.pop("C2A_EMUL_SEED", None)
    else:
        os.environ["C2A_EMUL_SEED"] = old


# (library build, levels behind the sinks done a whole level at once before the dataflow launch: k_peel_shallow)
SHALLOW_BACKENDS = [_variant("emul", "s1"), _variant("emul", "s2"), _variant("emul", "s16"), _variant("emul", "s48"),
                    _variant("hip", "s1"), _variant("hip", "s3"), _variant("hip", "s48")]


@pytest.fixture(params=SHALLOW_BACKENDS)
def backend_shallow(request, c2a):
    """The peel with 1 .. 48 bulk passes in front of the dataflow launch: short records, region overflow, strings up to 48 bits."""
    kind, mode = request.param
    with _Env(C2A_PEEL_SHALLOW=int(mode[1:])):
        be = c2a.Backend(0, lib_path=request.getfixturevalue("emul_lib")) if kind == "emul" else c2a.Backend(0)
    yield be
    be.close()


@pytest.fixture
def hip_backend(c2a):
    be = c2a.Backend(0)
    yield be
    be.close()


def random_gate_graph(rng, n, extra_nodes=6, p_dup_out=0.0, p_same=0.1, p_cycle=0.0, n_in=None, n_out=None):
    """Small adversarial flat gate list: raw node ids 1..K with gaps; fan-in 0..2 through un-produced nodes;
    optional duplicate out nodes (last writer wins, compiler.rs:403-406), lh==rh, back edges (cycles)."""
    K = n + extra_nodes
    ids = np.sort(rng.choice(np.arange(1, 2 * K + 2), size=K, replace=False))      # sparse ids
    rank_of = rng.permutation(K)                                                   # topological rank per node slot
    by_rank = np.argsort(rank_of)
    lh = np.empty(n, dtype=np.int64); rh = np.empty(n, dtype=np.int64); out = np.empty(n, dtype=np.int64)
    free_out = list(rng.permutation(K))
    used_out = []
    for g in range(n):
        if used_out and rng.random() < p_dup_out:
            o = used_out[rng.integers(len(used_out))]
        else:
            o = free_out.pop()
            used_out.append(o)
        r = rank_of[o]
        if r == 0 or rng.random() < p_cycle:
            a, b = rng.integers(K), rng.integers(K)                                  # may create back edges
        else:
            a, b = by_rank[rng.integers(r)], by_rank[rng.integers(r)]
        if rng.random() < p_same:
            b = a
        lh[g], rh[g], out[g] = a, b, o
    produced = set(out.tolist())
    others = [s for s in range(K) if s not in produced]
    rng.shuffle(others)
    if n_in is None:
        n_in = int(rng.integers(0, len(others) + 1)) if others else 0
    in_slots = others[:n_in]
    cand_out = [s for s in range(K) if s not in in_slots]
    if n_out is None:
        n_out = int(rng.integers(0, min(4, len(cand_out)) + 1))
    out_slots = list(rng.choice(cand_out, size=n_out, replace=False)) if n_out else []
    return dict(lh=ids[lh].astype(np.uint32), rh=ids[rh].astype(np.uint32), out=ids[out].astype(np.uint32),
                op=rng.integers(0, 20, n).astype(np.uint8), n_nodes=int(ids.max()) + 1 + int(rng.integers(0, 3)),
                input_nodes=ids[in_slots].astype(np.uint32) if len(in_slots) else np.empty(0, np.uint32),
                output_nodes=ids[out_slots].astype(np.uint32) if len(out_slots) else np.empty(0, np.uint32))
