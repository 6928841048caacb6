"""The C++ host mirror (circom-2-arithc_amd/host/) — the reference's integration tests in C++ (tests/cpp/integration.cpp)
and the CLI that mirrors src/main.rs — built against the emulated library for the CPU suite and against libc2a_hip.so
for the GPU suite."""
import json
import os
import subprocess

import pytest

from helpers import load_fixtures

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "circom-2-arithc_amd", "host")
FX = load_fixtures()


def _build(tmp_path, kind, emul_lib=None):
    if kind == "emul":
        libdir, lib = os.path.dirname(emul_lib), "c2a_emul"
    else:
        libdir, lib = os.path.join(ROOT, "circom-2-arithc_amd"), "c2a_hip"
    out = {}
    for name, src in (("integration", os.path.join(ROOT, "tests", "cpp", "integration.cpp")),
                      ("cli", os.path.join(HOST, "c2a_cli.cpp"))):
        exe = str(tmp_path / f"{name}_{kind}")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-o", exe, src, f"-L{libdir}", f"-l{lib}",
                               f"-Wl,-rpath,{libdir}"])
        out[name] = exe
    return out


KINDS = [pytest.param("emul", id="emul"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("kind", KINDS)
def test_reference_integration_tests_in_cpp(kind, tmp_path, request):
    exes = _build(tmp_path, kind, request.getfixturevalue("emul_lib") if kind == "emul" else None)
    env = dict(os.environ)
    res = subprocess.run([exes["integration"]], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "all checks passed" in res.stdout


@pytest.mark.parametrize("kind", KINDS)
def test_cli_writes_the_reference_artefacts(kind, tmp_path, request):
    exes = _build(tmp_path, kind, request.getfixturevalue("emul_lib") if kind == "emul" else None)
    env = dict(os.environ)
    fx = FX["infixOps"]
    calls = tmp_path / "calls.txt"
    with open(calls, "w") as f:
        for st in fx["script"]:
            if st[0] == "signal":
                f.write(f"signal {st[1]} {st[2]}" + (f" {st[3]}" if st[3] is not None else "") + "\n")
            elif st[0] == "gate":
                f.write(f"gate {st[1]} {st[2]} {st[3]} {st[4]}\n")
            else:
                f.write(f"connect {st[1]} {st[2]}\n")
        for p in fx["input_prefixes"]:
            f.write(f"inputs {p}\n")
        for p in fx["output_prefixes"]:
            f.write(f"outputs {p}\n")
    outdir = tmp_path / "output"
    subprocess.check_call([exes["cli"], "-i", str(calls), "-o", str(outdir)], env=env, timeout=600)
    text = (outdir / "circuit.txt").read_text().split("\n")
    assert text[0] == f"{len(fx['gates'])} {fx['expect']['hand']['wire_count']}"
    assert text[1].split()[0] == "6" and text[2].split()[0] == "29"
    info = json.loads((outdir / "circuit_info.json").read_text())
    assert info["input_name_to_wire_index"] == {f"0.x{i}": i for i in range(6)}
    assert set(info["output_name_to_wire_index"].values()) == set(range(6, 35))
    gate_lines = [ln for ln in text[4:] if ln]
    assert [ln.split()[-1] for ln in gate_lines] == [g[0] for g in fx["gates"]]
    # with --boolify-width the same CLI emits the boolean circuit
    subprocess.check_call([exes["cli"], "-i", str(calls), "-o", str(outdir), "--boolify-width", "8"], env=env, timeout=600)
    head = (outdir / "circuit.txt").read_text().split("\n")[:3]
    assert head[1] == "6 " + " ".join(["8"] * 6) and head[2].startswith("29 8 8")
    ops = {ln.split()[-1] for ln in (outdir / "circuit.txt").read_text().split("\n")[4:] if ln}
    assert ops <= {"XOR", "AND", "INV"}


@pytest.mark.parametrize("kind", KINDS)
def test_cli_gpu_writer_equals_host_writer_and_report_json(kind, tmp_path, request, orc):
    """circuit.txt printed on the GPU (c2a_format_bristol, the CLI's default) is byte-identical to the host writer
    (--host-writer), arithmetic and boolean; report.json (src/main.rs:46-47) equals the literal restatement of
    compiler.rs:287-319 + :502-531 driven by the same calls."""
    exes = _build(tmp_path, kind, request.getfixturevalue("emul_lib") if kind == "emul" else None)
    fx = FX["infixOps"]
    calls = tmp_path / "calls.txt"
    lit = orc.CompilerModel()
    with open(calls, "w") as f:
        for st in fx["script"]:
            if st[0] == "signal":
                f.write(f"signal {st[1]} {st[2]}" + (f" {st[3]}" if st[3] is not None else "") + "\n")
                lit.add_signal(st[1], st[2], st[3])
            elif st[0] == "gate":
                f.write(f"gate {st[1]} {st[2]} {st[3]} {st[4]}\n")
                lit.add_gate(orc.OP[st[1]], st[2], st[3], st[4])
            else:
                f.write(f"connect {st[1]} {st[2]}\n")
                lit.add_connection(st[1], st[2])
        for p in fx["input_prefixes"]:
            f.write(f"inputs {p}\n")
        for p in fx["output_prefixes"]:
            f.write(f"outputs {p}\n")
    for p in fx["input_prefixes"]:
        lit.add_inputs(lit.get_signals("0." + p))
    for p in fx["output_prefixes"]:
        lit.add_outputs(lit.get_signals("0." + p))
    lit_circ = lit.build_circuit()
    pay = lit.flat_payload()
    o_arith = orc.build_circuit(pay["lh"], pay["rh"], pay["out"], pay["op"], pay["n_nodes"], pay["input_nodes"], pay["output_nodes"], mode=0)
    o_bool = orc.boolify(o_arith, 5)
    # what both writers of the CLI must produce: the checker's own per-gate writer over the literal build_circuit / the oracle's bit-blast
    want = {0: (orc.bristol_text_of(lit_circ).encode(), orc.circuit_info_json(lit_circ).encode()),
            5: (orc.bristol_text_of(o_bool).encode(),
                orc.circuit_info_json(lit_circ, 5, lambda w: orc.bool_wire(o_arith, o_bool.wire_count - o_arith.wire_count * 5, 5, w)).encode())}
    for extra in ([], ["--boolify-width", "5"]):
        a, b = tmp_path / "gpu", tmp_path / "host"
        subprocess.check_call([exes["cli"], "-i", str(calls), "-o", str(a)] + extra, timeout=600)
        subprocess.check_call([exes["cli"], "-i", str(calls), "-o", str(b), "--host-writer"] + extra, timeout=600)
        assert (a / "circuit.txt").read_bytes() == (b / "circuit.txt").read_bytes()
        assert (a / "circuit_info.json").read_bytes() == (b / "circuit_info.json").read_bytes()
        text, info = want[5 if extra else 0]
        assert (a / "circuit.txt").read_bytes() == text
        assert (a / "circuit_info.json").read_bytes().rstrip(b"\n") == info
    rep = json.loads((a / "report.json").read_text())
    assert rep == lit.generate_circuit_report("sint")
    assert (a / "report.json").read_text() == json.dumps(lit.generate_circuit_report("sint"), indent=2)   # serde pretty layout
    assert all("random_" not in nm for r in rep["inputs"] + rep["outputs"] for nm in r["names"])
    assert len(rep["outputs"]) == 29 and len(rep["inputs"]) >= 6
