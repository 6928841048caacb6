"""The C-ABI library loads and exports every symbol include/c2a.h declares (no GPU call)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "c2a.h")).read()
    return sorted(set(re.findall(r"\b(c2a_[a-z_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for n in ("c2a_create", "c2a_destroy", "c2a_load_gates", "c2a_topo_sort", "c2a_assign_wires", "c2a_emit_gates",
              "c2a_build_circuit", "c2a_boolify", "c2a_bool_read"):
        assert n in names


def test_product_library_exports_every_symbol(c2a):
    path = c2a.library_path()
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "circom-2-arithc_amd", "csrc")])
    lib = ctypes.CDLL(path)
    for n in _declared():
        assert hasattr(lib, n), n
    lib.c2a_version.restype = ctypes.c_char_p
    assert b"hip" in lib.c2a_version()
    # host-only query works without a GPU
    g, a = ctypes.c_uint64(), ctypes.c_uint64()
    assert lib.c2a_template_size(10, 32, ctypes.byref(g), ctypes.byref(a)) == 0
    assert (g.value, a.value) == (32, 0)


def test_abi_version_is_one_number_everywhere(c2a):
    """include/c2a.h, the built library and the ctypes binding agree on C2A_ABI_VERSION (a signature change must
    be a loud load-time error, not undefined behaviour), and the device query works without a GPU."""
    import importlib
    src = open(os.path.join(ROOT, "include", "c2a.h")).read()
    header = int(re.search(r"#define\s+C2A_ABI_VERSION\s+(\d+)", src).group(1))
    backend_mod = importlib.import_module("circom-2-arithc_amd.backend")
    lib = ctypes.CDLL(c2a.library_path())
    assert lib.c2a_abi_version() == header == backend_mod.ABI_VERSION
    assert c2a.visible_devices() >= 0


def test_no_gpu_means_loud_failure(c2a):
    """On a box without a GPU the product path must raise, never fall back."""
    import shutil
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    with pytest.raises(c2a.BackendError):
        c2a.Backend(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "circom-2-arithc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", "").replace("No oracle", "") or f == "synth.py", \
                    os.path.join(dirpath, f)
