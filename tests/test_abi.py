"""The C-ABI library loads without a GPU and exports every symbol include/deadtrees_hip.h declares."""
import ctypes
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "deadtrees_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(dt_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_header_and_binding_table_agree():
    from deadtrees_amd import _lib
    hdr = _header_functions()
    assert len(hdr) >= 30
    assert sorted(_lib.SIGNATURES.keys()) == hdr


def test_library_builds_loads_and_exports_all_symbols():
    import __graft_entry__ as g
    g.build()
    from deadtrees_amd import _lib
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_functions():
        assert hasattr(raw, name), name
    assert lib.dt_version() >= 100
    assert isinstance(lib.dt_last_error(), bytes)


def test_host_side_validation_without_gpu():
    """shape preconditions are checked on the host before any launch (no GPU needed to see the error)"""
    from deadtrees_amd import _lib
    lib = _lib.load()
    bad = _lib.ConvDesc(1, 8, 8, 16, 0, 0, 9, 9, 16, 3, 1, 1, 0, 0)   # Ho/Wo inconsistent with Hin/pad/k
    assert lib.dt_conv2d_stat_rows(ctypes.byref(bad)) < 0
    assert b"Ho/Wo" in lib.dt_last_error()
    ok = _lib.ConvDesc(2, 64, 64, 64, 0, 0, 64, 64, 64, 3, 1, 1, 0, 0)
    assert lib.dt_conv2d_stat_rows(ctypes.byref(ok)) == 2 * 8 * 2
    assert lib.dt_conv2d_wgrad_workspace(ctypes.byref(ok)) > 0
    odd = _lib.ConvDesc(2, 64, 64, 64, 0, 0, 64, 64, 64, 5, 1, 2, 0, 0)
    assert lib.dt_conv2d_wgrad_workspace(ctypes.byref(odd)) == 0


def test_public_header_is_plain_c99(tmp_path):
    """include/deadtrees_hip.h is the drop-in boundary: it must compile as C (no C++ / torch types) with gcc."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    src = tmp_path / "t.c"
    src.write_text('#include "deadtrees_hip.h"\nint main(void) { dt_conv_desc d; dt_bn_bwd_fuse f; (void)d; (void)f; '
                   'return (int)sizeof(d); }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src), "-o",
                        str(tmp_path / "t.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
