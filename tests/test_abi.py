"""The C-ABI shared library loads on a machine without a GPU, exports every symbol that
include/mlst.h declares, and refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import __graft_entry__ as ge
from metamlst_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    ge.build()
    return engine.load_library()


def declared_functions(header="mlst.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mlst_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libmlst_hip.so does not export " + n
    # diagnostics and test hooks live in their own header (include/mlst_debug.h): exported and bound as well, but no part of
    # the drop-in boundary
    debug = declared_functions("mlst_debug.h")
    assert debug and not set(debug) & set(names)
    assert all(n.startswith(("mlst_selftest_", "mlst_debug_", "mlst_get_route_trace")) for n in debug), debug
    for n in debug:
        assert hasattr(lib, n), "libmlst_hip.so does not export " + n
    assert set(names) | set(debug) == set(lib._mlst_symbols), "engine.py binds a different set than the headers declare"


def test_struct_layouts_match_header():
    assert C.sizeof(engine.MlstParams) == 16 * 4 + 2 * 8 + 3 * 8
    assert C.sizeof(engine.MlstItem) == 24


def test_default_params_match_policy(lib):
    p = engine.MlstParams()
    lib.mlst_default_params(C.byref(p))
    q = engine.default_params()
    for f, _ in engine.MlstParams._fields_:
        assert getattr(p, f) == getattr(q, f), f


def test_no_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.mlst_create(0, None, C.byref(h))
    assert rc == -2 and not h.value                                   # MLST_E_NOGPU
    assert b"no CPU path" in lib.mlst_last_error(None)
    with pytest.raises(engine.MlstError):
        engine.Engine(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "metamlst_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and "mlst_oracle" not in txt, f
