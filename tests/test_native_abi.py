"""CPU-side checks of the C ABI: the shared library loads and exports every function that
include/diffsci_hip.h declares, and the ctypes binding covers exactly that set.  No kernel is
launched (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "diffsci_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ds_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    import build
    return build.build(force=False, verbose=False)


def test_header_declares_functions():
    names = header_functions()
    assert "ds_karras_heun" in names and "ds_conv2d" in names and len(names) >= 20


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for name in header_functions():
        assert hasattr(lib, name), f"{name} declared in include/diffsci_hip.h but not exported"


def test_ctypes_binding_matches_header(built_lib):
    from diffsci_amd import _native
    assert _native.exported_symbols() == header_functions()
    L = _native.lib()
    assert L.ds_version() == 2
    assert L.ds_last_error() is not None


def test_struct_layout_matches_header():
    from diffsci_amd._native import EvalCoef
    src = open(os.path.join(ROOT, "include", "diffsci_hip.h")).read()
    body = re.search(r"typedef struct ds_eval_coef \{(.*?)\} ds_eval_coef;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(float|int)\s+([a-z_]+)\s*;", body)
    assert [n for _, n in fields] == [n for n, _ in EvalCoef._fields_]
    assert ctypes.sizeof(EvalCoef) == 4 * len(fields)


def test_packed_weight_size_is_host_side(built_lib):
    from diffsci_amd import _native
    L = _native.lib()
    assert L.ds_conv2d_packed_floats(64, 64, 3) == 1 * 8 * 9 * 8 * 64
    assert L.ds_conv2d_packed_floats(1, 64, 3) == 1 * 8 * 9 * 8 * 64        # Cout padded to 64
    assert L.ds_conv2d_packed_floats(768, 256, 1) == 12 * 8 * 1 * 32 * 64
    assert L.ds_conv2d_packed_floats(8, 8, 5) == 0                          # unsupported kernel size


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from diffsci_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeLibraryError, match="no CPU or PyTorch fallback"):
        _native.lib()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under diffsci_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "diffsci_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), os.path.join(dirpath, f)
