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
    assert L.ds_version() == _native.ABI_VERSION == 4
    assert L.ds_last_error() is not None


def test_struct_layout_matches_header():
    from diffsci_amd._native import EvalCoef
    src = open(os.path.join(ROOT, "include", "diffsci_hip.h")).read()
    body = re.search(r"typedef struct ds_eval_coef \{(.*?)\} ds_eval_coef;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(float|int|uint32_t\*)\s+([a-z_]+)\s*;", body)
    assert [n for _, n in fields] == [n for n, _ in EvalCoef._fields_]
    kinds = {"float": ctypes.c_float, "int": ctypes.c_int, "uint32_t*": ctypes.c_void_p}
    assert [kinds[k] for k, _ in fields] == [t for _, t in EvalCoef._fields_]
    assert ctypes.sizeof(EvalCoef) == 4 * (len(fields) - 1) + 8 and EvalCoef.nonfinite.offset == 4 * (len(fields) - 1)


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


def test_image_layout_helpers_are_host_side(built_lib):
    """Sizes and shape predicates of the pre-split image route: plain host arithmetic, callable without a GPU."""
    import ctypes
    lib = ctypes.CDLL(str(built_lib))
    lib.ds_conv_images_bytes.restype = ctypes.c_size_t
    lib.ds_conv_images_bytes.argtypes = [ctypes.c_int] * 4
    for B, C, H, W in ((1, 1, 1, 1), (2, 16, 8, 8), (3, 40, 17, 33), (64, 256, 32, 32)):
        chunks = (C + 15) // 16
        # [b][chunk][piece 2][half 2][H+2][W+2] vectors of 16 bytes
        assert lib.ds_conv_images_bytes(B, C, H, W) == B * chunks * 4 * (H + 2) * (W + 2) * 16
    assert lib.ds_conv_images_bytes(0, 16, 8, 8) == 0
    assert lib.ds_inorm_silu_images_supported(64, 64) == 1 and lib.ds_inorm_silu_images_supported(128, 64) == 0
    assert lib.ds_inorm_silu_images_supported(3, 3) == 0                      # H*W must be a multiple of 4
    assert lib.ds_conv2d_h3_up_supported(8, 32) == 1 and lib.ds_conv2d_h3_up_supported(16, 48) == 1
    assert lib.ds_conv2d_h3_up_supported(12, 20) == 0
