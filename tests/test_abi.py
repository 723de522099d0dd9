"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/zhilight_amd.h declares (no compute: there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(experimental=False):
    src = open(os.path.join(ROOT, "include", "zhilight_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    if not experimental:                                    # entry points of a ZL_BUILD_EXPERIMENTAL=1 build only
        src = re.sub(r"#ifdef ZL_EXPERIMENTAL.*?#endif", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zl_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from zhilight_amd import _lib, build
    build.build()
    _lib.lib()
    names = _declared(_lib.experimental)
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.SO_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS + (_lib.EXPERIMENTAL_SYMBOLS if _lib.experimental else [])) == names
    # the default library carries neither the engine nor the digit-plane entry points
    assert sorted(set(_declared(True)) - set(_declared(False))) == sorted(_lib.EXPERIMENTAL_SYMBOLS)


def test_version_and_status_strings_without_gpu():
    from zhilight_amd import _lib
    l = _lib.lib()
    assert l.zl_version() == 100
    for code in (0, -1, -2, -3, -4):
        assert len(l.zl_status_string(ctypes.c_int(code))) > 0


def test_layout_arithmetic_host_only():
    from zhilight_amd._lib import W4Layout, lib
    L = W4Layout()
    assert lib().zl_w4_layout(ctypes.c_int64(4096), ctypes.c_int64(4096), ctypes.c_int64(128), ctypes.byref(L)) == 0
    assert (L.np, L.kp, L.q, L.c) == (4096, 4096, 4, 2)
    # exactly the canonical GPTQ bytes: 0.5 B/weight + 2 B scale and 0.5 B zero per 128-group
    assert L.qw_bytes + L.scales_bytes + L.zeros_bytes == int(4096 * 4096 * 0.51953125)
    assert lib().zl_w4_layout(ctypes.c_int64(4096), ctypes.c_int64(14336), ctypes.c_int64(128), ctypes.byref(L)) == 0
    assert (L.kp, L.q) == (14336, 14)
    assert lib().zl_w4_layout(ctypes.c_int64(5), ctypes.c_int64(2304), ctypes.c_int64(128), ctypes.byref(L)) == 0
    assert (L.np, L.kp, L.q) == (6, 3072, 3)
    assert lib().zl_w4_layout(ctypes.c_int64(8), ctypes.c_int64(1000), ctypes.c_int64(128), ctypes.byref(L)) == -2
    assert lib().zl_w4_layout(ctypes.c_int64(0), ctypes.c_int64(1024), ctypes.c_int64(128), ctypes.byref(L)) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import importlib
    from zhilight_amd import _lib
    monkeypatch.setattr(_lib, "SO_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    import pytest
    with pytest.raises(ImportError, match="no CPU/PyTorch fallback"):
        _lib.lib()
    importlib.reload(_lib)


def test_comm_library_exports_every_declared_symbol():
    """libzhilight_amd_comm.so (include/zhilight_amd_comm.h): loads without a GPU, exports every declared entry point"""
    import re
    from zhilight_amd import _lib
    lib = _lib.comm_lib()
    hdr = open(os.path.join(ROOT, "include", "zhilight_amd_comm.h")).read()
    declared = set(re.findall(r"\b(zl_(?:comm|ar)_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.COMM_SYMBOLS), declared ^ set(_lib.COMM_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
