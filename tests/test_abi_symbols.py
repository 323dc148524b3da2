"""CPU-side checks of the C-ABI: the library loads, exports every symbol include/vfik.h declares, and
refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from vfclik_amd import engine
    return engine.load_library()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "vfik.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(vfik_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 20
    raw = ctypes.CDLL(os.path.join(ROOT, "vfclik_amd", "csrc", "libvfik_hip.so"))
    for n in names:
        assert hasattr(raw, n), "libvfik_hip.so does not export " + n


def test_struct_sizes_match_header(lib):
    from vfclik_amd import _abi
    assert ctypes.sizeof(_abi.Field) == 152 == _abi.FIELD_DTYPE.itemsize
    assert ctypes.sizeof(_abi.Chain) == 4 + 4 * 16 + 4 + 8 * 12 * 17 + 8 * 16 * 2  # n, jtype, pad, B, limits
    assert ctypes.sizeof(_abi.Params) == 8 * 7 + 8 * 6 + 8 * 16 + 8 * 6 + 8 + 16
    from vfclik_amd import engine
    sizes = (ctypes.c_size_t * 4)()
    lib.vfik_struct_sizes(sizes)  # what the C compiler made of include/vfik_types.h
    assert list(sizes) == [ctypes.sizeof(_abi.Field), ctypes.sizeof(_abi.Chain), ctypes.sizeof(_abi.Params), ctypes.sizeof(engine.IO)]


def test_supported_joints(lib):
    m = lib.vfik_supported_joints()
    for n in (6, 7, 10, 14):
        assert (m >> n) & 1


def test_no_cpu_fallback(lib):
    """Without a GPU vfik_create must fail with a message, and the Python engine must raise."""
    if lib.vfik_device_count() > 0:
        pytest.skip("a GPU is visible")
    h = lib.vfik_create(0, 32, 7, 8, 64)
    assert not h
    assert b"no CPU path" in lib.vfik_last_error()
    from vfclik_amd import engine, robots
    with pytest.raises(engine.VfikError):
        engine.Engine(robots.lwr(), 64)


def test_create_rejects_bad_arguments(lib):
    assert not lib.vfik_create(0, 16, 7, 8, 64)
    assert b"io_dtype" in lib.vfik_last_error()
    assert not lib.vfik_create(0, 32, 5, 8, 64)
    assert b"no kernel built for 5 joints" in lib.vfik_last_error()
    assert not lib.vfik_create(0, 32, 7, 8, 0)


def test_missing_library_is_loud(tmp_path):
    from vfclik_amd import engine
    with pytest.raises(engine.VfikError):
        engine.load_library(str(tmp_path / "nope.so"))
