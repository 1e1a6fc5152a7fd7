"""CPU: the product library builds, loads, exports every symbol include/cc_hip.h declares, and refuses to run
without a GPU (no CPU fallback). No compute is called here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from continuous_clustering_amd import build, load_library
    build.build()
    return load_library()


def declared_functions():
    names = set()
    for header in ("cc_hip.h", "cc_kitti.h"):
        txt = open(os.path.join(ROOT, "include", header)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(cc_[a-z_0-9]+)\s*\(", txt))
    return sorted(names)


def test_header_symbols_are_exported(lib):
    names = declared_functions()
    assert len(names) >= 30 and "cc_kitti_convert_frames" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported by libcc_hip.so"


def test_struct_layouts_match_header(lib):
    from continuous_clustering_amd import capi
    # cc_config: 33 four-byte scalars; cc_event: 40 bytes; cc_stream_state: 4*4 + 4*8 + 4*8 + 2*4 + 2*8
    assert ctypes.sizeof(capi.Config) == 33 * 4
    assert ctypes.sizeof(capi.Event) == 40
    assert ctypes.sizeof(capi.StreamState) == 16 + 32 + 32 + 8 + 16
    assert ctypes.sizeof(capi.ColumnView) == 22 * 8  # 14 fields of ABI 0.1 + the 8 clustering fields the ROS packers read
    c = capi.Config()
    lib.cc_config_default(ctypes.byref(c))
    assert bytes(c) == bytes(capi.Config.default())
    lib.cc_config_kitti(ctypes.byref(c))
    assert bytes(c) == bytes(capi.Config.kitti())
    assert b"gfx950" in lib.cc_version()


def test_no_gpu_means_no_engine(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from continuous_clustering_amd import Engine, EngineError, capi
    with pytest.raises(EngineError) as ei:
        Engine(capi.Config.kitti(), 64)
    assert ei.value.code == capi.CC_ERR_NO_DEVICE


def test_product_does_not_reference_the_oracle():
    """The shipped package must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "continuous_clustering_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "cc_oracle" not in txt and "libcc_oracle" not in txt, f
