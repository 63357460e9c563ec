"""The C-ABI libraries load without a GPU and export every symbol their headers
declare (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from raytracer_amd import _ffi
from raytracer_amd._ffi import RtRenderChunk, RtSceneDesc, RtSettings

INCLUDE = os.path.join(_ffi.REPO_ROOT, "include")


def declared_functions(header):
    text = open(os.path.join(INCLUDE, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_rt_hip_exports_every_declared_symbol():
    names = declared_functions("rt_hip.h")
    assert len(names) >= 15
    lib = C.CDLL(_ffi.hip_library_path())
    for n in names:
        assert hasattr(lib, n), f"librt_hip.so does not export {n}"
    # and the Python prototypes cover the header exactly
    assert sorted(_ffi.HIP_PROTOTYPES) == names


def test_rt_samples_exports_every_declared_symbol():
    names = declared_functions("rt_samples.h")
    lib = C.CDLL(_ffi.samples_library_path())
    for n in names:
        assert hasattr(lib, n), f"librt_samples.so does not export {n}"
    assert sorted(_ffi.SAMPLES_PROTOTYPES) == names


def test_no_oracle_in_product():
    """The product never links or imports the oracle (it is test infrastructure)."""
    import subprocess
    for path in (_ffi.hip_library_path(), _ffi.samples_library_path()):
        out = subprocess.run(["nm", "-D", path], capture_output=True, text=True).stdout
        assert "orc_" not in out
        needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
        assert "oracle" not in needed
    pkg = os.path.join(_ffi.REPO_ROOT, "raytracer_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "rt_oracle" not in src and "tests.oracle" not in src and "import oracle" not in src, f


def test_error_reporting_without_gpu():
    lib = _ffi.load_hip()
    assert lib.rt_scene_create(None, 0) is None
    assert b"null description" in lib.rt_last_error()
    bad = RtSceneDesc()
    bad.abi_version = 99
    assert lib.rt_scene_create(C.byref(bad), 0) is None
    assert b"abi_version" in lib.rt_last_error()
    st = RtSettings(10, 10, 5, 1)
    ch = RtRenderChunk(0, 0, 10, 10)
    assert lib.rt_render_chunk(None, C.byref(st), C.byref(ch), 1, None, None, None) == _ffi.RT_E_INVALID
    out = (RtRenderChunk * 4)()
    assert lib.rt_create_render_chunks(10, 10, 0, 3, out, 4) == _ffi.RT_E_INVALID      # zero count
    assert lib.rt_create_render_chunks(10, 10, 3, 3, out, 4) == _ffi.RT_E_INVALID      # capacity too small
