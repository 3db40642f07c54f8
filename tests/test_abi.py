"""CPU: the C-ABI library loads and exports every symbol include/d3feat_hip.h declares; the ctypes table mirrors it."""
import ctypes
import os
import re

import d3feat_pytorch_amd
from d3feat_pytorch_amd import _native

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "d3feat_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(d3f_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libd3feat_hip.so does not export %s" % n


def test_ctypes_table_covers_header():
    assert sorted(_native.SIGNATURES) == _declared()


def test_no_gpu_calls_needed_for_size_queries():
    lib = _native.lib()
    assert lib.d3f_version().decode().startswith("d3feat-hip")
    assert lib.d3f_radius_grid_ws_bytes(1000) > 0
    assert lib.d3f_grid_subsample_ws_bytes(1000, 2) > 0
    assert lib.d3f_kpconv_ws_bytes(100, 100, 40, 15, 32, 32) >= 2 * 100 * 15 * 32 * 4
    assert lib.d3f_circle_det_loss_stats_floats(128) == 6 * 128
