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


def test_tunables_struct_round_trips_and_rejects_nonsense():
    """d3f_get_tunables / d3f_set_tunables: the library's only knobs (it reads no environment); defaults are all zero."""
    lib = _native.lib()
    t = _native.Tunables()
    lib.d3f_get_tunables(ctypes.byref(t))
    names = [n for n, _ in _native.Tunables._fields_ if n != "reserved"]
    assert names == ["atb_task_us", "atb_form", "atb_first_form_wgs", "match_wgs", "agg_through_lds", "atb_pipe",
                     "xw_rows", "xw_split", "rowgemm_wide", "rowgemm_rt"]
    assert [getattr(t, n) for n in names] == [0] * len(names)
    assert ctypes.sizeof(t) == 64
    old = _native.set_tunables(atb_task_us=33, atb_form=2)
    lib.d3f_get_tunables(ctypes.byref(t))
    assert (t.atb_task_us, t.atb_form) == (33, 2)
    _native.set_tunables(**old)
    bad = _native.Tunables()
    bad.atb_form = 7
    assert lib.d3f_set_tunables(ctypes.byref(bad)) == -1
    for field, value in (("xw_rows", 5), ("xw_split", 65), ("rowgemm_wide", 3), ("rowgemm_rt", 2)):
        bad = _native.Tunables()
        setattr(bad, field, value)
        assert lib.d3f_set_tunables(ctypes.byref(bad)) == -1, field
    assert lib.d3f_set_tunables(None) == -1
    lib.d3f_get_tunables(ctypes.byref(t))
    assert (t.atb_task_us, t.atb_form) == (0, 0)


def test_gemm_epilogue_support_and_workspace_are_host_computations():
    """d3f_gemm_epilogue_supported / _ws_bytes need no GPU: multiples of 16, the block form only for reduction-contiguous
    weights in blocks of 64; a split reduction (few rows against a long reduction) asks for its slabs."""
    lib = _native.lib()
    assert lib.d3f_gemm_epilogue_supported(512, 7680, 512, 1, 0) == 1
    assert lib.d3f_gemm_epilogue_supported(1792, 3840, 256, 0, 256) == 1
    assert lib.d3f_gemm_epilogue_supported(1792, 3840, 256, 1, 256) == 0
    assert lib.d3f_gemm_epilogue_supported(1792, 3840, 250, 0, 0) == 0
    assert lib.d3f_gemm_epilogue_supported(0, 64, 64, 0, 0) == 0
    assert lib.d3f_gemm_epilogue_ws_bytes(512, 7680, 512) >= 8 * 512 * 512 * 4       # 8 partitions of the reduction
    assert lib.d3f_gemm_epilogue_ws_bytes(114624, 128, 32) == 256                      # undivided: no slabs
    old = _native.set_tunables(xw_split=1)
    try:
        assert lib.d3f_gemm_epilogue_ws_bytes(512, 7680, 512) == 256
    finally:
        _native.set_tunables(**old)


def test_grouped_weight_gradient_plan_is_a_host_computation():
    """d3f_linear_grad_weight_group_ws_bytes needs no GPU: slab bytes of a queue; few rows against a large aligned target
    take no slab at all (undivided reduction written straight to the target); unsupported widths give 0."""
    lib = _native.lib()
    arr = (_native.AtbProblem * 2)()
    for q, (n, cin, cout) in zip(arr, ((114688, 32, 128), (512, 512, 7680))):
        q.x, q.grad_out, q.grad_w = 0x10000, 0x20000, 0x30000       # (addresses are only checked for alignment here)
        q.N, q.Cin, q.Cout, q.ldw = n, cin, cout, cin
    both = lib.d3f_linear_grad_weight_group_ws_bytes(arr, 2)
    first = lib.d3f_linear_grad_weight_group_ws_bytes(arr, 1)
    assert first > 256 and (first - 256) % (128 * 32 * 4) == 0     # whole slabs of the 128 x 32 gradient
    assert both == first                                            # the direct problem adds nothing
    arr[1].grad_w = 0x30004                                         # not 16-byte aligned: partitions + slabs after all
    assert lib.d3f_linear_grad_weight_group_ws_bytes(arr, 2) >= first + 8 * 7680 * 512 * 4
    arr[1].Cin = 24
    assert lib.d3f_linear_grad_weight_group_ws_bytes(arr, 2) == 0
    assert lib.d3f_linear_grad_weight_group_ws_bytes(None, 2) == 0
