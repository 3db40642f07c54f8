"""ctypes binding of ``libd3feat_hip.so`` (the C ABI declared in ``include/d3feat_hip.h``).

There is deliberately NO fallback: if the HIP library is missing or a tensor is not a CUDA/HIP tensor, every
operator raises ``RuntimeError`` (the reference's only error type, cpp_neighbors/wrapper.cpp:77).
"""
import ctypes as C
import os
import subprocess

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_PKG_DIR)
LIB_PATH = os.path.join(_PKG_DIR, "libd3feat_hip.so")
CSRC = os.path.join(_PKG_DIR, "csrc")
SOURCES = ["radius_neighbors.hip", "grid_subsample.hip", "kpconv.hip", "kpconv_fused.hip", "kpconv_aggregate.hip", "kpconv_small.hip", "kpconv_deform.hip", "pool.hip", "detection.hip", "loss.hip",
           "reverse_table.hip", "kpconv_dx_gather.hip", "matching.hip", "elementwise.hip", "batchnorm.hip", "linear.hip", "gemm_epilogue.hip", "optimizer.hip", "misc.hip"]

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class Tunables(C.Structure):
    """d3f_tunables of include/d3feat_hip.h: the library's knobs (it never reads the environment)."""
    _fields_ = [("atb_task_us", C.c_int32), ("atb_form", C.c_int32), ("atb_first_form_wgs", C.c_int32),
                ("match_wgs", C.c_int32), ("agg_through_lds", C.c_int32), ("atb_pipe", C.c_int32), ("xw_rows", C.c_int32), ("xw_split", C.c_int32),
                ("rowgemm_wide", C.c_int32), ("rowgemm_rt", C.c_int32), ("reserved", C.c_int32 * 6)]


class AtbProblem(C.Structure):
    """d3f_atb_problem of include/d3feat_hip.h: one queued weight gradient grad_w [Cout, ldw] = grad_out^T x."""
    _fields_ = [("x", _vp), ("grad_out", _vp), ("grad_w", _vp), ("N", C.c_int32), ("Cin", C.c_int32),
                ("Cout", C.c_int32), ("ldw", C.c_int32), ("bias_part", _vp), ("bias_blocks", C.c_int32),
                ("bias_cols", C.c_int32), ("grad_bias", _vp), ("grad_bias2", _vp)]


# name -> (restype, argtypes); mirrors include/d3feat_hip.h one to one
SIGNATURES = {
    "d3f_version": (C.c_char_p, []),
    "d3f_device_arch_ok": (_i, []),
    "d3f_device_arch_name": (_i, [C.c_char_p, _i]),
    "d3f_debug_set_flags": (None, [_i]),
    "d3f_get_tunables": (None, [C.POINTER(Tunables)]),
    "d3f_set_tunables": (_i, [C.POINTER(Tunables)]),
    "d3f_linear_grad_weight_group_ws_bytes": (_sz, [C.POINTER(AtbProblem), _i]),
    "d3f_linear_grad_weight_group": (_i, [C.POINTER(AtbProblem), _i, _vp, _sz, _vp]),
    "d3f_debug_set_phase_clock": (None, [_vp]),
    "d3f_debug_kernel_timing_begin": (_i, [_i, _i]),
    "d3f_debug_kernel_timing_end": (_i, [_vp, _vp, _i]),
    "d3f_radius_grid_ws_bytes": (_sz, [_i]),
    "d3f_radius_grid_build": (_i, [_vp, _i, _vp, _i, _f, _vp, _sz, _vp, _vp]),
    "d3f_radius_query": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "d3f_radius_query_ex": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "d3f_radius_grid_zero_bytes": (_sz, [_i]),
    "d3f_zero_buffers": (_i, [_vp, _vp, _i, _vp]),
    "d3f_copy_buffers": (_i, [_vp, _vp, _vp, _vp, _i, C.c_double, _vp]),
    "d3f_radius_grid_build_prezeroed": (_i, [_vp, _i, _vp, _i, _f, _vp, _sz, _vp, _vp]),
    "d3f_radius_query_prefix": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _f, _f, _f, _f, _i, _vp, _vp, _vp]),
    "d3f_radius_query_prefix_missing": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _f, _f, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "d3f_radius_query_pool_transposed": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _f, _f, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                              _vp]),
    "d3f_upsample_rows_rank": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "d3f_grid_subsample_ws_bytes": (_sz, [_i, _i]),
    "d3f_grid_subsample": (_i, [_vp, _i, _vp, _i, _f, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "d3f_grid_subsample_ex": (_i, [_vp, _i, _vp, _i, _f, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz,
                                   _vp, _vp]),
    "d3f_kpconv_deform_aggregate": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _f, _f, _i, _vp, _vp, _vp, _vp,
                                         _vp]),
    "d3f_kpconv_deform_grad": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "d3f_bias_act_packs": (_i, [_i]),
    "d3f_bias_act_forward_pack": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "d3f_batchnorm_ws_bytes": (_sz, [_i, _i]),
    "d3f_batchnorm_forward": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d3f_batchnorm_backward": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d3f_kpconv_ws_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "d3f_kpconv_forward": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp,
                                _vp, _sz, _vp]),
    "d3f_kpconv_packs_supports": (_i, [_i, _i, _i, _i, _i]),
    "d3f_kpconv_saves_wf": (_i, [_i, _i, _i, _i]),
    "d3f_kpconv_backward": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _i,
                                 _vp, _vp, _vp, _sz, _vp]),
    "d3f_kpconv_grad_input_supported": (_i, [_i, _i, _i, _i]),
    "d3f_kpconv_aggregate": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d3f_kpconv_aggregate_modes": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _f, _i, _vp, _vp, _vp]),
    "d3f_kpconv_grad_input_modes": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _f, _i, _vp, _vp, _vp]),
    "d3f_kpconv_grad_input": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _f, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "d3f_kpconv_aggregate_transposed_supported": (_i, [_i, _i]),
    "d3f_kpconv_aggregate_transposed": (_i, [_vp, _i, _i, _i, _vp, _i, _f, _vp, _vp, _i, _vp, _vp]),
    "d3f_reverse_table_ws_bytes": (_sz, [_i, _i, _i]),
    "d3f_reverse_table_build": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "d3f_kpconv_grad_input_gather_supported": (_i, [_i, _i, _i]),
    "d3f_kpconv_grad_input_gather": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _f, _vp, _vp, _i, _vp, _i, _i, _f, _vp, _vp, _vp,
                                          _vp, _vp]),
    "d3f_reverse_table_filter": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _f, _vp, _vp, _vp]),
    "d3f_linear_grad_weight_supported": (_i, [_i, _i, _i]),
    "d3f_linear_fused_supported": (_i, [_i, _i, _i]),
    "d3f_linear_bias_act_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp]),
    "d3f_linear_pair_supported": (_i, [_i, _i, _i, _i]),
    "d3f_linear_pair_bias_act_forward": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp]),
    "d3f_linear_grad_input": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "d3f_linear_grad_weight_ws_bytes": (_sz, [_i, _i, _i]),
    "d3f_linear_grad_weight": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "d3f_linear_grad_weight_bias": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp, _i, _i, _vp, _vp, _vp]),
    "d3f_permute_kpconv_weights": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "d3f_gemm_epilogue_supported": (_i, [_i, _i, _i, _i, _i]),
    "d3f_gemm_epilogue_ws_bytes": (_sz, [_i, _i, _i]),
    "d3f_gemm_epilogue": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _f, _vp, _i, _vp, _i, _vp, _sz,
                               _vp]),
    "d3f_bias_act_backward_blocks": (_i, [_i, _i]),
    "d3f_bias_act_backward_partial": (_i, [_vp, _vp, _f, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "d3f_bias_sum": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "d3f_max_pool_forward": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "d3f_max_pool_backward": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "d3f_closest_pool_forward": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "d3f_closest_pool_backward": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "d3f_bias_act_forward": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "d3f_bias_act_backward_ws_bytes": (_sz, [_i, _i]),
    "d3f_bias_act_backward": (_i, [_vp, _vp, _f, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "d3f_global_max": (_i, [_vp, _sz, _vp, _vp, _sz, _vp]),
    "d3f_global_max_rows": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _sz, _vp]),
    "d3f_global_max_groups": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "d3f_detection_scores_aux_floats": (_i, [_i]),
    "d3f_detection_scores_forward": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "d3f_detection_scores_ws_bytes": (_sz, [_i, _i]),
    "d3f_detection_scores_backward": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d3f_detection_scores_backward_groups": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "d3f_circle_det_loss_forward_pairs": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp,
                                               _vp, _vp, _vp, _vp]),
    "d3f_circle_det_loss_backward_pairs": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _f, _f, _f, _f, _f, _f, _vp, _vp,
                                                _vp, _vp, _vp, _vp, _vp, _vp]),
    "d3f_select_normalize_forward_pairs": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "d3f_select_normalize_backward_pairs": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "d3f_circle_det_loss_stats_floats": (_sz, [_i]),
    "d3f_circle_det_loss_ws_bytes": (_sz, [_i]),
    "d3f_circle_det_loss_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp,
                                         _vp]),
    "d3f_circle_det_loss_backward": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _vp, _vp, _vp, _sz, _vp]),
    "d3f_select_normalize_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "d3f_select_normalize_backward": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "d3f_mutual_nn_ws_bytes": (_sz, [_i, _i]),
    "d3f_mutual_nn": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d3f_mutual_nn_batched_ws_bytes": (_sz, [_i, _i]),
    "d3f_mutual_nn_batched": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "d3f_topk_scores": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp]),
    "d3f_sgd_guarded_step": (_i, [_vp, _vp, _vp, _sz, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "d3f_sgd_guarded_step_lanes": (_i, [_vp, _i, _vp, _vp, _sz, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "d3f_poison_gradient_if_status": (_i, [_vp, _vp, _vp, _vp]),
}

ERRORS = {-1: "invalid argument", -2: "workspace too small", -3: "HIP launch failure"}
STATUS_BITS = {1: "a query has more in-radius candidates than the kernel can rank (512)",
               2: "a point lies outside the addressable cell grid",
               4: "voxel hash table full",
               8: "a pyramid level needs more rows than its capacity (raise the capacities)",
               16: "a point has more in-radius neighbors than the reverse (wide) table holds",
               32: "an upsampling query has no coarse point within the guaranteed nearest bound (a coarse level lost "
                   "voxels, or the bound passed to query_prefix is wrong)"}


def build(verbose=False, jobs=None, extra_flags=(), out=None, objdir=None):
    """Compile every HIP source for gfx950 into the in-tree shared library (cross-compiles without a GPU).
    One object per source under ``build/`` (recompiled only when the source or a header changed), then one link.
    ``extra_flags`` / ``out`` / ``objdir``: a MEASUREMENT build of the same sources next to the product (e.g.
    -DD3F_ATB_PROBE=1 into profiles/experiments/, profiles/atb_loop_probe.py); the product build takes neither."""
    # -ffp-contract=off: HIP's __fmul_rn/__fadd_rn are plain operators on AMD, so with the default contract=fast
    # the compiler fuses the "exact" d2 = (dx*dx + dy*dy) + dz*dz into FMAs and the strict d2 < r2 test / the
    # distance order stop matching the reference's x86 arithmetic bit for bit.  Fusion is requested explicitly
    # (fmaf / MFMA) where it is wanted.
    from concurrent.futures import ThreadPoolExecutor
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-pass-failed", "-fPIC",
             "-I" + os.path.join(_REPO, "include")] + list(extra_flags)
    objdir = objdir or os.path.join(_PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(_REPO, "include", "d3feat_hip.h"))
    hdr_time = max(os.path.getmtime(h) for h in headers)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time):
            return obj
        cmd = ["hipcc"] + flags + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out or LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out or LIB_PATH


def _needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_REPO, "include", "d3feat_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


_lib = None


def lib():
    """The loaded library; raises RuntimeError if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        try:  # bring up torch's HIP context first: the library shares torch's bundled HIP runtime
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:  # pragma: no cover
            pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libd3feat_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  d3feat_pytorch_amd has no CPU or PyTorch fallback.")
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise RuntimeError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


TRACE = False     # debugging: name every C-ABI call on stderr and fence it (set _native.TRACE = True)


def check(rc, what):
    if TRACE:  # debugging aid: name every C-ABI call and fence it, so a device fault can be attributed
        import sys
        import torch
        sys.stderr.write("[d3f] %s\n" % what)
        sys.stderr.flush()
        torch.cuda.synchronize()
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, ERRORS.get(rc, "error %d" % rc)))


def set_tunables(**kw):
    """Change fields of the library's d3f_tunables (experiments / tests); returns the previous values as a dict."""
    L = lib()
    t = Tunables()
    L.d3f_get_tunables(C.byref(t))
    old = {name: getattr(t, name) for name, _ in Tunables._fields_ if name != "reserved"}
    for k, v in kw.items():
        if k not in old:
            raise RuntimeError("d3f_tunables has no field %r" % k)
        setattr(t, k, int(v))
    check(L.d3f_set_tunables(C.byref(t)), "d3f_set_tunables")
    return old


def status_message(word):
    return "; ".join(msg for bit, msg in STATUS_BITS.items() if word & bit)
