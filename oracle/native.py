"""ctypes front-end of the CPU checkers -- TEST INFRASTRUCTURE ONLY.

Two libraries, both built by ``oracle/Makefile``:

* ``oracle/libd3f_oracle.so``  -- our C++ restatement (``native_oracle.cpp``); travels everywhere.
* ``oracle/_ref/libd3f_ref.so`` -- the reference's own C++ compiled in place from ``/root/reference``
  (``ref_shim.cpp``); git-ignored, exists wherever it was prebuilt (it is shipped to the GPU box by gpurun).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Function names follow the reference's Python-visible native API
(``radius_neighbors.batch_query`` cpp_wrappers/cpp_neighbors/wrapper.cpp:27,71-75;
``grid_subsampling.subsample_batch`` cpp_wrappers/cpp_subsampling/wrapper.cpp:30,75-82).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "libd3f_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libd3f_ref.so")
_REFERENCE_ROOT = "/root/reference"

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)
_i64p = C.POINTER(C.c_int64)


def build(ref=True):
    """(Re)build the checker libraries.  ``ref`` is attempted only when /root/reference is mounted."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    if ref and os.path.isdir(_REFERENCE_ROOT):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_oracle = None
_ref = None


def _lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(_ORACLE_SO):
            build(ref=False)
        _oracle = C.CDLL(_ORACLE_SO)
        _oracle.orc_radius_neighbors.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _i32p, _i32p, C.c_int, C.c_float,
                                                 C.c_int, C.c_int, C.POINTER(_i32p), C.POINTER(_f32p), _i32p]
        _oracle.orc_radius_counts.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _i32p, _i32p, C.c_int, C.c_float, _i32p]
        _oracle.orc_grid_subsample.argtypes = [_f32p, C.c_int, _i32p, C.c_int, C.c_float, C.c_int, _f32p, _i32p,
                                               _i32p, _i32p, _i64p]
        _oracle.orc_grid_subsample_ex.argtypes = [_f32p, C.c_int, _i32p, C.c_int, C.c_float, C.c_int, _f32p, C.c_int,
                                                  _i32p, C.c_int, _f32p, _i32p, _i32p, _i32p, _i64p, _f32p, _i32p]
        _oracle.orc_free.argtypes = [C.c_void_p]
    return _oracle


def have_ref():
    return os.path.exists(_REF_SO)


def _reflib():
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libd3f_ref.so not built (needs /root/reference; run `make -C oracle ref`)")
        _ref = C.CDLL(_REF_SO)
        _ref.ref_batch_query.argtypes = [_f32p, C.c_int, _f32p, C.c_int, _i32p, _i32p, C.c_int, C.c_float,
                                         C.POINTER(_i32p), _i32p]
        _ref.ref_subsample_batch.argtypes = [_f32p, C.c_int, _i32p, C.c_int, C.c_float, C.c_int, C.POINTER(_f32p),
                                             _i32p, _i32p]
        if hasattr(_ref, 'ref_subsample_batch_ex'):
            _ref.ref_subsample_batch_ex.argtypes = [_f32p, C.c_int, _i32p, C.c_int, C.c_float, C.c_int, _f32p, C.c_int,
                                                    _i32p, C.c_int, C.POINTER(_f32p), _i32p, _i32p, C.POINTER(_f32p),
                                                    C.POINTER(_i32p)]
        _ref.ref_free.argtypes = [C.c_void_p]
    return _ref


def _f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32).reshape(-1)


def _ptr(a, t):
    return a.ctypes.data_as(t)


def _check_clouds(queries, supports, q_batches, s_batches):
    # the reference's shape errors (cpp_neighbors/wrapper.cpp:127-171) are RuntimeError
    if queries.ndim != 2 or queries.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : query.shape is not (N, 3)")
    if supports.ndim != 2 or supports.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : support.shape is not (N, 3)")
    if q_batches.shape[0] != s_batches.shape[0]:
        raise RuntimeError("Wrong number of batch elements: different for queries and supports ")


def batch_query(queries, supports, q_batches, s_batches, radius=0.1, method="grid", max_neighbors=0,
                return_d2=False):
    """Restated ``radius_neighbors.batch_query``: int32 [Nq, width], rows (d2, idx)-ascending, pad = Ns."""
    q, s, qb, sb = _f32(queries), _f32(supports), _i32(q_batches), _i32(s_batches)
    _check_clouds(q, s, qb, sb)
    idx_p, d2_p, width = _i32p(), _f32p(), C.c_int(0)
    rc = _lib().orc_radius_neighbors(_ptr(q, _f32p), q.shape[0], _ptr(s, _f32p), s.shape[0], _ptr(qb, _i32p),
                                     _ptr(sb, _i32p), qb.shape[0], float(radius), 0 if method == "brute" else 1,
                                     int(max_neighbors), C.byref(idx_p), C.byref(d2_p) if return_d2 else None,
                                     C.byref(width))
    if rc != 0:
        raise RuntimeError("Error")  # cpp_neighbors/wrapper.cpp:201-205
    n = q.shape[0] * width.value
    idx = np.ctypeslib.as_array(idx_p, shape=(n,)).reshape(q.shape[0], width.value).copy()
    _lib().orc_free(idx_p)
    if not return_d2:
        return idx
    d2 = np.ctypeslib.as_array(d2_p, shape=(n,)).reshape(q.shape[0], width.value).copy()
    _lib().orc_free(d2_p)
    return idx, d2


def neighbor_counts(queries, supports, q_batches, s_batches, radius):
    q, s, qb, sb = _f32(queries), _f32(supports), _i32(q_batches), _i32(s_batches)
    out = np.zeros(q.shape[0], dtype=np.int32)
    _lib().orc_radius_counts(_ptr(q, _f32p), q.shape[0], _ptr(s, _f32p), s.shape[0], _ptr(qb, _i32p), _ptr(sb, _i32p),
                             qb.shape[0], float(radius), _ptr(out, _i32p))
    return out


def subsample_batch(points, batches, sampleDl=0.1, max_p=0, return_meta=False):
    """Restated ``grid_subsampling.subsample_batch`` (points only): (points f32 [N',3], batches i32 [B]).

    ``return_meta`` adds (first_index int32 [N'], cell_key int64 [N']) for permutation bookkeeping."""
    p, b = _f32(points), _i32(batches)
    if p.ndim != 2 or p.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")
    out = np.zeros((p.shape[0], 3), dtype=np.float32)
    first = np.zeros(p.shape[0], dtype=np.int32)
    key = np.zeros(p.shape[0], dtype=np.int64)
    ob = np.zeros(b.shape[0], dtype=np.int32)
    n = C.c_int(0)
    rc = _lib().orc_grid_subsample(_ptr(p, _f32p), p.shape[0], _ptr(b, _i32p), b.shape[0], float(sampleDl),
                                   int(max_p), _ptr(out, _f32p), C.byref(n), _ptr(ob, _i32p), _ptr(first, _i32p),
                                   _ptr(key, _i64p))
    if rc != 0:
        raise RuntimeError("Error")
    if return_meta:
        return out[:n.value].copy(), ob, first[:n.value].copy(), key[:n.value].copy()
    return out[:n.value].copy(), ob


def subsample_batch_ex(points, batches, features=None, classes=None, sampleDl=0.1, max_p=0):
    """Restated ``subsample_batch(points, batches, features=, classes=)``: (points, batches[, features][, classes]);
    classes come back as int32 [N', ldim] like the reference's (wrapper.cpp:282-310)."""
    p, b = _f32(points), _i32(batches)
    f = _f32(features) if features is not None else None
    c = _i32(classes).reshape(p.shape[0], -1) if classes is not None else None
    fd, ld = (f.shape[1] if f is not None else 0), (c.shape[1] if c is not None else 0)
    out = np.zeros((p.shape[0], 3), dtype=np.float32)
    of = np.zeros((p.shape[0], max(fd, 1)), dtype=np.float32)
    oc = np.zeros((p.shape[0], max(ld, 1)), dtype=np.int32)
    ob = np.zeros(b.shape[0], dtype=np.int32)
    n = C.c_int(0)
    rc = _lib().orc_grid_subsample_ex(_ptr(p, _f32p), p.shape[0], _ptr(b, _i32p), b.shape[0], float(sampleDl),
                                      int(max_p), _ptr(f, _f32p) if f is not None else None, fd,
                                      _ptr(c, _i32p) if c is not None else None, ld, _ptr(out, _f32p), C.byref(n),
                                      _ptr(ob, _i32p), None, None, _ptr(of, _f32p), _ptr(oc, _i32p))
    if rc != 0:
        raise RuntimeError("Error")
    res = [out[:n.value].copy(), ob]
    if f is not None:
        res.append(of[:n.value, :fd].copy())
    if c is not None:
        res.append(oc[:n.value, :ld].copy())
    return tuple(res)


# ---------------------------------------------------------------------------------------------
# the real reference (compiled in place); same call shapes
# ---------------------------------------------------------------------------------------------
def ref_batch_query(queries, supports, q_batches, s_batches, radius=0.1):
    q, s, qb, sb = _f32(queries), _f32(supports), _i32(q_batches), _i32(s_batches)
    _check_clouds(q, s, qb, sb)
    idx_p, width = _i32p(), C.c_int(0)
    rc = _reflib().ref_batch_query(_ptr(q, _f32p), q.shape[0], _ptr(s, _f32p), s.shape[0], _ptr(qb, _i32p),
                                   _ptr(sb, _i32p), qb.shape[0], float(radius), C.byref(idx_p), C.byref(width))
    if rc != 0:
        raise RuntimeError("Error")
    idx = np.ctypeslib.as_array(idx_p, shape=(q.shape[0] * width.value,)).reshape(q.shape[0], width.value).copy()
    _reflib().ref_free(idx_p)
    return idx


def ref_subsample_batch(points, batches, sampleDl=0.1, max_p=0):
    p, b = _f32(points), _i32(batches)
    out_p, n = _f32p(), C.c_int(0)
    ob = np.zeros(b.shape[0], dtype=np.int32)
    rc = _reflib().ref_subsample_batch(_ptr(p, _f32p), p.shape[0], _ptr(b, _i32p), b.shape[0], float(sampleDl),
                                       int(max_p), C.byref(out_p), C.byref(n), _ptr(ob, _i32p))
    if rc != 0:
        raise RuntimeError("Error")
    pts = np.ctypeslib.as_array(out_p, shape=(n.value * 3,)).reshape(n.value, 3).copy()
    _reflib().ref_free(out_p)
    return pts, ob


def ref_subsample_batch_ex(points, batches, features=None, classes=None, sampleDl=0.1, max_p=0):
    """The reference's batch_grid_subsampling with features / classes.  Only meaningful for ldim == 1 or one cloud
    (the reference slices the classes of later clouds wrongly otherwise, grid_subsampling.cpp:157-158)."""
    p, b = _f32(points), _i32(batches)
    f = _f32(features) if features is not None else None
    c = _i32(classes).reshape(p.shape[0], -1) if classes is not None else None
    fd, ld = (f.shape[1] if f is not None else 0), (c.shape[1] if c is not None else 0)
    out_p, out_f, out_c, n = _f32p(), _f32p(), _i32p(), C.c_int(0)
    ob = np.zeros(b.shape[0], dtype=np.int32)
    rc = _reflib().ref_subsample_batch_ex(_ptr(p, _f32p), p.shape[0], _ptr(b, _i32p), b.shape[0], float(sampleDl),
                                          int(max_p), _ptr(f, _f32p) if f is not None else None, fd,
                                          _ptr(c, _i32p) if c is not None else None, ld, C.byref(out_p), C.byref(n),
                                          _ptr(ob, _i32p), C.byref(out_f), C.byref(out_c))
    if rc != 0:
        raise RuntimeError("Error")
    res = [np.ctypeslib.as_array(out_p, shape=(n.value * 3,)).reshape(n.value, 3).copy(), ob]
    _reflib().ref_free(out_p)
    if f is not None:
        res.append(np.ctypeslib.as_array(out_f, shape=(n.value * fd,)).reshape(n.value, fd).copy())
        _reflib().ref_free(out_f)
    if c is not None:
        res.append(np.ctypeslib.as_array(out_c, shape=(n.value * ld,)).reshape(n.value, ld).copy())
        _reflib().ref_free(out_c)
    return tuple(res)
