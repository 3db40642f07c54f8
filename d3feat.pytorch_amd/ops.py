"""Tensor-level operators over the C ABI (device tensors in, device tensors out, autograd where the reference
has it).  Every function here launches hand-written HIP kernels from ``libd3feat_hip.so``; none has a PyTorch or
CPU fallback.  Reference locations are cited per operator.
"""
import numpy as np
import ctypes
import threading

import torch

from . import _native

ORDER_REFERENCE = 0   # row order of the reference's std::unordered_map iteration (grid_subsampling.cpp:85)
ORDER_FIRST_SEEN = 1  # cells in order of their first input point


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------------------------------------------------------
# optional per-operator HIP-event timing (bench.py): events are recorded on the stream the kernels are launched on
# ---------------------------------------------------------------------------------------------------------------
_PROFILER = None


def set_profiler(p):
    global _PROFILER
    _PROFILER = p


class _Region:
    def __init__(self, prof, label, nbytes):
        self.prof, self.label, self.nbytes = prof, label, nbytes

    def __enter__(self):
        if self.prof is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if self.prof is not None:
            self.e1.record(torch.cuda.current_stream())
            self.prof.records.append((self.label, self.nbytes, self.e0, self.e1))
        return False


def _region(label, nbytes=0):
    return _Region(_PROFILER, label, nbytes)


class EventProfiler:
    """Collects (label, algorithmic bytes, start event, end event) per C-ABI call; summary() after a sync."""

    def __init__(self):
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for label, nbytes, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            d = out.setdefault(label, {"calls": 0, "total_ms": 0.0, "bytes_per_call": nbytes})
            d["calls"] += 1
            d["total_ms"] += ms
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / d["calls"]
        return out


def kpconv_fwd_bytes(Nq, Ns, H, K, Cin, Cout):
    """Algorithmic (gather-expanded, int32-index) bytes of one KPConv forward -- SURVEY.md section 8d."""
    return 12 * Nq + 4 * Nq * H + Nq * H * (12 + 4 * Cin) + 4 * K * Cin * Cout + 180 + 4 * Nq * Cout


def kpconv_bwd_bytes(Nq, Ns, H, K, Cin, Cout):
    return kpconv_fwd_bytes(Nq, Ns, H, K, Cin, Cout) + 4 * Nq * Cout + 4 * Ns * Cin + 4 * K * Cin * Cout


def _f32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA/HIP tensor (d3feat_pytorch_amd has no CPU path)" % name)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _i32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA/HIP tensor (d3feat_pytorch_amd has no CPU path)" % name)
    if t.dtype != torch.int32:
        t = t.to(torch.int32)  # the reference hands LongTensors (dataloader.py:161-163)
    return t.contiguous()


def _lens(t, device, name):
    """Stack lengths as an int32 device tensor (accepts lists / CPU tensors like the reference's q_batches)."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t, dtype=torch.int32)
    return t.to(device=device, dtype=torch.int32).contiguous().view(-1)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _p(t):
    return t.data_ptr() if t is not None else None


SPACK_READY = 1   # D3F_SPACK_READY of the C ABI: "spack_keep already holds the packed supports"


class PackedSupports(object):
    """What the epilogue that produced a KPConv's input features left behind for it (d3f_bias_act_forward_pack): the
    packed supports {x, y, z, [row sum > 0]} of (s_pts, x) and, optionally, the cleared grad_x scatter target.  Rides on
    the feature tensor as ``x._d3f_spack``; the KPConv operators use it instead of launching their own packing kernel."""
    __slots__ = ("spack", "gx_buf", "s_ptr", "rows", "cols")

    def __init__(self, spack, gx_buf, s_pts, rows, cols):
        self.spack, self.gx_buf, self.s_ptr, self.rows, self.cols = spack, gx_buf, s_pts.data_ptr(), int(rows), int(cols)

    def fits(self, s_pts, x):
        return self.s_ptr == s_pts.data_ptr() and self.rows == int(x.shape[0]) == int(s_pts.shape[0]) and \
            self.cols == int(x.shape[1])


class DeviceStatus:
    """int32 device word the kernels OR error bits into; checked at the caller's next natural sync."""

    def __init__(self, device):
        self.word = torch.zeros(1, dtype=torch.int32, device=device)

    def raise_if_set(self):
        v = int(self.word.item())
        if v:
            raise RuntimeError(_native.status_message(v) or ("device status %d" % v))


# ---------------------------------------------------------------------------------------------------------------
# radius neighbors (cpp_wrappers/cpp_neighbors; datasets/dataloader.py:52-67)
# ---------------------------------------------------------------------------------------------------------------
class RadiusGrid:
    """Cell list of one support cloud for one radius; serves any number of query sets."""

    def __init__(self, supports, s_len, radius, status=None, ws=None):
        """``ws``: a workspace (``RadiusGrid.workspace(Ns, device)``) whose bucket counters the caller has cleared already
        (``zero_buffers``): a pyramid build clears all of its cell lists with one launch."""
        self.supports = _f32(supports, "supports")
        if self.supports.dim() != 2 or self.supports.shape[1] != 3:
            raise RuntimeError("Wrong dimensions : support.shape is not (N, 3)")
        dev = self.supports.device
        self.s_len = _lens(s_len, dev, "s_batches")
        self.radius = float(radius)
        self.status = status if status is not None else DeviceStatus(dev)
        L = _native.lib()
        self.Ns = int(self.supports.shape[0])
        nbytes = L.d3f_radius_grid_ws_bytes(self.Ns)
        build = L.d3f_radius_grid_build if ws is None else L.d3f_radius_grid_build_prezeroed
        if ws is not None and ws.numel() < nbytes:
            raise RuntimeError("cell-list workspace of %d bytes, %d needed" % (ws.numel(), nbytes))
        self.ws = _ws(nbytes, dev) if ws is None else ws
        with _region("radius_grid_build[Ns=%d]" % self.Ns, 12 * self.Ns + 24 * self.Ns):
            _native.check(build(_p(self.supports), self.Ns, _p(self.s_len), int(self.s_len.numel()), self.radius,
                                _p(self.ws), nbytes, _p(self.status.word), _stream()), "d3f_radius_grid_build")

    @staticmethod
    def workspace(Ns, device):
        """(workspace for a cell list over ``Ns`` supports, the leading part of it that must be cleared before the build)"""
        L = _native.lib()
        ws = _ws(L.d3f_radius_grid_ws_bytes(int(Ns)), device)
        return ws, ws[:int(L.d3f_radius_grid_zero_bytes(int(Ns)))]

    def query_prefix(self, queries, q_len, width, prefix_radius, radius=None, nearest_bound=0.0):
        """Prefix form (d3f_radius_query_prefix): int32 [Nq, width] rows = the supports within ``prefix_radius``, ranked as
        the leading part of the ``query`` row -- or the single nearest support within the search radius when there is
        none.  For tables of which only column 0 and the part within ``prefix_radius`` are ever read (the upsampling
        tables inside the training engine).  ``nearest_bound`` > 0: the caller guarantees a support within that distance
        of every query; cells beyond it are not scanned."""
        q = _f32(queries, "queries")
        q_len = _lens(q_len, q.device, "q_batches")
        Nq = int(q.shape[0])
        r = self.radius if radius is None else float(radius)
        out = torch.empty((Nq, int(width)), dtype=torch.int32, device=q.device)
        with _region("radius_query_prefix[Nq=%d,Ns=%d]" % (Nq, self.Ns), 12 * Nq + 12 * self.Ns + 4 * Nq * int(width)):
            _native.check(_native.lib().d3f_radius_query_prefix(
                _p(self.ws), _p(q), Nq, _p(q_len), self.Ns, _p(self.s_len), int(q_len.numel()), self.radius, r,
                float(prefix_radius), float(nearest_bound), int(width), _p(out), _p(self.status.word), _stream()),
                "d3f_radius_query_prefix")
        return out

    def query_pool_transposed(self, queries, q_len, width, max_group=0, mx_out=None, counts=None):
        """A pooling search over this (fine) cloud -- (table, device max count(s), last kept keys) as
        ``query(want_max=True, want_last_key=True)`` -- that also leaves its transpose behind: ``(counts [Ns] int32, keys
        [32 Ns] int64)``, per fine point the (d2 bits, coarse query) keys of the coarse queries that found it
        (d3f_radius_query_pool_transposed); ``RadiusGrid.prefix_rows_from_transposed`` ranks them into upsampling rows."""
        q = _f32(queries, "queries")
        q_len = _lens(q_len, q.device, "q_batches")
        if q_len.numel() != self.s_len.numel():
            raise RuntimeError("Wrong number of batch elements: different for queries and supports ")
        Nq = int(q.shape[0])
        out = torch.empty((Nq, int(width)), dtype=torch.int32, device=q.device)
        n_mx = -(-int(q_len.numel()) // int(max_group)) if max_group else 1
        mx = mx_out if mx_out is not None else torch.zeros(n_mx, dtype=torch.int32, device=q.device)
        if mx.numel() != n_mx or mx.dtype != torch.int32:
            raise RuntimeError("mx_out must hold %d int32 counters" % n_mx)
        lkey = torch.empty(Nq, dtype=torch.int64, device=q.device)
        if counts is None:          # (``counts``: [Ns] int32 the caller has cleared -- a pyramid build clears all at once)
            counts = torch.empty(self.Ns, dtype=torch.int32, device=q.device)
            zero_buffers([counts])
        elif counts.numel() != self.Ns or counts.dtype != torch.int32:
            raise RuntimeError("counts must hold %d int32 counters" % self.Ns)
        keys = torch.empty(32 * max(self.Ns, 1), dtype=torch.int64, device=q.device)
        with _region("radius_query_pool_transposed[Nq=%d,Ns=%d]" % (Nq, self.Ns), 12 * Nq + 12 * self.Ns + 4 * Nq * int(width)):
            _native.check(_native.lib().d3f_radius_query_pool_transposed(
                _p(self.ws), _p(q), Nq, _p(q_len), self.Ns, _p(self.s_len), int(q_len.numel()), self.radius, self.radius,
                int(width), _p(out), _p(mx), _p(lkey), int(max_group), _p(counts), _p(keys), _p(self.status.word),
                _stream()), "d3f_radius_query_pool_transposed")
        return out, mx, lkey, (counts, keys)

    def prefix_rows_from_transposed(self, queries, q_len, width, prefix_radius, transposed, nearest_bound=0.0):
        """The rows of ``query_prefix`` (this grid = the COARSE cloud, ``queries`` = the fine cloud) ranked from the lists a
        pooling search at ``prefix_radius`` left behind (``query_pool_transposed`` on the fine cloud's grid): the pairs are
        the same and so are the bits of their distances, so nothing is searched for a second time
        (d3f_upsample_rows_rank); the few fine points without a coarse point inside the radius -- and the padding rows --
        are searched as before (d3f_radius_query_prefix_missing)."""
        q = _f32(queries, "queries")
        q_len = _lens(q_len, q.device, "q_batches")
        counts, keys = transposed
        Nq = int(q.shape[0])
        if int(counts.numel()) != Nq:
            raise RuntimeError("transposed lists of %d points for %d queries" % (int(counts.numel()), Nq))
        L = _native.lib()
        out = torch.empty((Nq, int(width)), dtype=torch.int32, device=q.device)
        with _region("upsample_rows_rank[Nf=%d,Nc=%d]" % (Nq, self.Ns), 256 * Nq + 4 * Nq * int(width)):
            _native.check(L.d3f_upsample_rows_rank(_p(counts), _p(keys), Nq, self.Ns, int(width), _p(out), _stream()),
                          "d3f_upsample_rows_rank")
            _native.check(L.d3f_radius_query_prefix_missing(
                _p(self.ws), _p(q), Nq, _p(q_len), self.Ns, _p(self.s_len), int(q_len.numel()), self.radius, self.radius,
                float(prefix_radius), float(nearest_bound), int(width), _p(out), _p(counts), _p(self.status.word),
                _stream()), "d3f_radius_query_prefix_missing")
        return out

    def query(self, queries, q_len, width, want_counts=False, want_max=False, radius=None, wide=0, want_last_key=False,
              table=True, max_group=0, mx_out=None):
        """int32 [Nq, width] neighbor table (+ per-query uncapped counts, + device max count).

        ``radius`` (<= the grid's): search radius when it differs from the one the cell list was built for.
        ``wide`` > 0: additionally the whole ranked list of every query as an int32 [Nq, wide] table; ``want_last_key``:
        uint64 [Nq] rank key of the last entry each capped row keeps -- together the transposed form of a table for the
        gather-form KPConv grad-input (d3f_radius_query_ex).  ``table=False`` skips the capped table itself.
        ``max_group`` > 0: the max count comes per group of that many consecutive clouds (int32 [ceil(B/max_group)]).
        ``mx_out``: where the max count(s) go -- int32 counters the caller has cleared (``zero_buffers``)."""
        q = _f32(queries, "queries")
        if q.dim() != 2 or q.shape[1] != 3:
            raise RuntimeError("Wrong dimensions : query.shape is not (N, 3)")
        q_len = _lens(q_len, q.device, "q_batches")
        if q_len.numel() != self.s_len.numel():
            raise RuntimeError("Wrong number of batch elements: different for queries and supports ")
        Nq = int(q.shape[0])
        r = self.radius if radius is None else float(radius)
        if r > self.radius:
            raise RuntimeError("search radius %g exceeds the cell list's %g" % (r, self.radius))
        out = torch.empty((Nq, int(width)), dtype=torch.int32, device=q.device) if table else None
        counts = torch.empty(Nq, dtype=torch.int32, device=q.device) if want_counts else None
        n_mx = -(-int(q_len.numel()) // int(max_group)) if max_group else 1
        mx = None
        if want_max:
            mx = mx_out if mx_out is not None else torch.zeros(n_mx, dtype=torch.int32, device=q.device)
            if mx.numel() != n_mx or mx.dtype != torch.int32:
                raise RuntimeError("mx_out must hold %d int32 counters" % n_mx)
        wtab = torch.empty((Nq, int(wide)), dtype=torch.int32, device=q.device) if wide else None
        lkey = torch.empty(Nq, dtype=torch.int64, device=q.device) if want_last_key else None
        with _region("radius_query[Nq=%d,Ns=%d]" % (Nq, self.Ns), 12 * Nq + 12 * self.Ns + 4 * Nq * (int(width) + int(wide))):
            _native.check(_native.lib().d3f_radius_query_ex(_p(self.ws), _p(q), Nq, _p(q_len), self.Ns, _p(self.s_len),
                                                            int(q_len.numel()), self.radius, r, int(width), _p(out),
                                                            _p(counts), _p(mx), _p(wtab), int(wide), _p(lkey),
                                                            int(max_group), _p(self.status.word), _stream()),
                          "d3f_radius_query_ex")
        res = (out,) if table else ()
        for flag, t in ((want_counts, counts), (want_max, mx), (wide, wtab), (want_last_key, lkey)):
            if flag:
                res += (t,)
        return res if len(res) != 1 else res[0]


def copy_buffers(jobs, threshold=0.0):
    """``jobs`` = [(src, dst)] device-to-device copies of equal byte size, or (src float64, dst uint8, 'mask') for
    dst = src > threshold -- all in ONE launch per 24 jobs (d3f_copy_buffers).  The caller guarantees contiguous tensors."""
    import ctypes
    jobs = [j for j in jobs if j[0].numel()]
    for k in range(0, len(jobs), 24):
        part = jobs[k:k + 24]
        n = len(part)
        srcs = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in part])
        dsts = (ctypes.c_void_p * n)(*[j[1].data_ptr() for j in part])
        sizes = (ctypes.c_size_t * n)(*[j[0].numel() * j[0].element_size() for j in part])
        kinds = (ctypes.c_int * n)(*[1 if len(j) > 2 else 0 for j in part])
        _native.check(_native.lib().d3f_copy_buffers(srcs, dsts, sizes, kinds, n, float(threshold), _stream()),
                      "d3f_copy_buffers")


def zero_buffers(tensors):
    """Clear up to 8 device buffers (4-byte multiples) with ONE launch (d3f_zero_buffers)."""
    import ctypes
    ts = [t for t in tensors if t is not None and t.numel()]
    for k in range(0, len(ts), 8):
        part = ts[k:k + 8]
        ptrs = (ctypes.c_void_p * len(part))(*[t.data_ptr() for t in part])
        sizes = (ctypes.c_size_t * len(part))(*[t.numel() * t.element_size() for t in part])
        _native.check(_native.lib().d3f_zero_buffers(ptrs, sizes, len(part), _stream()), "d3f_zero_buffers")


# ---------------------------------------------------------------------------------------------------------------
# grid subsampling (cpp_wrappers/cpp_subsampling; datasets/dataloader.py:12-22)
# ---------------------------------------------------------------------------------------------------------------
def grid_subsample_raw(points, lens, sampleDl, max_p=0, order=ORDER_REFERENCE, status=None, out_cap=0, features=None,
                       labels=None):
    """Sync-free form: returns (out_points [capacity,3], out_len [B] int32, out_total [1] int32, status), followed by
    out_features [capacity,fdim] / out_labels [capacity,ldim] int32 when ``features`` / ``labels`` are given
    (reference grid_subsampling.cpp:89-102: member mean, majority vote).

    ``points`` may itself be a capacity buffer: only the first sum(lens) rows are read, so pyramid levels chain
    on the device without reading lengths back."""
    p = _f32(points, "points")
    if p.dim() != 2 or p.shape[1] != 3:
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")
    dev = p.device
    lens = _lens(lens, dev, "batches")
    status = status if status is not None else DeviceStatus(dev)
    N, B = int(p.shape[0]), int(lens.numel())
    f = c = None
    if features is not None:
        f = _f32(features, "features")
        if f.dim() != 2 or f.shape[0] != N:
            raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")      # wrapper.cpp:172-215
    if labels is not None:
        c = labels
        if not (isinstance(c, torch.Tensor) and c.is_cuda and c.dtype == torch.int32):
            raise RuntimeError("classes must be an int32 device tensor")
        if c.dim() > 2 or c.shape[0] != N:
            raise RuntimeError("Wrong dimensions : classes.shape is not (N,) or (N, d)")  # wrapper.cpp:182-224
        c = c.reshape(N, -1).contiguous()
    L = _native.lib()
    nbytes = L.d3f_grid_subsample_ws_bytes(N, B)
    ws = _ws(nbytes, dev)
    cap = int(out_cap) if out_cap and out_cap > 0 else N
    out = torch.empty((cap, 3), dtype=torch.float32, device=dev)
    out_len = torch.empty(B, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    of = torch.empty((cap, f.shape[1]), dtype=torch.float32, device=dev) if f is not None else None
    oc = torch.empty((cap, c.shape[1]), dtype=torch.int32, device=dev) if c is not None else None
    with _region("grid_subsample[cap=%d]" % N, 24 * N):
        if f is None and c is None:
            _native.check(L.d3f_grid_subsample(_p(p), N, _p(lens), B, float(sampleDl), int(max_p), int(order), _p(out),
                                               cap, _p(out_len), _p(total), _p(ws), nbytes, _p(status.word), _stream()),
                          "d3f_grid_subsample")
        else:
            _native.check(L.d3f_grid_subsample_ex(
                _p(p), N, _p(lens), B, float(sampleDl), int(max_p), int(order), _p(f) if f is not None else None,
                int(f.shape[1]) if f is not None else 0, _p(c) if c is not None else None,
                int(c.shape[1]) if c is not None else 0, _p(out), cap, _p(out_len), _p(total),
                _p(of) if of is not None else None, _p(oc) if oc is not None else None, _p(ws), nbytes, _p(status.word),
                _stream()), "d3f_grid_subsample_ex")
    extra = tuple(t for t in (of, oc) if t is not None)
    return (out, out_len, total, status) + extra


# ---------------------------------------------------------------------------------------------------------------
# reverse neighbor tables (csrc/reverse_table.hip): CSR transpose of a table, for the gather-form KPConv grad-input
# ---------------------------------------------------------------------------------------------------------------
class ReverseTable(object):
    """The transpose of a neighbor table [Nq, H] over Ns supports, in one of two forms (int32 / int64 device tensors):
      CSR    ``ent[ptr[s] : ptr[s+1]]`` = the queries that list support s, ascending (build_reverse_table);
      search ``ent`` [Ns, width] = every query point within the radius of s, ranked, and ``last_key`` [Nq] = rank key of
             the last entry each table row kept: q lists s iff key(q, s) <= last_key[q] (RadiusGrid.query outputs)."""
    __slots__ = ("ptr", "ent", "last_key", "width", "Nq", "H", "Ns", "radius", "status", "rel")

    def __init__(self, ent, Nq, H, Ns, ptr=None, last_key=None, radius=0.0, status=None, rel=None):
        """``rel`` [Ns, width, 4] float32: the EXACT form (filter_reverse_table) -- row s = its true reverse neighbors
        compacted as {q - s, bits of q}; ``ent`` / ``ptr`` / ``last_key`` are then None."""
        if rel is not None:
            if ent is not None or ptr is not None or last_key is not None:
                raise ValueError("an exact-form reverse table carries only `rel`")
        elif (ptr is None) == (last_key is None):
            raise ValueError("a reverse table is either CSR (ptr) or search-form (last_key)")
        self.rel = rel
        self.ptr, self.ent, self.last_key = ptr, ent, last_key
        # radius > 0: `ent` comes from a search with a larger radius (an upsampling table); only its entries within
        # `radius` count.  status: DeviceStatus that receives D3F_ST_WIDE_OVERFLOW if such a row was cut short.
        self.radius, self.status = float(radius), status
        self.width = int(rel.shape[1]) if rel is not None else (int(ent.shape[1]) if ptr is None else 0)
        self.Nq, self.H, self.Ns = int(Nq), int(H), int(Ns)

    def matches(self, Nq, H, Ns):
        return (self.Nq, self.H, self.Ns) == (int(Nq), int(H), int(Ns))

    def tensors(self):
        return [t for t in (self.ptr, self.ent, self.last_key, self.rel) if t is not None]

    def edges(self):
        """Number of (query, support) pairs (one host read-back; measurement only)."""
        if self.rel is not None:
            return int((self.rel[:, :, 3].contiguous().view(torch.int32) < self.Nq).sum())
        if self.ptr is not None:
            return int(self.ptr[-1])
        return int((self.ent < self.Nq).sum())   # (search form: an upper bound -- the rows are supersets)


def filter_reverse_table(rev, q_pts, s_pts):
    """Search-form ReverseTable -> exact form (one launch, meant for the pyramid build): the membership test of every
    entry evaluated once, survivors compacted as {q - s, q}.  The search-form tensors are not kept."""
    if rev.rel is not None or rev.ptr is not None:
        return rev
    q, sp = _f32(q_pts, "q_pts"), _f32(s_pts, "s_pts")
    rel = torch.empty((rev.Ns, rev.width, 4), dtype=torch.float32, device=sp.device)
    with _region("reverse_table_filter[Ns=%d,W=%d]" % (rev.Ns, rev.width), 20 * rev.Ns * rev.width):
        _native.check(_native.lib().d3f_reverse_table_filter(
            _p(rev.ent), rev.width, _p(rev.last_key), _p(q), rev.Nq, _p(sp), rev.Ns, rev.radius, _p(rel),
            _p(rev.status.word) if rev.status is not None else None, _stream()), "d3f_reverse_table_filter")
    return ReverseTable(None, rev.Nq, rev.H, rev.Ns, rel=rel)


def attach_reverse_table(neighb_inds, rev):
    """Hand a table its transpose, so that operators given only the table (reference signatures) find it."""
    try:
        neighb_inds._d3f_rev = rev
    except AttributeError:  # pragma: no cover
        pass
    return rev


def build_reverse_table(neighb_inds, Ns):
    """CSR transpose of ``neighb_inds`` [Nq, H] (int32 device table, shadow entries >= Ns) over ``Ns`` supports, for
    tables that did not come with a search-form transpose.  Also attached to the table (``_d3f_rev``)."""
    idx = _i32(neighb_inds, "neighb_inds")
    Nq, H = int(idx.shape[0]), int(idx.shape[1])
    dev = idx.device
    ptr = torch.empty(int(Ns) + 1, dtype=torch.int32, device=dev)
    ent = torch.empty(max(Nq * H, 1), dtype=torch.int32, device=dev)
    L = _native.lib()
    nbytes = L.d3f_reverse_table_ws_bytes(Nq, H, int(Ns))
    ws = _ws(nbytes, dev)
    with _region("reverse_table[Nq=%d,H=%d,Ns=%d]" % (Nq, H, Ns), 12 * Nq * H + 8 * Ns):
        _native.check(L.d3f_reverse_table_build(_p(idx), Nq, H, int(Ns), _p(ptr), _p(ent), _p(ws), nbytes, _stream()),
                      "d3f_reverse_table_build")
    return attach_reverse_table(neighb_inds, ReverseTable(ent, Nq, H, Ns, ptr=ptr))


def reverse_table_of(neighb_inds, Nq, H, Ns, rev=None):
    """The reverse table to use for a table: the explicit one, else the one its builder attached; None otherwise."""
    if rev is None:
        rev = getattr(neighb_inds, "_d3f_rev", None)
    if rev is not None and not rev.matches(Nq, H, Ns):
        raise RuntimeError("reverse table of a [%d,%d] table over %d supports used with a [%d,%d] table over %d" % (
            rev.Nq, rev.H, rev.Ns, Nq, H, Ns))
    return rev


# KPConv layers whose grad-input runs as a gather over the reverse table: those with at least this many SUPPORT rows.
# Measured per layer of the S1 pair (profiles/r02b_timeline.txt vs r02a): 38k rows 177 -> 67 us, 8k rows x 64 ch
# 84 -> 36 us, the 8k -> 2k strided layer 43 -> 29 us.  Round 3 (exact-form tables, profiles/r03_dx_gather_threshold.txt):
# at the 2k-point level the two forms tie (4.096 vs 4.098 ms per step) -- the gather form is taken there too, it has no
# atomics and is bit-reproducible; below that (581 / 159 points x 256 / 512 channels) the scatter stays ahead (4.25 ms
# with the gather form at 581 points, 4.57 ms at 159: the chip is filled by splitting channels, which the gather form
# pays for with repeated aggregation).  Round 6 (3 stacked pairs: 114k / 24k / 6.2k / 1.7k / 0.5k rows; the wide layers
# now run the transposed aggregation + GEMM form, not the fused gather kernel): 1000 rows puts the 1.7k-row level on
# it as well -- 581.6 against 578.8 pairs/s at 2000, 575 / 574 at 300 / 100 -- and takes its float atomics away.
DX_GATHER_MIN_ROWS = 1000


# width of the search-form transpose of a conv table (the whole in-radius list of a point; S1: mean 41, max 68 at the
# 80 % limit of 42).  More than that sets D3F_ST_WIDE_OVERFLOW.  A pooling table's transpose is read off the
# upsampling table (coarse points within 2r of a fine point, nearest first: those within r -- mean 7, max 18 -- lead).
REV_WIDTH_CONV = 96
# the engine's upsampling rows ranked from the transpose the POOLING search leaves behind (RadiusGrid.query_pool_transposed
# / prefix_rows_from_transposed); False: the upsampling rows are searched for (rounds 4-6)
UPSAMPLES_FROM_POOL = True


def wants_reverse_table(Ns):
    """Whether a table over ``Ns`` supports is worth transposing (what the pyramid builders ask)."""
    return int(Ns) >= DX_GATHER_MIN_ROWS



# ---------------------------------------------------------------------------------------------------------------
# KPConv (models/blocks.py:237-382)
# ---------------------------------------------------------------------------------------------------------------
def _adoptable(t, slot):
    """autograd adopts an incoming gradient as ``p.grad`` without a copy only when nobody else holds the tensor
    object: hand it a fresh view of the slot."""
    return t.view_as(t) if (slot is not None and t is slot) else t


def _grad_slot(param):
    """Where the weight gradient of ``param`` is to be written, when its owner (train.FlatParams) reserved a place for
    it in a flat gradient buffer; None otherwise (a fresh tensor is allocated)."""
    return getattr(param, "_d3f_grad_slot", None)


# False: the backward pass recomputes the neighbor aggregation instead of reading it back (saves K*Cin*4 B/query)
SAVE_WEIGHTED_FEATURES = True
# below this many rows a weight gradient is a plain GEMM for the library; above, the reduction-parallel kernel
_SPLITK_MIN_ROWS = 4096
# below this many query points KPConv goes through library GEMMs (aggregate + wf@W forward; gW = (g/nn) W^T + scatter
# backward): measured better than the fused kernels up to the 2k-point level, not at 8k points
_GEMM_DX_MAX_ROWS = 4096
# ... and, whatever the row count, from this many input channels up (round 4): the contraction with W is then 60 % and
# more of a KPConv's arithmetic, and inside the fused kernels it runs on 16-row tiles against weights streamed from L2
# (0.15 - 0.2 of the f32 matrix rate, profiles/r04_step_timeline_stack4.txt); as aggregation kernel (registers -> HBM,
# csrc/kpconv_aggregate.hip) + tall GEMMs every contraction gets 128-row tiles.  Same split for the grad-input over
# the exact-form reverse table (transposed aggregation + GEMM with the permuted weights).
_GEMM_PATH_MIN_CIN = 32      # (round 6: 32 -- the level-0 layers' forward too: 266 -> 207 us alone, +1.2 % inside the 4 x 3 step)
_GEMM_DX_AGG_MIN_COUT = 64


_DW_LIBRARY_MIN_OUT = 1920 * 128
_DW_LIBRARY_MAX_ROWS = 16384
_DW_LIBRARY = True


def _takes_gemm_path(Nq, Cin):
    return 0 < Nq < _GEMM_DX_MAX_ROWS or (Nq > 0 and Cin >= _GEMM_PATH_MIN_CIN)


def _kpconv_gw(gon, weights, Nq, K, Cin, Cout):
    """gW [Nq, K Cin] = (g / nn) [Nq, Cout] @ W^T, W viewed [K Cin, Cout]: the per-query gradient of the weighted features
    the scatter-form grad-input kernel distributes (few-point layers)."""
    w2 = weights.view(K * Cin, Cout)
    if _own_gemm("kpconv_gw", gon, w2, GEMM_NT, Nq, Cout, K * Cin):
        return gemm_epilogue(gon, w2, GEMM_NT, Nq, Cout, K * Cin)
    return torch.mm(gon, w2.t())


class _KPConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_pts, s_pts, idx, x, kernel_points, weights, extent, rev=None, ready=None):
        L = _native.lib()
        Nq, Ns, H = int(q_pts.shape[0]), int(s_pts.shape[0]), int(idx.shape[1])
        K, Cin, Cout = int(weights.shape[0]), int(weights.shape[1]), int(weights.shape[2])
        if rev is not None and not (Ns >= DX_GATHER_MIN_ROWS and L.d3f_kpconv_grad_input_gather_supported(Cin, Cout, K)):
            rev = None
        ctx.rev = rev
        out = torch.empty((Nq, Cout), dtype=torch.float32, device=x.device)
        nn = torch.empty(Nq, dtype=torch.float32, device=x.device)
        nbytes = L.d3f_kpconv_ws_bytes(Nq, Ns, H, K, Cin, Cout)
        ws = _ws(nbytes, x.device)
        # training: keep the weighted features [Nq, K*Cin] for the backward pass (K*Cin*4 B per query of HBM) so the
        # weight gradient is one tall-skinny GEMM and the neighbor aggregation is not recomputed
        wf = None
        wf_row = L.d3f_kpconv_saves_wf(Cin, Cout, K, H)   # floats per query (the input-layer kernels pad K to 16 slots)
        if SAVE_WEIGHTED_FEATURES and ctx.needs_input_grad[5] and Nq > 0 and wf_row:
            wf = torch.empty((Nq, wf_row), dtype=torch.float32, device=x.device)
        # training: the packed supports are kept for the backward pass and its scatter target is cleared on the side
        keep = gx_buf = None
        packs = Nq > 0 and Ns > 0 and L.d3f_kpconv_packs_supports(Cin, Cout, K, H, Ns)
        need_clear = ctx.needs_input_grad[3] and rev is None   # (the gather form writes every row: nothing to clear)
        # the epilogue that produced x may have packed the supports already (PackedSupports): no packing launch then
        use_ready = packs and ready is not None and (not need_clear or ready.gx_buf is not None)
        if use_ready:
            keep, gx_buf = ready.spack, (ready.gx_buf if need_clear else None)
        elif (ctx.needs_input_grad[3] or ctx.needs_input_grad[5]) and packs:
            keep = torch.empty(16 * Ns, dtype=torch.uint8, device=x.device)
            if need_clear:
                gx_buf = torch.empty_like(x)
        with _region("kpconv_fwd[Nq=%d,Cin=%d,Cout=%d,H=%d]" % (Nq, Cin, Cout, H),
                     kpconv_fwd_bytes(Nq, Ns, H, K, Cin, Cout)):
            _native.check(L.d3f_kpconv_forward(_p(q_pts), Nq, _p(s_pts), Ns, _p(idx), H, _p(x), Cin,
                                               _p(kernel_points), K, _p(weights), Cout, float(extent), _p(out),
                                               _p(nn), _p(wf), _p(keep), SPACK_READY if use_ready else _p(gx_buf),
                                               _p(ws), nbytes, _stream()), "d3f_kpconv_forward")
        ctx.keep, ctx.gx_buf = keep, gx_buf
        ctx.gw_slot = _grad_slot(weights)
        ctx.has_wf = wf is not None
        ctx.save_for_backward(q_pts, s_pts, idx, x, kernel_points, weights, nn, *([wf] if wf is not None else []))
        ctx.extent = float(extent)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q_pts, s_pts, idx, x, kernel_points, weights, nn = ctx.saved_tensors[:7]
        wf = ctx.saved_tensors[7] if ctx.has_wf else None
        L = _native.lib()
        Nq, Ns, H = int(q_pts.shape[0]), int(s_pts.shape[0]), int(idx.shape[1])
        K, Cin, Cout = int(weights.shape[0]), int(weights.shape[1]), int(weights.shape[2])
        need_x, need_w = ctx.needs_input_grad[3], ctx.needs_input_grad[5]
        keep, pre = ctx.keep, 0
        gx = None
        if need_x:
            gx, ctx.gx_buf = ctx.gx_buf, None  # cleared by the forward; a second backward gets a fresh buffer
            pre = 1 if gx is not None else 0
            if gx is None:
                gx = torch.empty_like(x)
        gw = (ctx.gw_slot if ctx.gw_slot is not None else torch.empty_like(weights)) if need_w else None
        go = grad_out.contiguous().float() if (need_x or need_w) else None
        gw_native, gx_native = gw, gx
        gon = None
        big_dw = (need_w and wf is not None and Nq >= _SPLITK_MIN_ROWS and wf.shape[1] == K * Cin
                  and L.d3f_linear_grad_weight_supported(Nq, Cout, K * Cin))
        if big_dw:
            # g / nn ONCE (one small elementwise launch) for both consumers: inside the A^T B kernel the division sat in
            # every lane of every column block behind a dependent 4-byte load (round 4, bench per-launch table:
            # 23.9k x 480 x 32 in 63 us against 24 us without it), and the gather kernel divided once per edge
            gon = go / nn.unsqueeze(1)
        if need_x and ctx.rev is not None and Nq > 0:
            # gather over the reverse table: aggregate grad_out/nn around every support, then contract with W^T
            rev = ctx.rev
            with _region("kpconv_dx_gather[Ns=%d,Cin=%d,Cout=%d]" % (Ns, Cin, Cout),
                         kpconv_bwd_bytes(Nq, Ns, H, K, Cin, Cout)):
                _native.check(L.d3f_kpconv_grad_input_gather(_p(q_pts), Nq, _p(s_pts), Ns, _p(rev.ptr), _p(rev.ent),
                                                             _p(rev.last_key), rev.width, rev.radius, _p(rev.rel),
                                                             _p(kernel_points), K, _p(weights), Cin, Cout, ctx.extent,
                                                             _p(nn) if gon is None else None,
                                                             _p(go) if gon is None else _p(gon), _p(gx),
                                                             _p(rev.status.word) if rev.status is not None else None,
                                                             _stream()),
                              "d3f_kpconv_grad_input_gather")
            gx_native = None
        if big_dw:
            # x := g / nn [Nq, Cout], grad_out := wf [Nq, K Cin]: grad_W [K Cin, Cout] = wf^T (g / nn)
            _grad_weight_atb(gon, wf, Nq, Cout, K * Cin, gw.view(K * Cin, Cout), None, 0, None, None, "kpconv_dw_atb")
            gw_native = None
        if need_w and wf is not None and Nq < _SPLITK_MIN_ROWS and wf.shape[1] == K * Cin:
            # few points, wide layers (bottom of the U-Net): grad_W = wf^T (g/nn) is an ordinary GEMM with a short
            # reduction -- a library call; the reduction-parallel kernel is for the tall-skinny upper levels
            gon = go / nn.unsqueeze(1)
            if not _queue_small_weight_grad(gon, wf, Nq, Cout, K * Cin, gw.view(K * Cin, Cout)):
                with _region("kpconv_dw_gemm[Nq=%d,Cin=%d,Cout=%d]" % (Nq, Cin, Cout), 4 * Nq * (K * Cin + Cout)):
                    torch.mm(wf.t(), gon, out=gw.view(K * Cin, Cout))
            gw_native = None
        if gx_native is not None and 0 < Nq < _GEMM_DX_MAX_ROWS and L.d3f_kpconv_grad_input_supported(Cin, K, H, Ns):
            # same layers: gW = (g/nn) W^T over all queries is one library GEMM; the kernel only scatters
            if gon is None:
                gon = go / nn.unsqueeze(1)
            gwf = _kpconv_gw(gon, weights, Nq, K, Cin, Cout)
            nbytes = L.d3f_kpconv_ws_bytes(Nq, Ns, H, K, Cin, 64)
            ws = _ws(nbytes, x.device)
            with _region("kpconv_dx_scatter[Nq=%d,Cin=%d,H=%d]" % (Nq, Cin, H), 4 * Nq * K * Cin + 4 * Nq * H * (1 + Cin)):
                _native.check(L.d3f_kpconv_grad_input(_p(q_pts), Nq, _p(s_pts), Ns, _p(idx), H, _p(x), Cin,
                                                      _p(kernel_points), K, ctx.extent, _p(gwf), _p(keep), pre, _p(gx),
                                                      _p(ws), nbytes, _stream()), "d3f_kpconv_grad_input")
            gx_native = None
        if gx_native is not None or gw_native is not None:
            nbytes = L.d3f_kpconv_ws_bytes(Nq, Ns, H, K, Cin, Cout)
            ws = _ws(nbytes, x.device)
            with _region("kpconv_bwd[Nq=%d,Cin=%d,Cout=%d,H=%d]" % (Nq, Cin, Cout, H),
                         kpconv_bwd_bytes(Nq, Ns, H, K, Cin, Cout)):
                _native.check(L.d3f_kpconv_backward(_p(q_pts), Nq, _p(s_pts), Ns, _p(idx), H, _p(x), Cin,
                                                    _p(kernel_points), K, _p(weights), Cout, ctx.extent, _p(nn),
                                                    _p(go), _p(wf), _p(keep), pre, _p(gx_native), _p(gw_native),
                                                    _p(ws), nbytes, _stream()),
                              "d3f_kpconv_backward")
        return None, None, None, gx, None, (_adoptable(gw, ctx.gw_slot) if gw is not None else None), None, None, None


# The transposed-aggregation grad-input contracts with the permuted weights W'[k, o, c] = W[k, c, o].  One
# permute + copy launch per layer (9 per 3-pair stack, 63 us) became ONE launch per backward pass: the forward of every
# such layer queues its weights, the first grad-input that needs a permuted matrix launches d3f_permute_kpconv_weights
# for the whole queue.  The queue is process-wide, not per thread: a forward on the calling thread is followed by its
# backward on autograd's device thread (or, for a lane, on the lane's capture thread); steps are recorded / run one at a
# time, the lock only keeps the two dictionaries consistent.  False: one launch per layer.
BATCH_WEIGHT_PERMUTES = True
_WPERM = {'queue': {}, 'ready': {}}
_WPERM_LOCK = threading.Lock()


def _wperm_state():
    return _WPERM


def _queue_weight_permute(weights):
    st = _wperm_state()
    key = weights.data_ptr()
    with _WPERM_LOCK:
        st['ready'].pop(key, None)          # (a new forward: whatever an earlier backward left behind is stale)
        if len(st['queue']) >= 64:          # forwards without a backward: start over
            st['queue'].clear()
        st['queue'][key] = weights


def _permuted_weights(weights):
    """W' [K * Cout, Cin] of ``weights`` [K, Cin, Cout]."""
    K, Cin, Cout = int(weights.shape[0]), int(weights.shape[1]), int(weights.shape[2])
    st = _wperm_state()
    key = weights.data_ptr()
    with _WPERM_LOCK:
        wp = st['ready'].pop(key, None)
        jobs = []
        if wp is None and BATCH_WEIGHT_PERMUTES and key in st['queue']:
            jobs = [w for w in st['queue'].values()
                    if w.shape[1] % 32 == 0 and w.shape[2] % 32 == 0 and w.is_contiguous() and w.device == weights.device]
            st['queue'].clear()
            st['ready'].clear()
    if jobs:
        for j0 in range(0, len(jobs), 16):
            part = jobs[j0:j0 + 16]
            outs = [torch.empty((w.shape[0] * w.shape[2], w.shape[1]), dtype=torch.float32, device=w.device) for w in part]
            n = len(part)
            srcs = (ctypes.c_void_p * n)(*[w.data_ptr() for w in part])
            dsts = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
            ks = (ctypes.c_int * n)(*[int(w.shape[0]) for w in part])
            cis = (ctypes.c_int * n)(*[int(w.shape[1]) for w in part])
            cos = (ctypes.c_int * n)(*[int(w.shape[2]) for w in part])
            _native.check(_native.lib().d3f_permute_kpconv_weights(srcs, dsts, ks, cis, cos, n, _stream()),
                          "d3f_permute_kpconv_weights")
            with _WPERM_LOCK:
                for w, o in zip(part, outs):
                    st['ready'][w.data_ptr()] = o
        with _WPERM_LOCK:
            wp = st['ready'].pop(key, None)
    if wp is None:
        wp = weights.permute(0, 2, 1).contiguous().view(K * Cout, Cin)
    return wp


class _KPConvGemmBiasActFn(torch.autograd.Function):
    """act(KPConv(x) + bias) for the few-point / wide layers (bottom of the U-Net), as
        aggregation kernel -> wf [Nq, K*Cin], nn      library GEMM  raw = wf @ W       epilogue  act(raw/nn + bias)
    and backward   epilogue backward -> g/nn, bias gradient      GEMMs  grad_W = wf^T (g/nn),  gW = (g/nn) W^T
                   scatter kernel -> grad_x.
    The fused forward kernel would run its contraction on 10..40 workgroups there (58-71 us); this is ~40 us and the
    backward saves the separate g/nn pass."""

    @staticmethod
    def forward(ctx, q_pts, s_pts, idx, x, kernel_points, weights, bias, extent, slope, rev=None, ready=None):
        L = _native.lib()
        Nq, Ns, H = int(q_pts.shape[0]), int(s_pts.shape[0]), int(idx.shape[1])
        K, Cin, Cout = int(weights.shape[0]), int(weights.shape[1]), int(weights.shape[2])
        if rev is not None and not (Ns >= DX_GATHER_MIN_ROWS and L.d3f_kpconv_grad_input_gather_supported(Cin, Cout, K)):
            rev = None
        ctx.rev = rev
        dev = x.device
        wf = torch.empty((Nq, K * Cin), dtype=torch.float32, device=dev)
        nn = torch.empty(Nq, dtype=torch.float32, device=dev)
        nbytes = L.d3f_kpconv_ws_bytes(Nq, Ns, H, K, Cin, 64)
        ws = _ws(nbytes, dev)
        keep = gx_buf = None
        need_clear = ctx.needs_input_grad[3] and rev is None
        # the epilogue that produced x may have packed the supports already (PackedSupports): no packing launch then
        use_ready = ready is not None and (not need_clear or ready.gx_buf is not None)
        if use_ready:
            keep, gx_buf = ready.spack, (ready.gx_buf if need_clear else None)
        elif need_clear:
            keep = torch.empty(16 * Ns, dtype=torch.uint8, device=dev)
            gx_buf = torch.empty_like(x)
        with _region("kpconv_aggregate[Nq=%d,Cin=%d,H=%d]" % (Nq, Cin, H), 4 * Nq * H * (4 + Cin) + 4 * Nq * K * Cin):
            _native.check(L.d3f_kpconv_aggregate(_p(q_pts), Nq, _p(s_pts), Ns, _p(idx), H, _p(x), Cin,
                                                 _p(kernel_points), K, float(extent), _p(wf), _p(nn), _p(keep),
                                                 SPACK_READY if use_ready else _p(gx_buf), _p(ws), nbytes, _stream()),
                          "d3f_kpconv_aggregate")
        ctx.keep, ctx.gx_buf = keep, gx_buf
        want_b = bias is not None and ctx.needs_input_grad[6]
        gbuf = torch.empty((1, Cout), dtype=torch.float32, device=dev) if want_b else None
        if _own_gemm("kpconv_fwd", wf, weights, GEMM_NN, Nq, K * Cin, Cout, 0, None, None, bias):
            # contraction + / nn + bias + LeakyReLU in one launch (csrc/gemm_epilogue.hip)
            out = gemm_epilogue(wf, weights, GEMM_NN, Nq, K * Cin, Cout, row_div=nn, bias1=bias, slope=slope,
                                zero_init=gbuf if want_b else None)
        else:
            raw = torch.mm(wf, weights.view(K * Cin, Cout))
            out = torch.empty_like(raw)
            _native.check(L.d3f_bias_act_forward(_p(raw), _p(bias), None, None, float(slope), Nq, Cout, _p(out), _p(gbuf),
                                                 Cout if want_b else 0, _p(nn), None, 0, 0, _stream()),
                          "d3f_bias_act_forward")
        ctx.save_for_backward(q_pts, s_pts, idx, x, kernel_points, weights, nn, wf, out)
        ctx.gbuf, ctx.extent, ctx.slope, ctx.want_b = gbuf, float(extent), float(slope), want_b
        ctx.gw_slot = _grad_slot(weights)
        if BATCH_WEIGHT_PERMUTES and ctx.needs_input_grad[3] and rev is not None and rev.rel is not None and \
                Cout >= _GEMM_DX_AGG_MIN_COUT and L.d3f_kpconv_aggregate_transposed_supported(Cout, K):
            _queue_weight_permute(weights)      # (its grad-input contracts with W': one launch for all such layers)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q_pts, s_pts, idx, x, kernel_points, weights, nn, wf, out = ctx.saved_tensors
        L = _native.lib()
        Nq, Ns, H = int(q_pts.shape[0]), int(s_pts.shape[0]), int(idx.shape[1])
        K, Cin, Cout = int(weights.shape[0]), int(weights.shape[1]), int(weights.shape[2])
        go = grad_out.contiguous()
        gb, pre = None, 0
        if ctx.want_b:
            gb, pre = ctx.gbuf, 1
            ctx.gbuf = None
            if gb is None:
                gb, pre = torch.empty((1, Cout), dtype=torch.float32, device=go.device), 0
        gon = torch.empty_like(go)  # masked gradient / nn
        # many rows: the reduction over the points is what has to be spread over the chip (csrc/linear.hip); a large
        # output over a few thousand rows (1920 x 128 and up) is an ordinary GEMM again: 36 against 58 us at 6159 rows
        # (profiles/r04_dw_library_vs_atb.txt; D3F_DW_LIBRARY=0: never, round 5's kernel re-measures it)
        atb = (ctx.needs_input_grad[5] and Nq >= _SPLITK_MIN_ROWS
               and bool(L.d3f_linear_grad_weight_supported(Nq, Cout, K * Cin))
               and not (_DW_LIBRARY and K * Cin * Cout >= _DW_LIBRARY_MIN_OUT and Nq <= _DW_LIBRARY_MAX_ROWS))
        bias_part, bias_blocks = _epilogue_backward(go, out, ctx.slope, Nq, Cout, gon, gb, None, pre, nn,
                                                    _FOLD_BIAS_SUM and atb)
        gx = gw = None
        if ctx.needs_input_grad[5]:
            gw = ctx.gw_slot if ctx.gw_slot is not None else torch.empty_like(weights)
            if atb:
                # x := g / nn [Nq, Cout], grad_out := wf [Nq, K Cin]: grad_W [K Cin, Cout] = wf^T (g / nn)
                _grad_weight_atb(gon, wf, Nq, Cout, K * Cin, gw.view(K * Cin, Cout), bias_part, bias_blocks, gb, None,
                                 "kpconv_dw_atb")
            elif not _queue_small_weight_grad(gon, wf, Nq, Cout, K * Cin, gw.view(K * Cin, Cout)):
                torch.mm(wf.t(), gon, out=gw.view(K * Cin, Cout))
        rev = ctx.rev
        if ctx.needs_input_grad[3] and rev is not None and rev.rel is not None and Cout >= _GEMM_DX_AGG_MIN_COUT \
                and L.d3f_kpconv_aggregate_transposed_supported(Cout, K):
            # transposed aggregation (registers -> HBM) + one tall GEMM with the permuted weights W'[k, o, c] = W[k, c, o]:
            # every row of grad_x written once, no atomics (gon is already / nn)
            agg = torch.empty((Ns, K * Cout), dtype=torch.float32, device=x.device)
            with _region("kpconv_agg_transposed[Ns=%d,Cout=%d]" % (Ns, Cout), 4 * Ns * K * Cout + 16 * Ns * rev.width):
                _native.check(L.d3f_kpconv_aggregate_transposed(_p(rev.rel), rev.width, Ns, Nq, _p(kernel_points), K,
                                                                ctx.extent, None, _p(gon), Cout, _p(agg), _stream()),
                              "d3f_kpconv_aggregate_transposed")
            if _own_gemm("kpconv_dx", agg, weights, GEMM_NT, Ns, K * Cout, Cin, Cout):
                # W' read in place from W [K, Cin, Cout] (block form of the reduction): no permuted copy of the weights
                gx = gemm_epilogue(agg, weights, GEMM_NT, Ns, K * Cout, Cin, kblock=Cout)
            else:
                gx = torch.mm(agg, _permuted_weights(weights))
        elif ctx.needs_input_grad[3] and rev is not None:
            # gather form: one launch instead of the gW GEMM + atomic scatter (gon is already / nn)
            gx = torch.empty_like(x)
            with _region("kpconv_dx_gather[Ns=%d,Cin=%d,Cout=%d]" % (Ns, Cin, Cout),
                         kpconv_bwd_bytes(Nq, Ns, H, K, Cin, Cout)):
                _native.check(L.d3f_kpconv_grad_input_gather(_p(q_pts), Nq, _p(s_pts), Ns, _p(rev.ptr), _p(rev.ent),
                                                             _p(rev.last_key), rev.width, rev.radius, _p(rev.rel),
                                                             _p(kernel_points), K, _p(weights), Cin, Cout, ctx.extent,
                                                             None, _p(gon), _p(gx),
                                                             _p(rev.status.word) if rev.status is not None else None,
                                                             _stream()),
                              "d3f_kpconv_grad_input_gather")
        elif ctx.needs_input_grad[3]:
            gx, ctx.gx_buf = ctx.gx_buf, None
            pre = 1 if gx is not None else 0
            if gx is None:
                gx = torch.empty_like(x)
            gwf = _kpconv_gw(gon, weights, Nq, K, Cin, Cout)
            nbytes = L.d3f_kpconv_ws_bytes(Nq, Ns, H, K, Cin, 64)
            ws = _ws(nbytes, x.device)
            with _region("kpconv_dx_scatter[Nq=%d,Cin=%d,H=%d]" % (Nq, Cin, H), 4 * Nq * K * Cin + 4 * Nq * H * (1 + Cin)):
                _native.check(L.d3f_kpconv_grad_input(_p(q_pts), Nq, _p(s_pts), Ns, _p(idx), H, _p(x), Cin,
                                                      _p(kernel_points), K, ctx.extent, _p(gwf), _p(ctx.keep), pre,
                                                      _p(gx), _p(ws), nbytes, _stream()), "d3f_kpconv_grad_input")
        return (None, None, None, gx, None, (_adoptable(gw, ctx.gw_slot) if gw is not None else None),
                (gb.view(-1) if gb is not None else None), None, None, None, None)


def _ready_supports(x, s_pts):
    """The PackedSupports the producer of ``x`` left on it, if they belong to exactly this (s_pts, x)."""
    ready = getattr(x, '_d3f_spack', None)
    return ready if (ready is not None and ready.fits(s_pts, x)) else None


def kpconv_bias_act(q_pts, s_pts, neighb_inds, x, kernel_points, weights, extent, bias, slope=0.1, influence='linear',
                    aggregation='sum', rev=None):
    """LeakyReLU(KPConv(x) + bias): KPConv + the bias/activation that follows it in every block
    (reference blocks.py:594-598, 668-676)."""
    if kpconv_mode(influence, aggregation) != 0:
        return bias_act(kpconv(q_pts, s_pts, neighb_inds, x, kernel_points, weights, extent, influence, aggregation),
                        bias, slope=slope)
    q_pts, s_pts, x = _f32(q_pts, "q_pts"), _f32(s_pts, "s_pts"), _f32(x, "x")
    Nq, H = int(q_pts.shape[0]), int(neighb_inds.shape[1])
    K, Cin = int(weights.shape[0]), int(weights.shape[1])
    if _takes_gemm_path(Nq, Cin) and s_pts.shape[0] > 0 and \
            _native.lib().d3f_kpconv_grad_input_supported(Cin, K, H, int(s_pts.shape[0])):
        idx = _i32(neighb_inds, "neighb_inds")
        kp, w = _f32(kernel_points, "kernel_points"), _f32(weights, "weights")
        if x.shape[0] != s_pts.shape[0] or x.shape[1] != w.shape[1] or idx.shape[0] != q_pts.shape[0]:
            raise RuntimeError("KPConv: inconsistent shapes q%s s%s idx%s x%s W%s" % (
                tuple(q_pts.shape), tuple(s_pts.shape), tuple(idx.shape), tuple(x.shape), tuple(w.shape)))
        b = _f32(bias, "bias") if bias is not None else None
        rev = reverse_table_of(neighb_inds, Nq, H, int(s_pts.shape[0]), rev) if x.requires_grad else None
        return _KPConvGemmBiasActFn.apply(q_pts, s_pts, idx, x, kp, w, b, float(extent), float(slope), rev,
                                          _ready_supports(x, s_pts))
    return bias_act(kpconv(q_pts, s_pts, neighb_inds, x, kernel_points, weights, extent, rev=rev), bias, slope=slope)


KP_INFLUENCES = {'linear': 0, 'constant': 1, 'gaussian': 2}
KP_AGGREGATIONS = {'sum': 0, 'closest': 4}


def kpconv_mode(influence='linear', aggregation='sum'):
    """Mode word of the C ABI; the reference's error messages for unknown names (blocks.py:344,352)."""
    if influence not in KP_INFLUENCES:
        raise ValueError('Unknown influence function type (config.KP_influence)')
    if aggregation not in KP_AGGREGATIONS:
        raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
    return KP_INFLUENCES[influence] | KP_AGGREGATIONS[aggregation]


class _KPConvModesFn(torch.autograd.Function):
    """KPConv with a non-default influence / aggregation mode (blocks.py:327-352): mode-aware aggregation and
    grad-input kernels of the general path, contraction with the kernel weights by library GEMMs."""

    @staticmethod
    def forward(ctx, q_pts, s_pts, idx, x, kp, weights, extent, mode):
        L = _native.lib()
        Nq, Ns, H = int(q_pts.shape[0]), int(s_pts.shape[0]), int(idx.shape[1])
        K, Cin, Cout = (int(v) for v in weights.shape)
        wf = torch.empty((Nq, K * Cin), dtype=torch.float32, device=x.device)
        nn_ = torch.empty((Nq,), dtype=torch.float32, device=x.device)
        with _region("kpconv_modes_fwd[Nq=%d,Cin=%d,H=%d,mode=%d]" % (Nq, Cin, H, mode), 4 * Nq * H * (4 + Cin)):
            _native.check(L.d3f_kpconv_aggregate_modes(_p(q_pts), Nq, _p(s_pts), Ns, _p(idx), H, _p(x), Cin, _p(kp), K,
                                                       extent, mode, _p(wf), _p(nn_), _stream()),
                          "d3f_kpconv_aggregate_modes")
        out = torch.mm(wf, weights.reshape(K * Cin, Cout)) / nn_[:, None]
        ctx.save_for_backward(q_pts, s_pts, idx, kp, weights, wf, nn_)
        ctx.extent, ctx.mode, ctx.gw_slot = extent, mode, _grad_slot(weights)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q_pts, s_pts, idx, kp, weights, wf, nn_ = ctx.saved_tensors
        L = _native.lib()
        Nq, Ns, H = int(q_pts.shape[0]), int(s_pts.shape[0]), int(idx.shape[1])
        K, Cin, Cout = (int(v) for v in weights.shape)
        g = grad_out.contiguous() / nn_[:, None]
        gx = gw = None
        if ctx.needs_input_grad[5]:
            gw = torch.mm(wf.t(), g).reshape(K, Cin, Cout)
            if ctx.gw_slot is not None:
                ctx.gw_slot.copy_(gw)
                gw = _adoptable(ctx.gw_slot)
        if ctx.needs_input_grad[3]:
            gwf = torch.mm(g, weights.reshape(K * Cin, Cout).t()).contiguous()
            gx = torch.empty((Ns, Cin), dtype=torch.float32, device=g.device)
            with _region("kpconv_modes_dx[Nq=%d,Cin=%d,H=%d,mode=%d]" % (Nq, Cin, H, ctx.mode), 8 * Nq * H * Cin):
                _native.check(L.d3f_kpconv_grad_input_modes(_p(q_pts), Nq, _p(s_pts), Ns, _p(idx), H, Cin, _p(kp), K,
                                                            ctx.extent, ctx.mode, _p(gwf), _p(gx), _stream()),
                              "d3f_kpconv_grad_input_modes")
        return None, None, None, gx, None, gw, None, None


def kpconv(q_pts, s_pts, neighb_inds, x, kernel_points, weights, extent, influence='linear', aggregation='sum',
           rev=None):
    """Rigid KPConv.  Shapes as KPConv.forward (blocks.py:237); 'linear' / 'sum' (the D3Feat configuration) runs on
    the fused kernels, the other modes of blocks.py:327-352 on the general path.  ``rev``: the table's ReverseTable
    (build_reverse_table; found on the table itself when its builder attached it): grad_x is then a gather."""
    mode = kpconv_mode(influence, aggregation)
    q_pts, s_pts, x = _f32(q_pts, "q_pts"), _f32(s_pts, "s_pts"), _f32(x, "x")
    idx = _i32(neighb_inds, "neighb_inds")
    kp, w = _f32(kernel_points, "kernel_points"), _f32(weights, "weights")
    if x.shape[0] != s_pts.shape[0] or x.shape[1] != w.shape[1] or idx.shape[0] != q_pts.shape[0]:
        raise RuntimeError("KPConv: inconsistent shapes q%s s%s idx%s x%s W%s" % (
            tuple(q_pts.shape), tuple(s_pts.shape), tuple(idx.shape), tuple(x.shape), tuple(w.shape)))
    if mode != 0:
        if q_pts.shape[0] == 0 or s_pts.shape[0] == 0:
            return x.new_zeros((q_pts.shape[0], w.shape[2])) + 0.0 * (x.sum() + w.sum())
        return _KPConvModesFn.apply(q_pts, s_pts, idx, x, kp, w, float(extent), mode)
    if x.requires_grad:
        rev = reverse_table_of(neighb_inds, int(q_pts.shape[0]), int(idx.shape[1]), int(s_pts.shape[0]), rev)
    else:
        rev = None
    return _KPConvFn.apply(q_pts, s_pts, idx, x, kp, w, float(extent), rev, _ready_supports(x, s_pts))


class _KPConvDeformAggFn(torch.autograd.Function):
    """wf, nn, min-distance bookkeeping of a deformable KPConv (csrc/kpconv_deform.hip); differentiable w.r.t. the
    features and the deformed kernel points."""

    @staticmethod
    def forward(ctx, q_pts, s_pts, idx, x, kp_def, extent, extent_sq, mode):
        L = _native.lib()
        Nq, Ns, H = int(q_pts.shape[0]), int(s_pts.shape[0]), int(idx.shape[1])
        K, Cin = int(kp_def.shape[1]), int(x.shape[1])
        wf = torch.empty((Nq, K, Cin), dtype=torch.float32, device=x.device)
        nn_ = torch.empty((Nq,), dtype=torch.float32, device=x.device)
        min_idx = torch.empty((Nq, K), dtype=torch.int32, device=x.device)
        with _region("kpconv_deform_fwd[Nq=%d,Cin=%d,H=%d]" % (Nq, Cin, H), 4 * Nq * H * (4 + Cin)):
            _native.check(L.d3f_kpconv_deform_aggregate(_p(q_pts), Nq, _p(s_pts), Ns, _p(idx), H, _p(x), Cin,
                                                        _p(kp_def), K, extent, extent_sq, mode, _p(wf), _p(nn_), None,
                                                        _p(min_idx), _stream()), "d3f_kpconv_deform_aggregate")
        ctx.save_for_backward(q_pts, s_pts, idx, x, kp_def)
        ctx.extent, ctx.extent_sq, ctx.mode = extent, extent_sq, mode
        ctx.mark_non_differentiable(nn_, min_idx)
        ctx.set_materialize_grads(False)
        return wf, nn_, min_idx

    @staticmethod
    def backward(ctx, gwf, _gnn, _gidx):
        if gwf is None:
            return (None,) * 8
        q_pts, s_pts, idx, x, kp_def = ctx.saved_tensors
        L = _native.lib()
        Nq, Ns, H = int(q_pts.shape[0]), int(s_pts.shape[0]), int(idx.shape[1])
        K, Cin = int(kp_def.shape[1]), int(x.shape[1])
        gx = torch.empty((Ns, Cin), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[3] else None
        gkp = torch.empty((Nq, K, 3), dtype=torch.float32, device=x.device) if ctx.needs_input_grad[4] else None
        if gx is None and gkp is None:
            return (None,) * 8
        gwf = gwf.contiguous()
        with _region("kpconv_deform_bwd[Nq=%d,Cin=%d,H=%d]" % (Nq, Cin, H), 12 * Nq * H * Cin):
            _native.check(L.d3f_kpconv_deform_grad(_p(q_pts), Nq, _p(s_pts), Ns, _p(idx), H, _p(x), Cin, _p(kp_def), K,
                                                   ctx.extent, ctx.extent_sq, ctx.mode, _p(gwf), _p(gx), _p(gkp),
                                                   _stream()), "d3f_kpconv_deform_grad")
        return None, None, None, gx, gkp, None, None, None


def kpconv_deformable(q_pts, s_pts, neighb_inds, x, kernel_points, weights, extent, offsets, modulations=None,
                      influence='linear', aggregation='sum'):
    """Deformable (and modulated) KPConv, the deformable=True branch of KPConv.forward (blocks.py:243-387).

    ``offsets`` [Nq,K,3] are the SCALED kernel-point shifts (unscaled offsets * KP_extent, :256), ``modulations``
    [Nq,K] the 2*sigmoid factors (:250) or None.  Returns (out [Nq,Cout], min_d2 [Nq,K], deformed_KP [Nq,K,3]) --
    the last two are what the reference keeps on the module for its fitting / repulsive regulariser
    (architectures.py:22-55); min_d2 carries its gradient to the offsets."""
    mode = kpconv_mode(influence, aggregation)
    q_pts, s_pts, x = _f32(q_pts, "q_pts"), _f32(s_pts, "s_pts"), _f32(x, "x")
    idx = _i32(neighb_inds, "neighb_inds")
    kp, w = _f32(kernel_points, "kernel_points"), _f32(weights, "weights")
    K, Cin, Cout = (int(v) for v in w.shape)
    Nq = int(q_pts.shape[0])
    if x.shape[0] != s_pts.shape[0] or x.shape[1] != Cin or idx.shape[0] != Nq or tuple(offsets.shape) != (Nq, K, 3):
        raise RuntimeError("deformable KPConv: inconsistent shapes q%s s%s idx%s x%s W%s offsets%s" % (
            tuple(q_pts.shape), tuple(s_pts.shape), tuple(idx.shape), tuple(x.shape), tuple(w.shape),
            tuple(offsets.shape)))
    deformed = offsets + kp                                              # blocks.py:287
    extent = float(extent)
    extent_sq = float(np.float32(extent ** 2))                           # the float32 scalar of `sq_distances < ext**2`
    wf, nn_, min_idx = _KPConvDeformAggFn.apply(q_pts, s_pts, idx, x, deformed.contiguous(), extent, extent_sq, mode)
    # min over the neighbors of d2 (blocks.py:301), re-evaluated at the arg-min support so that autograd carries its
    # gradient to the offsets (the selection itself has none); the shadow support sits at 1e6 like the reference's
    s_pad = torch.cat((s_pts, torch.zeros_like(s_pts[:1, :]) + 1e6), 0)
    nearest = s_pad[min_idx.long()] - q_pts.unsqueeze(1)                 # [Nq,K,3]
    min_d2 = torch.sum((nearest - deformed) ** 2, dim=2)
    if modulations is not None:
        wf = wf * modulations.unsqueeze(2)                               # :365-366
    out = torch.mm(wf.reshape(Nq, K * Cin), w.reshape(K * Cin, Cout)) / nn_[:, None]
    return out, min_d2, deformed


# ---------------------------------------------------------------------------------------------------------------
class GradHolder(object):
    __slots__ = ("tensor", "closed")

    def __init__(self):
        self.tensor, self.closed = None, False

    def deposit(self, g):
        """True when `g` was taken (the caller then returns None as its gradient)."""
        if self.closed or self.tensor is not None or g is None:
            return False
        self.tensor = g
        return True

    def collect(self):
        self.closed = True
        g, self.tensor = self.tensor, None
        return g


class _GradTapFn(torch.autograd.Function):
    """Identity whose backward deposits the gradient (the identity shortcut of a bottleneck block)."""

    @staticmethod
    def forward(ctx, x, holder):
        ctx.holder = holder
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (None if ctx.holder.deposit(g) else g), None


def grad_tap(x, holder):
    return _GradTapFn.apply(x, holder) if holder is not None else x


def _add_deposited(holder, go, weight):
    """grad_x = go @ weight (+ the sibling branch's deposited gradient, accumulated by the GEMM itself)."""
    c = holder.collect() if holder is not None else None
    N, Cout, Cin = int(go.shape[0]), int(go.shape[1]), int(weight.shape[1])
    if go.is_contiguous() and weight.is_contiguous() and \
            (c is None or (c.is_contiguous() and c.dtype == go.dtype and c.shape == (N, Cin))) and \
            _own_gemm("unary_dx", go, weight, GEMM_NN, N, Cout, Cin, 0, None, None, c):
        return gemm_epilogue(go, weight, GEMM_NN, N, Cout, Cin, add=c)      # (out of place: safe beside queued operands)
    if c is None:
        return torch.mm(go, weight)
    if c.is_contiguous() and c.dtype == go.dtype and c.shape == (go.shape[0], weight.shape[1]):
        if _WG_GROUP is not None:
            # the deposited buffer may be a queued operand of a weight gradient that has not run yet (the masked gradient
            # of the block's last unary layer IS the shortcut's gradient): accumulate out of place -- on the row-streaming
            # kernel where it serves the shape (it reads `add` and writes a separate output anyway: no extra traffic; the
            # library's out-of-place addmm first copies c, 58 MB at level 0)
            N, Cout, Cin = int(go.shape[0]), int(go.shape[1]), int(weight.shape[1])
            L = _native.lib()
            if go.is_contiguous() and weight.is_contiguous() and L.d3f_linear_fused_supported(N, Cin, Cout):
                gx = torch.empty_like(c)
                _native.check(L.d3f_linear_grad_input(_p(go), _p(weight), N, Cin, Cout, _p(c), _p(gx), _stream()),
                              "d3f_linear_grad_input")
                return gx
            return torch.addmm(c, go, weight)
        return c.addmm_(go, weight)      # beta = 1, in place: the deposited buffer has no other reader left
    return torch.mm(go, weight).add_(c)


# ---------------------------------------------------------------------------------------------------------------
# Own f32-MFMA GEMM with the block's epilogue fused (csrc/gemm_epilogue.hip): the contractions of the wide / few-row
# layers -- KPConv's wf @ W (models/blocks.py:362-374) with / nn + bias + LeakyReLU, nn.Linear of the unary blocks
# (:481-541) with bias + residual + LeakyReLU, and their grad-input products
# ---------------------------------------------------------------------------------------------------------------
GEMM_NT, GEMM_NN = 0, 1
# False: every such product is a library GEMM followed by its epilogue launch (rounds 1-5; A/B measurements)
OWN_GEMM = True


def _al16(*ts):
    return all(t is None or t.data_ptr() % 16 == 0 for t in ts)


def gemm_epilogue_ok(x, w, mode, R, K, N, kblock=0, ldx=None, ldw=None, *others):
    """True when d3f_gemm_epilogue serves this product (shape, alignment, leading dimensions)."""
    if not OWN_GEMM or R < 1:
        return False
    ldx = K if ldx is None else ldx
    ldw = (K if mode == GEMM_NT else N) if ldw is None else ldw
    if (ldx | ldw) & 3 or not _al16(x, w, *others):
        return False
    return bool(_native.lib().d3f_gemm_epilogue_supported(int(R), int(K), int(N), int(mode), int(kblock)))


def gemm_epilogue(x, w, mode, R, K, N, kblock=0, ldx=None, ldw=None, row_div=None, bias1=None, add=None, bias2=None,
                  slope=1.0, zero_init=None, out=None):
    """out [R, N] = act(x [R, K] . B / row_div + bias1 + add + bias2) in one launch (two when the reduction is split).
    mode GEMM_NT: B = w [N, ldw] (y = x w^T); kblock > 0: w is [K / kblock][N][kblock] (KPConv weights read as the
    permuted matrix of the transposed-aggregation grad-input).  mode GEMM_NN: B = w [K, ldw].  ``out`` may be a
    row-strided view.  Raises RuntimeError when the product is not supported (callers ask gemm_epilogue_ok first)."""
    L = _native.lib()
    ldx = K if ldx is None else int(ldx)
    ldw = (K if mode == GEMM_NT else N) if ldw is None else int(ldw)
    if out is None:
        out = torch.empty((R, N), dtype=torch.float32, device=x.device)
    ldy = int(out.stride(0)) if out.dim() == 2 else N
    ldadd = int(add.stride(0)) if add is not None else 0
    nbytes = int(L.d3f_gemm_epilogue_ws_bytes(int(R), int(K), int(N)))
    ws = _ws(nbytes, x.device)
    zn = int(zero_init.numel()) if zero_init is not None else 0
    with _region("gemm_epilogue[R=%d,K=%d,N=%d,mode=%d]" % (R, K, N, mode), 4 * (R * K + K * N + R * N)):
        _native.check(L.d3f_gemm_epilogue(_p(x), ldx, _p(w), ldw, int(mode), int(kblock), int(R), int(K), int(N),
                                          _p(row_div), _p(bias1), _p(add), ldadd, _p(bias2), float(slope), _p(out), ldy,
                                          _p(zero_init), zn, _p(ws), nbytes, _stream()), "d3f_gemm_epilogue")
    return out


# which products go to the own kernel (the others stay library GEMMs).  Measured per shape against the library GEMM + its
# epilogue launch (profiles/gemm_epilogue_bench.py -> profiles/r06_gemm_epilogue_bench.txt) and per kind inside the 4 x 3
# step (profiles/calls/r06_gemm_kinds_ab.sh): the own kernel wins where the library's pick is poor or the fused epilogue
# is a large share -- KPConv contractions of the bottom levels (<= 1024 rows x 3840 / 7680: split reduction) and of the
# many-row levels -- and loses on the mid-size square-ish products of levels 2-3 and the decoder, where the library's
# shared-panel tiles reach 0.65-0.75 of the matrix rate.  Inside the step only the KPConv kind pays (+1.0 %); the unary
# kinds ("unary_fwd", "unary_dx": >= 16k rows by rule), "kpconv_dx", "kpconv_gw" and "decoder" measured 0 ... -1.4 %
# and stay library GEMMs (the kernel serves them all: tests/test_gpu_ops.py::test_gemm_epilogue_matches_float64).
OWN_GEMM_KINDS = {"kpconv_fwd"}
OWN_GEMM_RULES = True     # False: every product of an enabled kind (A/B measurements)
OWN_GEMM_TALL_ROWS = 1 << 30


def _own_gemm(kind, x, w, mode, R, K, N, kblock=0, ldx=None, ldw=None, *others):
    """Policy + capability: True when the product `kind` of this shape runs on d3f_gemm_epilogue."""
    if kind not in OWN_GEMM_KINDS:
        return False
    if OWN_GEMM_RULES:
        if kind == "kpconv_fwd" and not (R <= 1024 or 4096 <= R < 16384 or R >= OWN_GEMM_TALL_ROWS):
            return False       # (levels 1 and 3 of a 3-pair stack: 43 vs 33 + 5 us and 45 vs 38 + 5 us inside the step)
        if kind in ("unary_fwd", "unary_dx") and R < 16384:
            return False
        if kind == "kpconv_dx" and R > 1024:
            return False
    return gemm_epilogue_ok(x, w, mode, R, K, N, kblock, ldx, ldw, *others)


# ---------------------------------------------------------------------------------------------------------------
# 1x1 convolution of the unary blocks (models/blocks.py:481-515): library GEMMs for y = x W^T and grad_x, own
# reduction-parallel kernel for grad_W (tall-skinny: the reduction runs over the points)
# ---------------------------------------------------------------------------------------------------------------
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, holder=None, deposit=None):
        ctx.save_for_backward(x, weight)
        ctx.gw_slot = _grad_slot(weight)
        ctx.holder, ctx.dep = holder, deposit
        return torch.mm(x, weight.t())

    @staticmethod
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        go = grad_out.contiguous()
        gx = _add_deposited(ctx.holder, go, weight) if ctx.needs_input_grad[0] else None
        if gx is not None and ctx.dep is not None and ctx.dep.deposit(gx):
            gx = None
        gw = None
        if ctx.needs_input_grad[1]:
            L = _native.lib()
            N, Cin, Cout = int(x.shape[0]), int(x.shape[1]), int(weight.shape[0])
            slot = ctx.gw_slot
            if N >= _SPLITK_MIN_ROWS and L.d3f_linear_grad_weight_supported(N, Cin, Cout):
                gw = slot if slot is not None else torch.empty_like(weight)
                _grad_weight_atb(x, go, N, Cin, Cout, gw, None, 0, None, None, "linear_dw")
            else:
                gw = slot if slot is not None else torch.empty_like(weight)
                if not _queue_small_weight_grad(x, go, N, Cin, Cout, gw):
                    torch.mm(go.t(), x, out=gw)
        return gx, (_adoptable(gw, ctx.gw_slot) if gw is not None else None), None, None


class _LinearBiasActFn(torch.autograd.Function):
    """act(x W^T + b1 + add + b2) in ONE launch (d3f_linear_bias_act_forward) for the many-row / narrow layers; the
    backward is the epilogue's backward kernel followed by grad_x = g W (row-streaming kernel) and grad_W = g^T x
    (reduction-parallel kernel)."""

    @staticmethod
    def forward(ctx, x, weight, b1, add, b2, slope, holder=None, deposit=None):
        L = _native.lib()
        ctx.holder, ctx.dep = holder, deposit
        N, Cin, Cout = int(x.shape[0]), int(x.shape[1]), int(weight.shape[0])
        out = torch.empty((N, Cout), dtype=torch.float32, device=x.device)
        nb = int(b1 is not None and ctx.needs_input_grad[2]) + int(b2 is not None and ctx.needs_input_grad[4])
        gbuf = torch.empty((nb, Cout), dtype=torch.float32, device=x.device) if nb else None
        with _region("linear_fused_fwd[N=%d,Cin=%d,Cout=%d]" % (N, Cin, Cout), 4 * N * (Cin + Cout) + 4 * Cin * Cout):
            _native.check(L.d3f_linear_bias_act_forward(_p(x), _p(weight), N, Cin, Cout, _p(b1), _p(add), _p(b2),
                                                        float(slope), _p(out), _p(gbuf), nb * Cout, _stream()),
                          "d3f_linear_bias_act_forward")
        ctx.save_for_backward(x, weight, out)
        ctx.gbuf = gbuf
        ctx.gw_slot = _grad_slot(weight)
        ctx.slope = float(slope)
        ctx.has = (b1 is not None, add is not None, b2 is not None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, out = ctx.saved_tensors
        L = _native.lib()
        N, Cin, Cout = int(x.shape[0]), int(x.shape[1]), int(weight.shape[0])
        go = grad_out.contiguous()
        want1 = ctx.has[0] and ctx.needs_input_grad[2]
        want2 = ctx.has[2] and ctx.needs_input_grad[4]
        g1 = g2 = None
        pre = 0
        if want1 or want2:
            gbuf, pre = ctx.gbuf, 1
            ctx.gbuf = None
            if gbuf is None:
                gbuf, pre = torch.empty((int(want1) + int(want2), Cout), dtype=torch.float32, device=go.device), 0
            rows = list(gbuf.unbind(0))
            g1 = rows.pop(0) if want1 else None
            g2 = rows.pop(0) if want2 else None
        first, second = (g1, g2) if g1 is not None else (g2, None)
        atb = ctx.needs_input_grad[1] and bool(L.d3f_linear_grad_weight_supported(N, Cin, Cout))
        bias_part, bias_blocks = None, 0
        if ctx.slope == 1.0 and not (want1 or want2):
            gm = go
        else:
            gm = go if ctx.slope == 1.0 else torch.empty_like(go)
            bias_part, bias_blocks = _epilogue_backward(go, out, ctx.slope, N, Cout, gm if ctx.slope != 1.0 else None,
                                                        first, second, pre, None, _FOLD_BIAS_SUM and atb)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            c = ctx.holder.collect() if ctx.holder is not None else None
            if c is not None and not (c.is_contiguous() and c.shape == gx.shape and c.dtype == gx.dtype):
                c = c.contiguous().float().reshape(gx.shape)
            with _region("linear_dx[N=%d,Cin=%d,Cout=%d]" % (N, Cin, Cout), 4 * N * (Cin + Cout) + 4 * Cin * Cout):
                _native.check(L.d3f_linear_grad_input(_p(gm), _p(weight), N, Cin, Cout, _p(c), _p(gx), _stream()),
                              "d3f_linear_grad_input")
            if ctx.dep is not None and ctx.dep.deposit(gx):
                gx = None
        if ctx.needs_input_grad[1]:
            slot = ctx.gw_slot
            if atb:
                gw = slot if slot is not None else torch.empty_like(weight)
                _grad_weight_atb(x, gm, N, Cin, Cout, gw, bias_part, bias_blocks, first, second, "linear_dw")
                bias_part = None
            else:
                gw = slot if slot is not None else torch.empty_like(weight)
                if not _queue_small_weight_grad(x, gm, N, Cin, Cout, gw):
                    torch.mm(gm.t(), x, out=gw)
        if bias_part is not None:      # (not reached: the fold implies the A^T B path)
            _native.check(L.d3f_bias_sum(_p(bias_part), bias_blocks, Cout, _p(first), _p(second), _stream()),
                          "d3f_bias_sum")
        return (gx, (_adoptable(gw, ctx.gw_slot) if gw is not None else None), g1,
                gm if ctx.has[1] and ctx.needs_input_grad[3] else None, g2, None, None, None)


class _LinearPairBiasActFn(torch.autograd.Function):
    """act(x1 W1^T + x2 W2^T + b1a + b1b + b2a + b2b) in ONE launch (d3f_linear_pair_bias_act_forward): the last unary
    block of a bottleneck and its shortcut unary (reference blocks.py:658-686) -- the shortcut tensor is never formed.
    Backward: ONE epilogue pass (masked gradient + the column sums all four biases share), grad_x1 = g W1 and
    grad_x2 = g W2 on the row-streaming kernel, both weight gradients queued / launched on the reduction-parallel kernel."""

    @staticmethod
    def forward(ctx, x1, w1, b1a, b1b, x2, w2, b2a, b2b, slope, deposit2=None):
        L = _native.lib()
        ctx.dep2 = deposit2
        N, C1, C2, Cout = int(x1.shape[0]), int(x1.shape[1]), int(x2.shape[1]), int(w1.shape[0])
        out = torch.empty((N, Cout), dtype=torch.float32, device=x1.device)
        ctx.want = tuple(b is not None and ctx.needs_input_grad[i] for b, i in ((b1a, 2), (b1b, 3), (b2a, 6), (b2b, 7)))
        nb = 2 if any(ctx.want) else 0
        gbuf = torch.empty((nb, Cout), dtype=torch.float32, device=x1.device) if nb else None
        with _region("linear_pair_fwd[N=%d,Cin=%d|%d,Cout=%d]" % (N, C1, C2, Cout),
                     4 * N * (C1 + C2 + Cout) + 4 * (C1 + C2) * Cout):
            _native.check(L.d3f_linear_pair_bias_act_forward(_p(x1), _p(w1), C1, _p(x2), _p(w2), C2, N, Cout, _p(b1a),
                                                             _p(b1b), _p(b2a), _p(b2b), float(slope), _p(out), _p(gbuf),
                                                             nb * Cout, _stream()), "d3f_linear_pair_bias_act_forward")
        ctx.save_for_backward(x1, w1, x2, w2, out)
        ctx.gbuf, ctx.slope = gbuf, float(slope)
        ctx.slots = (_grad_slot(w1), _grad_slot(w2))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x1, w1, x2, w2, out = ctx.saved_tensors
        L = _native.lib()
        N, C1, C2, Cout = int(x1.shape[0]), int(x1.shape[1]), int(x2.shape[1]), int(w1.shape[0])
        go = grad_out.contiguous()
        first = second = None
        pre = 0
        if any(ctx.want):
            gbuf, pre = ctx.gbuf, 1
            ctx.gbuf = None
            if gbuf is None:
                gbuf, pre = torch.empty((2, Cout), dtype=torch.float32, device=go.device), 0
            first, second = gbuf.unbind(0)
        need_w1, need_w2 = ctx.needs_input_grad[1], ctx.needs_input_grad[5]
        atb1 = need_w1 and bool(L.d3f_linear_grad_weight_supported(N, C1, Cout))
        atb2 = need_w2 and bool(L.d3f_linear_grad_weight_supported(N, C2, Cout))
        fold = _FOLD_BIAS_SUM and atb1 and atb2 and _WG_GROUP is not None
        gm = torch.empty_like(go) if ctx.slope != 1.0 else go
        bias_part, bias_blocks = None, 0
        if ctx.slope != 1.0 or first is not None:
            bias_part, bias_blocks = _epilogue_backward(go, out, ctx.slope, N, Cout, gm if ctx.slope != 1.0 else None, first,
                                                        second, pre, None, fold and first is not None)
        third = fourth = None
        if first is not None:
            third, fourth = torch.empty_like(first), torch.empty_like(first)
        gx1 = gx2 = None
        if ctx.needs_input_grad[0]:
            gx1 = torch.empty_like(x1)
            _native.check(L.d3f_linear_grad_input(_p(gm), _p(w1), N, C1, Cout, None, _p(gx1), _stream()),
                          "d3f_linear_grad_input")
        if ctx.needs_input_grad[4]:
            gx2 = torch.empty_like(x2)
            _native.check(L.d3f_linear_grad_input(_p(gm), _p(w2), N, C2, Cout, None, _p(gx2), _stream()),
                          "d3f_linear_grad_input")
            if ctx.dep2 is not None and ctx.dep2.deposit(gx2):
                gx2 = None
        gw1 = gw2 = None
        if need_w1:
            gw1 = ctx.slots[0] if ctx.slots[0] is not None else torch.empty_like(w1)
            if atb1:
                _grad_weight_atb(x1, gm, N, C1, Cout, gw1, bias_part, bias_blocks, first, second, "linear_dw")
            else:
                torch.mm(gm.t(), x1, out=gw1)
        if need_w2:
            gw2 = ctx.slots[1] if ctx.slots[1] is not None else torch.empty_like(w2)
            if atb2:
                # (the second problem finishes the same bias partials into the shortcut's two bias gradients)
                _grad_weight_atb(x2, gm, N, C2, Cout, gw2, bias_part, bias_blocks, third, fourth, "linear_dw")
            else:
                torch.mm(gm.t(), x2, out=gw2)
        if first is not None and bias_part is None:     # (no fold: the epilogue pass finished first / second itself)
            third.copy_(first)
            fourth.copy_(first)
        elif first is not None and not (atb1 and atb2):  # (fold implies both problems are queued: not reached)
            _native.check(L.d3f_bias_sum(_p(bias_part), bias_blocks, Cout, _p(first), _p(second), _stream()), "d3f_bias_sum")
            third.copy_(first)
            fourth.copy_(first)
        g = [first if ctx.want[0] else None, second if ctx.want[1] else None, third if ctx.want[2] else None,
             fourth if ctx.want[3] else None]
        return (gx1, (_adoptable(gw1, ctx.slots[0]) if gw1 is not None else None), g[0], g[1],
                gx2, (_adoptable(gw2, ctx.slots[1]) if gw2 is not None else None), g[2], g[3], None, None)


# False: unary2 and the shortcut unary of a bottleneck stay two launches (A/B measurements)
FUSE_UNARY_PAIR = True


def linear_pair_supported(N, C1, C2, Cout, *tensors):
    """Whether linear_pair_bias_act serves a pair of these dimensions (the level-0 bottleneck's: (32 | 64) -> 128 from 4096
    rows); ``tensors``: operands already at hand, checked for device / dtype / layout."""
    if not FUSE_UNARY_PAIR:
        return False
    for t in tensors:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0):
            return False
    L = _native.lib()
    N, C1, C2, Cout = int(N), int(C1), int(C2), int(Cout)
    return bool(L.d3f_linear_pair_supported(N, C1, C2, Cout)) and bool(L.d3f_linear_fused_supported(N, C1, Cout)) \
        and bool(L.d3f_linear_fused_supported(N, C2, Cout))


def linear_pair_bias_act(x1, w1, b1a, b1b, x2, w2, b2a, b2b, slope=0.1, grad_deposit2=None):
    """act(x1 @ w1^T + b1a + b1b + x2 @ w2^T + b2a + b2b): unary2(x1) + unary_shortcut(x2) + LeakyReLU of a bottleneck
    block in one launch (reference blocks.py:658-686).  grad_deposit2: x2's gradient is handed to a sibling branch
    (GradHolder) instead of being returned."""
    f = lambda t, n: _f32(t, n) if t is not None else None
    if not linear_pair_supported(x1.shape[0], x1.shape[1], x2.shape[1], w1.shape[0], x1, x2, w1, w2):
        raise RuntimeError("linear_pair_bias_act: unsupported operands x1%s x2%s w1%s w2%s" % (
            tuple(x1.shape), tuple(x2.shape), tuple(w1.shape), tuple(w2.shape)))
    return _LinearPairBiasActFn.apply(_f32(x1, "x1"), _f32(w1, "w1"), f(b1a, "bias"), f(b1b, "bias"), _f32(x2, "x2"),
                                      _f32(w2, "w2"), f(b2a, "bias"), f(b2b, "bias"), float(slope), grad_deposit2)


# The bias gradient's second pass rides in the weight gradient's second-stage launch (csrc/linear.hip,
# atb_reduce_bias_kernel) whenever both are two-pass forms: N >= 4096 rows for the epilogue's backward and the A^T B
# kernel for grad_W.  False: two launches as before (experiments).
_FOLD_BIAS_SUM = True


def _epilogue_backward(go, out, slope, N, C, gm, first, second, pre, row_div, fold):
    """grad through act(. + biases): masked gradient into ``gm`` (None: not wanted) and the bias gradient(s).  With
    ``fold`` only the first pass runs; returns (partials, blocks) for d3f_linear_grad_weight_bias, else (None, 0)."""
    L = _native.lib()
    wsb, nws = _bias_bwd_ws(N, C, go.device) if first is not None else (None, 0)
    nblk = int(L.d3f_bias_act_backward_blocks(N, C)) if (fold and first is not None and wsb is not None) else 0
    if nblk > 0:
        _native.check(L.d3f_bias_act_backward_partial(_p(go), _p(out), float(slope), N, C, _p(gm), _p(row_div), _p(wsb),
                                                      nws, _stream()), "d3f_bias_act_backward_partial")
        return wsb, nblk
    _native.check(L.d3f_bias_act_backward(_p(go), _p(out), float(slope), N, C, _p(gm), _p(first), _p(second),
                                          pre if first is not None else 0, _p(row_div), _p(wsb), nws, _stream()),
                  "d3f_bias_act_backward")
    return None, 0


class WeightGradGroup(object):
    """The weight gradients of ONE backward stage, queued by the autograd nodes and computed together when the stage
    ends (d3f_linear_grad_weight_group: one launch over every problem's tasks + one that sums all slabs and finishes
    the queued bias gradients).  A weight gradient has no consumer before the optimizer (reference trainer.py:103-111),
    so nothing orders it between the grad-input kernels; queued, the 27 problems of a stacked step share one launch's
    ramp-up and tail and the memory-bound ones run beside the matrix-bound ones.

        with ops.weight_grad_group():
            torch.autograd.backward(loss, ...)
        # <- here every p.grad / flat-buffer slot is final

    The queue keeps every operand alive until the flush (under hipGraph capture a freed block would be handed to a
    later allocation of the same capture) and the nodes never modify a queued operand in place (``_add_deposited``).
    Backward runs on autograd's worker thread: the active group is a module global, one backward at a time."""

    def __init__(self):
        self.problems, self.keep = [], []

    def add(self, x, gm, N, Cin, Cout, gw, bias_part, bias_blocks, first, second):
        ldw = int(gw.stride(0)) if gw.dim() == 2 else int(Cin)
        if gw.dim() == 2 and (gw.stride(1) != 1 or gw.shape[0] != Cout or gw.shape[1] != Cin):
            raise RuntimeError("weight-gradient target of shape %s (strides %s) for a [%d, %d] gradient" % (
                tuple(gw.shape), tuple(gw.stride()), Cout, Cin))
        self.problems.append((_p(x), _p(gm), _p(gw), int(N), int(Cin), int(Cout), ldw,
                              _p(bias_part), int(bias_blocks) if bias_part is not None else 0,
                              int(first.numel()) if bias_part is not None else 0,
                              _p(first) if bias_part is not None else None,
                              _p(second) if bias_part is not None else None))
        # held through FRESH tensor objects on the same storage: autograd adopts an incoming gradient as p.grad without a
        # copy only while nobody else holds the tensor object -- a second reference to gw / first / second would make it
        # clone them here, before the flush has written them
        self.keep.append(tuple(t.detach() if t is not None else None for t in (x, gm, gw, bias_part, first, second)))

    def flush(self):
        """Launch the queued problems on the current stream and empty the queue."""
        n = len(self.problems)
        if n == 0:
            return 0
        L = _native.lib()
        arr = (_native.AtbProblem * n)()
        for q, t in zip(arr, self.problems):
            (q.x, q.grad_out, q.grad_w, q.N, q.Cin, q.Cout, q.ldw, q.bias_part, q.bias_blocks, q.bias_cols,
             q.grad_bias, q.grad_bias2) = t
        nbytes = int(L.d3f_linear_grad_weight_group_ws_bytes(arr, n))
        if nbytes == 0:
            raise RuntimeError("d3f_linear_grad_weight_group: unsupported problem in the queue")
        dev = self.keep[0][0].device
        ws = _ws(nbytes, dev)
        flops = sum(2 * t[3] * t[4] * t[5] for t in self.problems)
        with _region("weight_grad_group[n=%d,GFLOP=%.2f]" % (n, flops * 1e-9),
                     sum(4 * t[3] * (t[4] + t[5]) + 4 * t[4] * t[5] for t in self.problems)):
            _native.check(L.d3f_linear_grad_weight_group(arr, n, _p(ws), nbytes, _stream()),
                          "d3f_linear_grad_weight_group")
        self.problems, self.keep = [], []
        return n


_WG_GROUP = None
# False: every weight gradient is launched where autograd reaches it (rounds 1-5; experiments and A/B tests)
GROUP_WEIGHT_GRADS = True


class weight_grad_group(object):
    """Context manager around ONE backward pass (or one stage of a split backward): see WeightGradGroup."""

    def __enter__(self):
        global _WG_GROUP
        if _WG_GROUP is not None:
            raise RuntimeError("weight_grad_group: a backward stage is already collecting weight gradients")
        self.group = WeightGradGroup() if GROUP_WEIGHT_GRADS else None
        _WG_GROUP = self.group
        return self.group

    def __exit__(self, exc_type, exc, tb):
        global _WG_GROUP
        g, _WG_GROUP = _WG_GROUP, None
        if g is not None and exc_type is None:
            g.flush()
        return False


def _queue_small_weight_grad(x, gm, N, Cin, Cout, gw):
    """Few-row weight gradients (the bottom levels: 462 / 1713 rows against 512 ... 7680 x 512 outputs) are library
    GEMMs when launched where autograd reaches them; inside a weight_grad_group they join the stage's grouped launch
    (undivided reduction, one task per 64 x 64 output block, written straight to the target).  True when queued."""
    g = _WG_GROUP
    if g is None or not GROUP_SMALL_ROW_GRADS or N < 1 or Cin % 16 or Cout % 16:
        return False
    if x.data_ptr() % 16 or gm.data_ptr() % 16 or not x.is_contiguous() or not gm.is_contiguous():
        return False
    g.add(x, gm, N, Cin, Cout, gw, None, 0, None, None)
    return True


# False: weight gradients below _SPLITK_MIN_ROWS rows stay library GEMMs even inside a group (A/B measurements)
GROUP_SMALL_ROW_GRADS = True


def _grad_weight_atb(x, gm, N, Cin, Cout, gw, bias_part, bias_blocks, first, second, label):
    """grad_W [Cout, Cin] = gm^T x on the reduction-parallel kernels; with ``bias_part`` the second stage also sums the
    bias partials into ``first`` (/ ``second``).  Inside a weight_grad_group the problem is only queued."""
    L = _native.lib()
    g = _WG_GROUP
    if g is not None and x.data_ptr() % 16 == 0 and gm.data_ptr() % 16 == 0:
        g.add(x, gm, N, Cin, Cout, gw, bias_part, bias_blocks, first, second)
        return
    if gw.dim() == 2 and not gw.is_contiguous():      # (a column block of a wider matrix: only the group writes in place)
        tmp = torch.empty((Cout, Cin), dtype=torch.float32, device=x.device)
        _grad_weight_atb(x, gm, N, Cin, Cout, tmp, bias_part, bias_blocks, first, second, label)
        gw.copy_(tmp)
        return
    nbytes = L.d3f_linear_grad_weight_ws_bytes(N, Cin, Cout)
    ws = _ws(nbytes, x.device)
    with _region("%s[N=%d,Cin=%d,Cout=%d]" % (label, N, Cin, Cout), 4 * N * (Cin + Cout) + 4 * Cin * Cout):
        if bias_part is not None:
            _native.check(L.d3f_linear_grad_weight_bias(_p(x), _p(gm), N, Cin, Cout, _p(gw), _p(ws), nbytes,
                                                        _p(bias_part), int(bias_blocks), int(first.numel()),
                                                        _p(first), _p(second), _stream()), "d3f_linear_grad_weight_bias")
        else:
            _native.check(L.d3f_linear_grad_weight(_p(x), _p(gm), N, Cin, Cout, _p(gw), _p(ws), nbytes, _stream()),
                          "d3f_linear_grad_weight")


class _LinearLibBiasActFn(torch.autograd.Function):
    """act(x W^T + b1 + add + b2) as library GEMM + one epilogue launch, as ONE autograd node (round 5; it used to be
    _LinearFn followed by _BiasActFn): the backward runs the epilogue's backward, grad_x = g W (library; a sibling
    branch's deposited gradient accumulated by the GEMM) and grad_W = g^T x, and because both live in one node the bias
    gradient's second pass is folded into grad_W's second-stage launch.  ``pack``: see bias_act."""

    @staticmethod
    def forward(ctx, x, weight, b1, add, b2, slope, holder=None, deposit=None, pack=None):
        L = _native.lib()
        ctx.holder, ctx.dep = holder, deposit
        N, C = int(x.shape[0]), int(weight.shape[0])
        nb = int(b1 is not None and ctx.needs_input_grad[2]) + int(b2 is not None and ctx.needs_input_grad[4])
        gbuf = torch.empty((nb, C), dtype=torch.float32, device=x.device) if nb else None
        ctx.gbuf, ctx.slope = gbuf, float(slope)
        ctx.has = (b1 is not None, add is not None, b2 is not None)
        ctx.gw_slot = _grad_slot(weight)
        Cin = int(x.shape[1])
        if pack is None and x.is_contiguous() and weight.is_contiguous() and (add is None or add.is_contiguous()) and \
                _own_gemm("unary_fwd", x, weight, GEMM_NT, N, Cin, C, 0, None, None, b1, add, b2):
            # x W^T + both biases + residual + LeakyReLU in one launch (csrc/gemm_epilogue.hip)
            out = gemm_epilogue(x, weight, GEMM_NT, N, Cin, C, bias1=b1, add=add, bias2=b2, slope=slope, zero_init=gbuf)
            ctx.save_for_backward(x, weight, out)
            return out
        raw = torch.mm(x, weight.t())
        out = torch.empty_like(raw)
        if pack is not None:
            s_pts, want_clear = pack
            spack = torch.empty(16 * N, dtype=torch.uint8, device=x.device)
            gx_clear = torch.empty_like(raw) if want_clear else None
            _native.check(L.d3f_bias_act_forward_pack(
                _p(raw), _p(b1), _p(add), _p(b2), float(slope), N, C, _p(out), _p(gbuf), nb * C, None, None, 0, 0,
                _p(s_pts), _p(spack), _p(gx_clear), _stream()), "d3f_bias_act_forward_pack")
            ctx.save_for_backward(x, weight, out)
            ctx.mark_non_differentiable(spack)
            if gx_clear is not None:
                ctx.mark_non_differentiable(gx_clear)
            ctx.set_materialize_grads(False)
            return out, spack, gx_clear
        _native.check(L.d3f_bias_act_forward(_p(raw), _p(b1), _p(add), _p(b2), float(slope), N, C, _p(out), _p(gbuf),
                                             nb * C, None, None, 0, 0, _stream()), "d3f_bias_act_forward")
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad_out, *_unused):
        x, weight, out = ctx.saved_tensors
        none = (None,) * 9
        if grad_out is None:
            return none
        L = _native.lib()
        N, Cin, Cout = int(x.shape[0]), int(x.shape[1]), int(weight.shape[0])
        go = grad_out.contiguous()
        want1 = ctx.has[0] and ctx.needs_input_grad[2]
        want2 = ctx.has[2] and ctx.needs_input_grad[4]
        g1 = g2 = None
        pre = 0
        if want1 or want2:
            gbuf, pre = ctx.gbuf, 1
            ctx.gbuf = None
            if gbuf is None:
                gbuf, pre = torch.empty((int(want1) + int(want2), Cout), dtype=torch.float32, device=go.device), 0
            rows = list(gbuf.unbind(0))
            g1 = rows.pop(0) if want1 else None
            g2 = rows.pop(0) if want2 else None
        first, second = (g1, g2) if g1 is not None else (g2, None)
        need_w = ctx.needs_input_grad[1]
        atb = need_w and N >= _SPLITK_MIN_ROWS and bool(L.d3f_linear_grad_weight_supported(N, Cin, Cout))
        bias_part, bias_blocks = None, 0
        identity = ctx.slope == 1.0
        if identity and first is None:
            gm = go
        else:
            gm = go if identity else torch.empty_like(go)
            bias_part, bias_blocks = _epilogue_backward(go, out, ctx.slope, N, Cout, None if identity else gm, first,
                                                        second, pre, None, _FOLD_BIAS_SUM and atb)
        gx = _add_deposited(ctx.holder, gm, weight) if ctx.needs_input_grad[0] else None
        if gx is not None and ctx.dep is not None and ctx.dep.deposit(gx):
            gx = None
        gw = None
        if need_w:
            slot = ctx.gw_slot
            if atb:
                gw = slot if slot is not None else torch.empty_like(weight)
                _grad_weight_atb(x, gm, N, Cin, Cout, gw, bias_part, bias_blocks, first, second, "linear_dw")
                bias_part = None
            else:
                gw = slot if slot is not None else torch.empty_like(weight)
                if not _queue_small_weight_grad(x, gm, N, Cin, Cout, gw):
                    torch.mm(gm.t(), x, out=gw)
        if bias_part is not None:      # (not reached: fold implies the A^T B path)
            _native.check(L.d3f_bias_sum(_p(bias_part), bias_blocks, Cout, _p(first), _p(second), _stream()),
                          "d3f_bias_sum")
        return (gx, (_adoptable(gw, ctx.gw_slot) if gw is not None else None), g1,
                gm if ctx.has[1] and ctx.needs_input_grad[3] else None, g2, None, None, None, None)


class _UpsampleLinearFn(torch.autograd.Function):
    """Decoder unary block on [nearest_upsample(x_coarse) | skip] (reference architectures.py:311-314 + blocks.py:481):
        act( [x_c[idx] | skip] W^T + b )  =  act( (x_c W1^T)[idx] + skip W2^T + b ),   W = [W1 | W2].
    A row gather commutes with a row-wise linear map, so the wide half of the product (the 2048/1024/512-channel
    upsampled features) is computed on the COARSE rows -- 3.3x .. 4.8x fewer -- and upsampled inside the epilogue; the
    backward pools the masked gradient back to the coarse rows before the two GEMMs that involve W1."""

    @staticmethod
    def forward(ctx, xc, idx, skip, weight, b1, b2, slope, skip_deposit=None):
        L = _native.lib()
        ctx.skip_dep = skip_deposit
        Nc, Cc = int(xc.shape[0]), int(xc.shape[1])
        N, Cs = int(skip.shape[0]), int(skip.shape[1])
        Cout, H = int(weight.shape[0]), int(idx.shape[1])
        w1, w2 = weight[:, :Cc], weight[:, Cc:]
        ldw = int(weight.stride(0))
        if xc.is_contiguous() and _own_gemm("decoder", xc, w1, GEMM_NT, Nc, Cc, Cout, 0, None, ldw):
            t = gemm_epilogue(xc, w1, GEMM_NT, Nc, Cc, Cout, ldw=ldw)
        else:
            t = torch.mm(xc, w1.t())                  # [Nc, Cout] on the coarse rows
        if skip.is_contiguous() and _own_gemm("decoder", skip, w2, GEMM_NT, N, Cs, Cout, 0, None, ldw):
            y = gemm_epilogue(skip, w2, GEMM_NT, N, Cs, Cout, ldw=ldw)
        else:
            y = torch.mm(skip, w2.t())                # [N, Cout]
        out = torch.empty_like(y)
        nb = int(b1 is not None and ctx.needs_input_grad[4]) + int(b2 is not None and ctx.needs_input_grad[5])
        # backward targets, cleared by the forward launch on the side: bias-gradient rows + the pooled gradient [Nc, Cout]
        zbuf = torch.empty((nb + Nc) * Cout, dtype=torch.float32, device=xc.device)
        gbuf = zbuf[:nb * Cout].view(nb, Cout) if nb else None
        gt_buf = zbuf[nb * Cout:].view(Nc, Cout)
        _native.check(L.d3f_bias_act_forward(_p(y), _p(b1), _p(t), _p(b2), float(slope), N, Cout, _p(out), _p(zbuf),
                                             int(zbuf.numel()), None, _p(idx), H, Nc, _stream()), "d3f_bias_act_forward")
        ctx.save_for_backward(xc, idx, skip, weight, out)
        ctx.gbuf, ctx.gt_buf, ctx.slope = gbuf, gt_buf, float(slope)
        ctx.has = (b1 is not None, b2 is not None)
        ctx.gw_slot = _grad_slot(weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        xc, idx, skip, weight, out = ctx.saved_tensors
        L = _native.lib()
        Nc, Cc = int(xc.shape[0]), int(xc.shape[1])
        N, Cs = int(skip.shape[0]), int(skip.shape[1])
        Cout, H = int(weight.shape[0]), int(idx.shape[1])
        go = grad_out.contiguous()
        want1 = ctx.has[0] and ctx.needs_input_grad[4]
        want2 = ctx.has[1] and ctx.needs_input_grad[5]
        g1 = g2 = None
        pre = 0
        if want1 or want2:
            gbuf, pre = ctx.gbuf, 1
            ctx.gbuf = None
            if gbuf is None:
                gbuf, pre = torch.empty((int(want1) + int(want2), Cout), dtype=torch.float32, device=go.device), 0
            rows = list(gbuf.unbind(0))
            g1 = rows.pop(0) if want1 else None
            g2 = rows.pop(0) if want2 else None
        gm = torch.empty_like(go)
        first, second = (g1, g2) if g1 is not None else (g2, None)
        atb = (ctx.needs_input_grad[3] and N >= _SPLITK_MIN_ROWS
               and bool(L.d3f_linear_grad_weight_supported(N, Cs, Cout)))
        bias_part, bias_blocks = _epilogue_backward(go, out, ctx.slope, N, Cout, gm, first, second, pre, None,
                                                    _FOLD_BIAS_SUM and atb)
        # pooled gradient of the coarse product: g_t[m] = sum_{n: idx[n,0] = m} gm[n]
        gt, ctx.gt_buf = ctx.gt_buf, None
        pre_t = 1 if gt is not None else 0
        if gt is None:
            gt = torch.empty((Nc, Cout), dtype=torch.float32, device=go.device)
        _native.check(L.d3f_closest_pool_backward(_p(gm), Cout, _p(idx), N, H, Cout, Nc, _p(gt), pre_t, _stream()),
                      "d3f_closest_pool_backward")
        w1, w2 = weight[:, :Cc], weight[:, Cc:]
        ldw = int(weight.stride(0))
        gxc = gskip = None
        if ctx.needs_input_grad[0]:
            if _own_gemm("decoder", gt, w1, GEMM_NN, Nc, Cout, Cc, 0, None, ldw):
                gxc = gemm_epilogue(gt, w1, GEMM_NN, Nc, Cout, Cc, ldw=ldw)
            else:
                gxc = torch.mm(gt, w1)
        if ctx.needs_input_grad[2]:
            if _own_gemm("decoder", gm, w2, GEMM_NN, N, Cout, Cs, 0, None, ldw):
                gskip = gemm_epilogue(gm, w2, GEMM_NN, N, Cout, Cs, ldw=ldw)
            else:
                gskip = torch.mm(gm, w2)
        if gskip is not None and ctx.skip_dep is not None and ctx.skip_dep.deposit(gskip):
            gskip = None    # handed to the encoder block that consumes the same skip tensor (GradHolder)
        gw = None
        if ctx.needs_input_grad[3]:
            slot = ctx.gw_slot
            gw = slot if slot is not None else torch.empty_like(weight)
            if not _queue_small_weight_grad(xc, gt, Nc, Cc, Cout, gw[:, :Cc]):
                torch.mm(gt.t(), xc, out=gw[:, :Cc])  # the GEMMs write their column block of W's gradient in place
            if atb:   # (the grouped second stage writes the column block of W's gradient in place: row stride Cc + Cs)
                _grad_weight_atb(skip, gm, N, Cs, Cout, gw[:, Cc:], bias_part, bias_blocks, first, second, "linear_dw")
            elif not _queue_small_weight_grad(skip, gm, N, Cs, Cout, gw[:, Cc:]):
                torch.mm(gm.t(), skip, out=gw[:, Cc:])
            gw = _adoptable(gw, slot)
        return gxc, None, gskip, gw, g1, g2, None, None


def upsample_linear_bias_act(x_coarse, inds, skip, weight, bias1=None, bias2=None, slope=0.1, skip_grad_deposit=None):
    """act([x_coarse[inds[:,0]] | skip] @ weight^T + bias1 + bias2) without forming the upsampled matrix."""
    xc, sk, w = _f32(x_coarse, "x_coarse"), _f32(skip, "skip"), _f32(weight, "weight")
    idx = _i32(inds, "inds")
    if idx.dim() == 1:
        idx = idx.view(-1, 1)
    if w.shape[1] != xc.shape[1] + sk.shape[1] or idx.shape[0] != sk.shape[0]:
        raise RuntimeError("upsample_linear: shapes x_c%s skip%s W%s idx%s" % (
            tuple(xc.shape), tuple(sk.shape), tuple(w.shape), tuple(idx.shape)))
    b1 = _f32(bias1, "bias1") if bias1 is not None else None
    b2 = _f32(bias2, "bias2") if bias2 is not None else None
    return _UpsampleLinearFn.apply(xc, idx, sk, w, b1, b2, float(slope), skip_grad_deposit)


# rows from which the unary blocks use the fused row-streaming kernels instead of library GEMM + epilogue launch
_FUSED_LINEAR_MIN_ROWS = 4096
_FUSED_LINEAR_MAX_CIN = 64     # (wider inputs: the library GEMM's deeper tiling + an epilogue launch wins, re-measured in round 6)
# library GEMM + epilogue as ONE autograd node (_LinearLibBiasActFn); D3F_MERGED_UNARY=0: the two nodes of rounds 1-4
_MERGED_UNARY = True


def linear_bias_act(x, weight, bias1=None, add=None, bias2=None, slope=0.1, grad_holder=None, grad_deposit=None,
                    pack_for=None):
    """act(x @ weight^T + bias1 + add + bias2) -- the whole unary block (reference blocks.py:481-541,686).
    grad_holder: the gradient a sibling branch deposited for `x` is added by this op's grad-input GEMM;
    grad_deposit: this op hands its own grad_x to the sibling instead of returning it (see GradHolder);
    pack_for: see bias_act (honoured on the library-GEMM + epilogue path)."""
    x, weight = _f32(x, "x"), _f32(weight, "weight")
    N, Cin, Cout = int(x.shape[0]), int(x.shape[1]), int(weight.shape[0])
    # measured (profiles/unary_gemm_microbench.py): the fused kernel beats library GEMM + epilogue launch for
    # Cin <= 64 (7-21 us vs 10-27 us at 38k rows), not for Cin >= 128 where the library's deeper tiling wins
    if N >= _FUSED_LINEAR_MIN_ROWS and Cin <= _FUSED_LINEAR_MAX_CIN and _native.lib().d3f_linear_fused_supported(N, Cin, Cout):
        b1 = _f32(bias1, "bias1") if bias1 is not None else None
        b2 = _f32(bias2, "bias2") if bias2 is not None else None
        a = _f32(add, "add") if add is not None else None
        return _LinearBiasActFn.apply(x, weight, b1, a, b2, float(slope), grad_holder, grad_deposit)
    if _MERGED_UNARY and x.dim() == 2 and x.is_cuda and (add is None or add.shape == (N, Cout)):
        b1 = _f32(bias1, "bias1") if bias1 is not None else None
        b2 = _f32(bias2, "bias2") if bias2 is not None else None
        a = _f32(add, "add") if add is not None else None
        if pack_for is not None and N > 0 and _native.lib().d3f_bias_act_packs(Cout) and N * Cout < 2 ** 32 and \
                int(pack_for[0].shape[0]) == N:
            s_pts = _f32(pack_for[0], "s_pts")
            out, spack, gx_clear = _LinearLibBiasActFn.apply(x, weight, b1, a, b2, float(slope), grad_holder,
                                                             grad_deposit, (s_pts, bool(pack_for[1])))
            out._d3f_spack = PackedSupports(spack, gx_clear, s_pts, N, Cout)
            return out
        return _LinearLibBiasActFn.apply(x, weight, b1, a, b2, float(slope), grad_holder, grad_deposit)
    return bias_act(linear_nobias(x, weight, grad_holder, grad_deposit), bias1, add, bias2, slope=slope,
                    pack_for=pack_for)


def linear_nobias(x, weight, grad_holder=None, grad_deposit=None):
    """x [N, Cin] @ weight[Cout, Cin]^T on the device (fp32)."""
    return _LinearFn.apply(_f32(x, "x"), _f32(weight, "weight"), grad_holder, grad_deposit)


# ---------------------------------------------------------------------------------------------------------------
# pools (models/blocks.py:79-110)
# ---------------------------------------------------------------------------------------------------------------
class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, deposit=None, incoming=None, width=None, groups=None):
        ctx.dep, ctx.incoming = deposit, incoming
        g_len, g_n = groups if groups is not None else (None, 0)
        Ns, C = int(x.shape[0]), int(x.shape[1])
        Nq, H = int(idx.shape[0]), int(idx.shape[1])
        out = torch.empty((Nq, C), dtype=torch.float32, device=x.device)
        arg = torch.empty((Nq, C), dtype=torch.int32, device=x.device)
        gx_buf = torch.empty_like(x) if ctx.needs_input_grad[0] else None  # cleared by the forward launch
        with _region("max_pool_fwd[Nq=%d,C=%d]" % (Nq, C), 4 * Nq * H + 4 * Nq * H * C + 4 * Nq * C):
            _native.check(_native.lib().d3f_max_pool_forward(_p(x), Ns, C, _p(idx), Nq, H, _p(out), _p(arg),
                                                             _p(gx_buf), _p(width), _p(g_len),
                                                             int(g_len.numel()) if g_len is not None else 0, int(g_n),
                                                             _stream()), "d3f_max_pool_forward")
        ctx.save_for_backward(arg)
        ctx.shape = (Ns, C)
        ctx.gx_buf = gx_buf
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (arg,) = ctx.saved_tensors
        Ns, C = ctx.shape
        go = grad_out.contiguous().float()
        gx, ctx.gx_buf = ctx.gx_buf, None
        pre = 1 if gx is not None else 0
        # a gradient an older consumer of x already produced (the decoder, for a skip tensor): scatter on top of it
        c = ctx.incoming.collect() if ctx.incoming is not None else None
        if c is not None and c.is_contiguous() and c.dtype == torch.float32 and tuple(c.shape) == (Ns, C):
            gx, pre, c = c, 1, None
        if gx is None:
            gx = torch.empty((Ns, C), dtype=torch.float32, device=go.device)
        _native.check(_native.lib().d3f_max_pool_backward(_p(go), _p(arg), int(arg.shape[0]), C, Ns, _p(gx), pre,
                                                          _stream()), "d3f_max_pool_backward")
        if c is not None:
            gx.add_(c)
        if ctx.dep is not None and ctx.dep.deposit(gx):
            gx = None
        return gx, None, None, None, None, None


def _check_groups(width, groups, what):
    """(q_lens int32 [B] on the device, clouds per group) of a batch that stacks several reference batches; ``width``
    then holds one entry per group."""
    if groups is None:
        if width is not None and not (width.is_cuda and width.dtype == torch.int32 and width.numel() == 1):
            raise ValueError("width must be a device int32[1] tensor")
        return None
    g_len, g_n = groups
    if not (isinstance(g_len, torch.Tensor) and g_len.is_cuda and g_len.dtype == torch.int32) or int(g_n) < 1:
        raise ValueError("%s: groups = (device int32 stack lengths, clouds per group)" % what)
    n_groups = -(-int(g_len.numel()) // int(g_n))
    if width is not None and not (width.is_cuda and width.dtype == torch.int32 and width.numel() == n_groups):
        raise ValueError("%s: width must hold one device int32 per group (%d)" % (what, n_groups))
    return g_len, int(g_n)


def max_pool(x, inds, grad_deposit=None, grad_incoming=None, width=None, groups=None):
    """max over the neighbors of every query row, zero shadow row included (blocks.py:94-110).  ``width``: device
    int32[1] = the table's max neighbor count; only the first min(H, width) columns count -- the table the reference
    would have built (dataloader.py:64-66) when ``inds`` is kept at a wider, static width.  ``groups`` = (stack lengths
    of the QUERY level, clouds per group): the batch stacks several reference batches (8 pairs: groups of 2) and
    ``width`` holds one entry per group."""
    groups = _check_groups(width, groups, "max_pool")
    return _MaxPoolFn.apply(_f32(x, "x"), _i32(inds, "inds"), grad_deposit, grad_incoming, width, groups)


class _ClosestPoolFn(torch.autograd.Function):
    """closest_pool, optionally concatenated with a skip tensor ([upsampled | skip]) by the same launch."""

    @staticmethod
    def forward(ctx, x, idx, skip):
        Ns, C = int(x.shape[0]), int(x.shape[1])
        Nq, H = int(idx.shape[0]), int(idx.shape[1])
        Cs = int(skip.shape[1]) if skip is not None else 0
        out = torch.empty((Nq, C + Cs), dtype=torch.float32, device=x.device)
        gx_buf = torch.empty_like(x) if ctx.needs_input_grad[0] else None  # cleared by the forward launch
        _native.check(_native.lib().d3f_closest_pool_forward(_p(x), Ns, C, _p(idx), Nq, H, _p(skip), Cs, _p(out),
                                                             _p(gx_buf), _stream()), "d3f_closest_pool_forward")
        ctx.save_for_backward(idx)
        ctx.shape = (Ns, C, Cs)
        ctx.gx_buf = gx_buf
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        Ns, C, Cs = ctx.shape
        go = grad_out if grad_out.dtype == torch.float32 else grad_out.float()
        # a column slice of a wider row-major matrix (the [upsampled | skip] gradient) is read in place
        if not (go.dim() == 2 and go.stride(1) == 1 and go.stride(0) >= C + Cs):
            go = go.contiguous()
        ld = int(go.stride(0)) if go.shape[0] > 1 else C + Cs
        gx = None
        if ctx.needs_input_grad[0]:
            gx, ctx.gx_buf = ctx.gx_buf, None
            pre = 1 if gx is not None else 0
            if gx is None:
                gx = torch.empty((Ns, C), dtype=torch.float32, device=go.device)
            _native.check(_native.lib().d3f_closest_pool_backward(_p(go), ld, _p(idx), int(idx.shape[0]),
                                                                  int(idx.shape[1]), C, Ns, _p(gx), pre, _stream()),
                          "d3f_closest_pool_backward")
        g_skip = go[:, C:] if (Cs and ctx.needs_input_grad[2]) else None
        return gx, None, g_skip


def closest_pool(x, inds, skip=None):
    """x'[inds[:, 0]] (reference blocks.py:79-91); with ``skip`` [Nq, Cs] the result is torch.cat([pooled, skip], 1)."""
    idx = _i32(inds, "inds")
    if idx.dim() == 1:
        idx = idx.view(-1, 1)
    sk = _f32(skip, "skip") if skip is not None else None
    if sk is not None and (sk.dim() != 2 or sk.shape[0] != idx.shape[0]):
        raise RuntimeError("closest_pool: skip %s does not match %d query rows" % (tuple(sk.shape), idx.shape[0]))
    return _ClosestPoolFn.apply(_f32(x, "x"), idx, sk)


# ---------------------------------------------------------------------------------------------------------------
# block epilogue: bias (+ residual) (+ LeakyReLU) (models/blocks.py:473,497,598,676,686)
# ---------------------------------------------------------------------------------------------------------------
def _bias_bwd_ws(N, C, device):
    nb = int(_native.lib().d3f_bias_act_backward_ws_bytes(N, C))
    return (_ws(nb, device), nb) if nb else (None, 0)


class _BiasActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b1, add, b2, slope, pack=None):
        N, C = int(x.shape[0]), int(x.shape[1])
        out = torch.empty_like(x)
        if pack is not None:
            s_pts, want_clear = pack
            nb = int(b1 is not None and ctx.needs_input_grad[1]) + int(b2 is not None and ctx.needs_input_grad[3])
            gbuf = torch.empty((nb, C), dtype=torch.float32, device=x.device) if nb else None
            spack = torch.empty(16 * N, dtype=torch.uint8, device=x.device)
            gx_clear = torch.empty_like(x) if want_clear else None
            _native.check(_native.lib().d3f_bias_act_forward_pack(
                _p(x), _p(b1), _p(add), _p(b2), float(slope), N, C, _p(out), _p(gbuf), nb * C, None, None, 0, 0,
                _p(s_pts), _p(spack), _p(gx_clear), _stream()), "d3f_bias_act_forward_pack")
            ctx.save_for_backward(out)
            ctx.gbuf, ctx.slope = gbuf, float(slope)
            ctx.has = (b1 is not None, add is not None, b2 is not None)
            ctx.mark_non_differentiable(spack)
            if gx_clear is not None:
                ctx.mark_non_differentiable(gx_clear)
            # (without this autograd hands backward zero-FILLED gradients for the two by-products: two fill launches
            # per packed layer and step, visible in the step timeline as FillFunctor pairs)
            ctx.set_materialize_grads(False)
            return out, spack, gx_clear
        # the backward's bias-gradient accumulators [2, C] are cleared by the forward kernel on the side: no fill
        # launch in backward, and the two bias parameters get separate buffers (autograd would clone a shared one)
        nb = int(b1 is not None and ctx.needs_input_grad[1]) + int(b2 is not None and ctx.needs_input_grad[3])
        gbuf = torch.empty((nb, C), dtype=torch.float32, device=x.device) if nb else None
        _native.check(_native.lib().d3f_bias_act_forward(_p(x), _p(b1), _p(add), _p(b2), float(slope), N, C, _p(out),
                                                         _p(gbuf), nb * C, None, None, 0, 0, _stream()),
                      "d3f_bias_act_forward")
        ctx.save_for_backward(out)
        ctx.gbuf = gbuf
        ctx.slope = float(slope)
        ctx.has = (b1 is not None, add is not None, b2 is not None)
        return out

    @staticmethod
    def backward(ctx, grad_out, *_unused):
        (out,) = ctx.saved_tensors
        N, C = int(out.shape[0]), int(out.shape[1])
        if grad_out is None:     # (only by-products were used: possible with set_materialize_grads(False))
            return None, None, None, None, None, None
        go = grad_out.contiguous()
        need_gx = ctx.needs_input_grad[0] or (ctx.has[1] and ctx.needs_input_grad[2])
        want1 = ctx.has[0] and ctx.needs_input_grad[1]
        want2 = ctx.has[2] and ctx.needs_input_grad[3]
        identity = ctx.slope == 1.0
        gx = g1 = g2 = None
        if want1 or want2:
            gbuf, pre = ctx.gbuf, 1
            ctx.gbuf = None
            if gbuf is None:  # a second backward through the same node: fresh, not pre-cleared accumulators
                gbuf, pre = torch.empty((int(want1) + int(want2), C), dtype=torch.float32, device=go.device), 0
            rows = list(gbuf.unbind(0))
            g1 = rows.pop(0) if want1 else None
            g2 = rows.pop(0) if want2 else None
        if identity and not (want1 or want2):
            gx = go
        elif need_gx or want1 or want2:
            if need_gx and not identity:
                gx = torch.empty_like(go)
            first, second = (g1, g2) if g1 is not None else (g2, None)
            wsb, nws = _bias_bwd_ws(N, C, go.device) if first is not None else (None, 0)
            _native.check(_native.lib().d3f_bias_act_backward(_p(go), _p(out), ctx.slope, N, C, _p(gx), _p(first),
                                                              _p(second), pre if first is not None else 0, None,
                                                              _p(wsb), nws, _stream()), "d3f_bias_act_backward")
            if identity:
                gx = go
        return (gx if ctx.needs_input_grad[0] else None, g1, gx if ctx.has[1] and ctx.needs_input_grad[2] else None,
                g2, None, None)


def bias_act(x, bias1=None, add=None, bias2=None, slope=0.1, pack_for=None):
    """act(x + bias1 + add + bias2) with act = LeakyReLU(slope) (slope = 1.0: no activation); one launch each way.
    ``pack_for`` = (s_pts, want_clear): the result is the feature matrix of a KPConv over the supports ``s_pts``; the
    same launch leaves that KPConv's packed supports (and, with ``want_clear``, its cleared grad_x target) behind as
    ``result._d3f_spack`` (PackedSupports) -- one launch less per KPConv layer."""
    x = _f32(x, "x")
    if x.dim() != 2:
        raise RuntimeError("bias_act expects [N, C]")
    b1 = _f32(bias1, "bias1") if bias1 is not None else None
    b2 = _f32(bias2, "bias2") if bias2 is not None else None
    a = _f32(add, "add") if add is not None else None
    if a is not None and a.shape != x.shape:
        raise RuntimeError("bias_act: residual shape %s != %s" % (tuple(a.shape), tuple(x.shape)))
    if pack_for is not None and x.shape[0] > 0 and _native.lib().d3f_bias_act_packs(int(x.shape[1])) and \
            x.numel() < 2 ** 32 and int(pack_for[0].shape[0]) == int(x.shape[0]):
        s_pts = _f32(pack_for[0], "s_pts")
        out, spack, gx_clear = _BiasActFn.apply(x, b1, a, b2, float(slope), (s_pts, bool(pack_for[1])))
        out._d3f_spack = PackedSupports(spack, gx_clear, s_pts, x.shape[0], x.shape[1])
        return out
    return _BiasActFn.apply(x, b1, a, b2, float(slope))


# ---------------------------------------------------------------------------------------------------------------
# batch normalisation over the stacked points (models/blocks.py:454-471, use_bn=True) -- csrc/batchnorm.hip
# ---------------------------------------------------------------------------------------------------------------
class _BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, slope, n_live):
        N, C = int(x.shape[0]), int(x.shape[1])
        L = _native.lib()
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        nbytes = L.d3f_batchnorm_ws_bytes(N, C)
        ws = _ws(nbytes, x.device)
        with _region("batchnorm_fwd[N=%d,C=%d]" % (N, C), 16 * N * C):
            _native.check(L.d3f_batchnorm_forward(_p(x), N, C, _p(n_live), _p(weight), _p(bias), _p(running_mean),
                                                  _p(running_var), float(momentum), float(eps), 1 if training else 0,
                                                  float(slope), _p(y), _p(mean), _p(invstd), _p(ws), nbytes, _stream()),
                          "d3f_batchnorm_forward")
        ctx.save_for_backward(x, weight, bias, mean, invstd)
        ctx.n_live, ctx.slope, ctx.training = n_live, float(slope), bool(training)
        ctx.mark_non_differentiable(mean, invstd)
        ctx.set_materialize_grads(False)
        return y, mean, invstd

    @staticmethod
    def backward(ctx, gy, _gm, _gi):
        if gy is None:
            return (None,) * 10
        x, weight, bias, mean, invstd = ctx.saved_tensors
        N, C = int(x.shape[0]), int(x.shape[1])
        L = _native.lib()
        gy = gy.contiguous()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty(C, dtype=torch.float32, device=x.device) if weight is not None else None
        gb = torch.empty(C, dtype=torch.float32, device=x.device) if bias is not None else None
        nbytes = L.d3f_batchnorm_ws_bytes(N, C)
        ws = _ws(nbytes, x.device)
        with _region("batchnorm_bwd[N=%d,C=%d]" % (N, C), 20 * N * C):
            _native.check(L.d3f_batchnorm_backward(_p(x), N, C, _p(ctx.n_live), _p(weight), _p(bias), _p(mean),
                                                   _p(invstd), ctx.slope, 1 if ctx.training else 0, _p(gy), _p(gx),
                                                   _p(gw), _p(gb), _p(ws), nbytes, _stream()),
                          "d3f_batchnorm_backward")
        return gx, gw, gb, None, None, None, None, None, None, None


def batch_norm(x, weight, bias, running_mean, running_var, training, momentum=0.1, eps=1e-5, slope=1.0, n_live=None):
    """nn.BatchNorm1d over the N stacked points of x [N, C] (the reference's BatchNormBlock, blocks.py:465-471) with an
    optional LeakyReLU(slope) fused behind it.  Training mode normalises with the batch statistics and updates
    ``running_mean`` / ``running_var`` in place (momentum, unbiased variance); eval mode uses the running statistics.
    ``n_live`` (int32 device scalar) bounds the live rows when x is a capacity-shaped buffer."""
    x = _f32(x, "x")
    if x.dim() != 2:
        raise RuntimeError("batch_norm expects [N, C]")
    if not training and (running_mean is None or running_var is None):
        raise RuntimeError("batch_norm in eval mode needs running statistics")
    if momentum is None:
        raise RuntimeError("batch_norm: cumulative-average momentum (None) is not supported")
    w = _f32(weight, "weight") if weight is not None else None
    b = _f32(bias, "bias") if bias is not None else None
    return _BatchNormFn.apply(x, w, b, running_mean, running_var, bool(training), float(momentum), float(eps),
                              float(slope), n_live)[0]


# ---------------------------------------------------------------------------------------------------------------
# detector score (models/architectures.py:322-368)
# ---------------------------------------------------------------------------------------------------------------
def global_max(x, lens=None, group=0):
    """max(x) as a device scalar; with ``lens`` (int32 stack lengths) only the first sum(lens) rows count; with
    ``group`` > 0 one maximum per group of that many consecutive clouds (float32 [ceil(B/group)])."""
    x = _f32(x, "x")
    if group:
        G = -(-int(lens.numel()) // int(group))
        out = torch.empty(G, dtype=torch.float32, device=x.device)
        ws = _ws(4 * G, x.device)
        _native.check(_native.lib().d3f_global_max_groups(_p(x), int(x.shape[0]), int(x.shape[1]), _p(lens),
                                                          int(lens.numel()), int(group), _p(out), _p(ws),
                                                          max(4 * G, 256), _stream()), "d3f_global_max_groups")
        return out
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = _ws(256, x.device)
    if lens is None:
        _native.check(_native.lib().d3f_global_max(_p(x), x.numel(), _p(out), _p(ws), 256, _stream()),
                      "d3f_global_max")
    else:
        _native.check(_native.lib().d3f_global_max_rows(_p(x), int(x.shape[0]), int(x.shape[1]), _p(lens),
                                                        int(lens.numel()), _p(out), _p(ws), 256, _stream()),
                      "d3f_global_max_rows")
    return out


class _DetScoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, idx, training, lens=None, width=None, group=0):
        N, C, H = int(feat.shape[0]), int(feat.shape[1]), int(idx.shape[1])
        if group and lens is None:
            raise RuntimeError("detection_scores: the grouped form needs the stack lengths")
        fmax = global_max(feat, lens, group)
        scores = torch.empty((N, 1), dtype=torch.float32, device=feat.device)
        na = _native.lib().d3f_detection_scores_aux_floats(C) if (training and ctx.needs_input_grad[0] and H <= 64) else 0
        aux = torch.empty((N, na), dtype=torch.float32, device=feat.device) if na else None
        with _region("detection_fwd[N=%d]" % N, 4 * N * H + 4 * N * H * C + 4 * N * C + 4 * N):
            _native.check(_native.lib().d3f_detection_scores_forward(_p(feat), N, C, _p(idx), H, _p(fmax),
                                                                     1 if training else 0, _p(scores), _p(aux),
                                                                     _p(width), _p(lens) if group else None,
                                                                     int(lens.numel()) if group else 0, int(group),
                                                                     _stream()), "d3f_detection_scores_forward")
        ctx.aux = aux
        ctx.save_for_backward(feat, idx, fmax)
        ctx.training = bool(training)
        ctx.groups = (lens, int(group)) if group else None
        return scores

    @staticmethod
    def backward(ctx, grad_scores):
        feat, idx, fmax = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("detection_scores: backward is defined for training mode only")
        N, C, H = int(feat.shape[0]), int(feat.shape[1]), int(idx.shape[1])
        gs = grad_scores.contiguous().float()
        gf = torch.empty_like(feat)
        nws = int(_native.lib().d3f_detection_scores_ws_bytes(N, C))
        ws = _ws(nws, feat.device)
        with _region("detection_bwd[N=%d]" % N, 4 * N * H + 4 * N * H * C + 12 * N * C + 4 * N):
            if ctx.groups is not None:   # stacked pairs: the normaliser's gradient stays inside each pair
                lens, group = ctx.groups
                _native.check(_native.lib().d3f_detection_scores_backward_groups(
                    _p(feat), N, C, _p(idx), H, _p(fmax), _p(gs), _p(ctx.aux), _p(gf), _p(lens), int(lens.numel()),
                    group, _p(ws), nws, _stream()), "d3f_detection_scores_backward_groups")
            else:
                _native.check(_native.lib().d3f_detection_scores_backward(_p(feat), N, C, _p(idx), H, _p(fmax), _p(gs),
                                                                          _p(ctx.aux), _p(gf), _p(ws), nws, _stream()),
                              "d3f_detection_scores_backward")
        return gf, None, None, None, None, None


def detection_scores(features, neighbors, training=True, lens=None, width=None, group=0):
    """scores [N,1] from un-normalised descriptors [N,C] and the layer-0 neighbor table.  ``lens`` (device int32
    stack lengths) restricts the global-max normaliser to the live rows of a capacity-shaped batch; ``width`` (device
    int32[1], the table's max neighbor count) makes a table kept at the full limit behave like the reference's
    min(limit, max_count)-column table in the eval-mode local-maximum gate (as for max_pool).  ``group`` > 0: the batch
    stacks several reference batches of that many clouds (8 pairs: 2); the normaliser (architectures.py:342 takes the
    maximum of ONE pair) and ``width`` are then per group, in forward and backward."""
    _check_groups(width, (lens, group) if group else None, "detection_scores")
    return _DetScoreFn.apply(_f32(features, "features"), _i32(neighbors, "neighbors"), bool(training), lens, width,
                             int(group))


# ---------------------------------------------------------------------------------------------------------------
# circle + detector loss (utils/loss.py:8-44,111-141,149-158)
# ---------------------------------------------------------------------------------------------------------------
class _CircleDetFn(torch.autograd.Function):
    """Returns (scalars[6], dists[M,M], furthest_positive[M], average_negative[M]); scalars[0] = desc loss,
    scalars[1] = det loss are differentiable wrt anchor / positive / scores."""

    @staticmethod
    def forward(ctx, anchor, positive, neg_mask, anc_score, pos_score, log_scale, safe_radius, pos_margin,
                neg_margin):
        L = _native.lib()
        M, C = int(anchor.shape[0]), int(anchor.shape[1])
        dev = anchor.device
        dists = torch.empty((M, M), dtype=torch.float32, device=dev)
        fp = torch.empty(M, dtype=torch.float32, device=dev)
        an = torch.empty(M, dtype=torch.float32, device=dev)
        scalars = torch.empty(6, dtype=torch.float32, device=dev)
        stats = torch.empty(L.d3f_circle_det_loss_stats_floats(M), dtype=torch.float32, device=dev)
        _native.check(L.d3f_circle_det_loss_forward(_p(anchor), _p(positive), M, C, _p(neg_mask), _p(anc_score),
                                                    _p(pos_score), float(log_scale), float(safe_radius),
                                                    float(pos_margin), float(neg_margin), _p(dists), _p(fp), _p(an),
                                                    _p(scalars), _p(stats), _stream()), "d3f_circle_det_loss_forward")
        ctx.save_for_backward(anchor, positive, neg_mask, anc_score, pos_score, dists, stats)
        ctx.params = (float(log_scale), float(safe_radius), float(pos_margin), float(neg_margin))
        ctx.mark_non_differentiable(dists, fp, an)
        ctx.set_materialize_grads(False)
        return scalars, dists, fp, an

    @staticmethod
    def backward(ctx, g_scalars, g_dists, g_fp, g_an):
        if g_scalars is None:
            return (None,) * 9
        anchor, positive, neg_mask, anc_score, pos_score, dists, stats = ctx.saved_tensors
        L = _native.lib()
        M, C = int(anchor.shape[0]), int(anchor.shape[1])
        g = g_scalars.contiguous().float()
        ga, gp = torch.empty_like(anchor), torch.empty_like(positive)
        gsa, gsp = torch.empty_like(anc_score), torch.empty_like(pos_score)
        nbytes = L.d3f_circle_det_loss_ws_bytes(M)
        ws = _ws(nbytes, anchor.device)
        s, sr, pm, nm = ctx.params
        _native.check(L.d3f_circle_det_loss_backward(_p(anchor), _p(positive), M, C, _p(neg_mask), _p(anc_score),
                                                     _p(pos_score), s, sr, pm, nm, _p(dists), _p(stats),
                                                     g.data_ptr(), g.data_ptr() + 4, _p(ga), _p(gp), _p(gsa), _p(gsp),
                                                     _p(ws), nbytes, _stream()), "d3f_circle_det_loss_backward")
        return ga, gp, None, gsa, gsp, None, None, None, None


def circle_det_loss(anchor, positive, dist_keypts, anc_score, pos_score, log_scale=10.0, safe_radius=0.1,
                    pos_margin=0.1, neg_margin=1.4):
    anchor, positive = _f32(anchor, "anchor"), _f32(positive, "positive")
    if not dist_keypts.is_cuda:
        raise RuntimeError("dist_keypts must be a CUDA/HIP tensor")
    neg_mask = (dist_keypts > safe_radius).to(torch.uint8).contiguous()  # evaluated in the caller's dtype (f64)
    sa = _f32(anc_score, "anc_score").reshape(-1)
    sp = _f32(pos_score, "pos_score").reshape(-1)
    return _CircleDetFn.apply(anchor, positive, neg_mask, sa, sp, log_scale, safe_radius, pos_margin, neg_margin)


def _corr_columns(idx_a, idx_p):
    """(ptr_a, ptr_p, stride) of the two index vectors.  When they are the two columns of one contiguous [M,2] int64
    table (``corr[:, 0]``, ``corr[:, 1]``) they are read in place with stride 2; otherwise compacted."""
    if (idx_a.dtype == torch.int64 and idx_p.dtype == torch.int64 and idx_a.dim() == 1 and idx_p.dim() == 1
            and idx_a.stride(0) == 2 and idx_p.stride(0) == 2 and idx_p.data_ptr() == idx_a.data_ptr() + 8
            and idx_a.shape == idx_p.shape):
        return idx_a, idx_p, 2
    ia, ip = idx_a.contiguous(), idx_p.contiguous()
    if ia.dtype != torch.int64 or ip.dtype != torch.int64:
        ia, ip = ia.long(), ip.long()
    return ia, ip, 1


def _select_normalize_fwd(x, scores, ia, ip, stride, off):
    N, C, M = int(x.shape[0]), int(x.shape[1]), int(ia.shape[0])
    dev = x.device
    oa = torch.empty((M, C), dtype=torch.float32, device=dev)
    op = torch.empty((M, C), dtype=torch.float32, device=dev)
    sa = torch.empty(M, dtype=torch.float32, device=dev)
    sp = torch.empty(M, dtype=torch.float32, device=dev)
    _native.check(_native.lib().d3f_select_normalize_forward(_p(x), _p(scores), N, C, _p(ia), _p(ip), stride, M,
                                                             _p(off), _p(oa), _p(op), _p(sa), _p(sp), _stream()),
                  "d3f_select_normalize_forward")
    return oa, op, sa, sp


def _select_normalize_bwd(x, ia, ip, stride, off, g_a, g_p, g_sa, g_sp):
    N, C, M = int(x.shape[0]), int(x.shape[1]), int(ia.shape[0])
    buf = torch.empty(N * (C + 1), dtype=torch.float32, device=x.device)
    gx, gs = buf[:N * C].view(N, C), buf[N * C:].view(N, 1)
    cont = [g.contiguous() if g is not None else None for g in (g_a, g_p, g_sa, g_sp)]
    _native.check(_native.lib().d3f_select_normalize_backward(_p(x), N, C, _p(ia), _p(ip), stride, M, _p(off),
                                                              _p(cont[0]), _p(cont[1]), _p(cont[2]), _p(cont[3]),
                                                              _p(gx), _p(gs), _stream()), "d3f_select_normalize_backward")
    return gx, gs


class _SelectNormalizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scores, idx_a, idx_p, p_offset, stride):
        ctx.save_for_backward(x, idx_a, idx_p, p_offset if p_offset is not None else idx_a.new_empty(0))
        ctx.has_off, ctx.stride = p_offset is not None, stride
        return _select_normalize_fwd(x, scores, idx_a, idx_p, stride, p_offset)

    @staticmethod
    def backward(ctx, g_a, g_p, g_sa, g_sp):
        x, idx_a, idx_p, p_off = ctx.saved_tensors
        gx, gs = _select_normalize_bwd(x, idx_a, idx_p, ctx.stride, p_off if ctx.has_off else None, g_a, g_p, g_sa, g_sp)
        return gx, gs, None, None, None, None


def _p_offset(p_offset, device):
    if p_offset is None:
        return None
    off = p_offset if isinstance(p_offset, torch.Tensor) else torch.tensor([int(p_offset)], device=device)
    return off.reshape(-1)[:1].to(torch.int32).contiguous()


def select_normalize(x, scores, idx_a, idx_p, p_offset=None):
    """(normalize(x)[idx_a], normalize(x)[idx_p + p_offset], scores[idx_a], scores[idx_p + p_offset]) without
    normalising or differentiating through the other N - 2M rows.  idx_*: int64 [M] (the two columns of a [M,2] table
    are read in place); p_offset: device int32 scalar."""
    x = _f32(x, "x")
    sc = _f32(scores, "scores").reshape(-1, 1)
    ia, ip, stride = _corr_columns(idx_a, idx_p)
    return _SelectNormalizeFn.apply(x, sc, ia, ip, _p_offset(p_offset, x.device), stride)


# ---------------------------------------------------------------------------------------------------------------
# the whole loss of one training step (trainer.py:91-98) as ONE autograd node
# ---------------------------------------------------------------------------------------------------------------
class _TrainLossFn(torch.autograd.Function):
    """x [N,C] raw descriptors, scores [N,1], corr [M,2] -> (total, scalars[6], dists, furthest_positive,
    average_negative) with total = w_desc * desc + w_det * det.  Same three kernels as select_normalize +
    circle_det_loss; what disappears is the autograd glue between them (index selects and their zero-filled
    backward, the weighted sum and its backward: ~12 sub-5-us launches per step)."""

    @staticmethod
    def forward(ctx, x, scores, corr, p_offset, neg_mask, params, weights, gw):
        L = _native.lib()
        ia, ip, stride = _corr_columns(corr[:, 0], corr[:, 1])
        oa, op, sa, sp = _select_normalize_fwd(x, scores, ia, ip, stride, p_offset)
        M, C = int(oa.shape[0]), int(oa.shape[1])
        dev = x.device
        dists = torch.empty((M, M), dtype=torch.float32, device=dev)
        fp = torch.empty(M, dtype=torch.float32, device=dev)
        an = torch.empty(M, dtype=torch.float32, device=dev)
        scalars = torch.empty(6, dtype=torch.float32, device=dev)
        stats = torch.empty(L.d3f_circle_det_loss_stats_floats(M), dtype=torch.float32, device=dev)
        s, sr, pm, nm = params
        _native.check(L.d3f_circle_det_loss_forward(_p(oa), _p(op), M, C, _p(neg_mask), _p(sa), _p(sp), s, sr, pm, nm,
                                                    _p(dists), _p(fp), _p(an), _p(scalars), _p(stats), _stream()),
                      "d3f_circle_det_loss_forward")
        # unit weights (config.py:58-59): the kernel's own desc + det (scalars[5]); no launch for the sum
        total = scalars[5] if weights == (1.0, 1.0) else torch.dot(scalars[:2], gw)
        ctx.save_for_backward(x, ia, ip, p_offset if p_offset is not None else ia.new_empty(0), neg_mask, oa, op, sa,
                              sp, dists, stats, gw)
        ctx.meta = (stride, p_offset is not None, params, weights == (1.0, 1.0))
        ctx.mark_non_differentiable(scalars, dists, fp, an)
        ctx.set_materialize_grads(False)   # (else four zero-fill launches per step for the by-products' gradients)
        return total, scalars, dists, fp, an

    @staticmethod
    def backward(ctx, g_total, g_scalars, g_dists, g_fp, g_an):
        if g_total is None:
            return (None,) * 8
        x, ia, ip, p_off, neg_mask, oa, op, sa, sp, dists, stats, gw = ctx.saved_tensors
        stride, has_off, (s, sr, pm, nm), unit = ctx.meta
        L = _native.lib()
        M, C = int(oa.shape[0]), int(oa.shape[1])
        if unit:   # d total / d desc = d total / d det = g_total: both pointers read the same scalar
            g = g_total.contiguous().float().reshape(1)
            p_desc = p_det = g.data_ptr()
        else:
            g = (gw * g_total).contiguous()
            p_desc, p_det = g.data_ptr(), g.data_ptr() + 4
        ga, gp = torch.empty_like(oa), torch.empty_like(op)
        gsa, gsp = torch.empty_like(sa), torch.empty_like(sp)
        nbytes = L.d3f_circle_det_loss_ws_bytes(M)
        ws = _ws(nbytes, x.device)
        _native.check(L.d3f_circle_det_loss_backward(_p(oa), _p(op), M, C, _p(neg_mask), _p(sa), _p(sp), s, sr, pm, nm,
                                                     _p(dists), _p(stats), p_desc, p_det, _p(ga),
                                                     _p(gp), _p(gsa), _p(gsp), _p(ws), nbytes, _stream()),
                      "d3f_circle_det_loss_backward")
        gx, gs = _select_normalize_bwd(x, ia, ip, stride, p_off if has_off else None, ga, gp, gsa, gsp)
        return gx, gs, None, None, None, None, None, None


def train_loss(x, scores, corr, p_offset, dist_keypts, log_scale=10.0, safe_radius=0.1, pos_margin=0.1, neg_margin=1.4,
               w_desc=1.0, w_det=1.0, neg_mask=None, _gw_cache={}):
    """Loss of one training step on the un-normalised network output (trainer.py:91-98):
    ``w_desc * CircleLoss(normalize(x)[corr[:,0]], normalize(x)[corr[:,1] + p_offset]) + w_det * DetLoss(...)``.
    Returns (total, desc, det, accuracy, furthest_positive [M], average_negative [M]); only ``total`` carries
    gradient.  ``neg_mask``: ``dist_keypts > safe_radius`` as uint8 when the caller already has it (the pipelined step
    evaluates it with the pair's upload, off the training stream)."""
    x = _f32(x, "x")
    sc = _f32(scores, "scores").reshape(-1, 1)
    if not (corr.is_cuda and corr.dtype == torch.int64 and corr.dim() == 2 and corr.shape[1] == 2):
        raise ValueError("corr must be an int64 [M,2] device tensor")
    corr = corr.contiguous()
    if neg_mask is None:
        if not dist_keypts.is_cuda:
            raise RuntimeError("dist_keypts must be a CUDA/HIP tensor")
        neg_mask = (dist_keypts > safe_radius).to(torch.uint8).contiguous()  # evaluated in the caller's dtype (f64)
    elif not (neg_mask.is_cuda and neg_mask.dtype == torch.uint8 and neg_mask.is_contiguous()
              and tuple(neg_mask.shape) == (corr.shape[0], corr.shape[0])):
        raise ValueError("neg_mask must be a contiguous uint8 [M,M] device tensor (dist_keypts > safe_radius)")
    key = (x.device, float(w_desc), float(w_det))
    if key not in _gw_cache:
        _gw_cache[key] = torch.tensor([float(w_desc), float(w_det)], dtype=torch.float32, device=x.device)
    total, scalars, dists, fp, an = _TrainLossFn.apply(
        x, sc, corr, _p_offset(p_offset, x.device), neg_mask,
        (float(log_scale), float(safe_radius), float(pos_margin), float(neg_margin)), (float(w_desc), float(w_det)),
        _gw_cache[key])
    return total, scalars[0], scalars[1], scalars[2], fp, an


class _TrainLossPairsFn(torch.autograd.Function):
    """_TrainLossFn for P fragment pairs stacked into one batch: x [N,C], scores [N,1], corr [P*M,2] (every pair's own
    cloud-local table), lens int32 [2P] (level-0 stack lengths on the device), neg_mask [P,M,M] -> (total, scalars
    [P,6], dists [P,M,M], furthest_positive [P*M], average_negative [P*M]) with total = sum_p (w_desc desc_p + w_det
    det_p).  Three launches forward (select + normalise, strips of all pairs, finalize), two backward."""

    @staticmethod
    def forward(ctx, x, scores, corr, lens, neg_mask, params, weights):
        L = _native.lib()
        P, M = int(neg_mask.shape[0]), int(neg_mask.shape[1])
        N, C = int(x.shape[0]), int(x.shape[1])
        dev = x.device
        T = P * M
        oa = torch.empty((T, C), dtype=torch.float32, device=dev)
        op = torch.empty((T, C), dtype=torch.float32, device=dev)
        sa = torch.empty(T, dtype=torch.float32, device=dev)
        sp = torch.empty(T, dtype=torch.float32, device=dev)
        _native.check(L.d3f_select_normalize_forward_pairs(_p(x), _p(scores), N, C, _p(corr), M, P, _p(lens), _p(oa),
                                                           _p(op), _p(sa), _p(sp), _stream()),
                      "d3f_select_normalize_forward_pairs")
        dists = torch.empty((P, M, M), dtype=torch.float32, device=dev)
        fp = torch.empty(T, dtype=torch.float32, device=dev)
        an = torch.empty(T, dtype=torch.float32, device=dev)
        scalars = torch.empty((P, 6), dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)
        stats = torch.empty(P * L.d3f_circle_det_loss_stats_floats(M), dtype=torch.float32, device=dev)
        s, sr, pm, nm = params
        _native.check(L.d3f_circle_det_loss_forward_pairs(_p(oa), _p(op), M, C, P, _p(neg_mask), _p(sa), _p(sp), s, sr,
                                                          pm, nm, weights[0], weights[1], _p(dists), _p(fp), _p(an),
                                                          _p(scalars), _p(total), _p(stats), _stream()),
                      "d3f_circle_det_loss_forward_pairs")
        ctx.save_for_backward(x, corr, lens, neg_mask, oa, op, sa, sp, dists, stats)
        ctx.meta = (params, weights)
        ctx.mark_non_differentiable(scalars, dists, fp, an)
        ctx.set_materialize_grads(False)
        return total, scalars, dists, fp, an

    @staticmethod
    def backward(ctx, g_total, g_scalars, g_dists, g_fp, g_an):
        if g_total is None:
            return (None,) * 7
        x, corr, lens, neg_mask, oa, op, sa, sp, dists, stats = ctx.saved_tensors
        (s, sr, pm, nm), weights = ctx.meta
        L = _native.lib()
        P, M = int(neg_mask.shape[0]), int(neg_mask.shape[1])
        N, C = int(x.shape[0]), int(x.shape[1])
        g = g_total.contiguous().float().reshape(1)
        ga, gp = torch.empty_like(oa), torch.empty_like(op)
        gsa, gsp = torch.empty_like(sa), torch.empty_like(sp)
        _native.check(L.d3f_circle_det_loss_backward_pairs(_p(oa), _p(op), M, C, P, _p(neg_mask), _p(sa), _p(sp), s, sr,
                                                           pm, nm, weights[0], weights[1], _p(dists), _p(stats), _p(g),
                                                           _p(ga), _p(gp), _p(gsa), _p(gsp), _stream()),
                      "d3f_circle_det_loss_backward_pairs")
        buf = torch.empty(N * (C + 1), dtype=torch.float32, device=x.device)
        gx, gs = buf[:N * C].view(N, C), buf[N * C:].view(N, 1)
        _native.check(L.d3f_select_normalize_backward_pairs(_p(x), N, C, _p(corr), M, P, _p(lens), _p(ga), _p(gp),
                                                            _p(gsa), _p(gsp), _p(gx), _p(gs), _stream()),
                      "d3f_select_normalize_backward_pairs")
        return gx, gs, None, None, None, None, None


def train_loss_pairs(x, scores, corr, lens, dist_keypts=None, log_scale=10.0, safe_radius=0.1, pos_margin=0.1,
                     neg_margin=1.4, w_desc=1.0, w_det=1.0, neg_mask=None):
    """Loss of one training step on P fragment pairs STACKED into one batch (clouds 2p, 2p+1 = pair p): per pair the
    reference's ``CircleLoss`` + ``DetLoss`` on its own M sampled correspondences (trainer.py:91-98; the reference
    trains one pair per step, dataloader.py:73), ``total`` = their sum -- its gradient is the sum of the pairs'
    gradients, the optimizer's gradient scale makes the mean.  ``corr`` int64 [P,M,2] / [P*M,2] cloud-local rows as the
    dataset yields them, ``lens`` device int32 [2P] (level-0 stack lengths), ``dist_keypts`` [P,M,M] or ``neg_mask``
    uint8 [P,M,M] = dist_keypts > safe_radius.
    Returns (total, desc [P], det [P], accuracy [P], furthest_positive [P,M], average_negative [P,M])."""
    x = _f32(x, "x")
    sc = _f32(scores, "scores").reshape(-1, 1)
    if neg_mask is None:
        if dist_keypts is None or not dist_keypts.is_cuda:
            raise RuntimeError("dist_keypts must be a CUDA/HIP tensor")
        neg_mask = (dist_keypts > safe_radius).to(torch.uint8).contiguous()   # evaluated in the caller's dtype (f64)
    if not (neg_mask.is_cuda and neg_mask.dtype == torch.uint8 and neg_mask.is_contiguous() and neg_mask.dim() == 3
            and neg_mask.shape[1] == neg_mask.shape[2]):
        raise ValueError("neg_mask must be a contiguous uint8 [P,M,M] device tensor (dist_keypts > safe_radius)")
    P, M = int(neg_mask.shape[0]), int(neg_mask.shape[1])
    if not (corr.is_cuda and corr.dtype == torch.int64 and corr.numel() == 2 * P * M and corr.shape[-1] == 2):
        raise ValueError("corr must be an int64 [P,M,2] device tensor (P = %d, M = %d)" % (P, M))
    corr = corr.contiguous().view(P * M, 2)
    if not (isinstance(lens, torch.Tensor) and lens.is_cuda and lens.dtype == torch.int32 and lens.numel() == 2 * P):
        raise ValueError("lens must hold the 2P level-0 stack lengths as device int32")
    if M > 128 or int(x.shape[1]) > 64 or P > 32:
        raise ValueError("stacked loss: M <= 128 correspondences, C <= 64 channels, P <= 32 pairs")
    total, scalars, dists, fp, an = _TrainLossPairsFn.apply(
        x, sc, corr, lens.contiguous(), neg_mask,
        (float(log_scale), float(safe_radius), float(pos_margin), float(neg_margin)), (float(w_desc), float(w_det)))
    return total, scalars[:, 0], scalars[:, 1], scalars[:, 2], fp.view(P, M), an.view(P, M)


# ---------------------------------------------------------------------------------------------------------------
# dense mutual-NN matching (geometric_registration/common.py:5-21)
# ---------------------------------------------------------------------------------------------------------------
def mutual_nn(source_desc, target_desc):
    """(row_argmin [Ns], col_argmin [Nt], mutual [Ns]) int32 device tensors."""
    s, t = _f32(source_desc, "source_desc"), _f32(target_desc, "target_desc")
    Ns, Nt, C = int(s.shape[0]), int(t.shape[0]), int(s.shape[1])
    ra = torch.empty(Ns, dtype=torch.int32, device=s.device)
    ca = torch.empty(Nt, dtype=torch.int32, device=s.device)
    mu = torch.empty(Ns, dtype=torch.int32, device=s.device)
    nbytes = _native.lib().d3f_mutual_nn_ws_bytes(Ns, Nt)
    ws = _ws(nbytes, s.device)
    _native.check(_native.lib().d3f_mutual_nn(_p(s), Ns, _p(t), Nt, C, _p(ra), _p(ca), _p(mu), _p(ws), nbytes,
                                              _stream()), "d3f_mutual_nn")
    return ra, ca, mu


def mutual_nn_batched(source_desc, target_desc, seg, max_src, max_tgt):
    """P matchings by one pair of launches.  ``seg`` int32 [P,4] on the device = (src_off, src_len, tgt_off, tgt_len) per
    pair into the rows of the (possibly identical) stacked descriptor matrices; ``max_src`` / ``max_tgt`` host upper
    bounds of the lengths.  Returns (row_argmin [src rows], col_argmin [tgt rows], mutual [src rows]) indexed by stacked
    row, holding pair-local indices; rows outside every segment hold -1 / 0."""
    s, t = _f32(source_desc, "source_desc"), _f32(target_desc, "target_desc")
    if not (seg.is_cuda and seg.dtype == torch.int32 and seg.dim() == 2 and seg.shape[1] == 4 and seg.is_contiguous()):
        raise ValueError("seg must be a contiguous device int32 [P,4] tensor")
    Ns, Nt, C, P = int(s.shape[0]), int(t.shape[0]), int(s.shape[1]), int(seg.shape[0])
    ra = torch.full((Ns,), -1, dtype=torch.int32, device=s.device)
    ca = torch.full((Nt,), -1, dtype=torch.int32, device=s.device)
    mu = torch.zeros(Ns, dtype=torch.int32, device=s.device)
    nbytes = _native.lib().d3f_mutual_nn_batched_ws_bytes(Ns, Nt)
    ws = _ws(nbytes, s.device)
    _native.check(_native.lib().d3f_mutual_nn_batched(_p(s), Ns, _p(t), Nt, _p(seg), P, int(max_src), int(max_tgt), C,
                                                      _p(ra), _p(ca), _p(mu), _p(ws), nbytes, _stream()),
                  "d3f_mutual_nn_batched")
    return ra, ca, mu


TOPK_MAX = 6144


def topk_scores(scores, seg, k):
    """int32 [P,k]: the k highest-scoring rows of every cloud (cloud-local indices), ascending (score, index) like the
    tail of a stable argsort (test.py:56-57); ``seg`` int32 [P,2] on the device = (offset, length) per cloud; a cloud
    with fewer than k rows leads with -1."""
    sc = _f32(scores, "scores").reshape(-1)
    if not (seg.is_cuda and seg.dtype == torch.int32 and seg.dim() == 2 and seg.shape[1] == 2 and seg.is_contiguous()):
        raise ValueError("seg must be a contiguous device int32 [P,2] tensor")
    if not 1 <= int(k) <= TOPK_MAX:
        raise ValueError("k must be in 1..%d" % TOPK_MAX)
    out = torch.empty((int(seg.shape[0]), int(k)), dtype=torch.int32, device=sc.device)
    _native.check(_native.lib().d3f_topk_scores(_p(sc), int(sc.numel()), _p(seg), int(seg.shape[0]), int(k), _p(out),
                                                _stream()), "d3f_topk_scores")
    return out


# ---------------------------------------------------------------------------------------------------------------
# guarded SGD step on flat buffers (trainer.py:104-111 + training_3DMatch.py:62-76)
# ---------------------------------------------------------------------------------------------------------------
def sgd_guarded_step(grad, params, momentum_buf, lr, momentum, weight_decay, state, hyper=None, pair_status=None):
    """In place: params/momentum_buf updated unless grad holds a non-finite value (then state[1] += 1).
    ``grad``: the gradient buffer, or a list of up to four of them (one per pair in flight on this GPU,
    train.PairLanes): the step then uses their sum (d3f_sgd_guarded_step_lanes).  ``hyper``: optional fp32[4] device tensor {lr, momentum, weight_decay, grad_scale} read by the kernel when it
    runs.  ``pair_status``: optional device int32[1], the status word of the pair the gradient came from: non-zero
    skips the update too (state[2] |= flags, state[3] += 1)."""
    lanes = list(grad) if isinstance(grad, (list, tuple)) else [grad]
    if not 1 <= len(lanes) <= 4:
        raise ValueError("1..4 gradient buffers, got %d" % len(lanes))
    grad = lanes[0]
    for t, name in [(g, "grad") for g in lanes] + [(params, "params"), (momentum_buf, "momentum_buf")]:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == grad.numel()):
            raise ValueError("%s must be a contiguous fp32 device tensor of %d elements" % (name, grad.numel()))
    if not (state.is_cuda and state.dtype == torch.int32 and state.numel() >= 4):
        raise ValueError("state must be an int32[4] device tensor")
    if hyper is not None and not (hyper.is_cuda and hyper.dtype == torch.float32 and hyper.numel() == 4
                                  and hyper.is_contiguous()):
        raise ValueError("hyper must be a contiguous fp32[4] device tensor")
    with _region("sgd", (12 + 8 * len(lanes)) * grad.numel()):
        if len(lanes) == 1:
            _native.check(_native.lib().d3f_sgd_guarded_step(_p(grad), _p(params), _p(momentum_buf), grad.numel(),
                                                             float(lr), float(momentum), float(weight_decay),
                                                             _p(hyper) if hyper is not None else None, _p(state),
                                                             _p(pair_status), _stream()), "d3f_sgd_guarded_step")
        else:
            import ctypes
            ptrs = (ctypes.c_void_p * len(lanes))(*[g.data_ptr() for g in lanes])
            _native.check(_native.lib().d3f_sgd_guarded_step_lanes(
                ctypes.cast(ptrs, ctypes.c_void_p), len(lanes), _p(params), _p(momentum_buf), grad.numel(), float(lr),
                float(momentum), float(weight_decay), _p(hyper) if hyper is not None else None, _p(state),
                _p(pair_status), _stream()), "d3f_sgd_guarded_step_lanes")


def poison_gradient_if_status(grad, pair_status, state):
    """Data-parallel form of the pair-status gate (before the gradient exchange): see d3f_poison_gradient_if_status."""
    _native.check(_native.lib().d3f_poison_gradient_if_status(_p(grad), _p(pair_status), _p(state), _stream()),
                  "d3f_poison_gradient_if_status")
