"""GPU parity tests proper: every operator is called THROUGH the C ABI (libd3feat_hip.so) on cuda:0 and compared with
the CPU oracle on the same seeded inputs (bit-exact for indices / barycentres, tolerance stated for fp32).
Tolerance: the north star asks descriptors/scores within 1e-4 (fp32); operator outputs are compared at
<= 2e-5 relative to the tensor's max magnitude, gradients at <= 2e-4 (atomic accumulation order)."""
import numpy as np
import pytest
import torch
from contextlib import nullcontext as _nullcontext

from d3feat_pytorch_amd import config as cfgmod
from d3feat_pytorch_amd import _native, ops, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.geometric_registration.common import build_correspondence
from oracle import ops_ref
from util import assert_neighbors_equal_tie_aware, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FWD_TOL = 2e-5
BWD_TOL = 2e-4


def cu(a, dtype=None):
    t = torch.as_tensor(a)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def _gpu_subsample(points, lengths, dlen):
    p, b = dl.batch_grid_subsampling_kpconv(cu(points), cu(lengths), sampleDl=dlen)
    return p.cpu().numpy(), b.cpu().numpy()


def _cloud(rng, n, scale=(2.0, 1.5, 0.6)):
    return (rng.random((n, 3)) * np.asarray(scale)).astype(np.float32)


# ------------------------------------------------------------------------------------------------ preprocessing
@pytest.mark.parametrize("dl_", [0.05, 0.11, 0.3])
def test_grid_subsample_bitexact_with_reference_row_order(native, dl_):
    rng = np.random.default_rng(3)
    lens = np.array([2500, 1700, 900], np.int32)
    pts = _cloud(rng, int(lens.sum()))
    ref_p, ref_b = native.subsample_batch(pts, lens, sampleDl=dl_)
    p, b = dl.batch_grid_subsampling_kpconv(cu(pts), cu(lens), sampleDl=dl_)
    assert np.array_equal(b.cpu().numpy(), ref_b)
    assert np.array_equal(p.cpu().numpy().view(np.uint32), ref_p.view(np.uint32))
    # max_p keeps the first rows of every cloud
    ref_p, ref_b = native.subsample_batch(pts, lens, sampleDl=dl_, max_p=5)
    p, b = dl.batch_grid_subsampling_kpconv(cu(pts), cu(lens), sampleDl=dl_, max_p=5)
    assert np.array_equal(b.cpu().numpy(), ref_b) and np.array_equal(p.cpu().numpy(), ref_p)


def test_grid_subsample_first_seen_order_is_a_permutation(native):
    rng = np.random.default_rng(4)
    lens = np.array([3000], np.int32)
    pts = _cloud(rng, 3000)
    ref_p, _ = native.subsample_batch(pts, lens, sampleDl=0.1)
    p, b = dl.batch_grid_subsampling_kpconv(cu(pts), cu(lens), sampleDl=0.1, order=ops.ORDER_FIRST_SEEN)
    a = p.cpu().numpy()
    assert a.shape == ref_p.shape
    key = lambda x: x[np.lexsort(x.T)]  # noqa: E731
    assert np.array_equal(key(a), key(ref_p))


def test_grid_subsample_large_cloud_and_single_point(native):
    rng = np.random.default_rng(5)
    pts = _cloud(rng, 120000, scale=(3, 2.5, 2.5))
    lens = np.array([70000, 50000], np.int32)
    ref_p, ref_b = native.subsample_batch(pts, lens, sampleDl=0.03)
    p, b = dl.batch_grid_subsampling_kpconv(cu(pts), cu(lens), sampleDl=0.03)
    assert np.array_equal(b.cpu().numpy(), ref_b)
    assert np.array_equal(p.cpu().numpy().view(np.uint32), ref_p.view(np.uint32))
    one = np.array([[0.1, 0.2, 0.3]], np.float32)
    p, b = dl.batch_grid_subsampling_kpconv(cu(one), cu(np.array([1], np.int32)), sampleDl=0.05)
    assert np.array_equal(p.cpu().numpy(), one) and b.tolist() == [1]


@pytest.mark.parametrize("n,n_labels,ldim,dl_", [([2500, 1700, 900], 4, 1, 0.11), ([3000, 500], 40, 1, 0.45),
                                                  ([4000], 40, 3, 0.5), ([1500, 800], 3, 2, 0.05)])
def test_grid_subsample_features_and_labels_bitexact(native, n, n_labels, ldim, dl_):
    """subsample_batch(points, batches, features=, classes=) (reference dataloader.py:24-50): member-mean features
    and majority-vote labels, rows in the reference's order, ties of a vote resolved like the reference's
    unordered_map<int,int> iteration (40 distinct labels in big voxels -> the map rehashes twice)."""
    rng = np.random.default_rng(31)
    pts = _cloud(rng, sum(n))
    lens = np.array(n, np.int32)
    feats = rng.normal(size=(sum(n), 5)).astype(np.float32)
    labels = (rng.integers(0, n_labels, size=(sum(n), ldim)) * 7919 - 20000).astype(np.int32)
    for mp in (0, 6):
        ref = native.subsample_batch_ex(pts, lens, feats, labels, sampleDl=dl_, max_p=mp)
        got = dl.batch_grid_subsampling_kpconv(cu(pts), cu(lens), features=cu(feats), labels=cu(labels), sampleDl=dl_,
                                               max_p=mp)
        assert len(got) == 4
        for g, r in zip(got, ref):
            g = g.cpu().numpy()
            assert g.shape == r.shape and g.dtype == r.dtype and np.array_equal(g.view(np.uint32), r.view(np.uint32))
    p, b, f = dl.batch_grid_subsampling_kpconv(cu(pts), cu(lens), features=cu(feats), sampleDl=dl_)
    assert np.array_equal(f.cpu().numpy(), ref[2] if mp == 0 else native.subsample_batch_ex(pts, lens, feats, None,
                                                                                             sampleDl=dl_)[2])
    p, b, c = dl.batch_grid_subsampling_kpconv(cu(pts), cu(lens), labels=cu(labels[:, 0].copy()), sampleDl=dl_)
    assert np.array_equal(c.cpu().numpy(), native.subsample_batch_ex(pts, lens, None, labels[:, :1], sampleDl=dl_)[2])


def test_native_module_spelling_of_subsample_batch_with_features_and_classes(native):
    """grid_subsampling.subsample_batch(points, batches, features=, classes=, sampleDl=) and .subsample(...) -- the
    reference's CPython module (cpp_subsampling/wrapper.cpp:75-82,316-322,350-356,550-556): NumPy in, NumPy out, all
    four return shapes, through d3f_grid_subsample_ex."""
    from d3feat_pytorch_amd.cpp_wrappers.cpp_subsampling import grid_subsampling as gs
    rng = np.random.default_rng(77)
    n = [1800, 1100]
    pts, lens = _cloud(rng, sum(n)), np.array(n, np.int32)
    feats = rng.normal(size=(sum(n), 3)).astype(np.float32)
    labels = rng.integers(0, 6, size=(sum(n), 1)).astype(np.int32)
    ref = native.subsample_batch_ex(pts, lens, feats, labels, sampleDl=0.15)
    got = gs.subsample_batch(pts, lens, features=feats, classes=labels, sampleDl=0.15)
    assert len(got) == 4 and all(isinstance(g, np.ndarray) for g in got)
    for g, r in zip(got, ref):
        assert g.shape == r.shape and g.dtype == r.dtype and np.array_equal(g.view(np.uint32), r.view(np.uint32))
    pf = gs.subsample_batch(pts, lens, features=feats, sampleDl=0.15)
    assert len(pf) == 3 and np.array_equal(pf[2], native.subsample_batch_ex(pts, lens, feats, None, sampleDl=0.15)[2])
    pc = gs.subsample_batch(pts, lens, classes=labels, sampleDl=0.15)
    assert len(pc) == 3 and pc[2].dtype == np.int32 and np.array_equal(pc[2], ref[3])
    # single-cloud form: points alone, or (points, features, classes)
    one = native.subsample_batch_ex(pts[:n[0]], lens[:1], feats[:n[0]], labels[:n[0]], sampleDl=0.15)
    p_only = gs.subsample(pts[:n[0]], sampleDl=0.15)
    assert isinstance(p_only, np.ndarray) and np.array_equal(p_only, one[0])
    p3 = gs.subsample(pts[:n[0]], features=feats[:n[0]], classes=labels[:n[0]], sampleDl=0.15)
    assert len(p3) == 3 and np.array_equal(p3[0], one[0]) and np.array_equal(p3[1], one[2]) and np.array_equal(p3[2], one[3])
    with pytest.raises(RuntimeError):
        gs.subsample_batch(pts, lens, sampleDl=0.15, method="nonsense")


@pytest.mark.parametrize("radius,limit", [(0.12, 40), (0.2, 64), (0.2, 0), (0.35, 130)])
def test_radius_neighbors_exact(native, radius, limit):
    rng = np.random.default_rng(7)
    sl, ql = np.array([2500, 1700], np.int32), np.array([900, 1100], np.int32)
    s, q = _cloud(rng, int(sl.sum())), _cloud(rng, int(ql.sum()))
    ref = native.batch_query(q, s, ql, sl, radius=radius, max_neighbors=limit)
    got = dl.batch_neighbors_kpconv(cu(q), cu(s), cu(ql), cu(sl), radius, limit).cpu().numpy()
    assert got.dtype == np.int32
    assert np.array_equal(got, ref)  # oracle and kernel share the canonical (d2, index) order -> exact equality
    # self-search: column 0 is the point itself
    got = dl.batch_neighbors_kpconv(cu(s), cu(s), cu(sl), cu(sl), radius, 20).cpu().numpy()
    assert np.array_equal(got[:, 0], np.arange(s.shape[0]))


def test_radius_neighbors_edge_cases(native):
    s = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 0], [0, 0, 0]], np.float32)
    q = np.array([[0, 0, 0]], np.float32)
    # strict '<' at exactly the radius; duplicates ordered by index; padding = number of supports
    got = dl.batch_neighbors_kpconv(cu(q), cu(s), [1], [4], 1.0, 6).cpu().numpy()
    assert got.tolist() == [[0, 2, 3]]  # width = min(limit, max_count)
    with pytest.raises(RuntimeError):
        dl.batch_neighbors_kpconv(cu(np.array([[9, 9, 9]], np.float32)), cu(s), [1], [4], 0.5, 6)
    with pytest.raises(RuntimeError):
        dl.batch_neighbors_kpconv(cu(q[:, :2]), cu(s), [1], [4], 0.5, 6)
    # negative coordinates / far from origin
    rng = np.random.default_rng(1)
    s = (_cloud(rng, 3000) - np.array([50.0, -30.0, 7.0])).astype(np.float32)
    ref = native.batch_query(s, s, [3000], [3000], radius=0.15, max_neighbors=30)
    got = dl.batch_neighbors_kpconv(cu(s), cu(s), [3000], [3000], 0.15, 30).cpu().numpy()
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("seed,width,pool_voxels", [(3, 40, 2.5), (4, 12, 2.5), (5, 40, 1.0)])
def test_upsampling_rows_from_the_pooling_lists_equal_the_prefix_search(seed, width, pool_voxels):
    """The engine's upsampling rows ranked from the TRANSPOSE a pooling search leaves behind
    (d3f_radius_query_pool_transposed -> d3f_upsample_rows_rank) + a search for the rows without a coarse point inside the
    radius (d3f_radius_query_prefix_missing) == RadiusGrid.query_prefix, bit for bit: voxel-subsampled fragments (every fine
    point has its voxel's barycentre nearby -- some farther than the pooling radius), two clouds, padding rows."""
    rng = np.random.default_rng(seed)
    dl0 = 0.03
    frags = [synthetic.make_fragment(seed + k, _gpu_subsample, n_raw=60000, scale=0.3) for k in range(2)]
    fine = np.concatenate(frags, 0)
    fl = np.array([f.shape[0] for f in frags], np.int32)
    cpts, cl = dl.batch_grid_subsampling_kpconv(cu(fine), cu(fl), sampleDl=2 * dl0)
    pool_r, up_r = pool_voxels * dl0, 5.0 * dl0          # (1.0: some fine points have NO coarse point inside the radius)
    pad = 37                                             # capacity-shaped inputs: rows past the live count
    fine_cap = torch.cat([cu(fine), torch.zeros((pad, 3), device=DEV)])
    coarse_cap = torch.cat([cpts, torch.zeros((pad, 3), device=DEV)])
    fgrid = ops.RadiusGrid(fine_cap, cu(fl), pool_r)
    cgrid = ops.RadiusGrid(coarse_cap, cl, up_r)
    bound = min(up_r, max(pool_r, 1.1 * 2 * dl0 * 3.0 ** 0.5))
    want = cgrid.query_prefix(fine_cap, cu(fl), width, pool_r, nearest_bound=bound)
    tab, mx, lkey, transposed = fgrid.query_pool_transposed(coarse_cap, cl, width)
    got = cgrid.prefix_rows_from_transposed(fine_cap, cu(fl), width, pool_r, transposed, nearest_bound=bound)
    ref_tab, ref_mx, ref_lkey = fgrid.query(coarse_cap, cl, width, want_max=True, want_last_key=True)
    assert torch.equal(tab, ref_tab) and torch.equal(mx, ref_mx) and torch.equal(lkey, ref_lkey)   # the search itself: unchanged
    fgrid.status.raise_if_set()
    cgrid.status.raise_if_set()
    assert torch.equal(got, want)
    nc = int(coarse_cap.shape[0])
    counts = (want[:fine.shape[0]] < nc).sum(1)
    assert int(counts.max()) > (3 if pool_voxels > 2 else 1)              # ranked rows ...
    if pool_voxels < 2:                                                    # ... and rows that hold the nearest point only
        d = (fine_cap[:fine.shape[0]] - coarse_cap[want[:fine.shape[0], 0].long()]).norm(dim=1)
        assert int((d >= pool_r).sum()) > 5
    assert bool((want[fine.shape[0]:] == nc).all())                        # padding rows: all shadow


def test_upsampling_rows_with_more_coarse_points_than_a_transposed_list_holds():
    """A volumetric cloud: fine points with up to ~60 'coarse' points inside the radius -- more than the 32 slots of a
    transposed list.  Such a list is dropped by the ranking kernel and the row is searched for like an empty one: the rows
    still equal RadiusGrid.query_prefix bit for bit and no status bit is raised."""
    rng = np.random.default_rng(9)
    fl, cl = np.array([3000, 2000], np.int32), np.array([2500, 1800], np.int32)
    fine = (rng.random((int(fl.sum()), 3)) * 0.5).astype(np.float32)
    coarse = (rng.random((int(cl.sum()), 3)) * 0.5).astype(np.float32)
    r, width = 0.09, 48
    fgrid = ops.RadiusGrid(cu(fine), cu(fl), r)
    cgrid = ops.RadiusGrid(cu(coarse), cu(cl), 2 * r)
    want = cgrid.query_prefix(cu(fine), cu(fl), width, r)
    tab, mx, lkey, transposed = fgrid.query_pool_transposed(cu(coarse), cu(cl), width)
    assert int(transposed[0].max()) > 40                       # lists that outgrow their 32 slots exist ...
    got = cgrid.prefix_rows_from_transposed(cu(fine), cu(fl), width, r, transposed)
    fgrid.status.raise_if_set()
    cgrid.status.raise_if_set()
    assert torch.equal(got, want)
    long_rows = (want < coarse.shape[0]).sum(1) > 32
    assert int(long_rows.sum()) > 50 and bool((transposed[0][long_rows] == 0).all())     # ... and went the other way


@pytest.mark.parametrize("radius,prefix,width", [(0.30, 0.15, 40), (0.30, 0.05, 24), (0.2, 0.2, 64), (0.35, 0.12, 8)])
def test_radius_query_prefix_rows_are_the_leading_part_of_the_full_rows(native, radius, prefix, width):
    """d3f_radius_query_prefix (the upsampling tables inside the training engine): row q = the entries of the full ranked
    row (oracle: the reference's batch_query, neighbors.cpp:211-333) within the prefix radius, in the same order, or the
    single nearest entry when there is none; bit-exact."""
    rng = np.random.default_rng(11)
    sl, ql = np.array([2100, 1500], np.int32), np.array([1300, 900], np.int32)
    s, q = _cloud(rng, int(sl.sum())), _cloud(rng, int(ql.sum()))
    q[:40] += 3.0                                          # some queries with nothing in range at all
    full = native.batch_query(q, s, ql, sl, radius=radius, max_neighbors=0)     # uncapped ranked rows
    ns = s.shape[0]
    grid = ops.RadiusGrid(cu(s), cu(sl), radius)
    got = grid.query_prefix(cu(q), cu(ql), width, prefix).cpu().numpy()
    grid.status.raise_if_set()
    assert got.shape == (q.shape[0], width) and got.dtype == np.int32
    r2 = np.float32(prefix) * np.float32(prefix)
    some_empty = some_nearest_only = some_prefix = 0
    for i in range(q.shape[0]):
        row = full[i][full[i] < ns]
        d = (q[i] - s[row]).astype(np.float32)
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        want = row[d2 < r2]
        if want.size == 0:
            want = row[:1]
            some_nearest_only += int(row.size > 0)
            some_empty += int(row.size == 0)
        else:
            some_prefix += 1
        want = want[:width]
        assert np.array_equal(got[i, :want.size], want), i
        assert np.all(got[i, want.size:] == ns), i
    assert some_empty and some_prefix and (some_nearest_only or prefix > 0.06)   # (the 0.05 case has nearest-only rows)
    # with a nearest bound (cells beyond it are not scanned): the same rows wherever the nearest support is within it
    bound = max(prefix, 0.7 * radius)
    gotb = grid.query_prefix(cu(q), cu(ql), width, prefix, nearest_bound=bound).cpu().numpy()
    b2 = np.float32(bound) * np.float32(bound)
    checked = 0
    for i in range(q.shape[0]):
        row = full[i][full[i] < ns]
        if row.size == 0:
            assert np.all(gotb[i] == ns)
            continue
        d = (q[i] - s[row[0]]).astype(np.float32)
        if (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2] < b2:
            assert np.array_equal(gotb[i], got[i]), i
            checked += 1
    assert checked > q.shape[0] // 2
    # ... and a query WITHOUT a support inside the bound breaks the caller's promise: flagged (D3F_ST_NO_NEAREST), never
    # a silent all-shadow row (the 40 far-away queries above)
    with pytest.raises(RuntimeError, match="nearest bound"):
        grid.status.raise_if_set()
    grid.status.word.zero_()
    near = np.nonzero([full[i][0] < ns and np.sum((q[i] - s[full[i][0]]) ** 2) < 0.9 * b2 for i in range(q.shape[0])])[0]
    near = near[near < int(ql[0])][:64]       # (queries of the first cloud: they see the first cloud's supports)
    ok_rows = grid.query_prefix(cu(q[near]), cu(np.array([near.size, 0], np.int32)), width, prefix, nearest_bound=bound)
    grid.status.raise_if_set()                # every query has its nearest support within the bound: no flag
    assert ok_rows.shape[0] == near.size and near.size > 8


# ------------------------------------------------------------------------------------------------ KPConv
def _kpconv_case(rng, nq, ns, h, cin, cout, shadow_frac=0.15, k=15):
    q, s = _cloud(rng, nq, (1, 1, 1)), _cloud(rng, ns, (1, 1, 1))
    idx = rng.integers(0, ns, size=(nq, h))
    # make geometry meaningful: neighbors near the query so influence weights are non-trivial
    s_near = q[rng.integers(0, nq, size=ns)] + rng.normal(scale=0.03, size=(ns, 3)).astype(np.float32)
    s = s_near.astype(np.float32)
    shadow = rng.random((nq, h)) < shadow_frac
    idx[shadow] = ns
    idx.sort(axis=1)  # shadows (== ns) at the row end like real tables (not required by the kernel)
    x = rng.normal(size=(ns, cin)).astype(np.float32)
    x[rng.random(ns) < 0.1] = 0.0  # rows with zero feature sum exercise the neighbor_num rule
    kp = (rng.normal(size=(k, 3)) * 0.03).astype(np.float32)
    kp[0] = 0
    w = (rng.normal(size=(k, cin, cout)) / np.sqrt(cin * k)).astype(np.float32)
    return q, s, idx.astype(np.int64), x, kp, w


@pytest.mark.parametrize("nq,ns,h,cin,cout", [(700, 900, 42, 1, 64), (1000, 1000, 42, 32, 32), (333, 1000, 37, 64, 64),
                                              (257, 300, 45, 128, 128), (97, 154, 23, 512, 512), (500, 500, 9, 16, 8),
                                              (200, 260, 42, 24, 40), (300, 400, 42, 16, 16), (300, 400, 40, 32, 64),
                                              (150, 160, 42, 256, 128), (2100, 2100, 42, 64, 32), (571, 2053, 42, 128, 128),
                                              (900, 900, 42, 1, 32), (400, 500, 30, 2, 100), (300, 300, 42, 4, 64), (300, 300, 17, 3, 8)])
@pytest.mark.parametrize("gemm_dx_rows", [0, 1 << 30])  # grad_x: fused gW tile / library GEMM + scatter kernel
def test_kpconv_forward_backward(nq, ns, h, cin, cout, gemm_dx_rows, monkeypatch):
    monkeypatch.setattr(ops, "_GEMM_DX_MAX_ROWS", gemm_dx_rows)
    rng = np.random.default_rng(nq + cin)
    q, s, idx, x, kp, w = _kpconv_case(rng, nq, ns, h, cin, cout)
    ext = 0.05
    tx = torch.from_numpy(x).requires_grad_(True)
    tw = torch.from_numpy(w).requires_grad_(True)
    ref = ops_ref.kpconv(torch.from_numpy(q), torch.from_numpy(s), torch.from_numpy(idx), tx, torch.from_numpy(kp), tw,
                         ext)
    go = torch.from_numpy(rng.normal(size=ref.shape).astype(np.float32))
    ref.backward(go)
    gx = cu(x).requires_grad_(True)
    gw = cu(w).requires_grad_(True)
    out = ops.kpconv(cu(q), cu(s), cu(idx), gx, cu(kp), gw, ext)
    out.backward(cu(go))
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < FWD_TOL
    assert rel_err(gx.grad.cpu().numpy(), tx.grad.numpy()) < BWD_TOL
    assert rel_err(gw.grad.cpu().numpy(), tw.grad.numpy()) < BWD_TOL
    # int32 tables give the same result as the reference's int64 ones
    out32 = ops.kpconv(cu(q), cu(s), cu(idx, torch.int32), cu(x), cu(kp), cu(w), ext)
    assert rel_err(out32.cpu().numpy(), out.detach().cpu().numpy()) < 1e-6  # atomically combined partial sums


@pytest.mark.parametrize("nq,h,ns", [(1000, 42, 900), (37, 5, 4000), (3000, 64, 70), (1, 1, 1), (500, 42, 500)])
def test_reverse_table_is_the_sorted_transpose(nq, h, ns):
    """rev.ent[rev.ptr[s]:rev.ptr[s+1]] = ascending list of the queries whose row holds s (shadow entries skipped)."""
    rng = np.random.default_rng(nq + h)
    idx = np.stack([rng.permutation(max(ns, h) + 3)[:h] for _ in range(nq)]).astype(np.int32)  # no repeats within a row
    idx[idx >= ns] = ns                                                                       # -> shadow entries
    if ns == 70:
        idx[:, 0] = 7   # one support listed by every query: a 3000-entry row (the > 64 path of the sort kernel)
    tab = cu(idx)
    rev = ops.build_reverse_table(tab, ns)
    assert getattr(tab, "_d3f_rev") is rev and rev.edges() == int((idx < ns).sum())
    ptr, ent = rev.ptr.cpu().numpy(), rev.ent.cpu().numpy()
    qs, hs = np.nonzero(idx < ns)
    order = np.lexsort((qs, idx[qs, hs]))
    counts = np.bincount(idx[qs, hs], minlength=ns)
    assert np.array_equal(ptr, np.concatenate([[0], np.cumsum(counts)]))
    assert np.array_equal(ent[:ptr[-1]], qs[order])
    again = ops.build_reverse_table(cu(idx), ns)
    assert torch.equal(again.ent[:int(ptr[-1])], rev.ent[:int(ptr[-1])])


def _rev_sets_exact(rev):
    """[set of queries per support] of an exact-form reverse table (rows = float4 {q - s, bits of q}, padded with Nq)."""
    q = rev.rel[:, :, 3].contiguous().view(torch.int32).cpu().numpy()
    out = []
    for row in q:
        live = row[row < rev.Nq]
        assert (row[:live.size] < rev.Nq).all()          # compacted: the live entries lead
        out.append(set(int(v) for v in live))
    return out


def _rev_sets_csr(rev):
    ptr, ent = rev.ptr.cpu().numpy(), rev.ent.cpu().numpy()
    return [set(ent[ptr[s]:ptr[s + 1]].tolist()) for s in range(rev.Ns)]


def _rev_sets_search(rev, q, s, r):
    """Decode a search-form transpose on the host: entries of row s that pass key(q,s) <= last_key[q] (and d2 < r2
    when the rows come from a wider search)."""
    ent, lk = rev.ent.cpu().numpy(), rev.last_key.cpu().numpy().view(np.uint64)
    out = []
    for si in range(rev.Ns):
        row = ent[si][ent[si] < rev.Nq]
        d = (q[row] - s[si]).astype(np.float32)
        d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]).astype(np.float32)
        key = (d2.view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.uint64(si)
        ok = key <= lk[row]
        if rev.radius > 0:
            ok &= d2 < np.float32(rev.radius) * np.float32(rev.radius)
        out.append(set(row[ok].tolist()))
    return out


@pytest.mark.parametrize("n0,n1,r,lim", [(2500, 1800, 0.11, 20), (1500, 1, 0.2, 42), (3000, 2000, 0.07, 8)])
def test_search_form_transpose_equals_the_csr_transpose(n0, n1, r, lim):
    """The whole ranked list of s, filtered by key(q,s) <= last_key[q], is exactly the set of queries whose CAPPED row
    lists s -- for a cloud searched against itself (conv tables) and for coarse-over-fine tables (pooling)."""
    rng = np.random.default_rng(n0)
    lens = np.array([n0, n1], np.int32)
    fine = _cloud(rng, n0 + n1)
    coarse, clen = dl.batch_grid_subsampling_kpconv(cu(fine), cu(lens), sampleDl=r * 0.8)
    coarse_np = coarse.cpu().numpy()
    g_fine = ops.RadiusGrid(cu(fine), cu(lens), r)
    # conv: same cloud both ways
    tab, wide, lk = g_fine.query(cu(fine), cu(lens), lim, wide=ops.REV_WIDTH_CONV * 2, want_last_key=True)
    rev = ops.ReverseTable(wide, n0 + n1, lim, n0 + n1, last_key=lk)
    csr = ops.build_reverse_table(tab, n0 + n1)
    assert _rev_sets_search(rev, fine, fine, r) == _rev_sets_csr(csr)
    assert int((tab < n0 + n1).sum()) == rev.edges() or lim < int((wide < n0 + n1).sum(1).max())   # truncation happened
    # pooling: queries = coarse points, supports = fine points; transpose = fine points searched over the coarse cloud
    tabp, lkp = g_fine.query(coarse, clen, lim, want_last_key=True)
    g_coarse = ops.RadiusGrid(coarse, clen, 2 * r)
    widep = g_coarse.query(cu(fine), cu(lens), 1, radius=r, wide=64, table=False)
    revp = ops.ReverseTable(widep, coarse.shape[0], lim, n0 + n1, last_key=lkp)
    csrp = ops.build_reverse_table(tabp, n0 + n1)
    assert _rev_sets_search(revp, coarse_np, fine, r) == _rev_sets_csr(csrp)
    # ... or, as the pyramid does it, the leading part (d2 < r2) of the upsampling table's rows (radius 2r)
    up = g_coarse.query(cu(fine), cu(lens), 42)
    revu = ops.ReverseTable(up, coarse.shape[0], lim, n0 + n1, last_key=lkp, radius=r, status=g_coarse.status)
    assert _rev_sets_search(revu, coarse_np, fine, r) == _rev_sets_csr(csrp)
    xf = cu(rng.normal(size=(n0 + n1, 32)).astype(np.float32))
    wp = cu((rng.normal(size=(15, 32, 32)) / 20).astype(np.float32))
    kpp = cu((rng.normal(size=(15, 3)) * r / 3).astype(np.float32))
    gop = cu(rng.normal(size=(coarse.shape[0], 32)).astype(np.float32))
    # the exact form (d3f_reverse_table_filter: membership evaluated once, rows compacted as {q - s, q}) holds the same edges
    exu, exp_ = ops.filter_reverse_table(revu, coarse, cu(fine)), ops.filter_reverse_table(revp, coarse, cu(fine))
    assert exu.edges() == exp_.edges() == csrp.edges() == int((tabp < n0 + n1).sum())
    assert _rev_sets_exact(exu) == _rev_sets_csr(csrp) and _rev_sets_exact(exp_) == _rev_sets_csr(csrp)
    gp = []
    for rv in (revu, revp, csrp, exu, exp_):
        gx = xf.clone().requires_grad_(True)
        ops.kpconv(coarse, cu(fine), tabp, gx, kpp, wp, r * 0.8, rev=rv).backward(gop)
        gp.append(gx.grad)
    for gpi in gp[:2] + gp[3:]:
        assert rel_err(gpi.cpu().numpy(), gp[2].cpu().numpy()) < 1e-5
    g_fine.status.raise_if_set()
    g_coarse.status.raise_if_set()
    # and the two forms give the same gradient, bit for bit up to summation order inside a row
    x = cu(rng.normal(size=(n0 + n1, 32)).astype(np.float32))
    w = cu((rng.normal(size=(15, 32, 32)) / 20).astype(np.float32))
    kp = cu((rng.normal(size=(15, 3)) * r / 3).astype(np.float32))
    go = cu(rng.normal(size=(n0 + n1, 32)).astype(np.float32))
    grads = []
    ex = ops.filter_reverse_table(rev, cu(fine), cu(fine))
    assert ex.edges() == csr.edges() and _rev_sets_exact(ex) == _rev_sets_csr(csr)
    for rv in (rev, csr, ex):
        gx = x.clone().requires_grad_(True)
        ops.kpconv(cu(fine), cu(fine), tab, gx, kp, w, r * 0.8, rev=rv).backward(go)
        grads.append(gx.grad)
    assert rel_err(grads[0].cpu().numpy(), grads[1].cpu().numpy()) < 1e-5
    assert rel_err(grads[2].cpu().numpy(), grads[1].cpu().numpy()) < 1e-5
    # the gather form (taken from ops.DX_GATHER_MIN_ROWS support rows) is deterministic: same bits on a second backward
    if n0 + n1 >= ops.DX_GATHER_MIN_ROWS:
        gx = x.clone().requires_grad_(True)
        ops.kpconv(cu(fine), cu(fine), tab, gx, kp, w, r * 0.8, rev=ex).backward(go)
        assert torch.equal(gx.grad, grads[2])


@pytest.mark.parametrize("nq,ns,h,cin,cout", [(1000, 1000, 42, 32, 32), (333, 1000, 37, 64, 64), (4500, 4500, 42, 64, 64),
                                              (257, 300, 45, 128, 128), (97, 154, 23, 512, 512), (300, 400, 42, 16, 16),
                                              (300, 400, 40, 32, 64), (150, 160, 42, 256, 128), (2100, 2100, 42, 64, 32),
                                              (571, 2053, 42, 128, 128), (5000, 900, 64, 32, 32)])
def test_kpconv_grad_input_as_a_gather_over_the_reverse_table(nq, ns, h, cin, cout, monkeypatch):
    """grad_x = sum_k (sum_{q in rev(s)} w gn[q]) W[k]^T  ==  the oracle's autograd gradient; bit-identical run to run
    (no atomics), on the fused path and on the aggregate + GEMM path of the few-point layers."""
    monkeypatch.setattr(ops, "DX_GATHER_MIN_ROWS", 0)   # (the training step uses it from 4096 support rows)
    rng = np.random.default_rng(nq + cin)
    q, s, idx, x, kp, w = _kpconv_case(rng, nq, ns, h, cin, cout)
    ext = 0.05
    tx = torch.from_numpy(x).requires_grad_(True)
    ref = ops_ref.kpconv(torch.from_numpy(q), torch.from_numpy(s), torch.from_numpy(idx), tx, torch.from_numpy(kp),
                         torch.from_numpy(w), ext)
    go = torch.from_numpy(rng.normal(size=ref.shape).astype(np.float32))
    ref.backward(go)
    tab = cu(idx, torch.int32)
    rev = ops.build_reverse_table(tab, ns)
    grads = []
    for _ in range(2):
        gx, gw = cu(x).requires_grad_(True), cu(w).requires_grad_(True)
        out = ops.kpconv(cu(q), cu(s), tab, gx, cu(kp), gw, ext)          # finds the table's transpose on the table
        out.backward(cu(go))
        grads.append(gx.grad.clone())
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < FWD_TOL
    assert rel_err(grads[0].cpu().numpy(), tx.grad.numpy()) < BWD_TOL
    assert torch.equal(grads[0], grads[1])
    # block form (KPConv + bias + LeakyReLU), explicit rev argument, int64 table as the reference hands it
    bias = cu(rng.normal(size=cout).astype(np.float32))
    gx = cu(x).requires_grad_(True)
    y = ops.kpconv_bias_act(cu(q), cu(s), cu(idx), gx, cu(kp), cu(w), ext, bias, slope=0.1, rev=rev)
    y.backward(cu(go))
    tx2 = torch.from_numpy(x).requires_grad_(True)
    r2 = torch.nn.functional.leaky_relu(ops_ref.kpconv(torch.from_numpy(q), torch.from_numpy(s), torch.from_numpy(idx), tx2,
                                                        torch.from_numpy(kp), torch.from_numpy(w), ext) + bias.cpu(), 0.1)
    r2.backward(go)
    assert rel_err(y.detach().cpu().numpy(), r2.detach().numpy()) < FWD_TOL
    assert rel_err(gx.grad.cpu().numpy(), tx2.grad.numpy()) < BWD_TOL
    with pytest.raises(RuntimeError):   # a transpose of another table is refused
        ops.kpconv(cu(q)[:-1], cu(s), tab[:-1], cu(x).requires_grad_(True), cu(kp), cu(w), ext, rev=rev)


@pytest.mark.parametrize("n0,n1,r,lim,cin,cout", [(2600, 1900, 0.11, 42, 64, 64), (1500, 900, 0.2, 24, 128, 128),
                                                  (3000, 1, 0.07, 8, 64, 128), (2300, 2100, 0.11, 48, 32, 64)])
def test_kpconv_as_aggregation_kernels_plus_gemms(n0, n1, r, lim, cin, cout, monkeypatch):
    """Wide layers (round 4): LeakyReLU(KPConv(x) + b) as  direct aggregation kernel (registers -> HBM) + tall GEMM +
    row-divided epilogue, its grad-input as  TRANSPOSED aggregation over the exact-form reverse table + GEMM with the
    permuted weights, its weight gradient through the reduction-parallel A^T B kernel -- against the oracle's autograd
    (blocks.py:359-380) and against the fused kernels, conv tables (a cloud against itself) and pooling tables (coarse
    queries over fine supports); grad_x bit-identical on a second backward (no atomics anywhere)."""
    rng = np.random.default_rng(n0 + cin)
    lens = np.array([n0, n1], np.int32)
    fine = _cloud(rng, n0 + n1)
    ns = n0 + n1
    g_fine = ops.RadiusGrid(cu(fine), cu(lens), r)
    tab, wide, lk = g_fine.query(cu(fine), cu(lens), lim, wide=ops.REV_WIDTH_CONV * 2, want_last_key=True)
    ex = ops.filter_reverse_table(ops.ReverseTable(wide, ns, lim, ns, last_key=lk, status=g_fine.status), cu(fine), cu(fine))
    coarse, clen = dl.batch_grid_subsampling_kpconv(cu(fine), cu(lens), sampleDl=r * 0.8)
    nc = int(coarse.shape[0])
    tabp, lkp = g_fine.query(coarse, clen, lim, want_last_key=True)
    g_coarse = ops.RadiusGrid(coarse, clen, 2 * r)
    up = g_coarse.query(cu(fine), cu(lens), 42)
    exp_ = ops.filter_reverse_table(ops.ReverseTable(up, nc, lim, ns, last_key=lkp, radius=r, status=g_coarse.status),
                                    coarse, cu(fine))
    x = np.abs(rng.normal(size=(ns, cin))).astype(np.float32) * (rng.random((ns, 1)) > 0.05)   # some all-zero rows: nn
    w = (rng.normal(size=(15, cin, cout)) / np.sqrt(15 * cin)).astype(np.float32)
    kp = (rng.normal(size=(15, 3)) * r / 3).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    ext = r * 0.8
    for q_np, q_t, table, rev in ((fine, cu(fine), tab, ex), (coarse.cpu().numpy(), coarse, tabp, exp_)):
        nq = int(q_t.shape[0])
        go = rng.normal(size=(nq, cout)).astype(np.float32)
        tx, tw = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(w).requires_grad_(True)
        idx64 = table.cpu().long()
        ref = torch.nn.functional.leaky_relu(
            ops_ref.kpconv(torch.from_numpy(q_np), torch.from_numpy(fine), idx64, tx, torch.from_numpy(kp), tw, ext)
            + torch.from_numpy(b), 0.1)
        ref.backward(torch.from_numpy(go))
        res = []
        for split in (True, False, True):
            monkeypatch.setattr(ops, "_GEMM_PATH_MIN_CIN", 16 if split else 1 << 30)
            monkeypatch.setattr(ops, "_GEMM_DX_AGG_MIN_COUT", 16 if split else 1 << 30)
            monkeypatch.setattr(ops, "_GEMM_DX_MAX_ROWS", 0)
            monkeypatch.setattr(ops, "DX_GATHER_MIN_ROWS", 0)
            gx, gw, gb = cu(x).requires_grad_(True), cu(w).requires_grad_(True), cu(b).requires_grad_(True)
            y = ops.kpconv_bias_act(q_t, cu(fine), table, gx, cu(kp), gw, ext, gb, slope=0.1, rev=rev)
            y.backward(cu(go))
            res.append([t.detach().cpu().numpy() for t in (y, gx.grad, gw.grad, gb.grad)])
        assert rel_err(res[0][0], ref.detach().numpy()) < FWD_TOL
        assert rel_err(res[0][1], tx.grad.numpy()) < BWD_TOL
        assert rel_err(res[0][2], tw.grad.numpy()) < BWD_TOL
        for a, c in zip(res[0], res[1]):                 # split == fused
            assert rel_err(a, c) < 2e-5
        assert np.array_equal(res[0][1], res[2][1])      # deterministic grad_x
    g_fine.status.raise_if_set()
    g_coarse.status.raise_if_set()


@pytest.mark.parametrize("nq,ns,h,cin,cout", [(1000, 1000, 42, 32, 32), (97, 154, 23, 512, 512), (300, 400, 42, 16, 16),
                                              (150, 160, 42, 256, 128), (200, 260, 42, 24, 40)])
@pytest.mark.parametrize("min_rows", [1, 1 << 30])  # reduction-parallel kernel / library GEMM for grad_W
def test_kpconv_backward_saved_vs_recomputed_aggregation(nq, ns, h, cin, cout, min_rows, monkeypatch):
    """grad_W from the weighted features the forward leaves behind == grad_W with the aggregation recomputed."""
    monkeypatch.setattr(ops, "_SPLITK_MIN_ROWS", min_rows)
    rng = np.random.default_rng(nq * 7 + cin)
    q, s, idx, x, kp, w = _kpconv_case(rng, nq, ns, h, cin, cout)
    go = cu(rng.normal(size=(nq, cout)).astype(np.float32))
    grads = []
    for save in (True, False):
        monkeypatch.setattr(ops, "SAVE_WEIGHTED_FEATURES", save)
        gx, gw = cu(x).requires_grad_(True), cu(w).requires_grad_(True)
        ops.kpconv(cu(q), cu(s), cu(idx), gx, cu(kp), gw, 0.05).backward(go)
        grads.append((gx.grad.cpu().numpy(), gw.grad.cpu().numpy()))
    assert rel_err(grads[0][0], grads[1][0]) < 1e-5
    assert rel_err(grads[0][1], grads[1][1]) < 1e-5
    # weight gradient alone (frozen features) and feature gradient alone (frozen weights)
    monkeypatch.setattr(ops, "SAVE_WEIGHTED_FEATURES", True)
    gw = cu(w).requires_grad_(True)
    ops.kpconv(cu(q), cu(s), cu(idx), cu(x), cu(kp), gw, 0.05).backward(go)
    assert rel_err(gw.grad.cpu().numpy(), grads[0][1]) < 1e-6
    gx = cu(x).requires_grad_(True)
    ops.kpconv(cu(q), cu(s), cu(idx), gx, cu(kp), cu(w), 0.05).backward(go)
    assert rel_err(gx.grad.cpu().numpy(), grads[0][0]) < 1e-5


@pytest.mark.parametrize("n,cin,cout", [(38001, 64, 128), (7919, 128, 32), (2050, 32, 128), (159, 1024, 512), (3, 16, 16),
                                        (1000, 48, 80), (500, 3072, 1024), (601, 24, 40)])
@pytest.mark.parametrize("min_rows", [1, 4096])
def test_linear_weight_gradient(n, cin, cout, min_rows, monkeypatch):
    """y = x W^T: grad_W from the reduction-parallel kernel (or the library GEMM: few rows, odd widths) vs float64."""
    monkeypatch.setattr(ops, "_SPLITK_MIN_ROWS", min_rows)
    rng = np.random.default_rng(n + cin)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    w = (rng.normal(size=(cout, cin)) / np.sqrt(cin)).astype(np.float32)
    go = rng.normal(size=(n, cout)).astype(np.float32)
    tx, tw = cu(x).requires_grad_(True), cu(w).requires_grad_(True)
    y = ops.linear_nobias(tx, tw)
    y.backward(cu(go))
    assert rel_err(y.detach().cpu().numpy(), (x.astype(np.float64) @ w.astype(np.float64).T)) < 1e-5
    assert rel_err(tw.grad.cpu().numpy(), go.astype(np.float64).T @ x.astype(np.float64)) < 1e-5
    assert rel_err(tx.grad.cpu().numpy(), go.astype(np.float64) @ w.astype(np.float64)) < 1e-5
    # deterministic (no atomics): a second backward gives the same bits
    tw2 = cu(w).requires_grad_(True)
    ops.linear_nobias(cu(x), tw2).backward(cu(go))
    assert torch.equal(tw2.grad, tw.grad)


def _group_call(problems):
    """d3f_linear_grad_weight_group on [(x, grad_out, grad_w 2-D view, bias_part or None, bias rows, gb, gb2)]."""
    import ctypes
    L = _native.lib()
    n = len(problems)
    arr = (_native.AtbProblem * n)()
    for q, (x, go, gw, bpart, gb, gb2) in zip(arr, problems):
        q.x, q.grad_out, q.grad_w = x.data_ptr(), go.data_ptr(), gw.data_ptr()
        q.N, q.Cin, q.Cout, q.ldw = x.shape[0], x.shape[1], go.shape[1], gw.stride(0)
        if bpart is not None:
            q.bias_part, q.bias_blocks, q.bias_cols = bpart.data_ptr(), bpart.shape[0], bpart.shape[1]
            q.grad_bias, q.grad_bias2 = gb.data_ptr(), (gb2.data_ptr() if gb2 is not None else None)
    nbytes = L.d3f_linear_grad_weight_group_ws_bytes(arr, n)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    _native.check(L.d3f_linear_grad_weight_group(arr, n, ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream),
                  "d3f_linear_grad_weight_group")
    return nbytes


# (rows, Cin, Cout): every tile shape of the grouped kernel (16 / 32 / 64-wide panels on either side), >= 1.4 GFLOP
# problems, rows that are no multiple of 16, row counts whose last partition is ragged, partition counts that are no
# multiple of 8, a problem with fewer rows than one partition group, the S1 stack's largest shapes
_GROUP_SHAPES = [(114688, 384, 32), (23872, 256, 256), (6208, 512, 512), (6208, 128, 1920), (23871, 64, 960),
                 (114683, 32, 480), (38001, 64, 128), (4099, 16, 16), (5003, 48, 80), (7001, 32, 16), (9013, 16, 64),
                 (333, 64, 64), (61, 32, 32), (100003, 16, 48), (12345, 96, 160),
                 # few rows x large outputs: the undivided ("direct") form, written straight to the (strided) target
                 (462, 512, 7680), (1713, 256, 3840), (462, 2048, 1024), (159, 512, 2048), (1001, 1024, 512)]


@pytest.mark.parametrize("task_us,pipe", [(0, 0), (5, 0), (0, 1), (5, 1)])
def test_grouped_weight_gradients_match_float64(task_us, pipe):
    """d3f_linear_grad_weight_group: ALL problems in one launch pair == float64 grad_out^T x per problem, bit-identical
    on a second run and when a problem is computed alone; bias partial sums finished by the same second stage; a
    strided target (column block of a wider matrix) written in place with its neighbours untouched.  task_us = 5:
    four times as many row partitions (ragged last partitions, partition counts off the multiples of 8).  pipe = 0: the
    software-pipelined task body (buffer-load LDS-DMA, rows past the operand zero-filled by the range check); 1: the
    plain one (the fallback for operands beyond 4 GiB)."""
    old = _native.set_tunables(atb_task_us=task_us, atb_pipe=pipe)
    try:
        rng = np.random.default_rng(5)
        probs, refs = [], []
        for k, (n, cin, cout) in enumerate(_GROUP_SHAPES):
            x = torch.from_numpy(rng.normal(size=(n, cin)).astype(np.float32)).cuda()
            go = torch.from_numpy(rng.normal(size=(n, cout)).astype(np.float32)).cuda()
            pad = 16 if k % 3 == 1 else 0                       # every third target is a column block
            full = torch.full((cout, pad + cin + pad), 7.0, dtype=torch.float32, device="cuda")
            gw = full[:, pad:pad + cin]
            bpart = gb = gb2 = None
            if k % 2 == 0:
                bpart = torch.from_numpy(rng.normal(size=(37 + k, cout)).astype(np.float32)).cuda()
                gb = torch.empty(cout, device="cuda")
                gb2 = torch.empty(cout, device="cuda") if k % 4 == 0 else None
            probs.append((x, go, gw, bpart, gb, gb2))
            refs.append((full, pad, (go.double().t() @ x.double()).cpu().numpy()))
        _group_call(probs)
        torch.cuda.synchronize()
        first = []
        for (x, go, gw, bpart, gb, gb2), (full, pad, ref) in zip(probs, refs):
            assert rel_err(gw.cpu().numpy(), ref) < 2e-5, (tuple(x.shape), tuple(go.shape))
            if pad:
                assert bool((full[:, :pad] == 7.0).all()) and bool((full[:, -pad:] == 7.0).all())
            if bpart is not None:
                assert rel_err(gb.cpu().numpy(), bpart.double().sum(0).cpu().numpy()) < 1e-5
                if gb2 is not None:
                    assert torch.equal(gb, gb2)
            first.append((gw.clone(), gb.clone() if gb is not None else None))
            gw.fill_(0.0)
        _group_call(probs)                                      # no atomics: the same bits
        for (x, go, gw, bpart, gb, gb2), (w0, b0) in zip(probs, first):
            assert torch.equal(gw, w0)
            assert b0 is None or torch.equal(gb, b0)
        for k in (1, 4, 9, 16, 19):                             # a problem alone == the problem inside the group
            x, go, gw, bpart, gb, gb2 = probs[k]
            gw.fill_(0.0)
            _group_call([probs[k]])
            assert torch.equal(gw, first[k][0])
    finally:
        _native.set_tunables(**old)


# (R, K, N): the step's own shapes (KPConv contractions, unary blocks, decoder) + ragged rows, 16- and 32-wide outputs
_XW_SHAPES = [(512, 7680, 512), (477, 3840, 256), (1792, 1920, 128), (6208, 960, 64), (23808, 128, 256), (1743, 1024, 1024),
              (512, 2048, 512), (6159, 256, 512), (100, 64, 16), (37, 16, 32), (3001, 192, 96), (1, 128, 64), (257, 1024, 48)]


def _xw_ref(x, b, row_div, b1, add, b2, slope):
    v = x.double() @ b.double()
    if row_div is not None:
        v = v / row_div.double()[:, None]
    for t in (b1, add, b2):
        if t is not None:
            v = v + t.double()
    return torch.where(v > 0, v, v * slope).cpu().numpy()


@pytest.mark.parametrize("rows,split", [(0, 0), (2, 1), (4, 8), (2, 16), (4, 1)])
def test_gemm_epilogue_matches_float64(rows, split):
    """d3f_gemm_epilogue (own f32-MFMA GEMM with the block's epilogue): y = act(x B / row_div + b1 + add + b2) for
    B = w^T (nn.Linear, blocks.py:481-541), B = w (KPConv's wf @ W, blocks.py:362-374, and grad-input products) and the
    block form (KPConv weights read as the permuted matrix of the transposed-aggregation grad-input) vs float64; every
    tile shape (rows = 2 / 4: 32- / 64-row blocks), undivided and split reductions (8 / 16 partitions), ragged row
    counts, row-strided operands and outputs, all epilogue combinations; bit-identical on a second run."""
    old = _native.set_tunables(xw_rows=rows, xw_split=split)
    try:
        rng = np.random.default_rng(11)
        for k, (R, K, N) in enumerate(_XW_SHAPES):
            for mode in (ops.GEMM_NT, ops.GEMM_NN):
                padx, padw, pady = (16 if k % 3 == 1 else 0), (32 if k % 2 == 1 else 0), (16 if k % 4 == 2 else 0)
                xf = torch.from_numpy(rng.normal(size=(R, K + padx)).astype(np.float32)).cuda()
                x = xf[:, :K]
                if mode == ops.GEMM_NT:
                    wf_ = torch.from_numpy(rng.normal(size=(N, K + padw)).astype(np.float32)).cuda()
                    w, b = wf_[:, :K], wf_[:, :K].t()
                else:
                    wf_ = torch.from_numpy(rng.normal(size=(K, N + padw)).astype(np.float32)).cuda()
                    w, b = wf_[:, :N], wf_[:, :N]
                combo = (k + mode) % 4
                row_div = torch.from_numpy(rng.integers(1, 40, size=R).astype(np.float32)).cuda() if combo in (1, 3) else None
                b1 = torch.from_numpy(rng.normal(size=N).astype(np.float32)).cuda() if combo >= 1 else None
                add = torch.from_numpy(rng.normal(size=(R, N)).astype(np.float32)).cuda() if combo >= 2 else None
                b2 = torch.from_numpy(rng.normal(size=N).astype(np.float32)).cuda() if combo == 3 else None
                slope = 1.0 if combo == 0 else 0.1
                full = torch.full((R, N + 2 * pady), 7.0, dtype=torch.float32, device="cuda")
                out = full[:, pady:pady + N]
                zi = torch.full((3 * N,), 5.0, device="cuda") if combo == 1 else None
                assert ops.gemm_epilogue_ok(x, w, mode, R, K, N, 0, x.stride(0), w.stride(0), b1, add, b2)
                ops.gemm_epilogue(x, w, mode, R, K, N, 0, x.stride(0), w.stride(0), row_div, b1, add, b2, slope, zi, out)
                ref = _xw_ref(x, b, row_div, b1, add, b2, slope)
                assert rel_err(out.cpu().numpy(), ref) < 2e-5, (R, K, N, mode, combo)
                if pady:
                    assert bool((full[:, :pady] == 7.0).all()) and bool((full[:, -pady:] == 7.0).all())
                if zi is not None:
                    assert bool((zi == 0.0).all())
                first = out.clone()
                out.fill_(0.0)
                ops.gemm_epilogue(x, w, mode, R, K, N, 0, x.stride(0), w.stride(0), row_div, b1, add, b2, slope, None, out)
                assert torch.equal(out, first), (R, K, N, mode)
        # block form: A [Ns, Kp Cout] . W', W'[k, o, c] = W[k, c, o], read in place from W [Kp, Cin, Cout]
        for (Ns, Kp, Cin, Cout) in [(1792, 15, 256, 256), (6208, 15, 128, 128), (23801, 15, 64, 64), (300, 3, 16, 192)]:
            A = torch.from_numpy(rng.normal(size=(Ns, Kp * Cout)).astype(np.float32)).cuda()
            W = torch.from_numpy(rng.normal(size=(Kp, Cin, Cout)).astype(np.float32)).cuda()
            addend = torch.from_numpy(rng.normal(size=(Ns, Cin)).astype(np.float32)).cuda()
            assert ops.gemm_epilogue_ok(A, W, ops.GEMM_NT, Ns, Kp * Cout, Cin, Cout)
            out = ops.gemm_epilogue(A, W, ops.GEMM_NT, Ns, Kp * Cout, Cin, kblock=Cout, add=addend)
            ref = (A.double() @ W.permute(0, 2, 1).reshape(Kp * Cout, Cin).double() + addend.double()).cpu().numpy()
            assert rel_err(out.cpu().numpy(), ref) < 2e-5, (Ns, Kp, Cin, Cout)
    finally:
        _native.set_tunables(**old)


def test_batched_weight_permute_equals_torch_and_one_launch_serves_a_backward():
    """d3f_permute_kpconv_weights: W'[k, o, c] = W[k, c, o] for several KPConv weight tensors in one launch == torch's
    permute; ops._permuted_weights launches it once for every queued layer and never hands out a copy of an earlier step."""
    import ctypes
    rng = np.random.default_rng(4)
    ws = [torch.from_numpy(rng.normal(size=shp).astype(np.float32)).cuda()
          for shp in [(15, 64, 64), (15, 128, 128), (3, 32, 96), (15, 256, 256), (1, 32, 32)]]
    n = len(ws)
    outs = [torch.empty((w.shape[0] * w.shape[2], w.shape[1]), device="cuda") for w in ws]
    arr = lambda t, v: (t * n)(*v)
    rc = _native.lib().d3f_permute_kpconv_weights(
        arr(ctypes.c_void_p, [w.data_ptr() for w in ws]), arr(ctypes.c_void_p, [o.data_ptr() for o in outs]),
        arr(ctypes.c_int, [w.shape[0] for w in ws]), arr(ctypes.c_int, [w.shape[1] for w in ws]),
        arr(ctypes.c_int, [w.shape[2] for w in ws]), n, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    for w, o in zip(ws, outs):
        assert torch.equal(o, w.permute(0, 2, 1).contiguous().view(o.shape))
    # the queue: three layers registered by their forward, the first request serves all of them
    st = ops._wperm_state()
    st['queue'].clear()                               # (whatever earlier tests' forwards without a backward left behind)
    st['ready'].clear()
    for w in ws[:3]:
        ops._queue_weight_permute(w)
    first = ops._permuted_weights(ws[1])
    assert torch.equal(first, ws[1].permute(0, 2, 1).contiguous().view(first.shape))
    assert len(st['queue']) == 0 and set(st['ready']) == {ws[0].data_ptr(), ws[2].data_ptr()}
    assert torch.equal(ops._permuted_weights(ws[0]), outs[0]) and ws[0].data_ptr() not in st['ready']
    ws[2].mul_(2.0)                                   # a new step: the weights changed, the forward queues them again
    ops._queue_weight_permute(ws[2])
    assert ws[2].data_ptr() not in st['ready']        # the stale copy is gone
    again = ops._permuted_weights(ws[2])
    assert torch.equal(again, ws[2].permute(0, 2, 1).contiguous().view(again.shape))
    assert torch.equal(ops._permuted_weights(ws[3]), outs[3])      # never queued: permuted on the spot
    # bad arguments
    assert _native.lib().d3f_permute_kpconv_weights(None, None, None, None, None, 1, None) == -1
    bad = torch.zeros((2, 48, 32), device="cuda")      # Cin no multiple of 32
    assert _native.lib().d3f_permute_kpconv_weights(
        (ctypes.c_void_p * 1)(bad.data_ptr()), (ctypes.c_void_p * 1)(outs[0].data_ptr()), (ctypes.c_int * 1)(2),
        (ctypes.c_int * 1)(48), (ctypes.c_int * 1)(32), 1, torch.cuda.current_stream().cuda_stream) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("N", [4096 + 37, 70001])
def test_linear_pair_equals_the_two_unary_launches(N):
    """ops.linear_pair_bias_act (unary2 + shortcut unary + LeakyReLU of a bottleneck in one launch, reference
    blocks.py:658-686; the forward kernel keeps both weight matrices in registers) == linear_bias_act(x1, residual =
    linear_bias_act(x2)): outputs to rounding, all eight gradients (two inputs, two weights, four biases) against the
    composition and against float64; the shortcut input's gradient handed to a GradHolder when asked."""
    rng = np.random.default_rng(8)
    mk = lambda *shape: torch.from_numpy(rng.normal(size=shape).astype(np.float32)).cuda().requires_grad_(True)
    x1, x2 = mk(N, 32), mk(N, 64)
    w1, w2 = mk(128, 32), mk(128, 64)
    bs = [mk(128) for _ in range(4)]
    assert ops.linear_pair_supported(N, 32, 64, 128, x1, x2, w1, w2)
    g = torch.from_numpy(rng.normal(size=(N, 128)).astype(np.float32)).cuda()
    leaves = [x1, w1, bs[0], bs[1], x2, w2, bs[2], bs[3]]

    def grads(fn, **kw):
        for t in leaves:
            t.grad = None
        with ops.weight_grad_group() if kw.get('group') else _nullcontext():
            out = fn()
            torch.autograd.backward(out, g)
        return out.detach(), [t.grad.clone() if t.grad is not None else None for t in leaves]

    two = lambda: ops.linear_bias_act(x1, w1, bs[0], ops.linear_bias_act(x2, w2, bs[2], None, bs[3], slope=1.0), bs[1], slope=0.1)
    one = lambda: ops.linear_pair_bias_act(x1, w1, bs[0], bs[1], x2, w2, bs[2], bs[3], slope=0.1)
    o2, g2 = grads(two)
    o1, g1 = grads(one)
    assert rel_err(o1.cpu().numpy(), o2.cpu().numpy()) < 2e-6
    v = x1.detach().double() @ w1.detach().double().t() + x2.detach().double() @ w2.detach().double().t() + \
        sum(b.detach().double() for b in bs)
    ref = torch.where(v > 0, v, 0.1 * v)
    assert rel_err(o1.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    gm = g.double() * torch.where(v > 0, 1.0, 0.1)
    want = [gm @ w1.detach().double(), gm.t() @ x1.detach().double(), gm.sum(0), gm.sum(0),
            gm @ w2.detach().double(), gm.t() @ x2.detach().double(), gm.sum(0), gm.sum(0)]
    for a, b, w in zip(g1, g2, want):
        assert rel_err(a.cpu().numpy(), w.cpu().numpy()) < BWD_TOL and rel_err(b.cpu().numpy(), w.cpu().numpy()) < BWD_TOL
    # inside a weight_grad_group: the bias gradients ride in the grouped second stage; same values
    _, g3 = grads(one, group=True)
    for a, w in zip(g3, want):
        assert rel_err(a.cpu().numpy(), w.cpu().numpy()) < BWD_TOL
    # the shortcut input's gradient deposited with a sibling branch
    holder = ops.GradHolder()
    for t in leaves:
        t.grad = None
    out = ops.linear_pair_bias_act(x1, w1, bs[0], bs[1], x2, w2, bs[2], bs[3], slope=0.1, grad_deposit2=holder)
    torch.autograd.backward(out, g)
    assert x2.grad is None and rel_err(holder.collect().cpu().numpy(), want[4].cpu().numpy()) < BWD_TOL


def test_gemm_epilogue_rejects_bad_arguments():
    L = _native.lib()
    assert not L.d3f_gemm_epilogue_supported(100, 24, 64, 0, 0)        # K no multiple of 16
    assert not L.d3f_gemm_epilogue_supported(100, 64, 40, 0, 0)        # N no multiple of 16
    assert not L.d3f_gemm_epilogue_supported(100, 960, 64, 1, 64)      # the block form belongs to mode 0
    assert not L.d3f_gemm_epilogue_supported(100, 960, 64, 0, 48)      # blocks of 64 reduction indices
    x = torch.zeros((100, 64), device="cuda")
    w = torch.zeros((32, 64), device="cuda")
    y = torch.zeros((100, 32), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    args = lambda xp, ldx: (xp, ldx, w.data_ptr(), 64, 0, 0, 100, 64, 32, None, None, None, 0, None, 1.0, y.data_ptr(),
                            32, None, 0, None, 0, st)
    assert L.d3f_gemm_epilogue(*args(x.data_ptr(), 64)) == 0
    assert L.d3f_gemm_epilogue(*args(x.data_ptr() + 4, 64)) == -1      # 16-byte alignment
    assert L.d3f_gemm_epilogue(*args(x.data_ptr(), 32)) == -1          # leading dimension below K
    torch.cuda.synchronize()


def test_grouped_launch_with_more_problems_than_one_table_holds():
    """60 problems > the 48 entries of one kernel-argument table: the call splits into two launch pairs."""
    rng = np.random.default_rng(6)
    probs = []
    for k in range(60):
        n, cin, cout = 4096 + 37 * k, 16 * (1 + k % 3), 16 * (1 + (k // 3) % 4)
        x = torch.from_numpy(rng.normal(size=(n, cin)).astype(np.float32)).cuda()
        go = torch.from_numpy(rng.normal(size=(n, cout)).astype(np.float32)).cuda()
        probs.append((x, go, torch.empty((cout, cin), device="cuda"), None, None, None))
    _group_call(probs)
    for x, go, gw, *_ in probs:
        assert rel_err(gw.cpu().numpy(), (go.double().t() @ x.double()).cpu().numpy()) < 2e-5


def test_grouped_launch_rejects_bad_problems():
    L = _native.lib()
    x = torch.zeros((4096, 32), device="cuda")
    go = torch.zeros((4096, 24), device="cuda")       # 24 is no multiple of 16
    gw = torch.zeros((24, 32), device="cuda")
    arr = (_native.AtbProblem * 1)()
    arr[0].x, arr[0].grad_out, arr[0].grad_w = x.data_ptr(), go.data_ptr(), gw.data_ptr()
    arr[0].N, arr[0].Cin, arr[0].Cout, arr[0].ldw = 4096, 32, 24, 32
    assert L.d3f_linear_grad_weight_group_ws_bytes(arr, 1) == 0
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    assert L.d3f_linear_grad_weight_group(arr, 1, ws.data_ptr(), 1 << 20, None) == -1
    arr[0].Cout, arr[0].ldw = 32, 16                  # row stride below the row length
    assert L.d3f_linear_grad_weight_group(arr, 1, ws.data_ptr(), 1 << 20, None) == -1
    arr[0].ldw = 32
    assert L.d3f_linear_grad_weight_group(arr, 1, ws.data_ptr(), 16, None) == -2   # workspace too small
    assert L.d3f_linear_grad_weight_group(arr, 0, None, 0, None) == 0


def test_weight_grad_group_defers_and_equals_immediate_launches(monkeypatch):
    """ops.weight_grad_group around a backward pass: every unary / KPConv weight gradient and bias gradient equals the
    one-launch-per-layer result (rounds 1-5) -- including the shortcut whose deposited gradient is accumulated by the
    next GEMM (out of place while operands are queued) and the decoder's column-block target."""
    rng = np.random.default_rng(11)
    n, nc = 9000, 2300
    x0 = rng.normal(size=(n, 128)).astype(np.float32)
    xc = rng.normal(size=(nc, 256)).astype(np.float32)
    idx = rng.integers(0, nc, size=(n, 1)).astype(np.int32)
    # two 128 -> 128 layers on the library-GEMM path (the second adds the block input: its masked gradient is BOTH the
    # queued operand of its weight gradient and the buffer the first layer's grad-input GEMM would accumulate into)
    ws_ = [(rng.normal(size=s) / np.sqrt(s[1])).astype(np.float32) for s in ((128, 128), (128, 128), (64, 256 + 128))]
    bs = [rng.normal(size=c).astype(np.float32) for c in (128, 128, 64)]

    def run(grouped):
        monkeypatch.setattr(ops, "GROUP_WEIGHT_GRADS", grouped)
        t = cu(x0).requires_grad_(True)
        W = [cu(w).requires_grad_(True) for w in ws_]
        B = [cu(b).requires_grad_(True) for b in bs]
        holder = ops.GradHolder()     # the identity shortcut's gradient is accumulated by the first layer's grad-input GEMM
        h = ops.linear_bias_act(t, W[0], B[0], slope=0.1, grad_holder=holder)
        h = ops.linear_bias_act(h, W[1], B[1], add=ops.grad_tap(t, holder), slope=0.1)
        y = ops.upsample_linear_bias_act(cu(xc), cu(idx), h, W[2], B[2], slope=0.1)
        with ops.weight_grad_group() as g:
            y.sum().backward()
            assert (g is not None and len(g.problems) >= 3) if grouped else g is None
        torch.cuda.synchronize()
        return [p.grad.clone() for p in W + B + [t]]

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert rel_err(u.cpu().numpy(), v.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("n,cin,cout,with_add,slope", [(5000, 64, 32, False, 0.1), (4133, 32, 128, True, 0.1),
                                                       (8200, 64, 256, True, 0.1), (33000, 64, 128, False, 0.1),
                                                       (4097, 32, 64, False, 1.0), (70, 128, 128, True, 0.1),
                                                       (4500, 48, 64, False, 0.1)])
def test_fused_unary_block_matches_unfused(n, cin, cout, with_add, slope, monkeypatch):
    """act(x W^T + b1 + add + b2): row-streaming fused kernels (forward, grad_x) == library GEMM + epilogue path."""
    rng = np.random.default_rng(n + cin)
    x = rng.normal(size=(n, cin)).astype(np.float32)
    w = (rng.normal(size=(cout, cin)) / np.sqrt(cin)).astype(np.float32)
    b1, b2 = rng.normal(size=cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    add = rng.normal(size=(n, cout)).astype(np.float32) if with_add else None
    go = rng.normal(size=(n, cout)).astype(np.float32)
    res = []
    for min_rows in (1, 1 << 30):
        monkeypatch.setattr(ops, "_FUSED_LINEAR_MIN_ROWS", min_rows)
        t = [cu(a).requires_grad_(True) if a is not None else None for a in (x, w, b1, add, b2)]
        y = ops.linear_bias_act(t[0], t[1], t[2], t[3], t[4], slope=slope)
        y.backward(cu(go))
        res.append([y.detach().cpu().numpy()] + [a.grad.cpu().numpy() if a is not None else None for a in t])
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b1 + b2 + (add if with_add else 0.0)
    ref = np.where(ref > 0, ref, ref * slope)
    assert rel_err(res[0][0], ref) < 1e-5
    for a, b in zip(res[0], res[1]):
        if a is not None:
            assert rel_err(a, b) < 2e-5


@pytest.mark.parametrize("nq,ns,h,cin,cout", [(97, 154, 23, 512, 512), (150, 160, 42, 256, 128), (300, 400, 42, 16, 16),
                                              (257, 300, 45, 128, 128), (571, 2053, 42, 64, 256)])
def test_kpconv_bias_act_gemm_path_matches_fused_path(nq, ns, h, cin, cout, monkeypatch):
    """LeakyReLU(KPConv(x) + b): aggregation kernel + library GEMMs + row-divided epilogue (few-point layers) ==
    fused KPConv kernel + epilogue, values and all three gradients."""
    rng = np.random.default_rng(nq + cout)
    q, s, idx, x, kp, w = _kpconv_case(rng, nq, ns, h, cin, cout)
    b = rng.normal(size=cout).astype(np.float32)
    go = rng.normal(size=(nq, cout)).astype(np.float32)
    res = []
    for rows in (1 << 30, 0):
        monkeypatch.setattr(ops, "_GEMM_DX_MAX_ROWS", rows)
        tx, tw, tb = cu(x).requires_grad_(True), cu(w).requires_grad_(True), cu(b).requires_grad_(True)
        y = ops.kpconv_bias_act(cu(q), cu(s), cu(idx), tx, cu(kp), tw, 0.05, tb, slope=0.1)
        y.backward(cu(go))
        res.append([t.detach().cpu().numpy() for t in (y, tx.grad, tw.grad, tb.grad)])
    ref = ops_ref.kpconv(*[torch.from_numpy(a) for a in (q, s, idx, x, kp, w)], 0.05).numpy() + b
    ref = np.where(ref > 0, ref, 0.1 * ref)
    assert rel_err(res[0][0], ref) < FWD_TOL
    for a, c in zip(res[0], res[1]):
        assert rel_err(a, c) < 2e-5


@pytest.mark.parametrize("nc,n,cc,cs,cout,slope", [(192, 640, 2048, 1024, 1024, 0.1), (640, 2112, 64, 32, 128, 0.1),
                                                   (300, 5000, 128, 64, 64, 0.1), (50, 177, 48, 16, 32, 1.0)])
def test_upsample_linear_commutes_with_gather(nc, n, cc, cs, cout, slope):
    """act([x_c[idx] | skip] W^T + b1 + b2) with the upsampled half of the product computed on the coarse rows ==
    nearest upsample + concatenation + linear + epilogue, values and all five gradients."""
    rng = np.random.default_rng(n + cc)
    xc = rng.normal(size=(nc, cc)).astype(np.float32)
    skip = rng.normal(size=(n, cs)).astype(np.float32)
    w = (rng.normal(size=(cout, cc + cs)) / np.sqrt(cc + cs)).astype(np.float32)
    b1, b2 = rng.normal(size=cout).astype(np.float32), rng.normal(size=cout).astype(np.float32)
    idx = rng.integers(0, nc + 1, size=(n, 5)).astype(np.int64)  # nc = shadow -> zero row
    go = rng.normal(size=(n, cout)).astype(np.float32)
    txc, tsk, tw, tb1, tb2 = [torch.from_numpy(a).double().requires_grad_(True) for a in (xc, skip, w, b1, b2)]
    up = torch.cat([txc, torch.zeros(1, cc, dtype=torch.float64)], 0)[torch.from_numpy(idx[:, 0])]
    ref = torch.cat([up, tsk], 1) @ tw.t() + tb1 + tb2
    ref = torch.where(ref > 0, ref, ref * slope)
    ref.backward(torch.from_numpy(go).double())
    g = [cu(a).requires_grad_(True) for a in (xc, skip, w, b1, b2)]
    out = ops.upsample_linear_bias_act(g[0], cu(idx), g[1], g[2], g[3], g[4], slope=slope)
    out.backward(cu(go))
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    for a, b in zip(g, (txc, tsk, tw, tb1, tb2)):
        assert rel_err(a.grad.cpu().numpy(), b.grad.numpy()) < 2e-5


def test_kpconv_all_shadow_rows_and_empty():
    rng = np.random.default_rng(0)
    q, s, idx, x, kp, w = _kpconv_case(rng, 64, 80, 10, 32, 32)
    idx[:5] = 80
    out = ops.kpconv(cu(q), cu(s), cu(idx), cu(x), cu(kp), cu(w), 0.05).cpu().numpy()
    assert np.all(out[:5] == 0)
    ref = ops_ref.kpconv(*[torch.from_numpy(a) for a in (q, s, idx, x, kp, w)], 0.05).numpy()
    assert rel_err(out, ref) < FWD_TOL


# ------------------------------------------------------------------------------------------------ pools
@pytest.mark.parametrize("c", [128, 33, 1024])
def test_pools(c):
    rng = np.random.default_rng(c)
    ns, nq, h = 500, 211, 17
    x = rng.normal(size=(ns, c)).astype(np.float32)
    idx = rng.integers(0, ns + 1, size=(nq, h)).astype(np.int64)
    idx[:3] = ns
    for fn, ref_fn in ((ops.max_pool, ops_ref.max_pool), (ops.closest_pool, ops_ref.closest_pool)):
        tx = torch.from_numpy(x).requires_grad_(True)
        ref = ref_fn(tx, torch.from_numpy(idx))
        go = torch.from_numpy(rng.normal(size=ref.shape).astype(np.float32))
        ref.backward(go)
        gx = cu(x).requires_grad_(True)
        out = fn(gx, cu(idx))
        out.backward(cu(go))
        assert np.array_equal(out.detach().cpu().numpy(), ref.detach().numpy())
        assert rel_err(gx.grad.cpu().numpy(), tx.grad.numpy()) < 1e-6
        # second backward through the same node (fresh, not pre-cleared scatter target)
        gx2 = cu(x).requires_grad_(True)
        out2 = fn(gx2, cu(idx))
        out2.backward(cu(go), retain_graph=True)
        out2.backward(cu(go))
        assert rel_err(gx2.grad.cpu().numpy(), 2 * tx.grad.numpy()) < 1e-6
    # the decoder's pattern: upsample, concatenate with a skip, and let the gradient arrive as a COLUMN SLICE
    skip = rng.normal(size=(nq, 24)).astype(np.float32)
    wgt = rng.normal(size=(nq, c + 24)).astype(np.float32)
    tx = torch.from_numpy(x).requires_grad_(True)
    (torch.cat([ops_ref.closest_pool(tx, torch.from_numpy(idx)), torch.from_numpy(skip)], dim=1)
     * torch.from_numpy(wgt)).sum().backward()
    gx = cu(x).requires_grad_(True)
    (torch.cat([ops.closest_pool(gx, cu(idx)), cu(skip)], dim=1) * cu(wgt)).sum().backward()
    assert rel_err(gx.grad.cpu().numpy(), tx.grad.numpy()) < 1e-6
    # ... and the same with the concatenation done by the pooling launch itself
    gx, gsk = cu(x).requires_grad_(True), cu(skip).requires_grad_(True)
    y = ops.closest_pool(gx, cu(idx), skip=gsk)
    assert y.shape == (nq, c + 24)
    ref_y = torch.cat([ops_ref.closest_pool(torch.from_numpy(x), torch.from_numpy(idx)), torch.from_numpy(skip)], dim=1)
    assert np.array_equal(y.detach().cpu().numpy(), ref_y.numpy())
    (y * cu(wgt)).sum().backward()
    assert rel_err(gx.grad.cpu().numpy(), tx.grad.numpy()) < 1e-6
    assert np.array_equal(gsk.grad.cpu().numpy(), wgt[:, c:])


# ------------------------------------------------------------------------------------------------ block epilogue
@pytest.mark.parametrize("n,c,slope,with_add", [(1000, 128, 0.1, True), (333, 32, 0.1, False), (77, 6, 1.0, True),
                                                (4000, 2048, 0.1, True), (50, 64, 1.0, False), (38001, 32, 0.1, True),
                                                (5000, 256, 0.1, False), (600, 1024, 0.1, True), (333, 48, 0.1, True)])
def test_bias_act(n, c, slope, with_add):
    import torch.nn.functional as F
    rng = np.random.default_rng(n + c)
    x = rng.normal(size=(n, c)).astype(np.float32)
    b1, b2 = rng.normal(size=c).astype(np.float32), rng.normal(size=c).astype(np.float32)
    a = rng.normal(size=(n, c)).astype(np.float32)
    go = rng.normal(size=(n, c)).astype(np.float32)
    tx, tb1, tb2, ta = [torch.from_numpy(v).requires_grad_(True) for v in (x, b1, b2, a)]
    pre = tx + tb1 + tb2 + (ta if with_add else 0)
    ref = F.leaky_relu(pre, slope) if slope != 1.0 else pre
    ref.backward(torch.from_numpy(go))
    gx, gb1, gb2, ga = [cu(v).requires_grad_(True) for v in (x, b1, b2, a)]
    out = ops.bias_act(gx, gb1, ga if with_add else None, gb2, slope=slope)
    out.backward(cu(go))
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-6
    assert rel_err(gx.grad.cpu().numpy(), tx.grad.numpy()) < 1e-6
    assert rel_err(gb1.grad.cpu().numpy(), tb1.grad.numpy()) < 2e-5
    assert rel_err(gb2.grad.cpu().numpy(), tb2.grad.numpy()) < 2e-5
    if with_add:
        assert rel_err(ga.grad.cpu().numpy(), ta.grad.numpy()) < 1e-6


# ------------------------------------------------------------------------------------------------ detector score
@pytest.mark.parametrize("c", [32, 16, 48, 64])
def test_detection_scores(c):
    rng = np.random.default_rng(c)
    n, h = 1500, 30
    feat = rng.normal(size=(n, c)).astype(np.float32)
    feat[rng.random(n) < 0.05] = 0.0
    idx = rng.integers(0, n + 1, size=(n, h)).astype(np.int64)
    idx[:, 0] = np.arange(n)
    tf = torch.from_numpy(feat).requires_grad_(True)
    ref = ops_ref.detection_scores(tf, torch.from_numpy(idx), training=True)
    go = torch.from_numpy(rng.normal(size=ref.shape).astype(np.float32))
    ref.backward(go)
    gf = cu(feat).requires_grad_(True)
    out = ops.detection_scores(gf, cu(idx), training=True)
    out.backward(cu(go))
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < FWD_TOL
    assert rel_err(gf.grad.cpu().numpy(), tf.grad.numpy()) < BWD_TOL
    ref_e = ops_ref.detection_scores(torch.from_numpy(feat), torch.from_numpy(idx), training=False).numpy()
    out_e = ops.detection_scores(cu(feat), cu(idx), training=False).cpu().numpy()
    assert np.array_equal(out_e != 0, ref_e != 0)
    assert rel_err(out_e, ref_e) < FWD_TOL
    # a table kept wider than the reference would build it (static shapes): with the device-resident max count the
    # local-maximum gate sees the reference's columns only (extra all-shadow columns would add a zero candidate)
    full = np.sort(idx, axis=1)                                   # shadow entries (== n) at the row end
    full[:, -4:] = rng.integers(0, n, size=(n, 4))                # ... and no shadow entry in the last columns at all
    full = np.sort(full, axis=1)
    wide = np.concatenate([full, np.full((n, 7), n, np.int64)], axis=1)
    fneg = (-(np.abs(feat) + 0.1)).astype(np.float32)             # all-negative features: a zero candidate always wins
    fneg[0, 0] = 0.5                                               # (one positive value keeps the normaliser sane)
    ref_w = ops_ref.detection_scores(torch.from_numpy(fneg), torch.from_numpy(full), training=False).numpy()
    width = torch.tensor([h], dtype=torch.int32, device=DEV)
    out_w = ops.detection_scores(cu(fneg), cu(wide), training=False, width=width).cpu().numpy()
    assert np.array_equal(out_w != 0, ref_w != 0) and rel_err(out_w, ref_w) < FWD_TOL
    naive = ops.detection_scores(cu(fneg), cu(wide), training=False).cpu().numpy()
    assert (ref_w != 0).sum() > 10 and (naive != 0).sum() < (ref_w != 0).sum()   # the data does tell the two apart


def test_detection_backward_with_the_sparse_gradient_of_the_detector_loss():
    """The detector loss reads the scores of the correspondences only (reference utils/loss.py:140-158), so the gradient
    entering the backward is zero at most points; the backward kernel leaves at such a point.  Against the oracle's
    autograd on a gradient that is non-zero at 3 % of the points; rows that neither carry a gradient nor neighbor a point
    that does must come out exactly zero (but for the arg-max row of the normaliser)."""
    rng = np.random.default_rng(21)
    n, h, c = 5000, 40, 32
    feat = rng.normal(size=(n, c)).astype(np.float32)
    idx = rng.integers(0, n + 1, size=(n, h)).astype(np.int64)
    idx[:, 0] = np.arange(n)
    go = rng.normal(size=(n, 1)).astype(np.float32)
    go[rng.random(n) > 0.03] = 0.0
    assert 50 < int((go != 0).sum()) < 400
    tf = torch.from_numpy(feat).requires_grad_(True)
    ops_ref.detection_scores(tf, torch.from_numpy(idx), training=True).backward(torch.from_numpy(go))
    gf = cu(feat).requires_grad_(True)
    ops.detection_scores(gf, cu(idx), training=True).backward(cu(go))
    assert rel_err(gf.grad.cpu().numpy(), tf.grad.numpy()) < BWD_TOL
    # rows that neither carry a gradient nor neighbor a point that does receive the normaliser's term only
    touched = np.zeros(n + 1, bool)
    act = np.nonzero(go[:, 0])[0]
    touched[act] = True
    touched[idx[act].reshape(-1)] = True
    quiet = ~touched[:n]
    assert quiet.sum() > 100
    fmax_rows = np.any(feat == feat.max(), axis=1)
    g = gf.grad.cpu().numpy()
    assert np.all(g[quiet & ~fmax_rows] == 0.0)


# ------------------------------------------------------------------------------------------------ loss
@pytest.mark.parametrize("m", [128, 64, 37])
def test_circle_det_loss(m):
    rng = np.random.default_rng(m)
    c = 32
    a = rng.normal(size=(m, c)).astype(np.float32)
    p = (a + 0.3 * rng.normal(size=(m, c))).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    kp = rng.random((m, 3))
    dk = np.linalg.norm(kp[:, None] - kp[None], axis=-1)  # float64 like scipy cdist
    sa, sp = rng.random((m, 1)).astype(np.float32), rng.random((m, 1)).astype(np.float32)
    ta, tp = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(p).requires_grad_(True)
    tsa, tsp = torch.from_numpy(sa).requires_grad_(True), torch.from_numpy(sp).requires_grad_(True)
    loss, acc, fp, an, dists = ops_ref.circle_loss(ta, tp, torch.from_numpy(dk))
    det = ops_ref.det_loss(dists, tsa, tsp)
    (1.0 * loss + 0.7 * det).backward()
    ga, gp = cu(a).requires_grad_(True), cu(p).requires_grad_(True)
    gsa, gsp = cu(sa).requires_grad_(True), cu(sp).requires_grad_(True)
    scalars, d, fpo, ano = ops.circle_det_loss(ga, gp, cu(dk), gsa, gsp)
    (1.0 * scalars[0] + 0.7 * scalars[1]).backward()
    s = scalars.detach().cpu().numpy()
    assert abs(s[0] - loss.item()) < 1e-5 * max(1, abs(loss.item()))
    assert abs(s[1] - det.item()) < 1e-5
    assert abs(s[2] - float(acc)) < 1e-3
    assert rel_err(d.cpu().numpy(), dists.detach().numpy()) < 1e-5
    assert rel_err(fpo.cpu().numpy(), fp.detach().numpy()) < 1e-5
    assert rel_err(ano.cpu().numpy(), an.detach().numpy()) < 1e-5
    assert rel_err(ga.grad.cpu().numpy(), ta.grad.numpy()) < BWD_TOL
    assert rel_err(gp.grad.cpu().numpy(), tp.grad.numpy()) < BWD_TOL
    assert rel_err(gsa.grad.cpu().numpy(), tsa.grad.numpy()) < 1e-5
    assert rel_err(gsp.grad.cpu().numpy(), tsp.grad.numpy()) < 1e-5


@pytest.mark.parametrize("n,c,m", [(5000, 32, 128), (300, 16, 64), (77, 48, 20)])
def test_select_normalize(n, c, m):
    """normalize(x)[ia], normalize(x)[ip + n0], scores[...] in one launch each way == F.normalize + indexing."""
    rng = np.random.default_rng(n)
    x = rng.normal(size=(n, c)).astype(np.float32)
    x[3] = 0.0  # a zero row: divided by eps like F.normalize
    sc = rng.normal(size=(n, 1)).astype(np.float32)
    n0 = n // 2
    ia = rng.integers(0, n0, size=m)
    ip = rng.integers(0, n - n0, size=m)
    ia[0], ip[1] = 3, ip[0]  # the zero row, and a repeated positive
    ga, gp = rng.normal(size=(m, c)).astype(np.float32), rng.normal(size=(m, c)).astype(np.float32)
    gsa, gsp = rng.normal(size=m).astype(np.float32), rng.normal(size=m).astype(np.float32)
    tx, ts = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(sc).requires_grad_(True)
    f = torch.nn.functional.normalize(tx, p=2, dim=-1)
    ref = (f[ia], f[ip + n0], ts[ia].reshape(-1), ts[ip + n0].reshape(-1))
    torch.autograd.backward(ref, [torch.from_numpy(a) for a in (ga, gp, gsa, gsp)])
    gx, gs = cu(x).requires_grad_(True), cu(sc).requires_grad_(True)
    off = torch.tensor([n0, n - n0], dtype=torch.int32, device=DEV)[:1]
    out = ops.select_normalize(gx, gs, cu(ia), cu(ip), off)
    torch.autograd.backward(out, [cu(a) for a in (ga, gp, gsa, gsp)])
    for a, b in zip(out, ref):
        assert rel_err(a.detach().cpu().numpy(), b.detach().numpy()) < 1e-6
    mask = np.ones(n, bool)
    mask[3] = False  # d(x/eps)/dx = 1/eps = 1e12: compared separately
    assert rel_err(gx.grad.cpu().numpy()[mask], tx.grad.numpy()[mask]) < 1e-5
    assert rel_err(gx.grad.cpu().numpy()[3], tx.grad.numpy()[3]) < 1e-5
    assert rel_err(gs.grad.cpu().numpy(), ts.grad.numpy()) < 1e-6
    # host-int offset gives the same selection
    out2 = ops.select_normalize(cu(x), cu(sc), cu(ia), cu(ip), n0)
    assert torch.equal(out2[1], out[1].detach())


def test_loss_modules_follow_reference_call_order():
    from d3feat_pytorch_amd.utils.loss import CircleLoss, DetLoss
    rng = np.random.default_rng(9)
    m, c = 64, 32
    a = rng.normal(size=(m, c)).astype(np.float32)
    p = (a + 0.2 * rng.normal(size=(m, c))).astype(np.float32)
    kp = rng.random((m, 3))
    dk = np.linalg.norm(kp[:, None] - kp[None], axis=-1)
    sa, sp = rng.random((m, 1)).astype(np.float32), rng.random((m, 1)).astype(np.float32)
    ta, tp = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(p).requires_grad_(True)
    tsa, tsp = torch.from_numpy(sa).requires_grad_(True), torch.from_numpy(sp).requires_grad_(True)
    loss, acc, fp, an, dists = ops_ref.circle_loss(ta, tp, torch.from_numpy(dk))
    (loss + ops_ref.det_loss(dists, tsa, tsp)).backward()
    ga, gp = cu(a).requires_grad_(True), cu(p).requires_grad_(True)
    gsa, gsp = cu(sa).requires_grad_(True), cu(sp).requires_grad_(True)
    circle = CircleLoss(dist_type='euclidean', log_scale=10, safe_radius=0.1, pos_margin=0.1, neg_margin=1.4)
    desc, accuracy, l_fp, l_an, zero, d = circle(ga, gp, cu(dk))      # trainer.py:96
    det = DetLoss('euclidean')(d, gsa, gsp)                             # trainer.py:97
    (desc + det).backward()
    assert zero == 0 and len(l_fp) == m and abs(float(np.mean(l_fp)) - fp.mean().item()) < 1e-5
    assert abs(desc.item() - loss.item()) < 1e-5 and abs(float(accuracy) - float(acc)) < 1e-3
    assert rel_err(ga.grad.cpu().numpy(), ta.grad.numpy()) < BWD_TOL
    assert rel_err(gsa.grad.cpu().numpy(), tsa.grad.numpy()) < 1e-5


# ------------------------------------------------------------------------------------------------ matching
@pytest.mark.parametrize("ns,nt", [(250, 250), (1000, 777), (5000, 5000)])
def test_mutual_nn(ns, nt):
    rng = np.random.default_rng(ns)
    s = rng.normal(size=(ns, 32)).astype(np.float32)
    t = np.concatenate([s[: nt // 2] + 0.05 * rng.normal(size=(nt // 2, 32)), rng.normal(size=(nt - nt // 2, 32))], 0)
    s /= np.linalg.norm(s, axis=1, keepdims=True)
    t = (t / np.linalg.norm(t, axis=1, keepdims=True)).astype(np.float32)
    ref = ops_ref.build_correspondence(s, t)
    got = build_correspondence(s, t)
    # BLAS vs MFMA summation order can flip exact near-ties: demand > 99.5 % identical pairs
    a = set(map(tuple, ref.tolist()))
    b = set(map(tuple, got.tolist()))
    assert len(a & b) >= 0.995 * max(len(a), 1) and abs(len(a) - len(b)) <= 0.005 * max(len(a), 1) + 1
    # and the argmins are true argmins of the fp64 distance matrix up to fp32 rounding
    row, col, mutual = ops.mutual_nn(cu(s), cu(t))
    dot = s.astype(np.float64) @ t.astype(np.float64).T
    r = row.cpu().numpy()
    assert (dot.max(axis=1) - dot[np.arange(ns), r] < 2e-6).all()
    c = col.cpu().numpy()
    assert (dot.max(axis=0) - dot[c, np.arange(nt)] < 2e-6).all()


@pytest.mark.parametrize("influence", ["linear", "constant", "gaussian"])
@pytest.mark.parametrize("aggregation", ["sum", "closest"])
def test_kpconv_modes_match_reference_vectors(influence, aggregation):
    """Influence / aggregation modes of blocks.py:327-352 through the module API, against vectors computed by the real
    reference (tests/golden/kpconv_modes.npz) and against the oracle on a second, larger random problem."""
    import os
    from d3feat_pytorch_amd.models import blocks
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'kpconv_modes.npz'))
    K, cin, cout = g['weights'].shape
    conv = blocks.KPConv(K, 3, cin, cout, float(g['extent']), float(g['radius']), KP_influence=influence,
                         aggregation_mode=aggregation).to(DEV)
    with torch.no_grad():
        conv.kernel_points.copy_(torch.from_numpy(g['kernel_points']))
        conv.weights.copy_(torch.from_numpy(g['weights']))
    x = torch.from_numpy(g['x']).to(DEV).requires_grad_(True)
    out = conv(torch.from_numpy(g['q_pts']).to(DEV), torch.from_numpy(g['s_pts']).to(DEV),
               torch.from_numpy(g['inds']).to(DEV), x)
    out.backward(torch.from_numpy(g['gout']).to(DEV))
    tag = '%s.%s.' % (influence, aggregation)
    for got, key in ((out.detach(), 'out'), (x.grad, 'grad_x'), (conv.weights.grad, 'grad_w')):
        want = g[tag + key]
        err = np.abs(got.cpu().numpy() - want).max()
        assert err <= 1e-4 * max(1.0, np.abs(want).max()), (key, err)      # tolerance: 1e-4 abs/rel fp32

    # wider channels (two channel chunks per lane group), shadow-heavy table, vs the oracle
    gen = torch.Generator().manual_seed(11)
    ns, nq, H, ci, co = 900, 700, 33, 80, 24
    s_pts = torch.rand((ns, 3), generator=gen)
    q_pts = s_pts[torch.randperm(ns, generator=gen)[:nq]].contiguous()
    d2 = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    dist, order = torch.sort(d2, dim=1)
    inds = torch.where(dist[:, :H] < 0.2 ** 2, order[:, :H], torch.full_like(order[:, :H], ns))
    kp = torch.from_numpy(g['kernel_points']) * 0.8
    w = torch.randn((K, ci, co), generator=gen) * 0.1
    xx = torch.randn((ns, ci), generator=gen)
    go = torch.randn((nq, co), generator=gen)
    a = [t.clone().requires_grad_(True) for t in (xx, w)]
    ref = ops_ref.kpconv(q_pts, s_pts, inds, a[0], kp, a[1], 0.1, influence, aggregation)
    ref.backward(go)
    b = [t.clone().to(DEV).requires_grad_(True) for t in (xx, w)]
    got = ops.kpconv(q_pts.to(DEV), s_pts.to(DEV), inds.to(DEV), b[0], kp.to(DEV), b[1], 0.1, influence, aggregation)
    got.backward(go.to(DEV))
    for name, u, v in (("out", got.detach(), ref.detach()), ("grad_x", b[0].grad, a[0].grad), ("grad_w", b[1].grad, a[1].grad)):
        err = float((u.cpu() - v).abs().max())
        assert err <= 1e-4 * max(1.0, float(v.abs().max())), (name, err)


def test_guarded_sgd_step_matches_torch_sgd_and_skips_on_nonfinite():
    """d3f_sgd_guarded_step vs torch.optim.SGD(momentum, weight_decay) (training_3DMatch.py:62-76) + the guard of
    trainer.py:104-111; device-resident hyper-parameters incl. the gradient scale of the data-parallel mean."""
    from d3feat_pytorch_amd.train import FlatParams, GuardedSGD
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 7)).to(DEV)
    ref = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 7)).to(DEV)
    ref.load_state_dict(model.state_dict())
    flat = FlatParams(model)
    opt = GuardedSGD(flat, lr=0.01, momentum=0.98, weight_decay=1e-6)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.98, weight_decay=1e-6)
    gen = torch.Generator(device=DEV).manual_seed(1)
    for step in range(6):
        g = torch.randn(flat.numel, generator=gen, device=DEV)
        scale = 0.125 if step >= 3 else 1.0           # as if 8 ranks had summed their gradients
        if step == 3:
            opt.grad_scale = 0.125
            opt.lr = 0.004
            for grp in ropt.param_groups:
                grp['lr'] = 0.004
        poisoned = step == 4
        flat.grad.copy_(g)
        if poisoned:
            flat.grad[flat.numel // 2] = float('inf')
        off = 0
        for p in ref.parameters():
            p.grad = (g[off:off + p.numel()] * scale).view_as(p).clone()
            off += p.numel()
        before = flat.data.clone()
        ok = opt.step()
        if poisoned:
            assert not bool(ok) and torch.equal(before, flat.data)
            continue
        assert bool(ok)
        ropt.step()
        want = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
        assert float((flat.data - want).abs().max()) <= 1e-7 * max(1.0, float(want.abs().max())), step
    assert int(opt.skipped) == 1


def test_guarded_sgd_on_gradient_lanes_steps_on_their_sum():
    """d3f_sgd_guarded_step_lanes: the update uses lane 0 + lane 1 (+ ...) in that order, bit-identical to the
    single-buffer step on the pre-added sum; a non-finite value in ANY lane skips it -- also Inf and -Inf at the same
    place in two lanes, which a test of the sum alone would let through as NaN only by luck of the arithmetic."""
    from d3feat_pytorch_amd.train import FlatParams, GuardedSGD
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(41, 29), torch.nn.Linear(29, 6)).to(DEV)   # 1398 floats: scalar tail
    twin = torch.nn.Sequential(torch.nn.Linear(41, 29), torch.nn.Linear(29, 6)).to(DEV)
    twin.load_state_dict(model.state_dict())
    flat, tflat = FlatParams(model), FlatParams(twin)
    assert flat.numel % 4 != 0
    opt, topt = GuardedSGD(flat, lr=0.01, momentum=0.9, weight_decay=1e-4), GuardedSGD(tflat, lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt.grad_scale = topt.grad_scale = 1.0 / 3
    for k in (1, 2):
        assert flat.add_lane() == k
    gen = torch.Generator(device=DEV).manual_seed(2)
    for step in range(4):
        gs = [torch.randn(flat.numel, generator=gen, device=DEV) for _ in range(3)]
        for (buf, _), g in zip(flat.lanes, gs):
            buf.copy_(g)
        tflat.grad.copy_((gs[0] + gs[1]) + gs[2])
        opt.step(want_ok=False, grads=[l[0] for l in flat.lanes])
        topt.step(want_ok=False)
        assert torch.equal(flat.data, tflat.data) and torch.equal(opt.buf, topt.buf), step
    before = flat.data.clone()
    flat.lanes[1][0][7] = float('inf')
    flat.lanes[2][0][7] = float('-inf')
    assert not bool(opt.step(grads=[l[0] for l in flat.lanes])) and torch.equal(before, flat.data)
    flat.lanes[1][0][7] = 0.0
    flat.lanes[2][0][7] = 0.0
    flat.lanes[2][0][flat.numel - 1] = float('nan')          # in the scalar tail
    assert not bool(opt.step(grads=[l[0] for l in flat.lanes])) and int(opt.skipped) == 2
    with pytest.raises(ValueError):
        ops.sgd_guarded_step([flat.grad] * 5, flat.data, opt.buf, 0.1, 0.9, 0.0, opt.state)
    # bind(): weight-gradient slots follow the lane
    w = model[0].weight
    flat.bind(2)
    assert w._d3f_grad_slot.data_ptr() == flat.lanes[2][1][0].data_ptr() and flat.grad.data_ptr() == flat.lanes[2][0].data_ptr()
    flat.bind(0)
    assert w._d3f_grad_slot.data_ptr() == flat.lanes[0][1][0].data_ptr()


# ------------------------------------------------------------------------------------------------ full-size matching
def _golden_s1_match():
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    return np.load(os.path.join(here, 'golden', 's1_match.npz'), allow_pickle=False)


def test_mutual_nn_at_full_size_on_the_reference_descriptors():
    """BASELINE configs[3]: dense N x N matching at 19k x 19k x 32 on the descriptors the REFERENCE produced for the
    benchmark pair (eval mode).  Row / column argmins and the mutual set are compared with what the reference's
    build_correspondence (common.py:5-21, float32 sqrt(2 - 2 S T^T)) computed; every row that picks another column
    must be a distance tie at float32 resolution, proven in float64, and their number is bounded."""
    g = _golden_s1_match()
    n0 = int(g['n0'])
    fe = g['features_eval']
    S, T = fe[:n0], fe[n0:]
    ra, ca, mu = ops.mutual_nn(cu(S), cu(T))
    ra, ca, mu = ra.cpu().numpy().astype(np.int64), ca.cpu().numpy().astype(np.int64), mu.cpu().numpy().astype(bool)
    S64, T64 = S.astype(np.float64), T.astype(np.float64)

    def check(ours, ref, A, B, what):
        flip = np.nonzero(ours != ref)[0]
        d_o = np.linalg.norm(A[flip] - B[ours[flip]], axis=1)
        d_r = np.linalg.norm(A[flip] - B[ref[flip]], axis=1)
        # float32 dot products of unit vectors carry ~32 * 6e-8 of error: d^2 = 2 - 2 s moves by ~4e-6, so two
        # candidates closer than that in d^2 are a tie for BOTH implementations
        assert np.all(np.abs(d_o ** 2 - d_r ** 2) < 8e-6), (what, float(np.abs(d_o ** 2 - d_r ** 2).max()))
        # (random-init descriptors are strongly clustered: ~2 % of the rows have a second candidate within that band)
        assert len(flip) <= 0.05 * len(ours), (what, len(flip), len(ours))
        return len(flip)
    f_rows = check(ra, g['row_argmin'], S64, T64, 'row argmin')
    f_cols = check(ca, g['col_argmin'], T64, S64, 'column argmin')
    ours = set(map(tuple, np.stack([np.nonzero(mu)[0], ra[mu]], 1).tolist()))
    ref = set(map(tuple, g['corr_all'].tolist()))
    assert len(ref) > 1000 and len(ours ^ ref) <= 2 * (f_rows + f_cols), (len(ours), len(ref), f_rows, f_cols)
    print("full-size mutual NN: %d x %d, flipped rows %d, flipped columns %d, |ours ^ ref| = %d of %d" % (
        S.shape[0], T.shape[0], f_rows, f_cols, len(ours ^ ref), len(ref)))
    # keypoint selection + matching as the evaluation does it (test.py:56-57): top-250 and top-5000 by score
    from d3feat_pytorch_amd.geometric_registration.common import select_keypoints
    se = g['scores_eval']
    for k in (250, 5000):
        si = select_keypoints(torch.from_numpy(se[:n0]), k).numpy()
        ti = select_keypoints(torch.from_numpy(se[n0:]), k).numpy()
        # equal scores (the many exact zeros of the eval gate) may be ordered differently: compare on the reference's own
        # selection when the sets differ
        if not (np.array_equal(np.sort(si), np.sort(g['src_idx%d' % k])) and
                np.array_equal(np.sort(ti), np.sort(g['tgt_idx%d' % k]))):
            assert np.array_equal(np.sort(se[:n0][si]), np.sort(se[:n0][g['src_idx%d' % k]]))
        Sk, Tk = S[g['src_idx%d' % k]], T[g['tgt_idx%d' % k]]
        got = build_correspondence(Sk, Tk)
        a, b = set(map(tuple, np.asarray(got).tolist())), set(map(tuple, g['corr%d' % k].tolist()))
        # rows / columns whose two best candidates are closer than float32 can tell (in d^2) may resolve either way;
        # every differing match must involve one of them
        d2 = 2.0 - 2.0 * (Sk.astype(np.float64) @ Tk.astype(np.float64).T)
        r2 = np.partition(d2, 1, axis=1)[:, :2]
        c2 = np.partition(d2, 1, axis=0)[:2, :]
        tie_r = set(np.nonzero(r2[:, 1] - r2[:, 0] < 8e-6)[0].tolist())
        tie_c = set(np.nonzero(c2[1] - c2[0] < 8e-6)[0].tolist())
        for i, j in a ^ b:
            assert i in tie_r or j in tie_c or any(jj in tie_c for ii, jj in (a | b) if ii == i) or \
                any(ii in tie_r for ii, jj in (a | b) if jj == j), (k, i, j)
        assert len(a ^ b) <= 2 * (len(tie_r) + len(tie_c)), (k, len(a), len(b), len(a ^ b), len(tie_r), len(tie_c))
        print("top-%d: %d / %d mutual matches (ours / reference), %d differ; %d tie rows, %d tie columns" % (
            k, len(a), len(b), len(a ^ b), len(tie_r), len(tie_c)))


# ------------------------------------------------------------------------------------------------ batch norm
@pytest.mark.parametrize("N,C,slope,mean", [(5000, 64, 0.1, 0.0), (37, 48, 0.1, 3.0), (20000, 512, 1.0, 0.0),
                                            (20000, 512, 1.0, 50.0), (3, 7, 1.0, 0.0)])
def test_batch_norm_matches_torch(N, C, slope, mean):
    """ops.batch_norm vs torch's nn.BatchNorm1d semantics the reference block uses (blocks.py:465-471): training
    statistics, running-stat update (momentum, unbiased variance), eval mode, and all three gradients, optionally with
    the block's LeakyReLU fused behind.  fp32 tolerance 1e-5 relative (2e-4 on the gradients).  The large-mean case (the
    reason for the two-pass variance) runs without the activation: 1e-5-level differences in (x - mean) flip the
    LeakyReLU branch of a handful of near-zero elements, which no tolerance on the gradient can absorb."""
    rng = np.random.default_rng(N + C)
    x = (rng.normal(size=(N, C)) * rng.uniform(0.5, 2.0, size=C) + mean).astype(np.float32)
    w = rng.uniform(0.5, 1.5, size=C).astype(np.float32)
    b = rng.normal(size=C).astype(np.float32)
    go = rng.normal(size=(N, C)).astype(np.float32)
    rm0, rv0 = rng.normal(size=C).astype(np.float32), rng.uniform(0.5, 2.0, size=C).astype(np.float32)

    def run(device_op, training):
        dev = DEV if device_op else 'cpu'
        dt = torch.float32 if device_op else torch.float64
        xt = torch.tensor(x, dtype=dt, device=dev, requires_grad=True)
        wt = torch.tensor(w, dtype=dt, device=dev, requires_grad=True)
        bt = torch.tensor(b, dtype=dt, device=dev, requires_grad=True)
        rm, rv = torch.tensor(rm0, dtype=dt, device=dev), torch.tensor(rv0, dtype=dt, device=dev)
        if device_op:
            y = ops.batch_norm(xt, wt, bt, rm, rv, training, momentum=0.02, eps=1e-5, slope=slope)
        else:
            y = torch.nn.functional.batch_norm(xt.t().unsqueeze(0), rm, rv, wt, bt, training, 0.02, 1e-5)
            y = torch.nn.functional.leaky_relu(y.squeeze(0).t(), slope) if slope != 1.0 else y.squeeze(0).t()
        y.backward(torch.tensor(go, dtype=dt, device=dev))
        return [t.detach().cpu().double().numpy() for t in (y, rm, rv, xt.grad, wt.grad, bt.grad)]

    for training in (True, False):
        got, ref = run(True, training), run(False, training)
        tol = 1e-5 if mean < 10 else 2e-4          # a mean of 50 costs fp32 digits in (x - mean)
        assert rel_err(got[0], ref[0]) < tol
        assert rel_err(got[1], ref[1]) < 1e-6 and rel_err(got[2], ref[2]) < 1e-5
        for a, r in zip(got[3:], ref[3:]):
            assert rel_err(a, r) < 20 * tol


def test_batch_norm_live_rows_of_a_capacity_buffer():
    rng = np.random.default_rng(9)
    x = rng.normal(size=(700, 32)).astype(np.float32)
    w, b = torch.ones(32, device=DEV), torch.zeros(32, device=DEV)
    full = ops.batch_norm(cu(x[:500]), w, b, None, None, True, slope=0.1)
    xc = cu(x).clone()
    xc[500:] = float('nan')                       # rows past the live count must never be read
    n_live = torch.tensor([500], dtype=torch.int32, device=DEV)
    part = ops.batch_norm(xc, w, b, None, None, True, slope=0.1, n_live=n_live)
    assert torch.equal(part[:500], full) and float(part[500:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ deformable KPConv
@pytest.mark.parametrize("influence,aggregation,modulated", [('linear', 'sum', 0), ('linear', 'sum', 1),
                                                              ('gaussian', 'sum', 1), ('linear', 'closest', 0)])
def test_deformable_kpconv_matches_reference_vectors(influence, aggregation, modulated):
    """KPConv(deformable=True[, modulated=True]) through the module API (offset convolution on the ordinary kernels,
    per-query kernel points in csrc/kpconv_deform.hip) against vectors computed by the real reference
    (tests/golden/kpconv_deform.npz): outputs, min_d2, deformed_KP, and the gradients of the reference's two loss
    routes (output and fitting term) w.r.t. features, kernel weights, offset weights and offset bias.
    Tolerance 1e-4 abs/rel (fp32)."""
    import os
    from d3feat_pytorch_amd.models import blocks
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'kpconv_deform.npz'))
    tag = '%s.%s.%d.' % (influence, aggregation, modulated)
    K, cin, cout = g[tag + 'sd.weights'].shape
    conv = blocks.KPConv(K, 3, cin, cout, float(g['extent']), float(g['radius']), KP_influence=influence,
                         aggregation_mode=aggregation, deformable=True, modulated=bool(modulated))
    sd = {k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + 'sd.')}
    assert set(sd) == set(conv.state_dict())
    conv.load_state_dict(sd)
    conv = conv.to(DEV)
    x = torch.from_numpy(g['x']).to(DEV).requires_grad_(True)
    out = conv(cu(g['q_pts']), cu(g['s_pts']), cu(g['inds']), x)
    ((out * cu(g['gout'])).sum() + 0.7 * (conv.min_d2 * cu(g['gmin'])).sum()).backward()
    checks = [(out.detach(), 'out'), (conv.min_d2.detach(), 'min_d2'), (conv.deformed_KP.detach(), 'deformed_KP'),
              (x.grad, 'grad_x'), (conv.weights.grad, 'grad.weights'),
              (conv.offset_conv.weights.grad, 'grad.offset_conv.weights'), (conv.offset_bias.grad, 'grad.offset_bias')]
    for got, key in checks:
        want = g[tag + key]
        err = np.abs(got.cpu().numpy() - want).max()
        assert err <= 1e-4 * max(1.0, np.abs(want).max()), (key, err)


def test_deformable_kpconv_wide_channels_vs_oracle():
    """Two channel chunks per lane group, shadow-heavy table, offsets large enough to push a third of the neighbors
    out of range -- against the oracle restatement (pinned to the reference vectors in test_oracle_ops.py)."""
    gen = torch.Generator().manual_seed(12)
    ns, nq, H, ci, co, K = 900, 700, 33, 80, 24, 15
    s_pts = torch.rand((ns, 3), generator=gen)
    q_pts = s_pts[torch.randperm(ns, generator=gen)[:nq]].contiguous()
    d2 = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    dist, order = torch.sort(d2, dim=1)
    inds = torch.where(dist[:, :H] < 0.2 ** 2, order[:, :H], torch.full_like(order[:, :H], ns))
    kp = (torch.rand((K, 3), generator=gen) - 0.5) * 0.16
    w = torch.randn((K, ci, co), generator=gen) * 0.1
    xx = torch.randn((ns, ci), generator=gen)
    off = torch.randn((nq, 4 * K), generator=gen) * 0.5
    go, gm = torch.randn((nq, co), generator=gen), torch.randn((nq, K), generator=gen)
    ext = 0.1

    def run(dev, fn):
        a = [t.clone().to(dev).requires_grad_(True) for t in (xx, w, off)]
        o, m, dk = fn(q_pts.to(dev), s_pts.to(dev), inds.to(dev), a[0], kp.to(dev), a[1], a[2])
        ((o * go.to(dev)).sum() + (m * gm.to(dev)).sum()).backward()
        return [o.detach().cpu(), m.detach().cpu()] + [t.grad.cpu() for t in a]

    ref = run('cpu', lambda q, s, i, x, k, ww, of: ops_ref.kpconv_deformable(q, s, i, x, k, ww, ext, of, True))
    got = run(DEV, lambda q, s, i, x, k, ww, of: ops.kpconv_deformable(
        q, s, i, x, k, ww, ext, of[:, :3 * K].reshape(-1, K, 3) * ext, 2 * torch.sigmoid(of[:, 3 * K:])))
    for name, u, v in zip(("out", "min_d2", "grad_x", "grad_w", "grad_offsets"), got, ref):
        assert rel_err(u.numpy(), v.numpy()) < 2e-4, name


# ------------------------------------------------------------------------------------------------ 8 pairs per batch
def test_topk_scores_equals_the_stable_argsort_tail():
    """ops.topk_scores vs np.argsort(kind='stable')[-k:] per cloud (test.py:56-57 selects keypoints that way): gated
    eval scores are mostly exact zeros, so ties decide; k larger than a cloud pads with -1 in front."""
    rng = np.random.default_rng(41)
    lens = [19000, 3, 700, 5000, 1]
    sc = rng.random(sum(lens)).astype(np.float32)
    sc[rng.random(sc.size) < 0.6] = 0.0
    sc[100:140] = sc[100]                                    # a run of equal non-zero scores
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    seg = cu(np.stack([offs, lens], 1).astype(np.int32))
    for k in (1, 250, 5000, ops.TOPK_MAX):
        got = ops.topk_scores(cu(sc), seg, k).cpu().numpy()
        for c, (o, n) in enumerate(zip(offs, lens)):
            want = np.argsort(sc[o:o + n], kind='stable')[-k:]
            assert np.array_equal(got[c, k - len(want):], want), (k, c)
            assert np.all(got[c, :k - len(want)] == -1)


@pytest.mark.parametrize("C", [32, 64])
def test_mutual_nn_batched_equals_per_pair_calls(C):
    """One pair of launches over 5 stacked pairs (one of them empty on the target side, sizes from 1 to 6000 rows) gives
    exactly what 5 separate ops.mutual_nn calls give -- same arg-min rule, same tie order."""
    gen = torch.Generator().manual_seed(5)
    sizes = [(6000, 5500), (1, 300), (257, 64), (1000, 0), (4100, 4099)]
    total = sum(a + b for a, b in sizes)
    desc = torch.nn.functional.normalize(torch.randn((total, C), generator=gen), dim=1)
    desc[10] = desc[3]                                       # exact duplicates: distance ties
    desc[6000 + 7] = desc[6000 + 2]
    d = desc.to(DEV)
    seg, off = [], 0
    for a, b in sizes:
        seg.append([off, a, off + a, b])
        off += a + b
    ra, ca, mu = ops.mutual_nn_batched(d, d, cu(np.asarray(seg, np.int32)), 6000, 5500)
    for so, sn, to, tn in seg:
        if tn == 0:
            assert int(mu[so:so + sn].sum()) == 0
            continue
        r1, c1, m1 = ops.mutual_nn(d[so:so + sn], d[to:to + tn])
        assert torch.equal(ra[so:so + sn], r1) and torch.equal(ca[to:to + tn], c1) and torch.equal(mu[so:so + sn], m1)


def test_grouped_max_pool_and_detection_equal_per_group_calls():
    """A batch that stacks 3 reference batches of 2 clouds: max_pool with one table width per group and the detector
    score with one normaliser + width per group equal the calls on each group alone (widths below the static table
    width, so the trimmed columns matter; negative features make the shadow's zero a live candidate)."""
    rng = np.random.default_rng(6)
    lens_s = np.array([300, 280, 150, 160, 90, 400], np.int32)          # supports per cloud
    lens_q = np.array([120, 100, 60, 70, 30, 150], np.int32)            # queries per cloud
    ns, nq, H, C = int(lens_s.sum()), int(lens_q.sum()), 12, 32
    so, qo = np.concatenate([[0], np.cumsum(lens_s)]), np.concatenate([[0], np.cumsum(lens_q)])
    widths = np.array([5, 9, 12], np.int32)
    idx = np.full((nq, H), ns, np.int32)
    for c in range(6):
        w = widths[c // 2]
        for r in range(qo[c], qo[c + 1]):
            cnt = rng.integers(1, w + 1)
            idx[r, :cnt] = rng.integers(so[c], so[c + 1], cnt)
        idx[qo[c], :w] = rng.integers(so[c], so[c + 1], w)              # one full row per cloud
    x = -np.abs(rng.normal(size=(ns, C))).astype(np.float32)
    got = ops.max_pool(cu(x), cu(idx), width=cu(widths), groups=(cu(lens_q), 2)).cpu().numpy()
    for g in range(3):
        q0, q1, s0, s1 = qo[2 * g], qo[2 * g + 2], so[2 * g], so[2 * g + 2]
        sub = np.where(idx[q0:q1] >= ns, s1 - s0, idx[q0:q1] - s0).astype(np.int32)
        want = ops.max_pool(cu(x[s0:s1]), cu(sub), width=cu(widths[g:g + 1])).cpu().numpy()
        assert np.array_equal(got[q0:q1], want), g
    # detector: self tables over the support clouds
    tab = np.full((ns, H), ns, np.int32)
    for c in range(6):
        w = widths[c // 2]
        for r in range(so[c], so[c + 1]):
            cnt = rng.integers(1, w + 1)
            tab[r, :cnt] = rng.integers(so[c], so[c + 1], cnt)
            tab[r, 0] = r
    f = rng.normal(size=(ns, C)).astype(np.float32) * np.repeat([1.0, 1.0, 3.0, 3.0, 0.2, 0.2], lens_s)[:, None]
    for training in (True, False):
        got = ops.detection_scores(cu(f), cu(tab), training=training, lens=cu(lens_s), width=cu(widths), group=2)
        for g in range(3):
            s0, s1 = so[2 * g], so[2 * g + 2]
            sub = np.where(tab[s0:s1] >= ns, s1 - s0, tab[s0:s1] - s0).astype(np.int32)
            want = ops.detection_scores(cu(f[s0:s1]), cu(sub), training=training, width=cu(widths[g:g + 1]))
            assert torch.equal(got[s0:s1], want), (training, g)
    # training: the gradient through every group's own normaliser (its arg-max row) stays inside the group
    gs = rng.normal(size=(ns, 1)).astype(np.float32)
    ff = cu(f).requires_grad_(True)
    ops.detection_scores(ff, cu(tab), training=True, lens=cu(lens_s), width=cu(widths), group=2).backward(cu(gs))
    for g in range(3):
        s0, s1 = so[2 * g], so[2 * g + 2]
        sub = np.where(tab[s0:s1] >= ns, s1 - s0, tab[s0:s1] - s0).astype(np.int32)
        fg = cu(f[s0:s1]).requires_grad_(True)
        ops.detection_scores(fg, cu(sub), training=True, width=cu(widths[g:g + 1])).backward(cu(gs[s0:s1]))
        assert rel_err(ff.grad[s0:s1].cpu().numpy(), fg.grad.cpu().numpy()) < 1e-5, g


def test_train_loss_of_stacked_pairs_equals_the_per_pair_losses():
    """ops.train_loss_pairs (P pairs stacked: clouds 2p, 2p+1; every pair its own M x M circle + detector problem,
    trainer.py:91-98) == ops.train_loss on each pair alone: losses, statistics and the gradient wrt descriptors / scores
    (total = sum over the pairs, so gradients add)."""
    rng = np.random.default_rng(11)
    P, M, C = 3, 64, 32
    lens = np.array([700, 650, 300, 420, 510, 90], np.int32)
    off = np.concatenate([[0], np.cumsum(lens)])
    N = int(off[-1]) + 37                                       # capacity rows past the live ones
    x = rng.normal(size=(N, C)).astype(np.float32)
    sc = rng.random((N, 1)).astype(np.float32)
    corr = np.stack([np.stack([rng.integers(0, lens[2 * p], M), rng.integers(0, lens[2 * p + 1], M)], 1) for p in range(P)])
    dk = rng.random((P, M, M)) * 0.3
    for w_desc, w_det in ((1.0, 1.0), (0.7, 1.3)):
        tx, ts = cu(x).requires_grad_(True), cu(sc).requires_grad_(True)
        tot, desc, det, acc, fp, an = ops.train_loss_pairs(tx, ts, cu(corr), cu(lens), cu(dk), w_desc=w_desc, w_det=w_det)
        tot.backward()
        gx, gs = torch.zeros_like(tx), torch.zeros_like(ts)
        want = 0.0
        for p in range(P):
            a0, a1 = int(off[2 * p]), int(off[2 * p + 2])
            px, ps = cu(x[a0:a1]).requires_grad_(True), cu(sc[a0:a1]).requires_grad_(True)
            t1, d1, e1, c1, f1, n1 = ops.train_loss(px, ps, cu(corr[p]), int(lens[2 * p]), cu(dk[p]), w_desc=w_desc,
                                                    w_det=w_det)
            t1.backward()
            gx[a0:a1] += px.grad
            gs[a0:a1] += ps.grad
            want += float(t1)
            assert abs(float(desc[p]) - float(d1)) < 1e-6 and abs(float(det[p]) - float(e1)) < 1e-6
            assert abs(float(acc[p]) - float(c1)) < 1e-4
            assert torch.equal(fp[p], f1) and torch.equal(an[p], n1)
        assert abs(float(tot) - want) < 1e-5 * max(1.0, abs(want))
        assert float((tx.grad - gx).abs().max()) <= 1e-6 * float(gx.abs().max()) + 1e-9
        assert float((ts.grad - gs).abs().max()) <= 1e-6 * float(gs.abs().max()) + 1e-9


@pytest.mark.parametrize("ns,nq,cin", [(700, 500, 64), (700, 700, 512), (5000, 5000, 128), (4500, 1200, 32)])
def test_epilogue_packed_supports_equal_the_packing_launch(ns, nq, cin):
    """ops.bias_act(..., pack_for=(s_pts, clear)) leaves the packed supports of the KPConv it feeds behind
    (d3f_bias_act_forward_pack): the KPConv that picks them up (no packing launch, D3F_SPACK_READY) gives the same
    output and gradients as the one that packs for itself -- few-point GEMM-fed path and fused path, rows spanning
    8 lanes up to two waves."""
    gen = torch.Generator().manual_seed(ns + cin)
    H, K, cout = 20, 15, 64
    s_pts = torch.rand((ns, 3), generator=gen)
    q_pts = s_pts[torch.randperm(ns, generator=gen)[:nq]].contiguous()
    d2 = ((q_pts[:, None] - s_pts[None]) ** 2).sum(-1)
    dist, order = torch.sort(d2, dim=1)
    inds = torch.where(dist[:, :H] < 0.15 ** 2, order[:, :H], torch.full_like(order[:, :H], ns)).to(torch.int32)
    z = torch.randn((ns, cin), generator=gen)
    b = torch.randn(cin, generator=gen) * 0.1
    kp = (torch.rand((K, 3), generator=gen) - 0.5) * 0.16
    w = torch.randn((K, cin, cout), generator=gen) * 0.05
    bias = torch.randn(cout, generator=gen) * 0.1
    go = torch.randn((nq, cout), generator=gen)
    res = []
    for pack in (False, True):
        zt, bt = z.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
        wt = w.clone().to(DEV).requires_grad_(True)
        sp = s_pts.to(DEV)
        x = ops.bias_act(zt, bt, slope=0.1, pack_for=(sp, True) if pack else None)
        assert (getattr(x, '_d3f_spack', None) is not None) == pack
        y = ops.kpconv_bias_act(q_pts.to(DEV), sp, inds.to(DEV), x, kp.to(DEV), wt, 0.08, bias.to(DEV), slope=0.1)
        y.backward(go.to(DEV))
        res.append([y.detach().cpu(), zt.grad.cpu(), bt.grad.cpu(), wt.grad.cpu()])
    for name, u, v in zip(("out", "grad_x", "grad_bias", "grad_w"), res[1], res[0]):
        assert rel_err(u.numpy(), v.numpy()) < 1e-6, name
