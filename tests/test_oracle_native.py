"""CPU: the C++ restatement (oracle/native_oracle.cpp) against (a) the reference's own C++ compiled in place
(oracle/_ref, present only where /root/reference is) and (b) the committed golden vectors, which were produced by that
same reference build.  Parity bar: indices exact up to the order inside equal-d2 groups; barycentres and their ROW ORDER
bit-exact."""
import numpy as np
import pytest

import d3feat_pytorch_amd  # noqa: F401  (package must import without a GPU)
from d3feat_pytorch_amd import config as cfgmod
from util import assert_neighbors_equal_tie_aware, sha

LEVELS = 5


def _levels(g):
    pts = [g['batch.points.%d' % l] for l in range(LEVELS)]
    lens = [g['batch.stack_lengths.%d' % l] for l in range(LEVELS)]
    return pts, lens


def test_subsample_matches_golden_bitexact_including_row_order(golden_s0, native):
    pts, lens = _levels(golden_s0)
    dl = 0.03 * 2.5 * 2 / 2.5
    for l in range(LEVELS - 1):
        p, b = native.subsample_batch(pts[l], lens[l], sampleDl=dl * 2 ** l)
        assert np.array_equal(b, lens[l + 1])
        assert np.array_equal(p.view(np.uint32), pts[l + 1].view(np.uint32)), "level %d barycentres / row order" % l


def test_radius_matches_golden_tie_aware(golden_s0, native):
    pts, lens = _levels(golden_s0)
    g = golden_s0
    lim = g['limits']
    ties = 0
    for l in range(LEVELS):
        r = 0.075 * 2 ** l
        ours = native.batch_query(pts[l], pts[l], lens[l], lens[l], radius=r)[:, :lim[l]]
        t, _ = assert_neighbors_equal_tie_aware(pts[l], pts[l], ours, g['batch.neighbors.%d' % l], 'conv%d' % l)
        ties += t
        if l < LEVELS - 1:
            ours = native.batch_query(pts[l + 1], pts[l], lens[l + 1], lens[l], radius=r)[:, :lim[l]]
            assert_neighbors_equal_tie_aware(pts[l + 1], pts[l], ours, g['batch.pools.%d' % l], 'pool%d' % l)
            ours = native.batch_query(pts[l], pts[l + 1], lens[l], lens[l + 1], radius=2 * r)[:, :lim[l]]
            assert_neighbors_equal_tie_aware(pts[l], pts[l + 1], ours, g['batch.upsamples.%d' % l], 'up%d' % l)
    # uncapped tables (width = global max count) as well
    for name, (q, s, ql, sl, r) in {'conv0': (pts[0], pts[0], lens[0], lens[0], 0.075),
                                    'pool0': (pts[1], pts[0], lens[1], lens[0], 0.075),
                                    'up0': (pts[0], pts[1], lens[0], lens[1], 0.15)}.items():
        ours = native.batch_query(q, s, ql, sl, radius=r)
        assert_neighbors_equal_tie_aware(q, s, ours, g['uncapped.' + name], 'uncapped ' + name)


def test_brute_and_grid_methods_agree(native):
    rng = np.random.default_rng(5)
    s = rng.random((3000, 3)).astype(np.float32)
    q = rng.random((700, 3)).astype(np.float32)
    sl, ql = np.array([1800, 1200], np.int32), np.array([300, 400], np.int32)
    a = native.batch_query(q, s, ql, sl, radius=0.11, method='brute')
    b = native.batch_query(q, s, ql, sl, radius=0.11, method='grid')
    assert np.array_equal(a, b)
    # a query only sees its own cloud
    assert (a[:300][a[:300] < 3000] < 1800).all() and (a[300:][a[300:] < 3000] >= 1800).all()


def test_edge_cases(native):
    s = np.array([[0, 0, 0], [1, 0, 0]], np.float32)
    q = np.array([[10, 10, 10]], np.float32)
    with pytest.raises(RuntimeError):  # all-empty result is an error in the reference (wrapper.cpp:201-205)
        native.batch_query(q, s, [1], [2], radius=0.5)
    # strict '<': a support exactly at the radius is excluded
    q = np.array([[0, 0, 0]], np.float32)
    idx = native.batch_query(q, s, [1], [2], radius=1.0)
    assert idx.tolist() == [[0]]
    with pytest.raises(RuntimeError):
        native.batch_query(q[:, :2], s, [1], [2], radius=1.0)
    # duplicate points: ties resolved by index
    s = np.zeros((4, 3), np.float32)
    assert native.batch_query(q, s, [1], [4], radius=1.0).tolist() == [[0, 1, 2, 3]]
    # subsample of a single point / ragged batch with one empty-ish cloud
    p, b = native.subsample_batch(np.array([[0.1, 0.2, 0.3]], np.float32), [1], sampleDl=0.05)
    assert b.tolist() == [1] and np.array_equal(p, np.array([[0.1, 0.2, 0.3]], np.float32))
    # max_p keeps the first rows of each cloud
    rng = np.random.default_rng(0)
    pts = rng.random((500, 3)).astype(np.float32)
    full, fb = native.subsample_batch(pts, [200, 300], sampleDl=0.2)
    cut, cb = native.subsample_batch(pts, [200, 300], sampleDl=0.2, max_p=7)
    assert cb.tolist() == [7, 7]
    assert np.array_equal(cut[:7], full[:7]) and np.array_equal(cut[7:], full[fb[0]:fb[0] + 7])


def test_oracle_vs_compiled_reference(native):
    if not native.have_ref():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(17)
    for trial in range(3):
        n = [2500, 1700]
        pts = (rng.random((sum(n), 3)) * np.array([2.0, 1.5, 0.6])).astype(np.float32)
        lens = np.array(n, np.int32)
        for dl in (0.05, 0.11):
            a, ab = native.subsample_batch(pts, lens, sampleDl=dl)
            b, bb = native.ref_subsample_batch(pts, lens, sampleDl=dl)
            assert np.array_equal(ab, bb) and np.array_equal(a.view(np.uint32), b.view(np.uint32))
            r = 2.5 * dl
            ours = native.batch_query(a, pts, ab, lens, radius=r)
            ref = native.ref_batch_query(a, pts, ab, lens, radius=r)
            assert_neighbors_equal_tie_aware(a, pts, ours, ref, 'trial %d dl %g' % (trial, dl))


def _attr_case(rng, n, n_labels, ldim):
    pts = (rng.random((sum(n), 3)) * np.array([2.0, 1.5, 0.6])).astype(np.float32)
    feats = rng.normal(size=(sum(n), 5)).astype(np.float32)
    # label values spread over negative / large ints so that bucket = (size_t)(int) % B is exercised
    labels = (rng.integers(0, n_labels, size=(sum(n), ldim)) * 7919 - 20000).astype(np.int32)
    return pts, np.array(n, np.int32), feats, labels


def test_subsample_features_and_labels_vs_compiled_reference(native):
    """Feature / label branches (grid_subsampling.h:42-73, .cpp:89-102): the restatement equals the reference's C++ bit
    for bit, including which label wins a tied vote (iteration order of the per-cell unordered_map<int,int>); big voxels
    with up to 40 distinct labels drive that map through its rehashes."""
    if not native.have_ref():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(23)
    for n, n_labels, ldim, dl in (([2500, 1700], 4, 1, 0.11), ([3000, 500], 40, 1, 0.45), ([4000], 40, 3, 0.5),
                                  ([1500], 3, 2, 0.05)):
        pts, lens, feats, labels = _attr_case(rng, n, n_labels, ldim)
        for mp in (0, 6):
            a = native.subsample_batch_ex(pts, lens, feats, labels, sampleDl=dl, max_p=mp)
            b = native.ref_subsample_batch_ex(pts, lens, feats, labels, sampleDl=dl, max_p=mp)
            assert len(a) == len(b) == 4
            for x, y in zip(a, b):
                assert x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x.view(np.uint32), y.view(np.uint32))
        fa = native.subsample_batch_ex(pts, lens, feats, None, sampleDl=dl)
        fb = native.ref_subsample_batch_ex(pts, lens, feats, None, sampleDl=dl)
        la = native.subsample_batch_ex(pts, lens, None, labels, sampleDl=dl)
        lb = native.ref_subsample_batch_ex(pts, lens, None, labels, sampleDl=dl)
        assert len(fa) == len(la) == 3 and np.array_equal(fa[2], fb[2]) and np.array_equal(la[2], lb[2])
        assert la[2].shape[1] == ldim


def test_s1_pyramid_hashes(golden_s1, native):
    """The benchmark pair: regenerate fragments with the oracle subsampler and check SHA-256 of every level."""
    from d3feat_pytorch_amd import synthetic
    sub = lambda p, l, dl: native.subsample_batch(p, l, sampleDl=dl)  # noqa: E731
    a = synthetic.make_fragment(1, sub)
    assert sha(a) == str(golden_s1['pts0.sha'])
    item = synthetic.make_pair(1, 2, sub)
    assert sha(item[1]) == str(golden_s1['pts1.sha'])
    assert np.array_equal(item[4], golden_s1['sel_corr'])
    pts = np.concatenate([item[0], item[1]], 0)
    lens = np.array([len(item[0]), len(item[1])], np.int32)
    for l in range(4):
        pts, lens = native.subsample_batch(pts, lens, sampleDl=0.06 * 2 ** l)
        assert sha(pts) == str(golden_s1['batch.points.%d.sha' % (l + 1)])
        assert np.array_equal(lens, golden_s1['batch.stack_lengths.%d' % (l + 1)])
