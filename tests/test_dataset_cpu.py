"""CPU: the 3DMatch dataset front-end (reference datasets/ThreeDMatch.py) -- pickle format, augmentation semantics and
random-draw order, the >50k-point skip, PLY fragments and their numeric file order."""
import pickle
import random
import struct

import numpy as np
import pytest

from d3feat_pytorch_amd.datasets import ThreeDMatch as tdm


def _write_pickles(root, sizes, pairs, split='train', ds=0.03):
    rng = np.random.default_rng(0)
    pts = {name: rng.uniform(-1, 1, size=(n, 3)) for name, n in sizes.items()}
    corr = {}
    for a, b in pairs:
        m = min(sizes[a], sizes[b]) // 2
        corr['%s@%s' % (a, b)] = np.stack([rng.permutation(sizes[a])[:m], rng.permutation(sizes[b])[:m]], axis=1)
    with open(root / ('3DMatch_%s_%.3f_points.pkl' % (split, ds)), 'wb') as f:
        pickle.dump(pts, f)
    with open(root / ('3DMatch_%s_%.3f_keypts.pkl' % (split, ds)), 'wb') as f:
        pickle.dump(corr, f)
    return pts, corr


def test_training_item_follows_the_reference_recipe(tmp_path):
    sizes = {'s/a': 400, 's/b': 300, 's/c': 350}
    pts, corr = _write_pickles(tmp_path, sizes, [('s/a', 's/b'), ('s/a', 's/c'), ('s/b', 's/c')])
    ds = tdm.ThreeDMatchDataset(str(tmp_path), split='train', num_node=64, downsample=0.03, augment_noise=0.005,
                                augment_axis=1, augment_rotation=1.0, augment_translation=0.5)
    assert len(ds) == 2 and ds.src_to_tgt == {'s/a': ['s/b', 's/c'], 's/b': ['s/c']}
    random.seed(3)
    np.random.seed(3)
    p0, p1, f0, f1, sel, dk = ds[0]
    # the same draws, in the reference's order (ThreeDMatch.py:96-134)
    random.seed(3)
    np.random.seed(3)
    tgt = 's/b' if random.random() > 0.5 else random.choice(['s/b', 's/c'])
    ang = np.random.rand(3) * 2 * np.pi
    axis = random.choice([0, 1, 2])
    trans = np.random.rand(3) * 0.5
    n0 = np.random.rand(400, 3) * 0.005
    n1 = np.random.rand(sizes[tgt], 3) * 0.005
    pick = np.random.choice(len(corr['s/a@' + tgt]), 64, replace=False)
    c, s = np.cos(ang[axis]), np.sin(ang[axis])
    R = [np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
         np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])][axis].astype(np.float32).astype(np.float64)
    assert p0.dtype == p1.dtype == np.float64 and f0.dtype == f1.dtype == np.float32
    assert np.array_equal(p0, pts['s/a'] + n0)
    assert np.allclose(p1, pts[tgt] @ R.T + trans.astype(np.float32).astype(np.float64) + n1, rtol=0, atol=1e-12)
    assert np.array_equal(sel, corr['s/a@' + tgt][pick]) and sel.shape == (64, 2)
    a = p0[sel[:, 0]].astype(np.float32).astype(np.float64)
    assert dk.shape == (64, 64) and dk.dtype == np.float64
    assert np.allclose(dk, np.sqrt(((a[:, None] - a[None]) ** 2).sum(-1)), rtol=0, atol=1e-12)
    assert f0.shape == (400, 1) and f1.shape == (sizes[tgt], 1) and f0.min() == f1.min() == 1.0
    # fewer correspondences than num_node: all of them, in order
    few = tdm.ThreeDMatchDataset(str(tmp_path), num_node=10 ** 6)
    assert np.array_equal(few[1][4], corr['s/b@s/c'])
    # self-augmentation: identity correspondences, 99 % of the input features zeroed
    sa = tdm.ThreeDMatchDataset(str(tmp_path), num_node=16, self_augment=True)
    q0, q1, g0, g1, sc, _ = sa[0]
    assert q0.shape == q1.shape == (400, 3) and np.array_equal(sc[:, 0], sc[:, 1])
    assert int(g0.sum()) == 400 - 396 and int(g1.sum()) == 4
    with pytest.raises(FileNotFoundError):
        tdm.ThreeDMatchDataset(str(tmp_path), split='val')


def test_oversized_fragments_are_skipped(tmp_path):
    _write_pickles(tmp_path, {'big': 50001, 'x': 64, 'y': 64}, [('big', 'x'), ('x', 'y')])
    ds = tdm.ThreeDMatchDataset(str(tmp_path), num_node=8)
    np.random.seed(0)
    random.seed(0)
    for _ in range(5):
        item = ds[0]                       # index 0 is the oversized source: another pair is drawn instead
        assert item[0].shape[0] == 64 and item[1].shape[0] == 64


def _write_ply(path, pts, fmt):
    n = len(pts)
    head = ("ply\nformat %s 1.0\ncomment made by a test\nelement vertex %d\nproperty float x\nproperty float y\n"
            "property float z\nproperty float nx\nproperty uchar red\nelement face 0\n"
            "property list uchar int vertex_indices\nend_header\n" % (fmt, n))
    with open(path, 'wb') as f:
        f.write(head.encode())
        for p in pts:
            if fmt == 'ascii':
                f.write(("%r %r %r 0.5 7\n" % (float(p[0]), float(p[1]), float(p[2]))).encode())
            else:
                f.write(struct.pack(('<' if 'little' in fmt else '>') + 'ffffB', p[0], p[1], p[2], 0.5, 7))


def test_ply_reader_and_testset_order(tmp_path):
    rng = np.random.default_rng(1)
    pts = rng.uniform(-2, 2, size=(57, 3)).astype(np.float32)
    for fmt in ('ascii', 'binary_little_endian', 'binary_big_endian'):
        _write_ply(tmp_path / 'a.ply', pts, fmt)
        got = tdm.read_ply_points(str(tmp_path / 'a.ply'))
        assert got.dtype == np.float64 and np.array_equal(got, pts.astype(np.float64)), fmt
    (tmp_path / 'bad.ply').write_bytes(b"plx\n")
    with pytest.raises(ValueError):
        tdm.read_ply_points(str(tmp_path / 'bad.ply'))
    scene = tmp_path / 'fragments' / 'room'
    scene.mkdir(parents=True)
    clouds = {i: rng.uniform(0, 1, size=(20 + i, 3)).astype(np.float32) for i in (0, 2, 10)}
    for i, c in clouds.items():
        _write_ply(scene / ('cloud_bin_%d.ply' % i), c, 'binary_little_endian')
    seen = []

    def every_other(points, voxel):
        seen.append(voxel)
        return points[::2]
    ts = tdm.ThreeDMatchTestset(str(tmp_path), downsample=0.05, scene_list=['room'], subsample=every_other)
    assert len(ts) == 3 and ts.ids_list == ['room/cloud_bin_0.ply', 'room/cloud_bin_2.ply', 'room/cloud_bin_10.ply']
    assert seen == [0.05] * 3
    p, q, f, g, c, d = ts[2]
    assert p.dtype == np.float32 and np.array_equal(p, clouds[10][::2]) and p is not q or np.array_equal(p, q)
    assert f.shape == (p.shape[0], 1) and c.size == 0 and d.size == 0
    assert [len(v) for v in ts.fragments_by_scene().values()] == [3]


def test_training_items_equal_the_reference_datasets_items(tmp_path):
    """Item for item against what the REFERENCE ThreeDMatchDataset returned (tests/golden/make_golden_extra.py) for the
    same pickles and the same seeds of `random` / `numpy.random`: fragment choice, rotation axis and angle,
    translation, noise, correspondence sampling, keypoint distances, self_augment feature masking."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'dataset_items.npz'),
                allow_pickle=False)
    pts = {str(k): g['points%d' % i] for i, k in enumerate(g['ids'])}
    corr = {str(k): g['corr%d' % i] for i, k in enumerate(g['pair_keys'])}
    with open(tmp_path / '3DMatch_train_0.030_points.pkl', 'wb') as f:
        pickle.dump(pts, f)
    with open(tmp_path / '3DMatch_train_0.030_keypts.pkl', 'wb') as f:
        pickle.dump(corr, f)
    for r, (self_aug, num_node, axis, n_items) in enumerate(g['runs'].tolist()):
        ds = tdm.ThreeDMatchDataset(str(tmp_path), split='train', num_node=num_node, downsample=0.03,
                                    self_augment=bool(self_aug), augment_noise=0.005, augment_axis=axis,
                                    augment_rotation=1.0, augment_translation=0.5)
        assert len(ds) == int(g['run%d.len' % r])
        random.seed(100 + r)
        np.random.seed(100 + r)
        for j, index in enumerate(g['run%d.indices' % r].tolist()):
            item = ds[index]
            for name, v in zip(('pts0', 'pts1', 'feat0', 'feat1', 'sel_corr', 'dist_keypts'), item):
                want = g['run%d.item%d.%s' % (r, j, name)]
                assert v.shape == want.shape and v.dtype == want.dtype, (r, j, name, v.dtype, want.dtype)
                if name == 'pts1':     # the rigid transform: Open3D's float64 product vs ours, last-bit differences
                    assert np.allclose(v, want, rtol=0, atol=1e-12), (r, j, name)
                else:
                    assert np.array_equal(v, want), (r, j, name)
