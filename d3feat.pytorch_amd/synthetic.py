"""Deterministic 3DMatch-shaped synthetic fragment pairs.

There is no dataset in this environment, so benchmarks and parity tests use an indoor-like scene: five noisy planar
patches (floor, two walls, table top, cabinet face) sampled with `n_raw` points, voxel-subsampled at 0.03 m with the
SAME barycentre operator the pipeline uses (the caller supplies it: the HIP op on the GPU, the CPU oracle in tests)
-> ~19k points per fragment.  A pair = two samplings of the scene; fragment B is rotated 0.7 rad about z and
translated.  The tuple returned matches the reference dataset contract (datasets/ThreeDMatch.py:135-149):
``(pts0, pts1, feat0, feat1, sel_corr, dist_keypts)``.
"""
import numpy as np

# (origin, edge u, edge v, share of the points)
_PATCHES = [
    ((0.0, 0.0, 0.0), (3.0, 0.0, 0.0), (0.0, 2.5, 0.0), 0.30),   # floor 3.0 x 2.5
    ((0.0, 0.0, 0.0), (3.0, 0.0, 0.0), (0.0, 0.0, 2.5), 0.30),   # wall 3.0 x 2.5
    ((0.0, 0.0, 0.0), (0.0, 2.5, 0.0), (0.0, 0.0, 2.5), 0.20),   # wall 2.5 x 2.5
    ((1.0, 0.9, 0.7), (1.0, 0.0, 0.0), (0.0, 0.8, 0.0), 0.10),   # table top 1.0 x 0.8 at z = 0.7
    ((2.3, 2.4, 0.0), (0.5, 0.0, 0.0), (0.0, 0.0, 1.2), 0.10),   # cabinet face 0.5 x 1.2
]
SCENE_SCALE = 0.62
NOISE_SIGMA = 0.004
PAIR_ROT_Z = 0.7
PAIR_TRANSLATION = (0.3, 0.1, 0.2)


def raw_fragment(seed, n_raw=300000, scale=SCENE_SCALE):
    rng = np.random.default_rng(seed)
    chunks = []
    for org, u, v, share in _PATCHES:
        n = int(round(n_raw * share))
        ab = rng.random((n, 2))
        pts = np.asarray(org)[None, :] + ab[:, :1] * np.asarray(u)[None, :] + ab[:, 1:] * np.asarray(v)[None, :]
        chunks.append(pts)
    pts = np.concatenate(chunks, axis=0) * scale
    pts = pts + rng.normal(scale=NOISE_SIGMA, size=pts.shape)
    return pts.astype(np.float32)


def make_fragment(seed, subsample, voxel=0.03, n_raw=300000, scale=SCENE_SCALE):
    """`subsample(points[N,3] f32, lengths[1] i32, dl) -> (points, lengths)` is the voxel barycentre operator."""
    raw = raw_fragment(seed, n_raw=n_raw, scale=scale)
    pts, _ = subsample(raw, np.array([raw.shape[0]], dtype=np.int32), voxel)
    return np.ascontiguousarray(np.asarray(pts, dtype=np.float32))


def make_pair(seed_a, seed_b, subsample, voxel=0.03, n_raw=300000, scale=SCENE_SCALE, num_node=128,
              corr_radius=0.0375):
    from scipy.spatial import cKDTree
    from scipy.spatial.distance import cdist
    a = make_fragment(seed_a, subsample, voxel, n_raw, scale)
    b = make_fragment(seed_b, subsample, voxel, n_raw, scale)
    # correspondences before moving B: nearest neighbor within corr_radius
    dist, nn = cKDTree(b).query(a, k=1, distance_upper_bound=corr_radius)
    ok = np.nonzero(np.isfinite(dist))[0]
    corr = np.stack([ok, nn[ok]], axis=1)
    rs = np.random.RandomState(0)
    n_sel = min(num_node, corr.shape[0])
    sel = rs.choice(corr.shape[0], n_sel, replace=False)
    sel_corr = corr[sel].astype(np.int64)
    dist_keypts = cdist(a[sel_corr[:, 0]].astype(np.float64), a[sel_corr[:, 0]].astype(np.float64))
    c, s = np.cos(PAIR_ROT_Z), np.sin(PAIR_ROT_Z)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
    b_moved = (b @ R.T + np.asarray(PAIR_TRANSLATION, dtype=np.float32)).astype(np.float32)
    feat0 = np.ones((a.shape[0], 1), dtype=np.float32)
    feat1 = np.ones((b_moved.shape[0], 1), dtype=np.float32)
    return a, b_moved, feat0, feat1, sel_corr, dist_keypts


class SyntheticPairs:
    """Indexable stream of pairs: item i uses seeds (base + 2i + 1, base + 2i + 2)."""

    def __init__(self, config, subsample, length=8, base_seed=0, **kw):
        self.config, self.subsample, self.length, self.base_seed, self.kw = config, subsample, length, base_seed, kw
        self._cache = {}

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        if i not in self._cache:
            self._cache[i] = make_pair(self.base_seed + 2 * i + 1, self.base_seed + 2 * i + 2, self.subsample,
                                       voxel=self.config.first_subsampling_dl, num_node=self.config.num_node,
                                       **self.kw)
        return self._cache[i]
