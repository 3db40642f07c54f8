"""3DMatch dataset front-end -- the data format on the input side of the hot path (reference datasets/ThreeDMatch.py).

``ThreeDMatchDataset`` reads the reference's two pickles (``3DMatch_<split>_<downsample:.3f>_points.pkl``: {fragment id:
float [N,3]}, ``..._keypts.pkl``: {"src@tgt": int [M,2]}; ThreeDMatch.py:68-90) and produces the same item tuple
``(pts0, pts1, feat0, feat1, sel_corr, dist_keypts)`` (:135-149) with the same augmentation, drawing from the global
``random`` / ``numpy.random`` generators in the same order, so a seeded run sees the same stream of pairs.
Open3D is not a dependency: the rigid transform of the target fragment is the 4x4 float32 matrix applied in float64,
which is what ``PointCloud.transform`` does.

``ThreeDMatchTestset`` reads ``<root>/fragments/<scene>/cloud_bin_<i>.ply`` (:153-207) with the small PLY reader
below.  DIFFERENCE, stated rather than hidden: the reference voxel-downsamples test fragments with Open3D
(``voxel_down_sample``, grid anchored at the cloud's min bound, unordered output); here the same barycentre operator
as everywhere else in the pipeline is used (``batch_grid_subsampling_kpconv``: grid anchored at the origin), so the
retained points differ from the reference's by sub-voxel shifts.  Pass ``subsample=`` to plug in another operator.
"""
import os
import pickle
import random
from os.path import exists, join

import numpy as np

SCENES = ['7-scenes-redkitchen', 'sun3d-home_at-home_at_scan1_2013_jan_1', 'sun3d-home_md-home_md_scan9_2012_sep_30',
          'sun3d-hotel_uc-scan3', 'sun3d-hotel_umd-maryland_hotel1', 'sun3d-hotel_umd-maryland_hotel3',
          'sun3d-mit_76_studyroom-76-1studyroom2', 'sun3d-mit_lab_hj-lab_hj_tea_nov_2_2012_scan1_erika']


def rotation_matrix(augment_axis, augment_rotation):
    """Random rotation about one axis (``augment_axis == 1``) or all three (ThreeDMatch.py:14-30)."""
    a = np.random.rand(3) * 2 * np.pi * augment_rotation
    c, s = np.cos(a), np.sin(a)
    Rx = np.array([[1, 0, 0], [0, c[0], -s[0]], [0, s[0], c[0]]])
    Ry = np.array([[c[1], 0, s[1]], [0, 1, 0], [-s[1], 0, c[1]]])
    Rz = np.array([[c[2], -s[2], 0], [s[2], c[2], 0], [0, 0, 1]])
    if augment_axis == 1:
        return random.choice([Rx, Ry, Rz])
    return Rx @ Ry @ Rz


def translation_matrix(augment_translation):
    return np.random.rand(3) * augment_translation


def pairwise_distance(a):
    """[M,M] float64 Euclidean distances (scipy.spatial.distance.cdist(a, a) of the reference, :135)."""
    try:
        from scipy.spatial.distance import cdist
        return cdist(a, a)
    except ImportError:  # same arithmetic, written out
        a = np.asarray(a, dtype=np.float64)
        d = a[:, None, :] - a[None, :, :]
        return np.sqrt((d * d).sum(axis=-1))


class ThreeDMatchDataset(object):
    __type__ = 'descriptor'
    MAX_POINTS = 50000   # larger fragments are skipped (ThreeDMatch.py:117-118)

    def __init__(self, root, split='train', num_node=16, downsample=0.03, self_augment=False, augment_noise=0.005,
                 augment_axis=1, augment_rotation=1.0, augment_translation=0.001, config=None):
        self.root, self.split, self.num_node, self.downsample = root, split, num_node, downsample
        self.self_augment, self.augment_noise, self.augment_axis = self_augment, augment_noise, augment_axis
        self.augment_rotation, self.augment_translation, self.config = augment_rotation, augment_translation, config
        self.points, self.ids_list, self.correspondences, self.src_to_tgt = [], [], {}, {}
        pts_filename = join(root, '3DMatch_%s_%.3f_points.pkl' % (split, downsample))
        keypts_filename = join(root, '3DMatch_%s_%.3f_keypts.pkl' % (split, downsample))
        if not (exists(pts_filename) and exists(keypts_filename)):
            raise FileNotFoundError("3DMatch pickles not found: %s, %s" % (pts_filename, keypts_filename))
        with open(pts_filename, 'rb') as f:
            data = pickle.load(f)
        self.points, self.ids_list = list(data.values()), list(data.keys())
        self._index = {k: i for i, k in enumerate(self.ids_list)}
        with open(keypts_filename, 'rb') as f:
            self.correspondences = pickle.load(f)
        for idpair in self.correspondences.keys():
            src, tgt = idpair.split("@")[0], idpair.split("@")[1]
            self.src_to_tgt.setdefault(src, []).append(tgt)
        self._sources = list(self.src_to_tgt.keys())

    def __len__(self):
        return len(self._sources)

    def __getitem__(self, index):
        while True:
            src_id = self._sources[index]
            targets = self.src_to_tgt[src_id]
            tgt_id = targets[0] if random.random() > 0.5 else random.choice(targets)
            src_ind, tgt_ind = self._index[src_id], self._index[tgt_id]
            if self.self_augment:
                tgt_ind = src_ind
                n = self.points[src_ind].shape[0]
                corr = np.array([np.arange(n), np.arange(n)]).T
            else:
                corr = self.correspondences["%s@%s" % (src_id, tgt_id)]
            if self.points[src_ind].shape[0] <= self.MAX_POINTS and self.points[tgt_ind].shape[0] <= self.MAX_POINTS:
                break
            index = int(np.random.choice(len(self), 1)[0])
        gt_trans = np.eye(4).astype(np.float32)
        gt_trans[0:3, 0:3] = rotation_matrix(self.augment_axis, self.augment_rotation)
        gt_trans[0:3, 3] = translation_matrix(self.augment_translation)
        T = gt_trans.astype(np.float64)
        src_points = np.array(self.points[src_ind], dtype=np.float64)
        tgt_points = np.asarray(self.points[tgt_ind], dtype=np.float64) @ T[:3, :3].T + T[:3, 3]
        src_points += np.random.rand(src_points.shape[0], 3) * self.augment_noise
        tgt_points += np.random.rand(tgt_points.shape[0], 3) * self.augment_noise
        if len(corr) > self.num_node:
            sel_corr = corr[np.random.choice(len(corr), self.num_node, replace=False)]
        else:
            sel_corr = corr
        dist_keypts = pairwise_distance(src_points[sel_corr[:, 0], :].astype(np.float32))
        feat0 = np.ones_like(src_points[:, :1]).astype(np.float32)
        feat1 = np.ones_like(tgt_points[:, :1]).astype(np.float32)
        if self.self_augment:
            feat0[np.random.choice(src_points.shape[0], int(src_points.shape[0] * 0.99), replace=False)] = 0
            feat1[np.random.choice(tgt_points.shape[0], int(tgt_points.shape[0] * 0.99), replace=False)] = 0
        return src_points, tgt_points, feat0, feat1, sel_corr, dist_keypts


# ------------------------------------------------------------------------------------------------------ PLY input
_PLY_TYPES = {'char': 'i1', 'int8': 'i1', 'uchar': 'u1', 'uint8': 'u1', 'short': 'i2', 'int16': 'i2', 'ushort': 'u2',
              'uint16': 'u2', 'int': 'i4', 'int32': 'i4', 'uint': 'u4', 'uint32': 'u4', 'float': 'f4', 'float32': 'f4',
              'double': 'f8', 'float64': 'f8'}


def read_ply_points(filename):
    """[N,3] float64 vertex positions of an ascii / binary PLY file (only the ``vertex`` element is read, which must
    come first -- true for the 3DMatch fragments; list properties inside it are not supported)."""
    with open(filename, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise ValueError("%s: not a PLY file" % filename)
        fmt, n_vertex, props, element = None, None, [], None
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: truncated PLY header" % filename)
            tok = line.decode('ascii', 'replace').split()
            if not tok or tok[0] in ('comment', 'obj_info'):
                continue
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                element = tok[1]
                if element == 'vertex':
                    n_vertex = int(tok[2])
            elif tok[0] == 'property' and element == 'vertex':
                if tok[1] == 'list':
                    raise ValueError("%s: list property in the vertex element" % filename)
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == 'end_header':
                break
        if n_vertex is None or fmt is None or not {'x', 'y', 'z'} <= {p for p, _ in props}:
            raise ValueError("%s: no vertex element with x, y, z" % filename)
        if fmt == 'ascii':
            rows = np.loadtxt(f, max_rows=n_vertex, ndmin=2) if n_vertex else np.zeros((0, len(props)))
            names = [p for p, _ in props]
            return np.stack([rows[:, names.index(a)] for a in 'xyz'], axis=1).astype(np.float64)
        order = {'binary_little_endian': '<', 'binary_big_endian': '>'}[fmt]
        dtype = np.dtype([(p, order + t) for p, t in props])
        data = np.frombuffer(f.read(n_vertex * dtype.itemsize), dtype=dtype, count=n_vertex)
        return np.stack([data['x'], data['y'], data['z']], axis=1).astype(np.float64)


def _device_subsample(points, voxel):
    import torch
    from .dataloader import batch_grid_subsampling_kpconv
    p = torch.as_tensor(np.ascontiguousarray(points, dtype=np.float32)).cuda()
    lens = torch.tensor([p.shape[0]], dtype=torch.int32, device=p.device)
    out, _ = batch_grid_subsampling_kpconv(p, lens, sampleDl=voxel)
    return out.cpu().numpy()


class ThreeDMatchTestset(object):
    __type__ = 'descriptor'

    def __init__(self, root, downsample=0.03, config=None, last_scene=False, subsample=None, scene_list=None):
        self.root, self.downsample, self.config = root, downsample, config
        self.points, self.ids_list, self.num_test = [], [], 0
        self.scene_list = list(scene_list) if scene_list is not None else list(SCENES)
        if last_scene:
            self.scene_list = self.scene_list[-1:]
        subsample = subsample if subsample is not None else _device_subsample
        for scene in self.scene_list:
            path = '%s/fragments/%s' % (root, scene)
            files = sorted((f for f in os.listdir(path) if f.endswith('ply')),
                           key=lambda x: int(x[:-4].split("_")[-1]))
            self.num_test += len(files)
            for name in files:
                self.points.append(np.asarray(subsample(read_ply_points(join(path, name)), downsample)))
                self.ids_list.append(scene + '/' + name)

    def fragments_by_scene(self):
        """{scene: [points, ...]} in file order -- the input of geometric_registration.evaluate.generate_features."""
        out = {s: [] for s in self.scene_list}
        for ident, pts in zip(self.ids_list, self.points):
            out[ident.split('/')[0]].append(pts)
        return out

    def __getitem__(self, index):
        pts = self.points[index].astype(np.float32)
        feat = np.ones_like(pts[:, :1]).astype(np.float32)
        return pts, pts, feat, feat, np.array([]), np.array([])

    def __len__(self):
        return self.num_test
