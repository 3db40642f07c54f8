"""3DMatch registration protocol around the matching kernel -- the reference's ``test.py`` (:20-128) re-cast for the GPU.

Same on-disk contract as the reference, so either side can read the other's dumps:

* ``generate_features``   (test.py:79-128): per fragment ``descriptors/<scene>/cloud_bin_<i>.D3Feat.npy`` [N,32] f32,
  ``keypoints/<scene>/cloud_bin_<i>.npy`` [N,3] f32, ``scores/<scene>/cloud_bin_<i>.npy`` [N,1] f32, eval mode
  (scores gated by the local-maximum mask).  The reference pushes a fragment through the network stacked with itself
  (datasets/ThreeDMatch.py:203 returns ``(pts, pts, ...)``) and keeps the first half (test.py:118-120); one copy gives
  the same rows (see ``describe_fragment``), so that is what runs.
* ``loadlog`` / ``writelog`` (geometric_registration/common.py:44-58): the ``gt.log`` trajectory format -- a
  ``id1 \\t id2 \\t n`` line followed by the 4x4 transform, tab separated.
* ``register_one_scene``  (test.py:20-76): for every fragment pair listed in ``gt.log``: top-k keypoints by score (or a
  random draw), mutual nearest neighbors in descriptor space, target keypoints moved by the ground-truth transform,
  inliers = matches closer than ``distance_threshold``; a pair is registered when its inlier ratio exceeds
  ``inlier_ratio_threshold``.  Returns ``(recall %, mean inlier count, mean inlier ratio)``.

Matching and inlier counting stay on the device (``ops.mutual_nn`` = the MFMA N x N kernel); per scene ONE read-back
of the per-pair counters.  The reference loops over pairs in NumPy/Open3D and one process per scene.
"""
import os

import numpy as np
import torch

from .. import ops
from ..datasets import dataloader as dl
from .common import select_keypoints

DESC_NAME = 'D3Feat'


# ------------------------------------------------------------------------------------------------- file formats
def loadlog(gtpath):
    """{'i_j': 4x4 float64} from ``<gtpath>/gt.log`` (common.py:44-58)."""
    with open(os.path.join(gtpath, 'gt.log')) as f:
        rows = [ln.rstrip('\n') for ln in f if ln.strip()]
    if len(rows) % 5:
        raise ValueError("gt.log: %d non-empty lines, expected blocks of 5" % len(rows))
    out = {}
    for i in range(0, len(rows), 5):
        head = rows[i].split('\t')[0:3]
        mat = np.array([[float(x) for x in rows[i + r].split('\t')[0:4]] for r in range(1, 5)], dtype=np.float64)
        out['%d_%d' % (int(head[0]), int(head[1]))] = mat
    return out


def writelog(gtpath, transforms, num_frag):
    """Inverse of :func:`loadlog` (used for synthetic scenes; the benchmark ships its own files)."""
    os.makedirs(gtpath, exist_ok=True)
    with open(os.path.join(gtpath, 'gt.log'), 'w') as f:
        for key in sorted(transforms, key=lambda k: tuple(int(x) for x in k.split('_'))):
            a, b = key.split('_')
            f.write('%d\t %d\t %d\t\n' % (int(a), int(b), num_frag))
            for row in np.asarray(transforms[key], dtype=np.float64):
                f.write(''.join(' % .8e\t ' % v for v in row).rstrip(' ') + '\n')


def _paths(save_path, scene):
    return (os.path.join(save_path, 'descriptors', scene), os.path.join(save_path, 'keypoints', scene),
            os.path.join(save_path, 'scores', scene))


def get_keypts(keyptspath, filename):
    return np.load(os.path.join(keyptspath, filename + '.npy'))


def get_desc(descpath, filename, desc_name=DESC_NAME):
    return np.load(os.path.join(descpath, filename + '.%s.npy' % desc_name))


def get_scores(scorepath, filename, desc_name=DESC_NAME):
    return np.load(os.path.join(scorepath, filename + '.npy'))


# ------------------------------------------------------------------------------------------ descriptor generation
@torch.no_grad()
def describe_fragment(model, points, config, neighborhood_limits, device=None, stacked=False, exact_width=True):
    """(keypoints [N,3], descriptors [N,32], scores [N,1]) of one fragment, device tensors (test.py:107-120).

    The reference feeds the fragment stacked with itself and keeps the first half.  Every operator of the network
    works per cloud (neighbor tables never cross clouds) and the detector's global maximum over two copies equals the
    maximum over one, so a single copy gives the same rows for half the work; ``stacked=True`` runs the literal
    two-copy batch (used by the test that checks the equivalence).  ``exact_width=False`` keeps every neighbor table at
    the full calibrated width instead of the reference's min(limit, max_count) -- what the static-shape graph engine
    (``infer.InferStep``) does; see the note on table widths in DESIGN.md section 3."""
    dev = torch.device(device) if device is not None else next(model.parameters()).device
    pts = torch.as_tensor(np.ascontiguousarray(points) if isinstance(points, np.ndarray) else points,
                          dtype=torch.float32, device=dev)
    n = int(pts.shape[0])
    feat = torch.ones((n, 1), dtype=torch.float32, device=dev)
    if stacked:
        empty = torch.zeros((0, 2), dtype=torch.int64, device=dev)
        batch = dl.collate_fn_descriptor([(pts, pts, feat, feat, empty, torch.zeros((0, 0), device=dev))], config,
                                         neighborhood_limits, device=dev)
    else:
        lengths = torch.tensor([n], dtype=torch.int32, device=dev)
        batch = dl.build_pyramid(pts, lengths, config, neighborhood_limits, exact_width=exact_width)
        batch.pop('_status')
        batch['features'] = feat
    was_training = model.training
    model.eval()
    try:
        features, scores = model(batch)
    finally:
        model.train(was_training)
    return batch['points'][0][:n], features[:n], scores[:n]


def generate_features(model, scenes, save_path, config, neighborhood_limits, device=None, verbose=False, engine=None):
    """``scenes``: {scene name: sequence of fragment point arrays [N,3]} (already voxel-subsampled at
    ``config.downsample`` like ThreeDMatchTestset does).  Writes the three .npy files per fragment.

    ``engine``: an ``infer.InferStep(model, ..., clouds=1)`` with graphs enabled -- fragments that fit its capacities
    go through the pipelined graph replay (the next fragment's pyramid is built under the current one's network),
    the others through the eager ``describe_fragment``."""
    for scene, fragments in scenes.items():
        dpath, kpath, spath = _paths(save_path, scene)
        for p in (dpath, kpath, spath):
            os.makedirs(p, exist_ok=True)
        frags = [np.ascontiguousarray(f, dtype=np.float32) for f in fragments]
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        on_dev = [torch.from_numpy(f).to(dev) for f in frags] if engine is not None else None
        for ids, points in enumerate(frags):
            if engine is not None and engine.fits((on_dev[ids],)):
                nxt = (on_dev[ids + 1],) if ids + 1 < len(frags) and engine.fits((on_dev[ids + 1],)) else None
                features, scores = engine.describe((on_dev[ids],), nxt)
                pts = on_dev[ids]
            else:
                pts, features, scores = describe_fragment(model, points, config, neighborhood_limits, device)
            np.save(os.path.join(dpath, 'cloud_bin_%d.%s' % (ids, DESC_NAME)), features.cpu().numpy().astype(np.float32))
            np.save(os.path.join(kpath, 'cloud_bin_%d' % ids), pts.cpu().numpy().astype(np.float32))
            np.save(os.path.join(spath, 'cloud_bin_%d' % ids), scores.cpu().numpy().astype(np.float32))
            if verbose:
                print("Generate cloud_bin_%d for %s" % (ids, scene))
        if engine is not None:
            engine.check_status()


# ----------------------------------------------------------------------------------------------------- registration
def match_pair(source_keypts, source_desc, source_score, target_keypts, target_desc, target_score, gt_trans,
               num_points=250, distance_threshold=0.10, random_points=False, rng=None):
    """Device tensors in; returns device scalars ``(num_inliers, num_matches)`` of one fragment pair
    (test.py:47-70).  ``gt_trans`` maps the target fragment into the source frame."""
    if random_points:
        rng = rng if rng is not None else np.random
        si = torch.as_tensor(rng.choice(source_keypts.shape[0], num_points), device=source_desc.device)
        ti = torch.as_tensor(rng.choice(target_keypts.shape[0], num_points), device=source_desc.device)
    else:
        si = select_keypoints(source_score.reshape(-1), num_points)
        ti = select_keypoints(target_score.reshape(-1), num_points)
    sd = torch.nan_to_num(source_desc[si]).contiguous()
    td = torch.nan_to_num(target_desc[ti]).contiguous()
    row, _, mutual = ops.mutual_nn(sd, td)
    T = torch.as_tensor(gt_trans, dtype=torch.float64, device=sd.device)
    tgt = target_keypts[ti][row.long()].double() @ T[:3, :3].T + T[:3, 3]     # every source row's match, moved
    dist = torch.sqrt(((source_keypts[si].double() - tgt) ** 2).sum(dim=1))
    keep = mutual.bool()
    return ((dist < distance_threshold) & keep).sum(), keep.sum()


def register_one_scene(inlier_ratio_threshold, distance_threshold, save_path, scene, gtpath, num_frag=None,
                       num_points=250, random_points=False, device='cuda', seed=0):
    """Recall / mean inlier count / mean inlier ratio of one scene from the dumped files (test.py:20-76)."""
    gt = loadlog(gtpath)
    dpath, kpath, spath = _paths(save_path, scene)
    if num_frag is None:
        num_frag = len([f for f in os.listdir(kpath) if f.endswith('.npy')])
    dev = torch.device(device)
    cache = {}

    def load(i):
        if i not in cache:
            name = 'cloud_bin_%d' % i
            cache[i] = tuple(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev) for a in (
                get_keypts(kpath, name), get_desc(dpath, name), get_scores(spath, name).reshape(-1)))
        return cache[i]
    rng = np.random.RandomState(seed)
    counts = []
    for id1 in range(num_frag):
        for id2 in range(id1 + 1, num_frag):
            key = '%d_%d' % (id1, id2)
            if key not in gt:   # pairs with less than 30% overlap are not part of the benchmark (test.py:38-42)
                continue
            sk, sd, ss = load(id1)
            tk, td, ts = load(id2)
            counts.append(torch.stack(match_pair(sk, sd, ss, tk, td, ts, gt[key], num_points, distance_threshold,
                                                 random_points, rng)))
    if not counts:
        raise ValueError("no fragment pair of %s is listed in %s" % (scene, os.path.join(gtpath, 'gt.log')))
    c = torch.stack(counts).cpu().numpy().astype(np.float64)      # the scene's only read-back
    inliers, matches = c[:, 0], c[:, 1]
    ratio = np.divide(inliers, matches, out=np.zeros_like(inliers), where=matches > 0)
    recall = float((ratio > inlier_ratio_threshold).sum()) * 100.0 / len(ratio)
    return recall, float(inliers.mean()), float(ratio.mean())


def evaluate_scenes(save_path, scenes_gt, inlier_ratio_threshold=0.05, distance_threshold=0.10, num_points=250,
                    random_points=False, device='cuda'):
    """{scene: gt directory} -> ({scene: [recall, inlier num, inlier ratio]}, averages) -- test.py:169-188."""
    out = {}
    for scene, gtpath in scenes_gt.items():
        out[scene] = list(register_one_scene(inlier_ratio_threshold, distance_threshold, save_path, scene, gtpath,
                                             num_points=num_points, random_points=random_points, device=device))
    avg = np.mean(np.array(list(out.values()), dtype=np.float64), axis=0)
    return out, {'recall': float(avg[0]), 'inlier_num': float(avg[1]), 'inlier_ratio': float(avg[2])}
