#!/usr/bin/env python
"""Generate golden vectors from the REAL reference (run in the build container only; /root/reference never travels).

    python tests/golden/make_golden.py

* Python side: imports the reference's own ``models.architectures.KPFCNN``, ``models.blocks``, ``utils.loss``,
  ``datasets.dataloader.collate_fn_descriptor / calibrate_neighbors`` and ``geometric_registration.common`` unmodified
  (cwd = /root/reference because load_kernels uses a cwd-relative path, kernels/kernel_points.py:403; ``open3d`` is
  stubbed -- it is only used for file I/O).
* Native side: ``cpp_wrappers.cpp_neighbors.radius_neighbors`` and ``cpp_wrappers.cpp_subsampling.grid_subsampling``
  are provided by thin module objects that call ``oracle/_ref/libd3f_ref.so``, i.e. the reference's own
  neighbors.cpp / grid_subsampling.cpp / cloud.cpp / nanoflann.hpp compiled in place (oracle/Makefile); only the
  CPython marshalling layer (which does not build against NumPy 2) is bypassed.

Outputs (committed, data only):
  tests/golden/s0_small.npz   mini pair (~2k pts/fragment): FULL batch dict, state_dict of a first_features_dim=16
                              network, per-KPConv inputs/outputs, features/scores (train+eval), losses, gradients.
  tests/golden/s1_full.npz    the 19.3k-point pair at full width (first_features_dim=128): kernel points, parameter
                              checksums (weights are re-created from the seed), sampled rows + SHA-256 of the big tensors.
"""
import hashlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import native  # noqa: E402

native.build(ref=True)


def _install_reference():
    sys.modules['open3d'] = types.ModuleType('open3d')
    m_sub = types.ModuleType('cpp_wrappers.cpp_subsampling.grid_subsampling')

    def subsample_batch(points, batches, sampleDl=0.1, max_p=0, verbose=0, **kw):
        return native.ref_subsample_batch(np.asarray(points), np.asarray(batches), sampleDl=sampleDl, max_p=max_p)
    m_sub.subsample_batch = subsample_batch
    m_nei = types.ModuleType('cpp_wrappers.cpp_neighbors.radius_neighbors')

    def batch_query(queries, supports, q_batches, s_batches, radius=0.1):
        return native.ref_batch_query(np.asarray(queries), np.asarray(supports), np.asarray(q_batches),
                                      np.asarray(s_batches), radius=radius)
    m_nei.batch_query = batch_query
    for name in ['cpp_wrappers', 'cpp_wrappers.cpp_subsampling', 'cpp_wrappers.cpp_neighbors']:
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
    sys.modules['cpp_wrappers.cpp_subsampling.grid_subsampling'] = m_sub
    sys.modules['cpp_wrappers.cpp_neighbors.radius_neighbors'] = m_nei
    sys.modules['cpp_wrappers.cpp_subsampling'].grid_subsampling = m_sub
    sys.modules['cpp_wrappers.cpp_neighbors'].radius_neighbors = m_nei
    sys.path.insert(0, REF)
    os.chdir(REF)


_install_reference()

import torch  # noqa: E402
import importlib.util  # noqa: E402


def _load_pkg_module(rel, name):
    """Load one file of OUR package (for the synthetic generator / default config) without importing the HIP parts."""
    spec = importlib.util.spec_from_file_location(name, os.path.join(REPO, "d3feat.pytorch_amd", rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


synthetic = _load_pkg_module("synthetic.py", "_d3f_synthetic")
cfgmod = _load_pkg_module("config.py", "_d3f_config")

from datasets.dataloader import collate_fn_descriptor, calibrate_neighbors  # noqa: E402  (reference)
from models.architectures import KPFCNN  # noqa: E402  (reference)
from models.blocks import KPConv  # noqa: E402  (reference)
from utils.loss import CircleLoss, DetLoss  # noqa: E402  (reference)
from geometric_registration.common import build_correspondence  # noqa: E402  (reference)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_subsample(points, lengths, dl):
    return native.ref_subsample_batch(points, lengths, sampleDl=dl)


class OnePair:
    def __init__(self, item, config):
        self.item, self.config = item, config

    def __len__(self):
        return 1

    def __getitem__(self, i):
        return self.item


def run_reference(item, config, limits, seed, capture_blocks):
    """collate + KPFCNN fwd/bwd + losses with the reference code; returns a dict of numpy arrays."""
    out = {}
    batch = collate_fn_descriptor([item], config, limits)
    for key in ('points', 'neighbors', 'pools', 'upsamples', 'stack_lengths'):
        for l, t in enumerate(batch[key]):
            out['batch.%s.%d' % (key, l)] = t.numpy()
    np.random.seed(seed)
    torch.manual_seed(seed)
    model = KPFCNN(config)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    captured = {}

    def hook(name):
        def fn(mod, inp, outp):
            captured[name] = ([t.detach().clone() for t in inp], outp.detach().clone())
        return fn
    names = {}
    for n, m in model.named_modules():
        if isinstance(m, KPConv):
            names[n] = m
    for n, m in names.items():
        if any(n.startswith('encoder_blocks.%d.' % b) for b in capture_blocks):
            m.register_forward_hook(hook(n))
    model.train()
    feats, scores = model(batch)
    corr = batch['corr'].long()
    n0 = int(batch['stack_lengths'][0][0])
    anc_f, pos_f = feats[corr[:, 0]], feats[corr[:, 1] + n0]
    anc_s, pos_s = scores[corr[:, 0]], scores[corr[:, 1] + n0]
    circle = CircleLoss(dist_type='euclidean', log_scale=config.log_scale, safe_radius=config.safe_radius,
                        pos_margin=config.pos_margin, neg_margin=config.neg_margin)
    det = DetLoss('euclidean')
    desc_loss, acc, fp, an, _, dists = circle(anc_f, pos_f, batch['dist_keypts'])
    det_loss = det(dists, anc_s, pos_s)
    loss = desc_loss + det_loss
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    model.eval()
    with torch.no_grad():
        feats_e, scores_e = model(batch)
    out.update({'features_train': feats.detach().numpy(), 'scores_train': scores.detach().numpy(),
                'features_eval': feats_e.numpy(), 'scores_eval': scores_e.numpy(),
                'desc_loss': np.float32(desc_loss.item()), 'det_loss': np.float32(det_loss.item()),
                'accuracy': np.float32(float(acc)), 'furthest_positive': np.asarray(fp, np.float32),
                'average_negative': np.asarray(an, np.float32), 'dists': dists.detach().numpy(),
                'corr': batch['corr'].numpy(), 'dist_keypts': batch['dist_keypts'].numpy()})
    return out, sd, grads, captured, batch


def main():
    # ------------------------------------------------------------------ S0: mini pair, reduced width, everything
    cfg0 = cfgmod.default_config(first_features_dim=16)
    item0 = synthetic.make_pair(11, 12, ref_subsample, n_raw=40000, scale=0.2, num_node=64)
    print('S0 fragments:', item0[0].shape, item0[1].shape, 'corr', item0[4].shape)
    limits0 = calibrate_neighbors(OnePair(item0, cfg0), cfg0, collate_fn=collate_fn_descriptor, samples_threshold=10**9)
    print('S0 limits', limits0)
    res, sd, grads, cap, batch = run_reference(item0, cfg0, limits0, seed=0, capture_blocks=range(14))
    g = {'pts0': item0[0], 'pts1': item0[1], 'sel_corr': item0[4], 'dist_keypts_in': item0[5],
         'limits': np.asarray(limits0, np.int64)}
    g.update(res)
    for k, v in sd.items():
        g['sd.' + k] = v.numpy()
    for k, v in grads.items():
        g['grad.' + k] = v.numpy()
    for n, (inp, outp) in cap.items():
        q, s, idx, x = inp
        g['kpconv.%s.x' % n] = x.numpy()
        g['kpconv.%s.out' % n] = outp.numpy()
    # uncapped searches of the first two levels (tie statistics + bit-exactness of the search itself)
    p0, l0 = res['batch.points.0'], res['batch.stack_lengths.0']
    p1, l1 = res['batch.points.1'], res['batch.stack_lengths.1']
    r0 = cfg0.first_subsampling_dl * cfg0.conv_radius
    g['uncapped.conv0'] = native.ref_batch_query(p0, p0, l0, l0, radius=r0)
    g['uncapped.pool0'] = native.ref_batch_query(p1, p0, l1, l0, radius=r0)
    g['uncapped.up0'] = native.ref_batch_query(p0, p1, l0, l1, radius=2 * r0)
    # dense matching on the eval descriptors: top-250 by score per fragment (test.py:56-57)
    n0 = int(l0[0])
    fe, se = res['features_eval'], res['scores_eval'].reshape(-1)
    si = np.argsort(se[:n0])[-250:]
    ti = np.argsort(se[n0:])[-250:]
    g['match.src_idx'], g['match.tgt_idx'] = si, ti
    g['match.corr250'] = build_correspondence(fe[:n0][si], fe[n0:][ti])
    g['match.corr_all'] = build_correspondence(fe[:n0], fe[n0:])
    np.savez_compressed(os.path.join(HERE, 's0_small.npz'), **g)
    print('wrote s0_small.npz', os.path.getsize(os.path.join(HERE, 's0_small.npz')) / 1e6, 'MB')

    # ------------------------------------------------------------------ S1: the benchmark pair, full width, sampled
    cfg1 = cfgmod.default_config()
    item1 = synthetic.make_pair(1, 2, ref_subsample)
    print('S1 fragments:', item1[0].shape, item1[1].shape, 'corr', item1[4].shape)
    limits1 = calibrate_neighbors(OnePair(item1, cfg1), cfg1, collate_fn=collate_fn_descriptor, samples_threshold=10**9)
    print('S1 limits', limits1)
    res, sd, grads, cap, batch = run_reference(item1, cfg1, limits1, seed=0, capture_blocks=[0, 1, 2, 3, 12])
    rs = np.random.RandomState(123)
    g = {'limits': np.asarray(limits1, np.int64), 'n_frag': np.asarray([len(item1[0]), len(item1[1])]),
         'pts0.sha': sha(item1[0]), 'pts1.sha': sha(item1[1]), 'sel_corr': item1[4], 'dist_keypts_in': item1[5]}
    for key, v in res.items():
        if key.startswith('batch.'):
            g[key + '.sha'] = sha(v)
            g[key + '.shape'] = np.asarray(v.shape)
            if v.ndim == 2 and v.shape[0] > 0:
                rows = np.sort(rs.choice(v.shape[0], min(256, v.shape[0]), replace=False))
                g[key + '.rows'] = rows
                g[key + '.sample'] = v[rows]
            elif v.ndim == 1:
                g[key] = v
        elif key in ('features_train', 'scores_train', 'features_eval', 'scores_eval'):
            rows = np.sort(rs.choice(v.shape[0], 2048, replace=False))
            g[key + '.rows'] = rows
            g[key + '.sample'] = v[rows]
        else:
            g[key] = v
    for k, v in sd.items():
        if k.endswith('kernel_points'):
            g['sd.' + k] = v.numpy()
        g['sdsum.' + k] = np.asarray([float(v.double().sum()), float(v.double().abs().sum())])
    for k, v in grads.items():
        g['gradnorm.' + k] = np.float64(v.double().norm().item())
    for k in ['encoder_blocks.0.KPConv.weights', 'encoder_blocks.1.KPConv.weights',
              'decoder_blocks.7.mlp.weight', 'encoder_blocks.12.KPConv.weights']:
        v = grads[k].numpy()
        g['grad.' + k] = v if v.size <= 40000 else v.reshape(-1)[:40000]
    for n, (inp, outp) in cap.items():
        rows = np.sort(rs.choice(outp.shape[0], min(512, outp.shape[0]), replace=False))
        g['kpconv.%s.rows' % n] = rows
        g['kpconv.%s.out' % n] = outp.numpy()[rows]
        g['kpconv.%s.xsum' % n] = np.float64(inp[3].double().sum().item())
    np.savez_compressed(os.path.join(HERE, 's1_full.npz'), **g)
    print('wrote s1_full.npz', os.path.getsize(os.path.join(HERE, 's1_full.npz')) / 1e6, 'MB')


if __name__ == '__main__':
    main()
