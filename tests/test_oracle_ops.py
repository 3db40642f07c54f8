"""CPU: the PyTorch-fp32 restatement (oracle/ops_ref.py) against golden vectors produced by the real reference
(models/blocks.py, models/architectures.py, utils/loss.py, geometric_registration/common.py)."""
import os

import numpy as np
import pytest
import torch

from d3feat_pytorch_amd import config as cfgmod
from oracle import ops_ref
from util import rel_err

TOL = 2e-5  # oracle and reference are both fp32 on CPU; only summation order differs


def _batch(g, levels=5):
    t = torch.from_numpy
    return {'points': [t(g['batch.points.%d' % l]) for l in range(levels)],
            'neighbors': [t(g['batch.neighbors.%d' % l]) for l in range(levels)],
            'pools': [t(g['batch.pools.%d' % l]) for l in range(levels)],
            'upsamples': [t(g['batch.upsamples.%d' % l]) for l in range(levels)],
            'stack_lengths': [t(g['batch.stack_lengths.%d' % l]) for l in range(levels)],
            'features': torch.ones((g['batch.points.0'].shape[0], 1), dtype=torch.float32)}


def _sd(g):
    return {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd.')}


def test_kpconv_per_block(golden_s0):
    g = golden_s0
    cfg = cfgmod.default_config(first_features_dim=16)
    batch, sd = _batch(g), _sd(g)
    layer_of = [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4]
    strided = {2, 5, 8, 11}
    for b in range(14):
        name = 'encoder_blocks.%d.KPConv' % b
        l = layer_of[b]
        r = 0.075 * 2 ** l
        if b in strided:
            q, s, idx = batch['points'][l + 1], batch['points'][l], batch['pools'][l]
        else:
            q, s, idx = batch['points'][l], batch['points'][l], batch['neighbors'][l]
        out = ops_ref.kpconv(q, s, idx, torch.from_numpy(g['kpconv.%s.x' % name]), sd[name + '.kernel_points'],
                             sd[name + '.weights'], r * cfg.KP_extent / cfg.conv_radius)
        assert rel_err(out.numpy(), g['kpconv.%s.out' % name]) < TOL, name


def test_model_forward_losses_and_grads(golden_s0):
    g = golden_s0
    cfg = cfgmod.default_config(first_features_dim=16)
    batch = _batch(g)
    sd = {k: v.clone().requires_grad_(not k.endswith('kernel_points')) for k, v in _sd(g).items()}
    feats, scores = ops_ref.kpfcnn_forward(sd, batch, cfg, training=True)
    assert rel_err(feats.detach().numpy(), g['features_train']) < 1e-4
    assert rel_err(scores.detach().numpy(), g['scores_train']) < 1e-4
    corr = torch.from_numpy(g['corr']).long()
    n0 = int(g['batch.stack_lengths.0'][0])
    dk = torch.from_numpy(g['dist_keypts'])
    loss, acc, fp, an, dists = ops_ref.circle_loss(feats[corr[:, 0]], feats[corr[:, 1] + n0], dk)
    det = ops_ref.det_loss(dists, scores[corr[:, 0]], scores[corr[:, 1] + n0])
    assert abs(loss.item() - float(g['desc_loss'])) < 1e-4 * max(1.0, abs(float(g['desc_loss'])))
    assert abs(det.item() - float(g['det_loss'])) < 1e-4
    assert abs(float(acc) - float(g['accuracy'])) < 1e-3
    assert rel_err(dists.detach().numpy(), g['dists']) < 1e-4
    assert rel_err(fp.detach().numpy(), g['furthest_positive']) < 1e-4
    assert rel_err(an.detach().numpy(), g['average_negative']) < 1e-4
    (loss + det).backward()
    checked = 0
    for k, v in sd.items():
        if ('grad.' + k) in g.files:
            ref = g['grad.' + k]
            if np.abs(ref).max() > 0:
                assert rel_err(v.grad.numpy(), ref) < 2e-3, k
                checked += 1
    assert checked > 50
    with torch.no_grad():
        fe, se = ops_ref.kpfcnn_forward({k: v.detach() for k, v in sd.items()}, batch, cfg, training=False)
    assert rel_err(fe.numpy(), g['features_eval']) < 1e-4
    # eval scores are gated by an exact float equality -> compare the gate pattern, then the values where open
    ref = g['scores_eval']
    same_gate = ((se.numpy() != 0) == (ref != 0)).mean()
    assert same_gate > 0.999
    m = (se.numpy() != 0) & (ref != 0)
    assert rel_err(se.numpy()[m], ref[m]) < 1e-4


def test_matching(golden_s0):
    g = golden_s0
    n0 = int(g['batch.stack_lengths.0'][0])
    fe = g['features_eval']
    got = ops_ref.build_correspondence(fe[:n0][g['match.src_idx']], fe[n0:][g['match.tgt_idx']])
    assert np.array_equal(got, g['match.corr250'])
    assert np.array_equal(ops_ref.build_correspondence(fe[:n0], fe[n0:]), g['match.corr_all'])


def test_collate_on_oracle_equals_reference_collate(golden_s0, native):
    g = golden_s0
    cfg = cfgmod.default_config(first_features_dim=16)
    from util import assert_neighbors_equal_tie_aware
    d = ops_ref.collate(g['pts0'], g['pts1'], cfg, g['limits'], native)
    for l in range(5):
        assert np.array_equal(d['points'][l].numpy().view(np.uint32), g['batch.points.%d' % l].view(np.uint32))
        assert np.array_equal(d['stack_lengths'][l].numpy(), g['batch.stack_lengths.%d' % l])
        p = d['points'][l].numpy()
        assert_neighbors_equal_tie_aware(p, p, d['neighbors'][l].numpy(), g['batch.neighbors.%d' % l])
        assert d['pools'][l].shape == g['batch.pools.%d' % l].shape
        assert d['upsamples'][l].shape == g['batch.upsamples.%d' % l].shape


@pytest.mark.parametrize("influence", ["linear", "constant", "gaussian"])
@pytest.mark.parametrize("aggregation", ["sum", "closest"])
def test_kpconv_modes_against_reference_vectors(influence, aggregation):
    """Oracle restatement of the influence / aggregation modes (blocks.py:327-352) vs vectors from the real reference
    (tests/golden/make_golden_modes.py)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'kpconv_modes.npz'))
    x = torch.from_numpy(g['x']).requires_grad_(True)
    w = torch.from_numpy(g['weights']).requires_grad_(True)
    out = ops_ref.kpconv(torch.from_numpy(g['q_pts']), torch.from_numpy(g['s_pts']), torch.from_numpy(g['inds']), x,
                         torch.from_numpy(g['kernel_points']), w, float(g['extent']), influence, aggregation)
    out.backward(torch.from_numpy(g['gout']))
    tag = '%s.%s.' % (influence, aggregation)
    for got, key in ((out.detach(), 'out'), (x.grad, 'grad_x'), (w.grad, 'grad_w')):
        want = g[tag + key]
        assert np.abs(got.numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), key
    if (influence, aggregation) != ('linear', 'sum'):
        assert np.abs(g[tag + 'out'] - g['linear.sum.out']).max() > 1e-3    # the modes really differ on this input


DEFORM_CASES = [('linear', 'sum', 0), ('linear', 'sum', 1), ('gaussian', 'sum', 1), ('linear', 'closest', 0)]


@pytest.mark.parametrize("influence,aggregation,modulated", DEFORM_CASES)
def test_deformable_kpconv_against_reference_vectors(influence, aggregation, modulated):
    """Oracle restatement of the deformable / modulated KPConv (blocks.py:243-387) vs vectors from the real reference
    (tests/golden/make_golden_modes.py -> kpconv_deform.npz): outputs, min_d2, deformed_KP and every gradient of
    sum(out * gout) + 0.7 sum(min_d2 * gmin)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'kpconv_deform.npz'))
    tag = '%s.%s.%d.' % (influence, aggregation, modulated)
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    x = t('x').requires_grad_(True)
    w = t(tag + 'sd.weights').requires_grad_(True)
    ow = t(tag + 'sd.offset_conv.weights').requires_grad_(True)
    ob = t(tag + 'sd.offset_bias').requires_grad_(True)
    ext = float(g['extent'])
    off = ops_ref.kpconv(t('q_pts'), t('s_pts'), t('inds'), x, t(tag + 'sd.offset_conv.kernel_points'), ow, ext,
                         influence, aggregation) + ob
    out, min_d2, dkp = ops_ref.kpconv_deformable(t('q_pts'), t('s_pts'), t('inds'), x, t(tag + 'sd.kernel_points'), w,
                                                 ext, off, bool(modulated), influence, aggregation)
    ((out * t('gout')).sum() + 0.7 * (min_d2 * t('gmin')).sum()).backward()
    for got, key in ((out.detach(), 'out'), (min_d2.detach(), 'min_d2'), (dkp.detach(), 'deformed_KP'),
                     (x.grad, 'grad_x'), (w.grad, 'grad.weights'), (ow.grad, 'grad.offset_conv.weights'),
                     (ob.grad, 'grad.offset_bias')):
        want = g[tag + key]
        assert np.abs(got.numpy() - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), key
    assert 0.5 < float(g[tag + 'live_fraction']) < 0.995      # the range filter dropped some neighbors, not all
