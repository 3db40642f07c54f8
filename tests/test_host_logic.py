"""CPU: host-side logic of the package -- no kernel launches."""
import os
import sys

import numpy as np
import pytest
import torch

import d3feat_pytorch_amd as pkg
from d3feat_pytorch_amd import config as cfgmod
from d3feat_pytorch_amd import ops, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.kernels import kernel_points as kp_mod
from d3feat_pytorch_amd.kernels.kernel_points import base_disposition, load_kernels
from d3feat_pytorch_amd.models import blocks
from d3feat_pytorch_amd.models.architectures import KPFCNN
from d3feat_pytorch_amd.utils.loss import CircleLoss, DetLoss, LazyList

REF = "/root/reference"


def test_ops_refuse_cpu_tensors_loudly():
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError):
        ops.kpconv(x, x, torch.zeros(4, 2, dtype=torch.long), torch.zeros(4, 1), torch.zeros(15, 3),
                   torch.zeros(15, 1, 8), 0.06)
    with pytest.raises(RuntimeError):
        ops.max_pool(torch.zeros(4, 8), torch.zeros(2, 3, dtype=torch.long))
    with pytest.raises(RuntimeError):
        ops.detection_scores(torch.zeros(4, 32), torch.zeros(4, 3, dtype=torch.long))
    with pytest.raises(RuntimeError):
        ops.RadiusGrid(x, [4], 0.1)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            dl.batch_neighbors_kpconv(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32), [4], [4], 0.1, 10)


def test_block_walk_matches_reference_radii():
    cfg = cfgmod.default_config()
    w = dl._Walk(cfg)
    assert len(w.layers) == 5
    for l, e in enumerate(w.layers):
        assert e['conv_r'] == pytest.approx(0.075 * 2 ** l)
        assert e['pool'] == (l < 4)
        if l < 4:
            assert e['dl'] == pytest.approx(0.06 * 2 ** l) and e['up_r'] == pytest.approx(0.15 * 2 ** l)
    assert cfg.architecture[:5] == ['simple', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb']
    assert cfg.architecture[-2:] == ['nearest_upsample', 'last_unary'] and len(cfg.architecture) == 22


def test_kernel_points_contract():
    base = base_disposition(15, 3, 'center')
    assert base.shape == (15, 3) and np.allclose(base[0], 0)
    assert abs(np.linalg.norm(base[1:], axis=1).mean() - 0.66) < 1e-6
    d = np.linalg.norm(base[:, None] - base[None], axis=-1) + np.eye(15)
    assert d.min() > 0.3  # well spread
    np.random.seed(3)
    kp = load_kernels(0.075, 15, 3, 'center')
    assert kp.dtype == np.float32 and kp.shape == (15, 3) and np.abs(kp).max() < 0.075


def test_kernel_point_generators_equal_the_reference_runs():
    """tests/golden/kernel_points.npz = reference kernels/kernel_points.py under fixed global seeds
    (make_golden_extra.py kernels).  Same RNG draws, same arithmetic: equal to rounding of the last bits."""
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'kernel_points.npz'))
    for tag in 'abc':
        k, dim, f = (int(v) for v in g['opt.%s.args' % tag])
        np.random.seed(11)
        pts, hist = kp_mod.kernel_point_optimization(1.0, k, num_kernels=8, dimension=dim,
                                                     fixed=['center', 'none', 'verticals'][f])
        assert int((hist.max(axis=1) > 0).sum()) == int(g['opt.%s.iters' % tag]), tag
        np.testing.assert_allclose(pts, g['opt.%s.points' % tag], rtol=0, atol=1e-12)
        np.testing.assert_allclose(hist[-1], g['opt.%s.last' % tag], rtol=0, atol=1e-12)
    np.random.seed(12)
    np.testing.assert_allclose(kp_mod.spherical_lloyd(1.0, 32, max_iter=60), g['lloyd.points'], rtol=0, atol=1e-12)
    np.random.seed(13)
    assert np.array_equal(load_kernels(0.075, 15, 3, 'center'), g['load.15.a'])   # the shipped K=15 table + RNG order
    assert np.array_equal(load_kernels(1.2, 15, 3, 'center'), g['load.15.b'])


def test_model_state_dict_layout_matches_golden(golden_s0, golden_s1):
    np.random.seed(0)
    torch.manual_seed(0)
    model = KPFCNN(cfgmod.default_config(first_features_dim=16))
    sd = model.state_dict()
    gold = {k[3:]: golden_s0[k].shape for k in golden_s0.files if k.startswith('sd.')}
    assert set(sd) == set(gold)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(gold[k]), k
    # identical construction order + same torch CPU RNG => identical initial weights as the reference model
    # ... and same NumPy RNG + the shipped K=15 disposition => identical kernel points too
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), golden_s0['sd.' + k]), k
    np.random.seed(0)
    torch.manual_seed(0)
    full = KPFCNN(cfgmod.default_config())
    n_params = sum(p.numel() for p in full.parameters() if p.requires_grad)
    assert n_params == 24316320
    for k, v in full.state_dict().items():
        s = golden_s1['sdsum.' + k]
        assert abs(float(v.double().sum()) - s[0]) <= 1e-9 * max(1.0, s[1]), k


def test_lazy_list_and_loss_signatures():
    t = torch.tensor([1.0, 2.0, 3.0])
    ll = LazyList(t)
    assert len(ll) == 3 and float(ll.device_mean()) == 2.0 and list(ll) == [1.0, 2.0, 3.0] and ll[1] == 2.0
    assert float(np.mean(ll)) == 2.0
    c = CircleLoss(dist_type='euclidean', log_scale=10, safe_radius=0.1, pos_margin=0.1, neg_margin=1.4)
    assert c.pos_optimal == 0.1 and c.neg_optimal == 1.4
    with pytest.raises(NotImplementedError):
        CircleLoss(dist_type='chebyshev')                       # the reference's cdist rejects it too (loss.py:42-44)


@pytest.mark.parametrize("metric", ["cosine", "arccosine", "sqeuclidean", "cityblock"])
def test_losses_with_the_other_cdist_metrics(metric):
    """CircleLoss / DetLoss with the metrics of the reference's cdist other than the configured 'euclidean'
    (loss.py:8-44; CircleLoss' constructor default is 'cosine'): metric-agnostic tensor algebra, equal to the oracle
    restatement -- and, where the reference tree is mounted, to the reference's own modules -- values and gradients."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from oracle import ops_ref
    gen = torch.Generator().manual_seed(3)
    a0 = torch.nn.functional.normalize(torch.randn(48, 32, generator=gen), dim=1) * 0.98
    p0 = torch.nn.functional.normalize(a0 + 0.3 * torch.randn(48, 32, generator=gen), dim=1) * 0.98
    dk = torch.rand(48, 48, generator=gen) * 0.3
    sa, sp = torch.rand(48, 1, generator=gen), torch.rand(48, 1, generator=gen)

    def run(circle_fn, det_fn):
        a, p = a0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
        out = circle_fn(a, p, dk)
        loss, dists = out[0], out[-1]
        det = det_fn(dists, sa, sp)
        (loss + det).backward()
        return [loss.detach(), det.detach(), torch.as_tensor(float(out[1])), torch.as_tensor(list(out[2])),
                torch.as_tensor(list(out[3])), a.grad, p.grad]

    ours = run(CircleLoss(dist_type=metric, log_scale=10, safe_radius=0.1, pos_margin=0.1, neg_margin=1.4), DetLoss())
    orc = run(lambda a, p, d: ops_ref.circle_loss(a, p, d, metric=metric), ops_ref.det_loss)
    for u, v in zip(ours, orc):
        assert torch.allclose(u.float(), v.float(), rtol=1e-5, atol=1e-6)
    if os.path.isdir(REF):
        saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == 'utils' or k.startswith('utils.')}
        for k in saved:
            sys.modules.pop(k, None)
        sys.path.insert(0, REF)
        try:
            import importlib
            ref_loss = importlib.import_module('utils.loss')
            ref = run(ref_loss.CircleLoss(dist_type=metric, log_scale=10, safe_radius=0.1, pos_margin=0.1,
                                          neg_margin=1.4), ref_loss.DetLoss())
        finally:
            sys.path.remove(REF)
            for k in [k for k in sys.modules if k == 'utils' or k.startswith('utils.')]:
                sys.modules.pop(k, None)
            sys.modules.update({k: v for k, v in saved.items() if v is not None})
        for u, v in zip(ours, ref):
            assert torch.allclose(u.float(), v.float(), rtol=1e-5, atol=1e-6)


def test_synthetic_is_deterministic_and_shaped(native):
    sub = lambda p, l, d: native.subsample_batch(p, l, sampleDl=d)  # noqa: E731
    a = synthetic.make_pair(11, 12, sub, n_raw=40000, scale=0.2, num_node=64)
    b = synthetic.make_pair(11, 12, sub, n_raw=40000, scale=0.2, num_node=64)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[0].dtype == np.float32 and a[4].shape == (64, 2) and a[5].shape == (64, 64) and a[5].dtype == np.float64
    assert a[2].shape == (a[0].shape[0], 1)


def test_unsupported_modes_raise():
    with pytest.raises(NotImplementedError):
        blocks.KPConv(20, 3, 8, 8, 0.06, 0.075)              # more kernel points than a 16-lane group
    np.random.seed(1)
    torch.manual_seed(1)
    d = blocks.KPConv(15, 3, 8, 12, 0.06, 0.075, deformable=True, modulated=True)   # reference blocks.py:187-203
    assert d.offset_dim == 60 and tuple(d.offset_conv.weights.shape) == (15, 8, 60) and d.offset_bias.shape == (60,)
    assert set(d.state_dict()) == {'weights', 'kernel_points', 'offset_bias', 'offset_conv.weights',
                                   'offset_conv.kernel_points'}
    with pytest.raises(ValueError, match='Unknown influence'):
        blocks.KPConv(15, 3, 8, 8, 0.06, 0.075, KP_influence='cubic')
    with pytest.raises(ValueError, match='Unknown convolution mode'):
        blocks.KPConv(15, 3, 8, 8, 0.06, 0.075, aggregation_mode='mean')
    assert ops.kpconv_mode('gaussian', 'closest') == 6 and ops.kpconv_mode() == 0
    conv = blocks.KPConv(15, 3, 8, 8, 0.06, 0.075, KP_influence='gaussian', aggregation_mode='closest')
    assert conv.KP_influence == 'gaussian' and conv.aggregation_mode == 'closest'
    with pytest.raises(ValueError):
        blocks.block_decider('nonsense', 0.1, 8, 8, 0, cfgmod.default_config())


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
def test_reference_architectures_runs_on_our_blocks_unchanged():
    """Drop-in boundary: the reference's models/architectures.py imported as-is, with `models.blocks` resolved to
    this package, builds the same network (same parameter names and shapes)."""
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == 'models' or k.startswith('models.')}
    for k in saved:
        sys.modules.pop(k, None)
    try:
        pkg.install_reference_aliases(['models', 'models.blocks'], overwrite=True)
        import importlib.util
        spec = importlib.util.spec_from_file_location('ref_architectures', os.path.join(REF, 'models', 'architectures.py'))
        mod = importlib.util.module_from_spec(spec)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(mod)
            np.random.seed(0)
            torch.manual_seed(0)
            ref_model = mod.KPFCNN(cfgmod.default_config(first_features_dim=16))
        assert isinstance(ref_model.encoder_blocks[0].KPConv, blocks.KPConv)
        np.random.seed(0)
        torch.manual_seed(0)
        ours = KPFCNN(cfgmod.default_config(first_features_dim=16))
        a, b = ref_model.state_dict(), ours.state_dict()
        assert list(a) == list(b)
        for k in a:
            assert torch.equal(a[k], b[k]), k
    finally:
        for k in [k for k in sys.modules if k == 'models' or k.startswith('models.')]:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


def test_grad_holder_protocol():
    """ops.GradHolder: first deposit wins, collect closes, a late deposit is refused (its owner then returns the
    gradient the ordinary way) -- pure host logic of the two-branch gradient fusion."""
    h = ops.GradHolder()
    a, b = torch.ones(2), torch.zeros(2)
    assert not h.deposit(None)
    assert h.deposit(a) and not h.deposit(b)          # one slot
    assert h.collect() is a and h.closed
    assert h.collect() is None                         # emptied
    assert not h.deposit(b)                            # closed: late branch keeps its gradient
    # the identity tap defers to the holder only while it is open
    x = torch.randn(3, requires_grad=True)
    h2 = ops.GradHolder()
    y = ops.grad_tap(x, h2)
    (y * 2).sum().backward()
    assert x.grad is None and torch.equal(h2.collect(), torch.full((3,), 2.0))
    h3 = ops.GradHolder()
    h3.collect()                                       # consumer ran first
    y = ops.grad_tap(x, h3)
    (y * 3).sum().backward()
    assert torch.equal(x.grad, torch.full((3,), 3.0))
    assert ops.grad_tap(x, None) is x
