#!/usr/bin/env python
"""Golden vectors for the non-default KPConv modes, from the REAL reference (build container only).

    python tests/golden/make_golden_modes.py    ->  tests/golden/kpconv_modes.npz, kpconv_deform.npz   (data only)

Runs the reference's own ``models.blocks.KPConv`` (imported unmodified, cwd = /root/reference like make_golden.py) on
one small random neighborhood problem for every (KP_influence, aggregation_mode) pair of models/blocks.py:327-352:
outputs, and the gradients w.r.t. the features and the kernel weights for a fixed upstream gradient.

kpconv_deform.npz: the same problem through ``KPConv(deformable=True, modulated=False/True)`` (blocks.py:187-203,
243-324,365-366) for 'linear'/'sum', 'gaussian'/'sum' and 'linear'/'closest': outputs, min_d2, deformed_KP, and the
gradients of  sum(out * gout) + 0.7 * sum(min_d2 * gmin)  w.r.t. the features, the kernel weights, the offset
convolution's weights and the offset bias (the min_d2 term is how the reference's fitting regulariser,
architectures.py:35-40, reaches the offsets).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.modules['open3d'] = types.ModuleType('open3d')       # file I/O only; never reached
for name in ['cpp_wrappers', 'cpp_wrappers.cpp_subsampling', 'cpp_wrappers.cpp_neighbors',
             'cpp_wrappers.cpp_subsampling.grid_subsampling', 'cpp_wrappers.cpp_neighbors.radius_neighbors']:
    sys.modules[name] = types.ModuleType(name)
sys.path.insert(0, REF)
os.chdir(REF)

import torch  # noqa: E402
from models.blocks import KPConv  # noqa: E402  (reference)


def main():
    rng = np.random.default_rng(2024)
    ns, nq, H, cin, cout, K = 400, 150, 24, 6, 10, 15
    radius, extent = 0.25, 0.12
    s_pts = rng.uniform(0, 1, size=(ns, 3)).astype(np.float32)
    q_ids = rng.permutation(ns)[:nq]
    q_pts = s_pts[q_ids]
    d2 = ((q_pts[:, None, :] - s_pts[None, :, :]) ** 2).sum(-1)
    order = np.argsort(d2, axis=1, kind='stable')[:, :H]
    inds = np.where(np.take_along_axis(d2, order, 1) < radius * radius, order, ns).astype(np.int64)   # shadow = ns
    x = rng.normal(size=(ns, cin)).astype(np.float32)
    x[rng.random(ns) < 0.1] = -np.abs(x[rng.random(ns) < 0.1][:1])          # some rows with a non-positive sum
    gout = rng.normal(size=(nq, cout)).astype(np.float32)
    g = {'q_pts': q_pts, 's_pts': s_pts, 'inds': inds, 'x': x, 'gout': gout, 'extent': np.float32(extent),
         'radius': np.float32(radius)}
    np.random.seed(7)
    torch.manual_seed(7)
    base = KPConv(K, 3, cin, cout, extent, radius)
    g['kernel_points'] = base.kernel_points.detach().numpy().copy()
    g['weights'] = base.weights.detach().numpy().copy()
    for infl in ('linear', 'constant', 'gaussian'):
        for agg in ('sum', 'closest'):
            np.random.seed(7)
            torch.manual_seed(7)
            conv = KPConv(K, 3, cin, cout, extent, radius, KP_influence=infl, aggregation_mode=agg)
            with torch.no_grad():
                conv.kernel_points.copy_(base.kernel_points)
                conv.weights.copy_(base.weights)
            xt = torch.from_numpy(x).requires_grad_(True)
            out = conv(torch.from_numpy(q_pts), torch.from_numpy(s_pts), torch.from_numpy(inds), xt)
            out.backward(torch.from_numpy(gout))
            tag = '%s.%s.' % (infl, agg)
            g[tag + 'out'] = out.detach().numpy()
            g[tag + 'grad_x'] = xt.grad.numpy()
            g[tag + 'grad_w'] = conv.weights.grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'kpconv_modes.npz'), **g)
    print({k: v.shape for k, v in g.items()})

    d = {k: g[k] for k in ('q_pts', 's_pts', 'inds', 'x', 'gout', 'extent', 'radius')}
    d['gmin'] = rng.normal(size=(nq, K)).astype(np.float32)
    for infl, agg, mod in (('linear', 'sum', False), ('linear', 'sum', True), ('gaussian', 'sum', True),
                           ('linear', 'closest', False)):
        np.random.seed(9)
        torch.manual_seed(9)
        conv = KPConv(K, 3, cin, cout, extent, radius, KP_influence=infl, aggregation_mode=agg, deformable=True,
                      modulated=mod)
        with torch.no_grad():
            conv.offset_conv.weights.mul_(0.6)      # offsets of a fraction of the extent: some neighbors leave the range
            conv.offset_bias.copy_(0.05 * torch.randn(conv.offset_dim))
        tag = '%s.%s.%d.' % (infl, agg, int(mod))
        for name, p in conv.state_dict().items():
            d[tag + 'sd.' + name] = p.detach().numpy().copy()
        xt = torch.from_numpy(x).requires_grad_(True)
        out = conv(torch.from_numpy(q_pts), torch.from_numpy(s_pts), torch.from_numpy(inds), xt)
        loss = (out * torch.from_numpy(gout)).sum() + 0.7 * (conv.min_d2 * torch.from_numpy(d['gmin'])).sum()
        loss.backward()
        d[tag + 'out'] = out.detach().numpy()
        d[tag + 'min_d2'] = conv.min_d2.detach().numpy()
        d[tag + 'deformed_KP'] = conv.deformed_KP.detach().numpy()
        d[tag + 'grad_x'] = xt.grad.numpy()
        for name, p in conv.named_parameters():
            if p.grad is not None:
                d[tag + 'grad.' + name] = p.grad.numpy()
        sq = ((torch.from_numpy(np.concatenate([s_pts, np.full((1, 3), 1e6, np.float32)]))[inds]
               - torch.from_numpy(q_pts)[:, None, :])[:, :, None, :] - conv.deformed_KP.detach()[:, None, :, :]).pow(2).sum(-1)
        live = (sq < extent ** 2).any(dim=2) & torch.from_numpy(inds < ns)
        d[tag + 'live_fraction'] = np.float32(live.float().sum() / max(1, int((inds < ns).sum())))
    np.savez_compressed(os.path.join(HERE, 'kpconv_deform.npz'), **d)
    print({k: (v.shape if hasattr(v, 'shape') else v) for k, v in d.items() if 'sd.' not in k})
    print({k: float(v) for k, v in d.items() if k.endswith('live_fraction')})


if __name__ == '__main__':
    main()
