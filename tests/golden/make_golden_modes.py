#!/usr/bin/env python
"""Golden vectors for the non-default KPConv modes, from the REAL reference (build container only).

    python tests/golden/make_golden_modes.py    ->  tests/golden/kpconv_modes.npz   (data only)

Runs the reference's own ``models.blocks.KPConv`` (imported unmodified, cwd = /root/reference like make_golden.py) on
one small random neighborhood problem for every (KP_influence, aggregation_mode) pair of models/blocks.py:327-352:
outputs, and the gradients w.r.t. the features and the kernel weights for a fixed upstream gradient.
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


if __name__ == '__main__':
    main()
