"""Do the level-0 gather kernels speed up when the points are stored in a spatially coherent (Morton) order, i.e. is the
gather rate limited by L2 locality?  Same cloud, same tables up to the row permutation.
python profiles/locality_experiment.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from d3feat_pytorch_amd import ops, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl

dev = torch.device("cuda:0")


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


def morton(p, cell):
    q = np.floor((p - p.min(0)) / cell).astype(np.uint64)
    code = np.zeros(len(p), np.uint64)
    for b in range(16):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return np.argsort(code, kind='stable')


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


it = synthetic.make_pair(1, 2, sub)
for name in ("reference order", "morton order"):
    clouds = [it[0], it[1]]
    if name.startswith("morton"):
        clouds = [c[morton(c, 0.075)] for c in clouds]
    pts = torch.from_numpy(np.concatenate(clouds)).to(dev)
    lens = torch.tensor([len(c) for c in clouds], dtype=torch.int32, device=dev)
    grid = ops.RadiusGrid(pts, lens, 0.075)
    tab = grid.query(pts, lens, 42)
    n = pts.shape[0]
    gen = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((n, 32), device=dev, generator=gen)
    kp = (torch.rand((15, 3), device=dev, generator=gen) - 0.5) * 0.1
    W = torch.randn((15, 32, 32), device=dev, generator=gen) * 0.1
    with torch.no_grad():
        t_kp = timeit(lambda: ops.kpconv(pts, pts, tab, x, kp, W, 0.06))
        t_det = timeit(lambda: ops.detection_scores(x, tab, training=True))
        sub1, len1, _, _ = ops.grid_subsample_raw(pts, lens, 0.06)
        n1 = int(len1.sum())
        ptab = ops.RadiusGrid(pts, lens, 0.075).query(sub1[:n1].contiguous(), len1, 42)
        x64 = torch.randn((n, 64), device=dev, generator=gen)
        t_mp = timeit(lambda: ops.max_pool(x64, ptab))
    print("%-16s kpconv fwd 32->32 %.1f us   detector fwd %.1f us   max_pool 38k->8k x64 %.1f us" % (name, t_kp, t_det, t_mp))
