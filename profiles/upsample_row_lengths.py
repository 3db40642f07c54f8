import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from d3feat_pytorch_amd import ops, synthetic, config as cfgmod
from d3feat_pytorch_amd.datasets import dataloader as dl
DEV = torch.device("cuda:0")
def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(DEV), torch.as_tensor(l).to(DEV), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()
worst = [0] * 4
seeds = [(100 * r + 2 * i + 1) for r in range(8) for i in range(3)]
for sd in seeds:
    f = synthetic.make_fragment(sd, sub)
    pts = torch.from_numpy(f).to(DEV)
    lens = torch.tensor([f.shape[0]], dtype=torch.int32, device=DEV)
    dl0 = 0.03
    for lvl in range(4):
        cp, cl = dl.batch_grid_subsampling_kpconv(pts, lens, sampleDl=2 * dl0)
        r = 2.5 * dl0
        g = ops.RadiusGrid(pts, lens, r)
        tab, mx, lkey, (counts, keys) = g.query_pool_transposed(cp, cl, 40)
        g.status.raise_if_set()
        worst[lvl] = max(worst[lvl], int(counts.max()))
        pts, lens, dl0 = cp, cl, 2 * dl0
print("max coarse points within the pooling radius of a fine point, per level, over %d fragments: %s" % (len(seeds), worst))
