"""Level sizes of the S1-class synthetic pairs used by bench.py on ranks 0..7 (seeds 100r+2i+1, 100r+2i+2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import TrainStep

dev = torch.device("cuda:0")
cfg = cfgmod.default_config()


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


ts = TrainStep(cfg, [42] * 5, dev, seed=0)
allsz = []
for r in range(8):
    for i in range(4):
        it = synthetic.make_pair(100 * r + 2 * i + 1, 100 * r + 2 * i + 2, sub)
        item = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in it)
        b = ts.build_batch(item)
        allsz.append([int(t.shape[0]) for t in b['points']])
a = np.array(allsz)
print("min", a.min(0).tolist())
print("max", a.max(0).tolist())
print("per-rank max:", [a[4 * r:4 * r + 4].max(0).tolist() for r in range(8)])
