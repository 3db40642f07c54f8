#!/usr/bin/env python
"""cProfile of the host side of the training step (where do the ~11 ms of enqueue time per step go?)."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from d3feat_pytorch_amd import config as cfgmod, synthetic  # noqa: E402
from d3feat_pytorch_amd.datasets import dataloader as dl  # noqa: E402
from d3feat_pytorch_amd.train import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
cfg = cfgmod.default_config()


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


items = []
for i in range(2):
    it = synthetic.make_pair(2 * i + 1, 2 * i + 2, sub)
    items.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in it))
ts = TrainStep(cfg, [42] * 5, dev)
for k in range(4):
    ts.step(items[k % 2], next_item=items[(k + 1) % 2])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
N = 10
for k in range(N):
    ts.step(items[k % 2], next_item=items[(k + 1) % 2])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
