#!/usr/bin/env python
"""Runs only the layer-0 32->32 KPConv (forward + backward) of the S1 pair a few times -- target for rocprofv3 --pmc."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from d3feat_pytorch_amd import config as cfgmod, ops, synthetic  # noqa: E402
from d3feat_pytorch_amd.datasets import dataloader as dl  # noqa: E402

dev = torch.device("cuda:0")
cfg = cfgmod.default_config()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 0
C = int(sys.argv[2]) if len(sys.argv) > 2 else 32


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


item = synthetic.make_pair(1, 2, sub)
batch = dl.collate_fn_descriptor([item], cfg, [42] * 5, exact_width=False)
s = batch['points'][L]
idx = batch['neighbors'][L]
rng = np.random.default_rng(0)
r = 0.075 * 2 ** L
x = torch.from_numpy(np.abs(rng.normal(size=(s.shape[0], C))).astype(np.float32)).to(dev).requires_grad_(True)
w = torch.from_numpy((rng.normal(size=(15, C, C)) / np.sqrt(15 * C)).astype(np.float32)).to(dev).requires_grad_(True)
kp = torch.from_numpy((rng.normal(size=(15, 3)) * r * 0.4).astype(np.float32)).to(dev)
go = torch.ones((s.shape[0], C), device=dev)
for _ in range(5):
    out = ops.kpconv(s, s, idx, x, kp, w, r * 0.8)
    out.backward(go)
torch.cuda.synchronize()
