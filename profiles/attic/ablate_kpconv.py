#!/usr/bin/env python
"""Ablation timing of the fused KPConv forward on the layer-0 32->32 shape: which phase owns the time?"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from d3feat_pytorch_amd import _native, config as cfgmod, ops, synthetic  # noqa: E402
from d3feat_pytorch_amd.datasets import dataloader as dl  # noqa: E402

dev = torch.device("cuda:0")
cfg = cfgmod.default_config()


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


item = synthetic.make_pair(1, 2, sub)
batch = dl.collate_fn_descriptor([item], cfg, [42] * 5, exact_width=False)
rng = np.random.default_rng(0)
busy = torch.randn(8192, 8192, device=dev)
for L, C in ((0, 32), (1, 64), (2, 128), (3, 256), (4, 512)):
    s = batch['points'][L]
    idx = batch['neighbors'][L]
    r = 0.075 * 2 ** L
    x = torch.from_numpy(np.abs(rng.normal(size=(s.shape[0], C))).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.normal(size=(15, C, C)) / np.sqrt(15 * C)).astype(np.float32)).to(dev)
    kp = torch.from_numpy((rng.normal(size=(15, 3)) * r * 0.4).astype(np.float32)).to(dev)
    for flags, name in ((0, "full"), (1, "no phase A"), (2, "no phase B"), (3, "no A, no B"), (7, "empty (setup only)"),
                        (4, "A+B, no stores")):
        _native.lib().d3f_debug_set_flags(flags)
        for _ in range(3):
            ops.kpconv(s, s, idx, x, kp, w, r * 0.8)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.mm(busy, busy)  # the host runs ahead: launch gaps are hidden, the number is GPU time
        e0.record()
        for _ in range(20):
            ops.kpconv(s, s, idx, x, kp, w, r * 0.8)
        e1.record()
        torch.cuda.synchronize()
        print("L%d C=%d %-22s %8.1f us per call (pack kernel + fused kernel, GPU time)" % (L, C, name, e0.elapsed_time(e1) / 20 * 1e3))
_native.lib().d3f_debug_set_flags(0)
