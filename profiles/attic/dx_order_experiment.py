#!/usr/bin/env python
"""Does the ORDER in which queries are processed matter for the scatter-bound KPConv grad-input kernel?
Level-0/1 shapes of the S1 pair, queries in storage order vs Morton (Z-curve) order; also reports how many distinct
support rows the 16 queries of a tile touch (the factor an LDS pre-combination could save).

Round-1 findings (MI355X):
  * order alone does not help: L0 174 us (storage order) vs 195 us (Morton), L1 79 vs 78 us -- although a Morton tile
    touches only ~116 distinct support rows for its 672 (query, neighbor) slots (629 in storage order);
  * an LDS pre-combination was built and measured (open-addressing table support row -> slot, ds_add_f32 into one row
    of CC floats per slot, one set of global atomics per distinct row): correct, but 377 us at L0 -- the LDS float
    atomics serialise (384 ds_add_f32 per tile, 4-way same-bank) and cost more than the global atomics they replace
    (ablation: table + flush without accumulation 175 us, kernel without any atomics 71 us).  Not kept.
  * profiles/experiments/atomic_width.hip: the L2 float-atomic rate is ~320 G floats/s chip-wide whatever the width
    of the contiguous segment (64 / 128 / 256 B per row), and ~20 G requests/s for single-float requests."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from d3feat_pytorch_amd import _native, config as cfgmod, synthetic  # noqa: E402
from d3feat_pytorch_amd.datasets import dataloader as dl  # noqa: E402

dev = torch.device("cuda:0")
cfg = cfgmod.default_config()
L_ = _native.lib()


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


def morton(p, cell):
    q = np.floor((p - p.min(0)) / cell).astype(np.int64)
    code = np.zeros(len(p), np.int64)
    for b in range(16):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return np.argsort(code, kind="stable")


item = synthetic.make_pair(1, 2, sub)
batch = dl.collate_fn_descriptor([item], cfg, [42] * 5, exact_width=False)
rng = np.random.default_rng(0)
busy = torch.randn(8192, 8192, device=dev)
for L, C in ((0, 32), (1, 64)):
    s = batch['points'][L]
    idx0 = batch['neighbors'][L].contiguous()
    Nq, H, K = int(s.shape[0]), int(idx0.shape[1]), 15
    r = 0.075 * 2 ** L
    x = torch.from_numpy(np.abs(rng.normal(size=(Nq, C))).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.normal(size=(K, C, C)) / np.sqrt(K * C)).astype(np.float32)).to(dev)
    kp = torch.from_numpy((rng.normal(size=(K, 3)) * r * 0.4).astype(np.float32)).to(dev)
    nn0 = torch.ones(Nq, device=dev)
    go0 = torch.randn(Nq, C, device=dev)
    perm = torch.from_numpy(morton(s.cpu().numpy(), r)).to(dev)
    res = {}
    for name, order in (("storage order", None), ("morton order", perm)):
        q = s if order is None else s[order].contiguous()
        idx = idx0 if order is None else idx0[order].contiguous()
        go = go0 if order is None else go0[order].contiguous()
        gx = torch.empty(Nq, C, device=dev)
        nb = L_.d3f_kpconv_ws_bytes(Nq, Nq, H, K, C, C)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def call():
            rc = L_.d3f_kpconv_backward(q.data_ptr(), Nq, s.data_ptr(), Nq, idx.data_ptr(), H, x.data_ptr(), C,
                                        kp.data_ptr(), K, w.data_ptr(), C, r * 0.8, nn0.data_ptr(), go.data_ptr(), None,
                                        None, 0, gx.data_ptr(), None, ws.data_ptr(), nb, st)
            assert rc == 0
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.mm(busy, busy)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        tiles = idx.cpu().numpy()[: (Nq // 16) * 16].reshape(-1, 16 * H)
        uniq = np.mean([len(np.unique(t[t < Nq])) for t in tiles[::37]])
        res[name] = gx.clone()
        print("L%d C=%d %-14s %7.1f us   distinct supports per 16-query tile: %.0f of %d slots" % (
            L, C, name, e0.elapsed_time(e1) / 10 * 1e3, uniq, 16 * H))
    d = (res["storage order"] - res["morton order"]).abs().max().item() / res["storage order"].abs().max().item()
    print("    same gradient either way: rel diff %.1e" % d)
