#!/usr/bin/env python
"""Ablation timing of the fused KPConv grad-input kernel per pyramid level: scatter atomics vs gW tile vs the rest.

Round-1 findings (MI355X): levels 0/1 are bound by the fp32 scatter atomics (L0: 176 us full, 71 us without the
atomics; ~0.5 T float atomics/s chip-wide), levels 3/4 by the gW tile GEMM running on 10..40 workgroups.  XCD-private
accumulation copies with workgroup-scope atomics were tried and are NOT faster (198 us): the limit is the L2 atomic
unit's per-float rate, not cross-XCD traffic."""
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
L_ = _native.lib()


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


item = synthetic.make_pair(1, 2, sub)
batch = dl.collate_fn_descriptor([item], cfg, [42] * 5, exact_width=False)
rng = np.random.default_rng(0)
busy = torch.randn(8192, 8192, device=dev)
for L, C in ((0, 32), (1, 64), (2, 128), (3, 256), (4, 512)):
    s = batch['points'][L]
    idx = batch['neighbors'][L].contiguous()
    Nq, H, K = int(s.shape[0]), int(idx.shape[1]), 15
    r = 0.075 * 2 ** L
    x = torch.from_numpy(np.abs(rng.normal(size=(Nq, C))).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.normal(size=(K, C, C)) / np.sqrt(K * C)).astype(np.float32)).to(dev)
    kp = torch.from_numpy((rng.normal(size=(K, 3)) * r * 0.4).astype(np.float32)).to(dev)
    nn = torch.ones(Nq, device=dev)
    go = torch.randn(Nq, C, device=dev)
    gx = torch.empty(Nq, C, device=dev)
    gw = torch.empty(K, C, C, device=dev)
    nb = L_.d3f_kpconv_ws_bytes(Nq, Nq, H, K, C, C)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def call(want_x=True, want_w=False):
        rc = L_.d3f_kpconv_backward(s.data_ptr(), Nq, s.data_ptr(), Nq, idx.data_ptr(), H, x.data_ptr(), C, kp.data_ptr(), K,
                                    w.data_ptr(), C, r * 0.8, nn.data_ptr(), go.data_ptr(), None, None, 0,
                                    gx.data_ptr() if want_x else None, gw.data_ptr() if want_w else None, ws.data_ptr(), nb, st)
        assert rc == 0, rc

    def timed(fn, n=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.mm(busy, busy)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    row = []
    for flags, name in ((0, "full"), (8, "no atomics"), (16, "no phase1"), (32, "no phase2"), (48, "setup only")):
        L_.d3f_debug_set_flags(flags)
        row.append("%s %.1f" % (name, timed(call)))
    L_.d3f_debug_set_flags(0)
    row.append("| dW recompute %.1f" % timed(lambda: call(False, True)))
    print("L%d Nq=%d C=%d (zero-fill + pack + dx kernel, us): %s" % (L, Nq, C, ", ".join(row)))
