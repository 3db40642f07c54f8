#!/usr/bin/env python
"""Per-layer KPConv timing on the S1-class benchmark pair (HIP events on the launch stream, L2-warm steady state).

    python profiles/kpconv_microbench.py [--reps 20]
Prints, for every KPConv shape of the network, forward / backward time and the achieved algorithmic GB/s
(SURVEY.md section 8d byte model) -- the numbers behind bench.py's `roofline` object."""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from d3feat_pytorch_amd import config as cfgmod, ops, synthetic  # noqa: E402
from d3feat_pytorch_amd.datasets import dataloader as dl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--morton", action="store_true", help="experiment: spatially sort every level first")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = cfgmod.default_config()

    def sub(p, l, d):
        a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
        return a.cpu().numpy(), b.cpu().numpy()
    item = synthetic.make_pair(1, 2, sub)
    batch = dl.collate_fn_descriptor([item], cfg, [42] * 5, exact_width=False)
    if args.morton:
        def morton(p, cell):
            c = ((p - p.min(0, keepdim=True)[0]) / cell).long()
            code = torch.zeros(p.shape[0], dtype=torch.long, device=p.device)
            for b in range(10):
                for a in range(3):
                    code |= ((c[:, a] >> b) & 1) << (3 * b + a)
            return torch.argsort(code)
        pts, lens = [], []
        for l in range(5):
            p = batch['points'][l]
            n0 = int(batch['stack_lengths'][l][0])
            o0 = morton(p[:n0], 0.075 * 2 ** l / 2)
            o1 = morton(p[n0:], 0.075 * 2 ** l / 2) + n0
            pts.append(p[torch.cat([o0, o1])].contiguous())
        for l in range(5):
            r = 0.075 * 2 ** l
            g = ops.RadiusGrid(pts[l], batch['stack_lengths'][l], r)
            batch['neighbors'][l] = g.query(pts[l], batch['stack_lengths'][l], 42)
            if l < 4:
                batch['pools'][l] = g.query(pts[l + 1], batch['stack_lengths'][l + 1], 42)
            batch['points'][l] = pts[l]
    # (level, strided, Cin, Cout, how many times the shape occurs in the network)
    layers = [(0, False, 1, 64, 1), (0, False, 32, 32, 1), (0, True, 32, 32, 1), (1, False, 64, 64, 2),
              (1, True, 64, 64, 1), (2, False, 128, 128, 2), (2, True, 128, 128, 1), (3, False, 256, 256, 2),
              (3, True, 256, 256, 1), (4, False, 512, 512, 2)]
    rng = np.random.default_rng(0)
    tot_f = tot_b = 0.0
    print("%-34s %9s %9s %9s %9s" % ("layer", "fwd_us", "fwd_GB/s", "bwd_us", "bwd_GB/s"))
    for (l, strided, cin, cout, mult) in layers:
        s = batch['points'][l]
        q = batch['points'][l + 1] if strided else s
        idx = batch['pools'][l] if strided else batch['neighbors'][l]
        r = 0.075 * 2 ** l
        x = torch.from_numpy(np.abs(rng.normal(size=(s.shape[0], cin))).astype(np.float32)).to(dev).requires_grad_(True)
        w = torch.from_numpy((rng.normal(size=(15, cin, cout)) / np.sqrt(15 * cin)).astype(np.float32)).to(dev)
        w.requires_grad_(True)
        kp = torch.from_numpy((rng.normal(size=(15, 3)) * r * 0.4).astype(np.float32)).to(dev)
        ext = r * 2.0 / 2.5
        go = torch.ones((q.shape[0], cout), device=dev)
        for _ in range(3):
            out = ops.kpconv(q, s, idx, x, kp, w, ext)
            out.backward(go)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(args.reps):
            x.grad = None
            w.grad = None
            e[0].record()
            out = ops.kpconv(q, s, idx, x, kp, w, ext)
            e[1].record()
            out.backward(go)
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1])
            tb += e[1].elapsed_time(e[2])
        tf, tb = tf / args.reps * 1e3, tb / args.reps * 1e3
        nq, ns, h = q.shape[0], s.shape[0], idx.shape[1]
        bf, bb = ops.kpconv_fwd_bytes(nq, ns, h, 15, cin, cout), ops.kpconv_bwd_bytes(nq, ns, h, 15, cin, cout)
        name = "L%d%s %d->%d Nq=%d" % (l, "s" if strided else " ", cin, cout, nq)
        tot_f += tf * mult
        tot_b += tb * mult
        print("%-34s %9.1f %9.0f %9.1f %9.0f" % (name, tf, bf / tf / 1e3, tb, bb / tb / 1e3))
    print("sum over the 14 KPConvs of the net: fwd %.0f us, bwd %.0f us" % (tot_f, tot_b))


if __name__ == "__main__":
    main()
