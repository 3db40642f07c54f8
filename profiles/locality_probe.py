"""How much do the gather-bound kernels (KPConv forward / grad-input, max-pool, detector) lose to the ORDER of the points?

The pyramid keeps the reference's row order (libstdc++ unordered_map iteration order of the voxel keys: runs along x,
scattered in y / z), and every kernel walks the queries in row order: the 16 queries of a workgroup are not neighbours in
space, their ~40 neighbours each are ~600 distinct rows.  This probe times the same operators on the same cloud with the
rows (a) as the pipeline has them, (b) sorted by the cell of the conv radius (what the radius search's cell list holds
anyway), (c) sorted along a Morton curve of the voxel grid -- an upper bound of what a processing-order permutation inside
the kernels could buy.  Not a product path."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3feat_pytorch_amd import ops, synthetic  # noqa: E402
from d3feat_pytorch_amd.datasets import dataloader as dl  # noqa: E402
from d3feat_pytorch_amd.models import blocks  # noqa: E402

DEV = torch.device("cuda:0")


def gpu_subsample(points, lengths, dlen):
    p, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(points).to(DEV), torch.as_tensor(lengths).to(DEV), sampleDl=dlen)
    return p.cpu().numpy(), b.cpu().numpy()


def morton(ix, iy, iz):
    def spread(v):
        v = v.astype(np.uint64) & 0x1fffff
        v = (v | (v << 32)) & 0x1f00000000ffff
        v = (v | (v << 16)) & 0x1f0000ff0000ff
        v = (v | (v << 8)) & 0x100f00f00f00f00f
        v = (v | (v << 4)) & 0x10c30c30c30c30c3
        v = (v | (v << 2)) & 0x1249249249249249
        return v
    return spread(ix) | (spread(iy) << 1) | (spread(iz) << 2)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    frags = [synthetic.make_fragment(s, gpu_subsample) for s in (1, 2, 3, 4, 5, 6)]   # 3 pairs' worth of level-0 points
    lens = np.array([f.shape[0] for f in frags], dtype=np.int32)
    pts = np.concatenate(frags, 0)
    r0, dl0 = 0.075, 0.03

    def order(kind):
        out = []
        off = 0
        for f in frags:
            if kind == "pipeline":
                o = np.arange(f.shape[0])
            else:
                edge = r0 if kind == "cell" else dl0
                c = np.floor((f - f.min(0)) / edge).astype(np.int64)
                if kind == "cell":
                    key = (c[:, 0] * 73856093) ^ (c[:, 1] * 19349663) ^ (c[:, 2] * 83492791)    # hashed cell: cells scattered,
                    o = np.argsort(key & 0xffff, kind="stable")                                 # their members together
                else:
                    o = np.argsort(morton(c[:, 0], c[:, 1], c[:, 2]), kind="stable")
            out.append(o + off)
            off += f.shape[0]
        return np.concatenate(out)

    print("# %d points in %d clouds" % (pts.shape[0], len(frags)))
    kinds = ("pipeline", "cell", "morton")
    if len(sys.argv) > 1:          # e.g. `gemm32`: the pipeline's order only, KPConv from 32 channels up on the aggregation + GEMM path
        kinds = ("pipeline",)
        if sys.argv[1] == "gemm32":
            ops._GEMM_PATH_MIN_CIN = 32
            ops._GEMM_DX_AGG_MIN_COUT = 32
    for kind in kinds:
        o = order(kind)
        p = torch.from_numpy(pts[o]).to(DEV)
        ln = torch.from_numpy(lens).to(DEV)
        nb = dl.batch_neighbors_kpconv(p, p, ln, ln, r0, 42)
        # coarse level + pooling / upsampling tables
        p1, l1 = dl.batch_grid_subsampling_kpconv(p, ln, sampleDl=2 * dl0)
        if kind != "pipeline":      # the coarse level sorted the same way
            off, oo = 0, []
            for n in l1.cpu().numpy():
                f = p1[off:off + n].cpu().numpy()
                edge = 2 * r0 if kind == "cell" else 2 * dl0
                c = np.floor((f - f.min(0)) / edge).astype(np.int64)
                if kind == "cell":
                    key = (c[:, 0] * 73856093) ^ (c[:, 1] * 19349663) ^ (c[:, 2] * 83492791)
                    oo.append(np.argsort(key & 0xffff, kind="stable") + off)
                else:
                    oo.append(np.argsort(morton(c[:, 0], c[:, 1], c[:, 2]), kind="stable") + off)
                off += n
            p1 = p1[torch.from_numpy(np.concatenate(oo)).to(DEV)].contiguous()
        pool = dl.batch_neighbors_kpconv(p1, p, l1, ln, r0, 40)
        nb1 = dl.batch_neighbors_kpconv(p1, p1, l1, l1, 2 * r0, 40)
        N, N1 = p.shape[0], p1.shape[0]
        res = {}
        torch.manual_seed(0)
        for (name, q, s, tab, cin, cout, rad) in [("kpconv 32>32 L0", p, p, nb, 32, 32, r0), ("kpconv 32>32 L0>1", p1, p, pool, 32, 32, r0),
                                                  ("kpconv 64>64 L1", p1, p1, nb1, 64, 64, 2 * r0)]:
            conv = blocks.KPConv(15, 3, cin, cout, rad * 2.0 / 2.5, rad).to(DEV)
            x = torch.randn(s.shape[0], cin, device=DEV, requires_grad=True)
            bias = torch.zeros(cout, device=DEV, requires_grad=True)

            def fwd():
                return ops.kpconv_bias_act(q, s, tab, x, conv.kernel_points, conv.weights, conv.KP_extent, bias)
            y = fwd()
            g = torch.randn_like(y)

            def fb():
                yy = fwd()
                torch.autograd.backward(yy, g)
            res[name + " fwd"] = timed(fwd)
            res[name + " fwd+bwd"] = timed(fb)
        x128 = torch.randn(N, 128, device=DEV, requires_grad=True)
        res["max_pool 128 L0>1 fwd"] = timed(lambda: ops.max_pool(x128, pool))
        y = ops.max_pool(x128, pool)
        g = torch.randn_like(y)
        res["max_pool 128 L0>1 fwd+bwd"] = timed(lambda: torch.autograd.backward(ops.max_pool(x128, pool), g))
        f32 = torch.randn(N, 32, device=DEV, requires_grad=True)
        res["detector fwd"] = timed(lambda: ops.detection_scores(f32, nb, training=True))
        sc = ops.detection_scores(f32, nb, training=True)
        gs = torch.randn_like(sc)
        res["detector fwd+bwd"] = timed(lambda: torch.autograd.backward(ops.detection_scores(f32, nb, training=True), gs))
        print("== order: %s  (N0 = %d, N1 = %d, widths %d / %d / %d)" % (kind, N, N1, nb.shape[1], pool.shape[1], nb1.shape[1]))
        for k, v in res.items():
            print("   %-28s %8.1f us" % (k, v))


if __name__ == "__main__":
    t0 = time.time()
    main()
    print("# %.1f s" % (time.time() - t0))
