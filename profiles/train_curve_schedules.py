"""Round 6: what does training on 12 pairs per optimizer step do to convergence?

Full-width D3Feat network, the reference's hyper-parameters (SGD lr 0.01, momentum 0.98, weight decay 1e-6, circle +
detector loss; training_3DMatch.py:62-81), the SAME stream of synthetic S1-class pairs (NPAIRS distinct pairs, cycled)
and the same initial parameters in every arm:
    ref       1 pair per update (the reference's schedule, dataloader.py:73), lr 0.01
    fast      4 lanes x 3 stacked pairs = 12 pairs per update (mean gradient), lr 0.01
    fast xS   the same at lr 0.01 x S for the factors given on the command line (default 2 and sqrt(12) = 3.46)
Every arm runs UPDATES optimizer steps; printed per 50 updates: loss, descriptor loss, detector loss, accuracy (means over
the pairs of those updates) and the number of pairs seen, so the arms can be compared per update AND per pair seen.

    python profiles/train_curve_schedules.py [updates [npairs [scale ...]]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from d3feat_pytorch_amd import config as cfgmod, synthetic  # noqa: E402
from d3feat_pytorch_amd.datasets import dataloader as dl  # noqa: E402
from d3feat_pytorch_amd.train import PairLanes, TrainStep  # noqa: E402

updates = int(sys.argv[1]) if len(sys.argv) > 1 else 600
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 48
scales = [float(a) for a in sys.argv[3:]] or [2.0, 12 ** 0.5]
dev = torch.device("cuda:0")
cfg = cfgmod.default_config()
L, Q = 4, 3


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


t0 = time.time()
items = []
for i in range(npairs):
    it = synthetic.make_pair(2 * i + 1, 2 * i + 2, sub)
    items.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in it))
print("# %d distinct pairs (%.0f s to generate), %d updates per arm, lanes x stack = %d x %d" % (
    npairs, time.time() - t0, updates, L, Q))


def arm(name, pairs_per_update, lr):
    np.random.seed(0)
    torch.manual_seed(0)
    ts = TrainStep(cfg, [42] * 5, dev, seed=0)
    ts.opt.lr = lr
    sizes = [[int(t.shape[0]) for t in ts.build_batch(it)['points']] for it in items[:8]]
    sizes = [[max(s[l] for s in sizes) for l in range(5)]]
    log = []
    t1 = time.perf_counter()
    if pairs_per_update == 1:
        ts.enable_graph(TrainStep.capacities_for(sizes, slack=1.04), num_corr=int(items[0][4].shape[0]))
        ts.capture(items[0])
        ts.flat.data.copy_(init) if init is not None else None
        ts.opt.buf.zero_()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(updates):
            out = ts.step_graph(items[k % npairs], items[(k + 1) % npairs])
            log.append(torch.stack([o.reshape(()) for o in out]).clone())
        torch.cuda.synchronize()
        ts.check_status()
    else:
        lanes = PairLanes(ts, L, stack=Q)
        lanes.enable_graph(TrainStep.capacities_for([[Q * n for n in sizes[0]]], slack=1.04), num_corr=int(items[0][4].shape[0]))
        lanes.capture(tuple(items[j % npairs] for j in range(L * Q)))
        ts.flat.data.copy_(init)
        ts.opt.buf.zero_()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        P = L * Q
        for k in range(updates):
            cur = [items[(P * k + j) % npairs] for j in range(P)]
            nxt = [items[(P * (k + 1) + j) % npairs] for j in range(P)]
            outs = lanes.step_graph(cur, nxt)
            lanes.make_visible()
            # a stacked lane reports (sum of losses, desc [Q], det [Q], acc [Q])
            row = torch.stack([torch.stack([o[0] / Q, o[1].mean(), o[2].mean(), o[3].mean()]) for o in outs]).mean(0)
            log.append(row.clone())
            lanes.resync()
        lanes.synchronize()
        torch.cuda.synchronize()
        lanes.check_status()
    dt = time.perf_counter() - t1
    vals = torch.stack(log).cpu().numpy()
    print("\n## %s: %d pair(s) per update, lr %.4g -- %d updates = %d pairs in %.1f s (%.0f pairs/s incl. the per-step "
          "scalar reads), skipped updates %d" % (name, pairs_per_update, lr, updates, updates * pairs_per_update, dt,
                                                 updates * pairs_per_update / dt, int(ts.opt.skipped)))
    for b in range(0, updates, 50):
        m = vals[b:b + 50].mean(axis=0)
        print("updates %4d-%4d  pairs seen %6d  loss %.4f  desc %.4f  det %+.4f  acc %5.1f" % (
            b, min(updates, b + 50) - 1, min(updates, b + 50) * pairs_per_update, m[0], m[1], m[2], m[3]))
    return vals


# the same initial parameters in every arm
np.random.seed(0)
torch.manual_seed(0)
_probe = TrainStep(cfg, [42] * 5, dev, seed=0)
init = _probe.flat.data.clone()
del _probe
arm("ref", 1, cfg.lr)
arm("fast", L * Q, cfg.lr)
for s in scales:
    arm("fast x%.2f" % s, L * Q, cfg.lr * s)
