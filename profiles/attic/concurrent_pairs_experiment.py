"""How much of the GPU does ONE pair's training step leave idle?  P independent engines (own model copy, own graphs, own
training + side streams) are replayed concurrently on one MI355X and the aggregate pairs/s is compared with one engine
alone.  The kernels of the coarse levels launch 50-200 workgroups on 256 CUs and the fine-level KPConv kernels have no
saturated unit (profiles/r03_pmc_kpconv.txt), so two steps in flight should overlap.
    python profiles/concurrent_pairs_experiment.py [P=2] [steps=40]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import TrainStep

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
d3f.enable_tuned_gemms()
dev = torch.device("cuda:0")
cfg = cfgmod.default_config()


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


items = []
for i in range(4):
    it = synthetic.make_pair(2 * i + 1, 2 * i + 2, sub)
    items.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in it))


class _DS:
    config = cfg

    def __len__(self):
        return 1

    def __getitem__(self, i):
        return tuple(t.cpu().numpy() for t in items[0])


limits = [int(x) for x in dl.calibrate_neighbors(_DS(), cfg, samples_threshold=10 ** 9)]
engines, streams = [], []
for p in range(P):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        ts = TrainStep(cfg, limits, dev, seed=0)
        sizes = []
        for it in items:
            b = ts.build_batch(it)
            sizes.append([int(t.shape[0]) for t in b['points']])
        ts.enable_graph(TrainStep.capacities_for(sizes, slack=1.0), num_corr=int(items[0][4].shape[0]))
        ts.capture(items[0])
    torch.cuda.synchronize()
    engines.append(ts)
    streams.append(s)


def run(active, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        for p in active:
            with torch.cuda.stream(streams[p]):
                engines[p].step_graph(items[(k + p) % 4], items[(k + p + 1) % 4])
    torch.cuda.synchronize()
    return len(active) * n / (time.perf_counter() - t0)


run(list(range(P)), 5)
for rep in range(3):
    one = run([0], steps)
    many = run(list(range(P)), steps)
    print("one engine %.1f pairs/s   %d engines in flight %.1f pairs/s   ratio %.3f" % (one, P, many, many / one))
for ts in engines:
    ts.check_status()
