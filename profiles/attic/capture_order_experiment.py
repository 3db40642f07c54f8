"""Does the ORDER in which engines are captured change how fast their graphs replay?  (bench.py: two pairs in flight
ran at 361 pairs/s when the lanes were the first graphs of the process and at 303 when the one-pair engine had been
captured before them.)  Captures, in the order given on the command line, engines named  one | lanesA | lanesB  and then
measures each of them.
    python profiles/capture_order_experiment.py one lanesA lanesB"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import PairLanes, TrainStep

order = sys.argv[1:] or ["one", "lanesA"]
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
ts = TrainStep(cfg, [42] * 5, dev, seed=0)
sizes = [[int(t.shape[0]) for t in ts.build_batch(it)['points']] for it in items]
caps, ncorr = TrainStep.capacities_for(sizes, slack=1.0), int(items[0][4].shape[0])
engines = {}
for name in order:
    if name == "one":
        ts.enable_graph(caps, ncorr)
        ts.capture(items[0])
        engines[name] = ts
    else:
        pl = PairLanes(ts, 2)
        pl.enable_graph(caps, ncorr)
        pl.capture(items[0])
        engines[name] = pl
    torch.cuda.synchronize()


# D3F_DUMMY_STREAMS=K: K further streams that have each run one kernel (= K more hardware queues alive), idle afterwards
dummies = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("D3F_DUMMY_STREAMS", "0")))]
for s in dummies:
    with torch.cuda.stream(s):
        torch.zeros(8, device=dev).add_(1)
torch.cuda.synchronize()


def rate(eng, steps=20):
    P = eng.P if isinstance(eng, PairLanes) else 1

    def step(k):
        if P == 1:
            eng.step_graph(items[k % 4], items[(k + 1) % 4])
        else:
            eng.step_graph([items[(P * k + j) % 4] for j in range(P)], [items[(P * (k + 1) + j) % 4] for j in range(P)])
    for k in range(4):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(4 + k)
    torch.cuda.synchronize()
    return P * steps / (time.perf_counter() - t0)


for rep in range(2):
    print("dummy streams %d, capture order %s:  " % (len(dummies), order) + "   ".join("%s %.1f pairs/s" % (n, rate(engines[n])) for n in order))
