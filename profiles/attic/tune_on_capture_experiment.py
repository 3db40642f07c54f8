"""Capacities other than the ones the shipped TunableOp table was tuned for (a trainer's classes, head-room): how much do
the library GEMMs lose on the library's default picks, how long does tuning the missing shapes during the capture warm-up
take, and what does it give back?
    python profiles/tune_on_capture_experiment.py [slack=1.10] [max_ms=10] [max_iter=20]"""
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

slack = float(sys.argv[1]) if len(sys.argv) > 1 else 1.10
max_ms = int(sys.argv[2]) if len(sys.argv) > 2 else 10
max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 20
assert d3f.enable_tuned_gemms()
dev = torch.device("cuda:0")
cfg = cfgmod.default_config()


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


items = []
for i in range(4):
    it = synthetic.make_pair(2 * i + 1, 2 * i + 2, sub)
    items.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in it))


def rate(ts, steps=20):
    for k in range(4):
        ts.step_graph(items[k % 4], items[(k + 1) % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        ts.step_graph(items[k % 4], items[(k + 1) % 4])
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


ts = TrainStep(cfg, [42] * 5, dev, seed=0)
sizes = [[int(t.shape[0]) for t in ts.build_batch(it)['points']] for it in items]
caps = TrainStep.capacities_for(sizes, slack=slack)
ts.enable_graph(caps, num_corr=int(items[0][4].shape[0]))
ts.capture(items[0])
print("capacities %s: library-default GEMMs for the shapes missing from the table: %.1f pairs/s" % (caps, rate(ts)))
n0 = len(torch.cuda.tunable.get_results())
torch.cuda.tunable.set_max_tuning_duration(max_ms)
torch.cuda.tunable.set_max_tuning_iterations(max_iter)
torch.cuda.tunable.tuning_enable(True)
t0 = time.perf_counter()
eng = ts.clone_for_capacities(caps, num_corr=int(items[0][4].shape[0]))
eng.capture(items[0])          # the warm-up steps of the capture tune what is missing
torch.cuda.synchronize()
t1 = time.perf_counter()
torch.cuda.tunable.tuning_enable(False)
print("tuned %d shapes during the capture warm-up in %.1f s (max %d ms / %d iterations per solution): %.1f pairs/s" % (
    len(torch.cuda.tunable.get_results()) - n0, t1 - t0, max_ms, max_iter, rate(eng)))
