"""End-to-end evidence that the pipelined training step learns: the full-width D3Feat network trained for 600 hipGraph
steps on 4 synthetic S1-class pairs (38k points each), reference hyper-parameters (SGD lr 0.01, momentum 0.98, circle +
detector loss).  Prints the mean loss / accuracy of every block of 50 steps.   python profiles/train_curve.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
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
ts.enable_graph(TrainStep.capacities_for(sizes, slack=1.0), num_corr=int(items[0][4].shape[0]))
ts.capture(items[0])
torch.cuda.synchronize()
log = []
t0 = time.perf_counter()
for k in range(steps):
    out = ts.step_graph(items[k % 4], items[(k + 1) % 4])
    log.append(torch.stack([o.reshape(()) for o in out]).clone())     # loss, desc, det, accuracy -- device scalars
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ts.check_status()
vals = torch.stack(log).cpu().numpy()
print("# %d steps in %.2f s (%.1f pairs/s incl. the per-step scalar copies); columns: loss desc det accuracy%%" % (steps, dt, steps / dt))
for b in range(0, steps, 50):
    m = vals[b:b + 50].mean(axis=0)
    print("steps %4d-%4d  loss %.4f  desc %.4f  det %+.4f  acc %5.1f" % (b, min(steps, b + 50) - 1, m[0], m[1], m[2], m[3]))
