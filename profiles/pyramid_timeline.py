"""The pyramid graph (4 voxel levels + 13 searches + reverse tables) of one pair, replayed alone, for a per-dispatch
timeline:  rocprofv3 --kernel-trace -d gpurun_out/pt -o pt -- python profiles/pyramid_timeline.py 12 [reverse=1 [Q]]
(Q pairs stacked into one pyramid: third argument)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import TrainStep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
TrainStep.reverse_tables = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
dev = torch.device("cuda:0")
cfg = cfgmod.default_config()


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


it = synthetic.make_pair(1, 2, sub)
item = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in it)
ts = TrainStep(cfg, [42] * 5, dev, seed=0)
b = ts.build_batch(item)
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if Q > 1:
    others = [tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in synthetic.make_pair(2 * q + 1, 2 * q + 2, sub))
              for q in range(1, Q)]
    stack = (item,) + tuple(others)
    sizes = [[sum(int(ts.build_batch(x)['points'][l].shape[0]) for x in stack) for l in range(5)]]
    ts.enable_graph(TrainStep.capacities_for(sizes, slack=1.0), num_corr=int(item[4].shape[0]), stack=Q)
    ts.capture(stack)
else:
    ts.enable_graph(TrainStep.capacities_for([[int(t.shape[0]) for t in b['points']]], slack=1.0),
                    num_corr=int(item[4].shape[0]))
    ts.capture(item)
torch.cuda.synchronize()
torch.cuda._sleep(2000000)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ts.g_pyr[0].replay()
e1.record()
torch.cuda.synchronize()
print("replays:", n, "ms/replay: %.3f" % (e0.elapsed_time(e1) / n))
