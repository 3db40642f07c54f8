"""Only the network step (forward, loss, backward, optimizer) of one pair -- or of a stack of Q pairs (second argument)
-- eagerly, N times: for a rocprofv3 kernel-count/-time table or --pmc counter passes of the TRAINING STREAM alone (the
pyramid runs once, before):
    rocprofv3 --kernel-trace --stats -d gpurun_out/net -o net -- python profiles/net_step_only.py 20 [Q]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import TrainStep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
d3f.enable_tuned_gemms()
dev = torch.device("cuda:0")
cfg = cfgmod.default_config()


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


it = synthetic.make_pair(1, 2, sub)
item = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in it)
ts = TrainStep(cfg, [42] * 5, dev, seed=0)
b = ts.build_batch(item)
sizes = [[int(t.shape[0]) for t in b['points']]]
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if Q > 1:
    others = [tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in synthetic.make_pair(2 * q + 1, 2 * q + 2, sub))
              for q in range(1, Q)]
    item = (item,) + tuple(others)
    sizes = [[sum(int(ts.build_batch(it)['points'][l].shape[0]) for it in item) for l in range(5)]]
ts.enable_graph(TrainStep.capacities_for(sizes, slack=1.0), num_corr=int(item[0][4].shape[0] if Q > 1 else item[4].shape[0]),
                stack=Q)
ts._use_scale()
st = ts.sets[0]
ts._load_inputs(st, item)
ts._build_set(st)
for _ in range(n):
    ts._net_step(st)
torch.cuda.synchronize()
print("steps:", n, "pairs per step:", Q)
