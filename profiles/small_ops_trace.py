"""Which Python lines of this package issue the small ATen launches (fill / zero / copy / add / mul ...) of one network
step?  TorchDispatchMode + traceback; run on the GPU box: python profiles/small_ops_trace.py"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import TrainStep

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
ts.enable_graph(TrainStep.capacities_for(sizes, slack=1.0), num_corr=int(item[4].shape[0]))
st = ts.sets[0]
ts._load_inputs(st, item)
ts._build_set(st)
for _ in range(2):
    ts._net_step(st)
torch.cuda.synchronize()
WATCH = ("fill_", "zero_", "copy_", "add", "add_", "mul", "clone", "zeros", "gt", "_to_copy", "cat", "sum", "div",
         "contiguous", "_foreach_copy_", "ones_like", "zeros_like", "index", "select_backward")
seen = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            big = [a for a in args if isinstance(a, torch.Tensor) and a.is_cuda]
            if big:
                frames = [f for f in traceback.extract_stack() if "/d3feat.pytorch_amd/" in f.filename]
                where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in frames[-3:][::-1])
                seen[(name, where or "(autograd engine / no package frame)", tuple(big[0].shape))] += 1
        return func(*args, **(kwargs or {}))


with Log():
    ts._net_step(st)
torch.cuda.synchronize()
for (name, where, shape), c in sorted(seen.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print("%-16s x%-3d %-22s %s" % (name, c, shape, where))
names = [n for n, p in ts.model.named_parameters() if p.requires_grad and p.grad is None]
print("parameters without a gradient after backward:", names)
