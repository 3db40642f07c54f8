"""Which ATen ops (PyTorch glue around the library calls) still run inside one training step, and how often?
Run on the GPU box: python profiles/aten_ops_in_step.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import TrainStep

import d3feat_pytorch_amd as d3f
if "--tuned" in sys.argv:
    print("tuned GEMM table loaded:", d3f.enable_tuned_gemms())
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
STACKS = "--stacks" in sys.argv
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=STACKS) as prof:
    ts._net_step(st)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total)
        for e in prof.key_averages()]
rows = [r for r in rows if r[0].startswith("aten::") and r[2] > 0]
rows.sort(key=lambda r: -r[2])
print("%-40s %6s %12s" % ("op", "calls", "device_us"))
for k, c, t in rows[:40]:
    print("%-40s %6d %12.1f" % (k, c, t))
print("total aten device time: %.1f us in %d op calls" % (sum(r[2] for r in rows), sum(r[1] for r in rows)))

if STACKS:   # where do the small ATen launches come from?  (frames inside this repository only)
    want = ("aten::add_", "aten::add", "aten::copy_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::clone",
            "aten::mul", "aten::select_backward", "aten::gt", "aten::_to_copy")
    seen = {}
    for e in prof.key_averages(group_by_stack_n=12):
        dt = e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
        if e.key in want and dt > 0:
            frames = [f for f in e.stack if "d3feat" in f or "train.py" in f]
            where = " <- ".join(f.split("/")[-1].strip() for f in frames[:3]) or "(autograd engine)"
            k = (e.key, where)
            seen[k] = (seen.get(k, (0, 0))[0] + e.count, seen.get(k, (0, 0))[1] + dt)
    for (k, where), (c, t) in sorted(seen.items(), key=lambda kv: -kv[1][1]):
        print("%-22s %3d %8.1f us  %s" % (k, c, t, where))
