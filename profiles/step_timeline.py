"""Network step of one pair (or of a stack of Q pairs: third argument) as a hipGraph replay, N times, for a per-dispatch
timeline:
    rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python profiles/step_timeline.py 12 [slack [stack]]
    python profiles/timeline_rocpd.py gpurun_out/tl/*/*.db   -> kernels of the last replay in order, duration + gap"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import TrainStep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
# A/B switches of THIS script (the package reads no environment for them): PROF_GROUP=0 launches every weight gradient
# where autograd reaches it (rounds 1-5), PROF_SMALL=0 keeps the few-row weight gradients on the library GEMM
from d3feat_pytorch_amd import ops as _ops
_ops.GROUP_WEIGHT_GRADS = os.environ.get("PROF_GROUP", "1") != "0"
_ops.GROUP_SMALL_ROW_GRADS = os.environ.get("PROF_SMALL", "1") != "0"
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
slack = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0     # capacity head-room (trainer default 1.10)
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 1             # pairs stacked into the one network graph
if Q > 1:
    others = [tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in synthetic.make_pair(2 * q + 1, 2 * q + 2, sub))
              for q in range(1, Q)]
    stack = (item,) + tuple(others)
    sizes = [[sum(int(t.shape[0]) for t in [ts.build_batch(it)['points'][l] for it in stack]) for l in range(5)]]
    ts.enable_graph(TrainStep.capacities_for(sizes, slack=slack), num_corr=int(item[4].shape[0]), stack=Q)
    ts.capture(stack)
else:
    ts.enable_graph(TrainStep.capacities_for(sizes, slack=slack), num_corr=int(item[4].shape[0]))
    ts.capture(item)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(2000000)   # marker kernel: the timeline tool keeps what follows it
e0.record()
for _ in range(n):
    ts.g_net[0].replay()
e1.record()
torch.cuda.synchronize()
print("replays:", n, "pairs per replay:", Q, "ms/replay: %.3f" % (e0.elapsed_time(e1) / n))
