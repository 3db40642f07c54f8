"""Which library-GEMM shapes of the current training / inference steps are NOT in the shipped TunableOp table?  Runs the
eager step with tuning enabled for missing shapes only and writes the merged table:
    python profiles/tune_missing.py gpurun_out/tunableop_merged.csv"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import TrainStep

out = sys.argv[1]
assert d3f.enable_tuned_gemms(tune_missing=True)
before = len(torch.cuda.tunable.get_results())
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
st = ts.sets[0]
ts._load_inputs(st, items[0])
ts._build_set(st)
for _ in range(2):
    ts._net_step(st)            # eager network step on the capacity shapes (what the graphs replay)
torch.cuda.synchronize()
res = torch.cuda.tunable.get_results()
print("entries before", before, "after", len(res))
for r in res[before:]:
    print("  new:", r)
with open(out, "w") as fh:          # the table format PyTorch reads back: validators, then one line per tuned shape
    for k, v in torch.cuda.tunable.get_validators():
        fh.write("Validator,%s,%s\n" % (k, v))
    for r in res:
        fh.write("%s,%s,%s,%s\n" % (r[0], r[1], r[2], r[3]))
print("wrote", out, len(res), "entries")
