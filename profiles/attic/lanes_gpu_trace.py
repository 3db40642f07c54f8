"""GPU-side picture of a PairLanes step from HIP events (rocprofv3's kernel trace serialises dispatches and cannot show
it): when each lane's network graph starts and ends, when the joint SGD step ends, how long the GPU is down to fewer
lanes at the ends of a step.
    python profiles/lanes_gpu_trace.py [lanes=3] [steps=8]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import PairLanes, TrainStep

P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d3f.enable_tuned_gemms()
dev = torch.device("cuda:0")
cfg = cfgmod.default_config()


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


items = []
for i in range(6):
    it = synthetic.make_pair(2 * i + 1, 2 * i + 2, sub)
    items.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in it))
ts = TrainStep(cfg, [42] * 5, dev, seed=0)
sizes = [[int(t.shape[0]) for t in ts.build_batch(it)['points']] for it in items]
lanes = PairLanes(ts, P)
lanes.enable_graph(TrainStep.capacities_for(sizes, slack=1.0), num_corr=int(items[0][4].shape[0]))
lanes.capture(items[0])
torch.cuda.synchronize()

marks = []      # (step, lane, start event, end event)
state = {"step": 0}
for eng in lanes.engines:
    for gi, g in enumerate(eng.g_net):
        def timed(g=g, eng=eng):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(torch.cuda.current_stream(dev))
            g.replay()
            b.record(torch.cuda.current_stream(dev))
            marks.append((state["step"], eng.lane, a, b))

        class _G:
            replay = staticmethod(timed)
        eng.g_net[gi] = _G()
joins = []


def step(k):
    state["step"] = k
    cur = [items[(P * k + j) % 6] for j in range(P)]
    nxt = [items[(P * (k + 1) + j) % 6] for j in range(P)]
    lanes.step_graph(cur, nxt)
    e = torch.cuda.Event(enable_timing=True)
    e.record(lanes.engines[0].stream)       # behind the joint SGD step
    joins.append((k, e))


for k in range(4):
    step(k)
torch.cuda.synchronize()
del marks[:], joins[:]
origin = torch.cuda.Event(enable_timing=True)
origin.record(lanes.engines[0].stream)
for k in range(steps):
    step(4 + k)
torch.cuda.synchronize()
print("times in ms from the first step's origin; per step: lane start..end, join end, [GPU on fewer than %d lanes]" % P)
prev_join = 0.0
for k, je in joins:
    rows = sorted((l, origin.elapsed_time(a), origin.elapsed_time(b)) for s, l, a, b in marks if s == k)
    jend = origin.elapsed_time(je)
    first_start, last_start = min(r[1] for r in rows), max(r[1] for r in rows)
    first_end, last_end = min(r[2] for r in rows), max(r[2] for r in rows)
    print("step %2d: " % k + "  ".join("L%d %.3f..%.3f (%.3f)" % (l, a, b, b - a) for l, a, b in rows) +
          "  join end %.3f   [ramp-up %.3f, tail %.3f, join+gap %.3f; step %.3f]" % (
              jend, last_start - first_start, last_end - first_end, jend - last_end + (first_start - prev_join), jend - prev_join))
    prev_join = jend
