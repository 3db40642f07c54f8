"""Where does the HOST spend a PairLanes step?  Wall-clock of every host-side phase (graph launches, event waits) over a
few steps -- a graph launch that blocks the host while the other lane waits to be launched serialises the lanes.
    python profiles/lanes_host_trace.py [lanes=2] [steps=12]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import d3feat_pytorch_amd as d3f   # (sets GPU_MAX_HW_QUEUES before the first HIP call)
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import PairLanes, TrainStep

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
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
# what differs between this script and bench.py (D3F_TRACE_PRE): "capture" = the one-pair engine is captured first (as
# bench.py does for its one_pair_in_flight leg), "streamsN" = N streams are taken from torch's pool first
pre = os.environ.get("D3F_TRACE_PRE", "")
held = []
keep = (ts.flat.data.clone(), ts.opt.buf.clone())
if "capture" in pre:
    ts.enable_graph(TrainStep.capacities_for(sizes, slack=1.0), num_corr=int(items[0][4].shape[0]))
    ts.capture(items[0])
if "eager" in pre:         # three optimizer steps, no graphs
    for k in range(3):
        ts.step(items[k % 4])
    ts._pending = None
if "restore" in pre:       # ... and the parameters put back
    torch.cuda.synchronize()
    ts.flat.data.copy_(keep[0])
    ts.opt.buf.copy_(keep[1])
if "drop" in pre:          # ... and the one-pair engine's graphs and buffer sets released
    for name in ('sets', 'g_net', 'g_net_b', 'g_pyr', '_graph_out', '_graph_dist'):
        ts.__dict__.pop(name, None)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
if "streams" in pre:
    held = [torch.cuda.Stream(device=dev) for _ in range(int(pre.split("streams")[1].split(",")[0]))]
lanes = PairLanes(ts, P)
lanes.enable_graph(TrainStep.capacities_for(sizes, slack=1.0), num_corr=int(items[0][4].shape[0]))
lanes.capture(items[0])
torch.cuda.synchronize()

marks = []
for eng in lanes.engines:          # wrap the graph replays of every lane
    for gi, g in enumerate(eng.g_net):
        def timed(g=g, lane=eng.lane):
            t0 = time.perf_counter()
            g.replay()
            marks.append(("replay lane %d" % lane, t0, time.perf_counter()))

        class _G:
            replay = staticmethod(timed)
        eng.g_net[gi] = _G()


def step(k):
    t0 = time.perf_counter()
    cur = [items[(P * k + j) % 4] for j in range(P)]
    nxt = [items[(P * (k + 1) + j) % 4] for j in range(P)]
    lanes.step_graph(cur, nxt)
    marks.append(("step", t0, time.perf_counter()))


for k in range(4):
    step(k)
torch.cuda.synchronize()
del marks[:]
T0 = time.perf_counter()
for k in range(steps):
    step(4 + k)
torch.cuda.synchronize()
T1 = time.perf_counter()
print("%d steps of %d pairs: %.3f ms per step, %.1f pairs/s   [pre=%r, lane streams %s]" % (
    steps, P, (T1 - T0) / steps * 1e3, P * steps / (T1 - T0), pre,
    [(hex(e.stream.cuda_stream), hex(e._side.cuda_stream)) for e in lanes.engines]))
for name, a, b in marks[:4 * (P + 1)]:
    print("%-16s start %8.3f ms   took %8.3f ms" % (name, (a - T0) * 1e3, (b - a) * 1e3))
