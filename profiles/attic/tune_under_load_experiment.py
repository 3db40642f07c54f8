"""The shipped TunableOp table was tuned with every GEMM running ALONE on the GPU.  With four pairs in flight a GEMM
shares the chip with three other lanes' kernels: does tuning UNDER that load pick different (better co-running)
solutions?  Three lanes replay their network graphs from a background thread while the fourth lane's stream runs the
eager network step with tuning enabled, from an EMPTY table; the result is written as a TunableOp CSV.
    python profiles/tune_under_load_experiment.py out.csv [max_ms=10] [max_iter=20] [load=1]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import d3feat_pytorch_amd as d3f   # noqa: F401  (GPU_MAX_HW_QUEUES before the first HIP call)
from d3feat_pytorch_amd import config as cfgmod, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import PairLanes, TrainStep

out = sys.argv[1]
max_ms = int(sys.argv[2]) if len(sys.argv) > 2 else 10
max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 20
load = int(sys.argv[4]) if len(sys.argv) > 4 else 1
tunable = torch.cuda.tunable
tunable.enable(True)
tunable.tuning_enable(False)
tunable.set_filename(os.path.join("/tmp", "d3f_tune_under_load_%d.csv" % os.getpid()))
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
lanes = PairLanes(ts, 4)
lanes.enable_graph(TrainStep.capacities_for(sizes, slack=1.0), num_corr=int(items[0][4].shape[0]))
tunable.enable(False)          # (TrainStep.capture would tune the missing shapes in its warm-up: not yet)
lanes.capture(items[0])        # library-default picks
torch.cuda.synchronize()
tunable.enable(True)
stop = threading.Event()


def background():
    evs = [[None, None] for _ in range(3)]
    k = 0
    while not stop.is_set():
        for j, eng in enumerate(lanes.engines[:3]):
            with torch.cuda.stream(eng.stream):
                eng.g_net[0].replay()
                ev = torch.cuda.Event()
                ev.record(eng.stream)
            old, evs[j][k % 2] = evs[j][k % 2], ev
            if old is not None:
                old.synchronize()
        k += 1


th = threading.Thread(target=background, daemon=True)
if load:
    th.start()
    time.sleep(0.05)
eng = lanes.engines[3]
tunable.set_max_tuning_duration(max_ms)
tunable.set_max_tuning_iterations(max_iter)
t0 = time.perf_counter()
with torch.cuda.stream(eng.stream):
    tunable.tuning_enable(True)
    for _ in range(2):
        eng._net_step(eng.sets[0])
    eng.stream.synchronize()
    tunable.tuning_enable(False)
dt = time.perf_counter() - t0
stop.set()
if load:
    th.join()
torch.cuda.synchronize()
res = tunable.get_results()
with open(out, "w") as f:
    for name, val in tunable.get_validators():
        f.write("Validator,%s,%s\n" % (name, val))
    for r in res:
        f.write(",".join(str(x) for x in r) + "\n")
print("tuned %d shapes in %.1f s %s -> %s" % (len(res), dt, "under three lanes of load" if load else "alone", out))
