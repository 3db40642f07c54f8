"""Where does a pipelined graph step spend its time?  Host time of each replay call, GPU time of each graph alone,
and the overlapped step.  Run on the GPU box: python profiles/graph_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from d3feat_pytorch_amd import _native, config as cfgmod, ops, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl
from d3feat_pytorch_amd.train import TrainStep


def main():
    dev = torch.device("cuda:0")
    _native.lib()
    cfg = cfgmod.default_config()

    def gpu_subsample(points, lengths, dlen):
        p, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(points).to(dev), torch.as_tensor(lengths).to(dev), sampleDl=dlen)
        return p.cpu().numpy(), b.cpu().numpy()
    items = []
    for i in range(2):
        it = synthetic.make_pair(2 * i + 1, 2 * i + 2, gpu_subsample)
        items.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in it))
    limits = [42] * 5
    ts = TrainStep(cfg, limits, dev, seed=0)
    sizes = []
    for it in items:
        b = ts.build_batch(it)
        sizes.append([int(t.shape[0]) for t in b['points']])
    ts.enable_graph(TrainStep.capacities_for(sizes), num_corr=int(items[0][4].shape[0]))
    ts.capture(items[0])
    torch.cuda.synchronize()

    def gpu_time(fn, n=10):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
    print("net graph alone   : host %.2f ms/launch, wall %.2f ms" % gpu_time(lambda: ts.g_net[0].replay()))
    print("pyramid graph alone: host %.2f ms/launch, wall %.2f ms" % gpu_time(lambda: ts.g_pyr[1].replay()))
    k = [0]

    def step():
        ts.step_graph(items[k[0] % 2], items[(k[0] + 1) % 2]); k[0] += 1
    print("pipelined step    : host %.2f ms, wall %.2f ms" % gpu_time(step, 20))
    side = torch.cuda.Stream()

    def both():
        with torch.cuda.stream(side):
            ts.g_pyr[1].replay()
        ts.g_net[0].replay()
    print("net || pyramid (no events): host %.2f ms, wall %.2f ms" % gpu_time(both, 20))


if __name__ == "__main__":
    main()
