"""What does a dependent kernel boundary cost inside a replayed hipGraph on this box -- the number that decides
between "fuse stages into fewer launches" and "one persistent kernel with grid barriers" for the few-point levels.
A chain of N dependent tiny kernels (each reads the previous one's output) is captured and replayed; time / N is the
all-in price of one small dependent launch.  Three bodies: the library's own epilogue kernel on a [64, 64] matrix, on a
[192, 512] matrix (a level-4 activation), and on [640, 1024] (level 3).
    python profiles/launch_floor.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d3feat_pytorch_amd import ops

dev = torch.device("cuda:0")


def chain_time(rows, cols, n=200, replays=20):
    x = torch.randn(rows, cols, device=dev)
    b = torch.zeros(cols, device=dev)

    def run():
        y = x
        for _ in range(n):
            y = ops.bias_act(y, b, slope=0.5)
        return y
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * replays)


for rows, cols in ((64, 64), (192, 512), (640, 1024), (2112, 512)):
    print("chain of dependent bias_act on [%d, %d]: %.2f us per launch (graph replay)" % (rows, cols, chain_time(rows, cols)))
