"""Dense mutual-NN matching alone: python profiles/matching_microbench.py [N] [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d3feat_pytorch_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 19100
c = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
a = torch.nn.functional.normalize(torch.randn(n, c, device=dev, generator=g), dim=1)
b = torch.nn.functional.normalize(torch.randn(n - 29, c, device=dev, generator=g), dim=1)
for _ in range(2):
    ops.mutual_nn(a, b)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.mutual_nn(a, b)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
# one sweep over the S x T tiles since round 4: 2 N^2 C flop
print("N=%d C=%d: %.3f ms, %.1f TFLOP/s" % (n, c, ms, 2.0 * n * (n - 29) * c / ms / 1e9))
