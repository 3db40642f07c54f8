"""Own GEMM and torch.mm on a few shapes, eager, for rocprofv3 --kernel-trace (kernel-only durations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import ops
d3f.enable_tuned_gemms()
dev = torch.device("cuda:0")
CASES = [(7961, 256, 256), (2053, 512, 512), (2053, 128, 256), (571, 1024, 1024), (154, 512, 1024), (64, 64, 64), (64, 64, 4096)]
for M, N, K in CASES:
    A = torch.randn((M, K), device=dev)
    B = torch.randn((N, K), device=dev)
    for _ in range(10):
        ops.gemm(A, B)
    torch.cuda.synchronize()
    for _ in range(10):
        torch.mm(A, B.t())
    torch.cuda.synchronize()
