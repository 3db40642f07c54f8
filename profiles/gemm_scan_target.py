"""Own GEMM over a sweep of reduction lengths / row counts, eager, for rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from d3feat_pytorch_amd import ops
dev = torch.device("cuda:0")
CASES = [(m, 1024, 4096) for m in (64, 128, 256, 571, 1142, 2284, 4568)] + [(571, n, 4096) for n in (64, 256, 512)]
for M, N, K in CASES:
    A = torch.randn((M, K), device=dev)
    B = torch.randn((N, K), device=dev)
    for _ in range(6):
        ops.gemm(A, B)
    torch.cuda.synchronize()
    for _ in range(6):
        torch.mm(A, B.t())
    torch.cuda.synchronize()
