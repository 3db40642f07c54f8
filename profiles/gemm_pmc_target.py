"""One GEMM shape, own kernel, repeated -- target for rocprofv3 --pmc.  args: M N K a_ks b_ks reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from d3feat_pytorch_amd import ops
M, N, K, aks, bks, reps = (int(v) for v in sys.argv[1:7])
dev = torch.device("cuda:0")
A = torch.randn((K, M) if aks else (M, K), device=dev)
B = torch.randn((K, N) if bks else (N, K), device=dev)
bias = torch.randn(N, device=dev)
for _ in range(reps):
    ops.gemm(A, B, a_ks=bool(aks), b_ks=bool(bks), bias1=bias, slope=0.1)
torch.cuda.synchronize()
