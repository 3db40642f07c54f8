"""Two GEMM shapes on the own kernel (forced plan) for rocprofv3 --pmc passes:
    rocprofv3 --kernel-trace --pmc <counters> -d out -- python profiles/gemm_pmc_target.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from d3feat_pytorch_amd import _native, ops

dev = torch.device("cuda:0")
lib = _native.lib()
CASES = [(640, 1024, 1024, 0, 0, (2, 4, 4, 1)), (640, 1024, 1024, 0, 0, (1, 4, 8, 1)), (192, 512, 1024, 0, 0, (1, 2, 8, 1)),
         (640, 256, 3840, 0, 1, (1, 0, 8, 1))]
for M, N, K, aks, bks, plan in CASES:
    A = torch.randn((K, M) if aks else (M, K), device=dev)
    B = torch.randn((K, N) if bks else (N, K), device=dev)
    bias = torch.randn(N, device=dev)
    lib.d3f_debug_set_gemm_plan(*plan)
    for _ in range(10):
        ops.gemm(A, B, a_ks=bool(aks), b_ks=bool(bks), bias1=bias, slope=0.1)
    torch.cuda.synchronize()
