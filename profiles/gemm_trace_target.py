"""A few GEMM shapes, own kernel (heuristic plan) and torch.mm, N eager launches each -- for
    rocprofv3 --kernel-trace --stats -d out -- python profiles/gemm_trace_target.py
(per-kernel device durations without graph / launch effects)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import _native, ops

d3f.enable_tuned_gemms()
dev = torch.device("cuda:0")
lib = _native.lib()
CASES = [(640, 512, 256, 0, 1, (1, 0, 4, 1)), (192, 1024, 512, 0, 1, (1, 0, 8, 1)), (192, 512, 1024, 0, 0, (1, 2, 8, 1)),
         (2112, 512, 512, 0, 0, (2, 2, 2, 1)), (640, 1024, 1024, 0, 0, (2, 4, 4, 1)), (192, 512, 7680, 0, 1, (2, 0, 8, 4)),
         (192, 7680, 512, 0, 0, (2, 2, 1, 1)), (2048, 512, 192, 1, 1, (0, 0, 4, 1))]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for M, N, K, aks, bks, plan in CASES:
    A = torch.randn((K, M) if aks else (M, K), device=dev)
    B = torch.randn((K, N) if bks else (N, K), device=dev)
    bias = torch.randn(N, device=dev)
    Am = A.t() if aks else A
    Bm = B if bks else B.t()
    lib.d3f_debug_set_gemm_plan(*plan)
    for _ in range(n):
        ops.gemm(A, B, a_ks=bool(aks), b_ks=bool(bks), bias1=bias, slope=0.1)
    torch.cuda.synchronize()
    for _ in range(n):
        torch.mm(Am, Bm)
    torch.cuda.synchronize()
    print("done", M, N, K)
