"""csrc/gemm.hip against torch.mm (+ the separate epilogue launch it replaces) on the shapes of the network's
levels 1-4, hipGraph-replayed (what the training step does).  python profiles/gemm_microbench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import ops

d3f.enable_tuned_gemms()
dev = torch.device("cuda:0")
# (M, N, K, a_ks, b_ks, label)
N1, N2, N3, N4 = 7961, 2053, 571, 154
CASES = [
    (N1, 256, 128, 0, 0, "L1 fwd 128->256"), (N1, 64, 256, 0, 0, "L1 fwd 256->64"), (N1, 256, 256, 0, 0, "L1 dec 256->256"),
    (N2, 128, 256, 0, 0, "L2 fwd 256->128"), (N2, 512, 128, 0, 0, "L2 fwd 128->512"), (N2, 512, 512, 0, 0, "L2 dec 512->512"),
    (N2, 128, 1920, 0, 1, "L2 kpconv wf.W 1920->128"),
    (N3, 256, 512, 0, 0, "L3 fwd 512->256"), (N3, 1024, 256, 0, 0, "L3 fwd 256->1024"), (N3, 1024, 1024, 0, 0, "L3 dec 1024->1024"),
    (N3, 256, 3840, 0, 1, "L3 kpconv wf.W 3840->256"),
    (N4, 512, 1024, 0, 0, "L4 fwd 1024->512"), (N4, 2048, 512, 0, 0, "L4 fwd 512->2048"), (N4, 1024, 2048, 0, 0, "L4 dec 2048->1024"),
    (N4, 512, 7680, 0, 1, "L4 kpconv wf.W 7680->512"),
    (N3, 512, 256, 0, 1, "L3 dx  g[571,256].W[256,512]"), (N3, 256, 1024, 0, 1, "L3 dx g[571,1024].W[1024,256]"),
    (256, 512, N3, 1, 1, "L3 dW  g^T x [256x512] over 571"), (1024, 256, N3, 1, 1, "L3 dW [1024x256] over 571"),
    (2048, 512, N4, 1, 1, "L4 dW [2048x512] over 154"), (3840, 256, N3, 1, 1, "L3 kpconv dW [3840x256] over 571"),
    (64, 256, N1, 1, 1, "L1 dW [64x256] over 7961"),
]


def graph_time(fn, iters=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters // 10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters // 10 * 10)


print("%-36s %9s %9s %9s %8s" % ("case", "own_us", "mm_us", "mm+epi", "TFLOP/s"))
for M, N, K, aks, bks, label in CASES:
    A = torch.randn((K, M) if aks else (M, K), device=dev)
    B = torch.randn((K, N) if bks else (N, K), device=dev)
    bias = torch.randn(N, device=dev)
    own = lambda: ops.gemm(A, B, a_ks=bool(aks), b_ks=bool(bks), bias1=bias, slope=0.1)   # noqa: E731
    Am = A.t() if aks else A
    Bm = B if bks else B.t()
    mm = lambda: torch.mm(Am, Bm)   # noqa: E731
    mme = lambda: ops.bias_act(torch.mm(Am, Bm), bias, slope=0.1)   # noqa: E731
    t_own, t_mm, t_mme = graph_time(own), graph_time(mm), graph_time(mme)
    print("%-36s %9.2f %9.2f %9.2f %8.1f" % (label, t_own, t_mm, t_mme, 2.0 * M * N * K / t_own / 1e6))
