"""Phase ablations of csrc/gemm.hip (d3f_gemm_debug_set_flags): which of global loads / staging / fragment reads / MFMA
a shape's time is made of.  hipGraph-replayed like the training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from d3feat_pytorch_amd import ops, _native
dev = torch.device("cuda:0")
L = _native.lib()


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


FLAGS = [(0, "full"), (1, "no global loads"), (3, "no loads, no staging"), (4, "no fragment reads"), (8, "no MFMA"),
         (12, "no frag reads, no MFMA"), (15, "nothing but the loop"), (15 + 16, "loop, no epilogue"), (15 + 32, "loop, no prologue"), (63, "bare"), (16, "full minus epilogue")]
for M, N, K in [(7961, 256, 256), (571, 1024, 1024), (2053, 512, 512)]:
    A = torch.randn((M, K), device=dev)
    B = torch.randn((N, K), device=dev)
    row = []
    for f, name in FLAGS:
        L.d3f_gemm_debug_set_flags(f)
        row.append("%s %.1f" % (name, graph_time(lambda: ops.gemm(A, B))))
    L.d3f_gemm_debug_set_flags(0)
    print("%dx%dx%d: " % (M, N, K) + " | ".join(row))
