"""GPU time of the library GEMMs behind the unary blocks (forward x W^T and grad_x = g W) per network shape.
Run on the GPU box: python profiles/unary_gemm_microbench.py"""
import torch

SHAPES = [(38180, 64, 32), (38180, 32, 128), (38180, 64, 128), (38180, 128, 32), (38180, 128, 128), (38180, 384, 128),
          (7920, 128, 64), (7920, 64, 256), (7920, 128, 256), (7920, 256, 64), (7920, 768, 256),
          (2054, 256, 128), (2054, 128, 512), (2054, 256, 512), (2054, 1536, 512),
          (580, 512, 256), (580, 256, 1024), (580, 512, 1024), (580, 3072, 1024),
          (159, 1024, 512), (159, 512, 2048), (159, 1024, 2048)]
dev = torch.device("cuda:0")
busy = torch.randn(8192, 8192, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.mm(busy, busy)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("%-20s %9s %9s %12s" % ("N,Cin,Cout", "fwd_us", "dx_us", "stream_us@5TB/s"))
tf = td = 0.0
for n, cin, cout in SHAPES:
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(cout, cin, device=dev)
    g = torch.randn(n, cout, device=dev)
    f = timeit(lambda: torch.mm(x, w.t()))
    d = timeit(lambda: torch.mm(g, w))
    tf += f
    td += d
    print("%-20s %9.1f %9.1f %12.1f" % ("%d,%d,%d" % (n, cin, cout), f, d, 4 * (n * (cin + cout) + cin * cout) / 5e6))
print("sum fwd %.1f us, sum dx %.1f us" % (tf, td))

# fused row-streaming kernels of this library on the same shapes
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3feat_pytorch_amd import _native
L = _native.lib()
print("%-20s %9s %9s   (fused forward incl. bias+act epilogue / grad_x; 0 = shape not supported)" % ("N,Cin,Cout", "fwd_us", "dx_us"))
for n, cin, cout in SHAPES:
    if not L.d3f_linear_fused_supported(n, cin, cout):
        continue
    x = torch.randn(n, cin, device=dev); w = torch.randn(cout, cin, device=dev); g = torch.randn(n, cout, device=dev)
    b = torch.randn(cout, device=dev); y = torch.empty(n, cout, device=dev); gx = torch.empty(n, cin, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    f = timeit(lambda: L.d3f_linear_bias_act_forward(x.data_ptr(), w.data_ptr(), n, cin, cout, b.data_ptr(), None, b.data_ptr(), 0.1, y.data_ptr(), None, 0, st))
    d = timeit(lambda: L.d3f_linear_grad_input(g.data_ptr(), w.data_ptr(), n, cin, cout, None, gx.data_ptr(), st))
    print("%-20s %9.1f %9.1f" % ("%d,%d,%d" % (n, cin, cout), f, d))
