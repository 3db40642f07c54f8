"""GPU time and achieved bandwidth of the block epilogue kernels per network shape.
Run on the GPU box: python profiles/epilogue_microbench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from d3feat_pytorch_amd import _native

L = _native.lib()
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


print("%-14s %9s %9s %9s %9s" % ("N,C", "fwd_us", "fwd_GB/s", "bwd_us", "bwd_GB/s"))
for n, c in ((38272, 32), (38272, 64), (38272, 128), (8000, 64), (8000, 256), (2112, 128), (2112, 512), (640, 256),
             (640, 1024), (192, 512), (192, 2048)):
    x = torch.randn(n, c, device=dev); b = torch.randn(c, device=dev); out = torch.empty_like(x)
    go = torch.randn(n, c, device=dev); gx = torch.empty_like(x); gb = torch.zeros(2, c, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    f = timeit(lambda: L.d3f_bias_act_forward(x.data_ptr(), b.data_ptr(), None, b.data_ptr(), 0.1, n, c, out.data_ptr(),
                                              None, 0, None, None, 0, 0, st))
    nb = L.d3f_bias_act_backward_ws_bytes(n, c)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    w = timeit(lambda: L.d3f_bias_act_backward(go.data_ptr(), out.data_ptr(), 0.1, n, c, gx.data_ptr(), gb.data_ptr(),
                                               gb[1].data_ptr(), 1, None, ws.data_ptr(), nb, st))
    w1 = timeit(lambda: L.d3f_bias_act_backward(go.data_ptr(), out.data_ptr(), 0.1, n, c, gx.data_ptr(), gb.data_ptr(),
                                                gb[1].data_ptr(), 1, None, None, 0, st))
    print("%-14s %9.1f %9.0f %9.1f %9.0f   (one-pass atomics: %.1f us)" % (
        "%d,%d" % (n, c), f, 8.0 * n * c / f / 1e3, w, 12.0 * n * c / w / 1e3, w1))
