"""Times the reduction-parallel grad_W kernel against the library GEMM for the unary-block shapes of the full net.
Run on the GPU box: python profiles/linear_microbench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import d3feat_pytorch_amd as d3
from d3feat_pytorch_amd import ops, _native

SHAPES = [(38180, 64, 32), (38180, 32, 128), (38180, 64, 128), (38180, 128, 32), (7920, 128, 64), (7920, 64, 256),
          (7920, 128, 256), (7920, 256, 64), (2054, 256, 128), (2054, 128, 512), (2054, 256, 512), (2054, 512, 128),
          (580, 512, 256), (580, 256, 1024), (580, 512, 1024), (580, 1024, 256), (159, 1024, 512), (159, 512, 2048),
          (159, 1024, 2048), (159, 2048, 512), (580, 3072, 1024), (2054, 1536, 512), (7920, 768, 256),
          (38180, 384, 128), (38180, 128, 128), (38180, 128, 32)]


_BUSY = None


def timeit(fn, n=20):
    """GPU time per call: a long GEMM is queued first so the host runs ahead and launch latency is hidden."""
    global _BUSY
    if _BUSY is None:
        _BUSY = torch.randn(8192, 8192, device="cuda:0")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.mm(_BUSY, _BUSY)
    torch.mm(_BUSY, _BUSY)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    L = _native.lib()
    dev = torch.device("cuda:0")
    print("%-22s %10s %10s %8s" % ("N,Cin,Cout", "own_us", "lib_us", "GB/s"))
    for n, cin, cout in SHAPES:
        x = torch.randn(n, cin, device=dev)
        g = torch.randn(n, cout, device=dev)
        gw = torch.empty(cout, cin, device=dev)
        nb = L.d3f_linear_grad_weight_ws_bytes(n, cin, cout)
        ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        own = timeit(lambda: L.d3f_linear_grad_weight(x.data_ptr(), g.data_ptr(), n, cin, cout, gw.data_ptr(),
                                                      ws.data_ptr(), nb, st))
        lib = timeit(lambda: torch.mm(g.t(), x))
        err = (gw - torch.mm(g.t(), x)).abs().max().item()
        print("%-22s %10.1f %10.1f %8.0f  maxdiff %.2e" % ("%d,%d,%d" % (n, cin, cout), own, lib,
                                                          4 * (n * (cin + cout) + cin * cout) / own / 1e3, err))


if __name__ == "__main__":
    main()
