"""The reduction-parallel A^T B kernel (weight gradients) on the shapes it runs on inside one training step, replayed
as a hipGraph: python profiles/atb_microbench.py   (tunables: D3F_ATB_U, D3F_ATB_WGS, D3F_ATB_FAN)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from d3feat_pytorch_amd import _native

# (rows, cout(M), cin(N)): KPConv dW = wf^T g as (M = K*Cin, N = Cout); unary dW of levels 0 / 1
SHAPES = [(38180, 480, 32), (7920, 960, 64), (7920, 480, 64), (38180, 64, 32), (38180, 32, 64), (38180, 128, 64),
          (7920, 128, 64), (7920, 64, 256), (7920, 256, 128), (38180, 32, 32), (38180, 16, 32)]
if len(sys.argv) > 1:   # Q pairs stacked (round 4): the rows of every level times Q, plus the level-2 KPConv gradients
    Q = int(sys.argv[1])
    SHAPES = [(Q * n, a, b) for n, a, b in SHAPES] + [(Q * 2053, 1920, 128), (Q * 2053, 960, 64), (Q * 2053, 128, 512),
                                                       (Q * 2053, 512, 128)]
L = _native.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
tot = 0.0
for n, cout, cin in SHAPES:
    x = torch.randn(n, cin, device=dev)
    g = torch.randn(n, cout, device=dev)
    gw = torch.empty(cout, cin, device=dev)
    nb = L.d3f_linear_grad_weight_ws_bytes(n, cin, cout)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    fn = lambda: L.d3f_linear_grad_weight(x.data_ptr(), g.data_ptr(), n, cin, cout, gw.data_ptr(), ws.data_ptr(), nb,
                                          torch.cuda.current_stream().cuda_stream)   # noqa: E731
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    err = (gw - torch.mm(g.t(), x)).abs().max().item() / max(1.0, torch.mm(g.t(), x).abs().max().item())
    tot += us
    print("%6d x %4d x %4d  %7.2f us  %6.0f GB/s  relerr %.1e" % (n, cout, cin, us, 4.0 * n * (cin + cout) / us / 1e3, err))
print("sum %.1f us" % tot)
