"""Backward of one unary block (models/blocks.py:481-515), hipGraph-replayed:
   library : bias_act_backward (mask + bias sums) -> mm (grad_x) -> mm (grad_W)
   own     : d3f_gemm with the mask in the operand staging (grad_x) + d3f_gemm with mask and row sums (grad_W, grad_b)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import ops, _native
d3f.enable_tuned_gemms()
dev = torch.device("cuda:0")
L = _native.lib()
N1, N2, N3, N4 = 7961, 2053, 571, 154
CASES = [(N1, 128, 64), (N1, 64, 256), (N1, 128, 256), (N1, 256, 64), (N2, 256, 128), (N2, 128, 512), (N2, 512, 128), (N2, 256, 512),
         (N3, 512, 256), (N3, 256, 1024), (N3, 1024, 256), (N3, 512, 1024), (N4, 1024, 512), (N4, 512, 2048), (N4, 2048, 512),
         (N4, 1024, 2048), (N3, 1024, 1024), (N2, 512, 512)]


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


print("%-22s %9s %9s   %9s %9s" % ("rows Cin->Cout", "lib_bwd", "own_bwd", "lib_fwd", "own_fwd"))
tl = to = 0.0
for N, Cin, Cout in CASES:
    x = torch.randn(N, Cin, device=dev); W = torch.randn(Cout, Cin, device=dev)
    out = torch.randn(N, Cout, device=dev); go = torch.randn(N, Cout, device=dev)
    b = torch.randn(Cout, device=dev)

    def lib_bwd():
        gm = torch.empty_like(go); gb = torch.zeros(Cout, device=dev)
        _native.check(L.d3f_bias_act_backward(go.data_ptr(), out.data_ptr(), 0.1, N, Cout, gm.data_ptr(), gb.data_ptr(), None, 1, None,
                                              None, 0, torch.cuda.current_stream().cuda_stream), "x")
        return torch.mm(gm, W), torch.mm(gm.t(), x)

    def own_bwd():
        gb = torch.empty(Cout, device=dev)
        gx = ops.gemm(go, W, b_ks=True, a_mask=out, mask_slope=0.1)
        gw = ops.gemm(go, x, a_ks=True, b_ks=True, a_mask=out, mask_slope=0.1, rowsum=gb)
        return gx, gw

    lib_fwd = lambda: ops.bias_act(torch.mm(x, W.t()), b, slope=0.1)   # noqa: E731
    own_fwd = lambda: ops.gemm(x, W, bias1=b, slope=0.1)               # noqa: E731
    a, c, d, e = graph_time(lib_bwd), graph_time(own_bwd), graph_time(lib_fwd), graph_time(own_fwd)
    tl += a; to += c
    print("%5d %5d->%-5d     %9.2f %9.2f   %9.2f %9.2f" % (N, Cin, Cout, a, c, d, e))
print("sum bwd: lib %.1f own %.1f" % (tl, to))
