"""Round 6: which side of the grouped weight-gradient kernel's loop sets its rate?  The same large problems through
MEASUREMENT builds of the library (csrc/linear.hip compiled with -DD3F_ATB_PROBE=1: no LDS-DMA, =2: no MFMAs, =3:
neither -- results are garbage) next to the product build.  Build them first (CPU container or GPU box):
    python -c "from d3feat_pytorch_amd import _native as n; [n.build(extra_flags=['-DD3F_ATB_PROBE=%d' % p],
               out='profiles/experiments/libd3f_probe%d.so' % p, objdir='/tmp/d3f_probe%d' % p) for p in (1, 2, 3)]"
    python profiles/atb_loop_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from d3feat_pytorch_amd import _native  # noqa: E402

dev = torch.device("cuda:0")
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("64x64 split  16384 x 1024 x 1024", [(16384, 1024, 1024)]),
         ("64x64 direct 512 x 7680 x 512 x8", [(512, 7680, 512)] * 8),
         ("32x64        114624 x 32 x 384 x8", [(114624, 32, 384)] * 8),
         ("32x32        114624 x 480 x 32 x8", [(114624, 480, 32)] * 8)]
libs = [("product", _native.lib())]
for p, what in ((1, "no LDS-DMA"), (2, "no MFMA"), (3, "neither")):
    path = os.path.join(HERE, "experiments", "libd3f_probe%d.so" % p)
    if os.path.exists(path):
        h = C.CDLL(path)
        for name in ("d3f_linear_grad_weight_group_ws_bytes", "d3f_linear_grad_weight_group"):
            res, args = _native.SIGNATURES[name]
            getattr(h, name).restype, getattr(h, name).argtypes = res, args
        libs.append((what, h))
for label, shapes in CASES:
    g = torch.Generator(device=dev).manual_seed(1)
    R, M, N = shapes[0]
    A, B = torch.randn(R, M, device=dev, generator=g), torch.randn(R, N, device=dev, generator=g)
    ps = [(A, B, torch.empty(M, N, device=dev)) for _ in shapes]
    arr = (_native.AtbProblem * len(ps))()
    for q, (a, b, c) in zip(arr, ps):
        q.x, q.grad_out, q.grad_w = b.data_ptr(), a.data_ptr(), c.data_ptr()
        q.N, q.Cin, q.Cout, q.ldw = a.shape[0], b.shape[1], a.shape[1], b.shape[1]
    fl = sum(2.0 * r * m * n for r, m, n in shapes)
    row = []
    for what, L in libs:
        nb = L.d3f_linear_grad_weight_group_ws_bytes(arr, len(ps))
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        L.d3f_linear_grad_weight_group(arr, len(ps), ws.data_ptr(), nb, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(5):
                L.d3f_linear_grad_weight_group(arr, len(ps), ws.data_ptr(), nb, st)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        row.append("%s %7.1f us (%.2f)" % (what, best * 1e3, fl / best / 1e9 / 157.3))
    print("%-36s  %s" % (label, "   ".join(row)))
print("(in brackets: 2 R M N / time over the f32 MFMA peak -- meaningful for the product and the no-DMA build only)")
