"""Weight gradients C = A^T B of a stacked step: the reduction-parallel A^T B kernel (csrc/linear.hip) against the
library GEMM torch.mm(A.t(), B) with TunableOp tuning on (rocBLAS candidates), hipGraph-replayed:
    python profiles/dw_library_vs_atb.py [Q]        (Q pairs stacked: rows = Q x the one-pair level sizes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import _native

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 3
# (rows, M, N): C [M, N] = A [rows, M]^T B [rows, N]
SHAPES = [(38180 * Q, 480, 32), (7920 * Q, 960, 64), (7920 * Q, 480, 32), (2053 * Q, 1920, 128), (2053 * Q, 960, 64),
          (38180 * Q, 128, 64), (38180 * Q, 128, 32), (7920 * Q, 256, 128), (7920 * Q, 256, 64), (7920 * Q, 64, 256),
          (2053 * Q, 512, 128), (2053 * Q, 128, 512), (2053 * Q, 512, 256)]
d3f.enable_tuned_gemms(tune_missing=True)
torch.cuda.tunable.set_max_tuning_duration(30)
torch.cuda.tunable.set_max_tuning_iterations(100)
L = _native.lib()
dev = torch.device("cuda:0")


def timed(fn):
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
    return e0.elapsed_time(e1) * 1e3 / 50


tot_a = tot_l = tot_best = 0.0
for R, M, N in SHAPES:
    A = torch.randn(R, M, device=dev)
    B = torch.randn(R, N, device=dev)
    C = torch.empty(M, N, device=dev)
    nb = L.d3f_linear_grad_weight_ws_bytes(R, N, M)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    atb = lambda: L.d3f_linear_grad_weight(B.data_ptr(), A.data_ptr(), R, N, M, C.data_ptr(), ws.data_ptr(), nb,
                                           torch.cuda.current_stream().cuda_stream)   # noqa: E731
    C2 = torch.empty(M, N, device=dev)
    lib = lambda: torch.mm(A.t(), B, out=C2)   # noqa: E731
    ua, ul = timed(atb), timed(lib)
    err = (C - C2).abs().max().item() / max(1.0, C2.abs().max().item())
    fl = 2.0 * R * M * N
    tot_a += ua; tot_l += ul; tot_best += min(ua, ul)
    print("%7d x %4d x %4d   atb %7.2f us (%5.1f TF/s)   library %7.2f us (%5.1f TF/s)   relerr %.1e" % (
        R, M, N, ua, fl / ua / 1e6, ul, fl / ul / 1e6, err))
print("sum: atb %.1f us, library %.1f us, best of both %.1f us" % (tot_a, tot_l, tot_best))
