"""Row-streaming unary kernels (linear.hip: rowgemm_kernel) on the level-0 / level-1 shapes of a 3-pair stack, column
dealing A/B (tunables().rowgemm_wide).  Times: `reps` launches captured in one hipGraph, replayed (no host gaps)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3feat_pytorch_amd import _native  # noqa: E402


def graph_timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


def main():
    L = _native.lib()
    rng = np.random.default_rng(0)
    # (rows, Cin, Cout, residual)
    shapes = [(114624, 64, 32, False), (114624, 32, 128, True), (114624, 64, 128, False), (23808, 32, 128, True),
              (23808, 64, 256, True), (6208, 64, 256, True)]
    print("%-28s %10s %10s | %10s %10s   (us; GB/s of x + add + y)" % ("shape", "fwd", "fwd wide", "dgrad", "dgrad wide"))
    for (N, Cin, Cout, res) in shapes:
        x = torch.from_numpy(rng.normal(size=(N, Cin)).astype(np.float32)).cuda()
        w = torch.from_numpy(rng.normal(size=(Cout, Cin)).astype(np.float32)).cuda()
        b = torch.zeros(Cout, device="cuda")
        add = torch.from_numpy(rng.normal(size=(N, Cout)).astype(np.float32)).cuda() if res else None
        y = torch.empty((N, Cout), device="cuda")
        g = torch.from_numpy(rng.normal(size=(N, Cout)).astype(np.float32)).cuda()
        gx = torch.empty((N, Cin), device="cuda")
        ts = []
        for wide in (1, 2):
            old = _native.set_tunables(rowgemm_wide=wide)
            try:
                def fwd():
                    st = torch.cuda.current_stream().cuda_stream
                    _native.check(L.d3f_linear_bias_act_forward(x.data_ptr(), w.data_ptr(), N, Cin, Cout, b.data_ptr(),
                                                                add.data_ptr() if add is not None else None, None, 0.1,
                                                                y.data_ptr(), None, 0, st), "fwd")

                def dgrad():
                    st = torch.cuda.current_stream().cuda_stream
                    _native.check(L.d3f_linear_grad_input(g.data_ptr(), w.data_ptr(), N, Cin, Cout, None, gx.data_ptr(), st), "dx")
                ts.append((graph_timed(fwd), graph_timed(dgrad)))
            finally:
                _native.set_tunables(**old)
        bf = 4.0 * N * (Cin + Cout * (2 if res else 1))
        bd = 4.0 * N * (Cin + Cout)
        print("%-28s %6.1f %4.0f %6.1f %4.0f | %6.1f %4.0f %6.1f %4.0f" % (
            "%d x %d -> %d%s" % (N, Cin, Cout, " + res" if res else ""), ts[0][0], bf / ts[0][0] * 1e-3, ts[1][0],
            bf / ts[1][0] * 1e-3, ts[0][1], bd / ts[0][1] * 1e-3, ts[1][1], bd / ts[1][1] * 1e-3))


if __name__ == "__main__":
    main()
