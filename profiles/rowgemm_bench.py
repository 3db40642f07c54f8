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


RT_SWEEP = (1, 0)


def main():
    L = _native.lib()
    rng = np.random.default_rng(0)
    # (rows, Cin, Cout, residual)
    shapes = [(114624, 64, 32, False), (114624, 32, 128, True), (114624, 64, 128, False), (23808, 32, 128, True),
              (23808, 64, 256, True), (6208, 64, 256, True)]
    print("%-28s | fwd: rowgemm_kernel, rowgemm_rt_kernel (us GB/s) | dgrad: the same two settings   (GB/s of x + add + y)" % "shape")
    for (N, Cin, Cout, res) in shapes:
        x = torch.from_numpy(rng.normal(size=(N, Cin)).astype(np.float32)).cuda()
        w = torch.from_numpy(rng.normal(size=(Cout, Cin)).astype(np.float32)).cuda()
        b = torch.zeros(Cout, device="cuda")
        add = torch.from_numpy(rng.normal(size=(N, Cout)).astype(np.float32)).cuda() if res else None
        y = torch.empty((N, Cout), device="cuda")
        g = torch.from_numpy(rng.normal(size=(N, Cout)).astype(np.float32)).cuda()
        gx = torch.empty((N, Cin), device="cuda")
        ts = []
        for wide in RT_SWEEP:
            old = _native.set_tunables(rowgemm_rt=wide)
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
        print("%-28s | " % ("%d x %d -> %d%s" % (N, Cin, Cout, " + res" if res else "")) +
              "  ".join("%6.1f %4.0f" % (t[0], bf / t[0] * 1e-3) for t in ts) + " | " +
              "  ".join("%6.1f %4.0f" % (t[1], bd / t[1] * 1e-3) for t in ts))


def pair():
    """unary2 + shortcut unary as one launch (d3f_linear_pair_bias_act_forward) against the two launches it replaces."""
    L = _native.lib()
    rng = np.random.default_rng(1)
    print("%-34s %10s %10s" % ("pair shape", "two", "one"))
    for (N, K1, K2, M) in [(114624, 32, 64, 128), (23808, 32, 64, 128), (114624, 16, 32, 64)]:
        x1 = torch.from_numpy(rng.normal(size=(N, K1)).astype(np.float32)).cuda()
        x2 = torch.from_numpy(rng.normal(size=(N, K2)).astype(np.float32)).cuda()
        w1 = torch.from_numpy(rng.normal(size=(M, K1)).astype(np.float32)).cuda()
        w2 = torch.from_numpy(rng.normal(size=(M, K2)).astype(np.float32)).cuda()
        b = torch.from_numpy(rng.normal(size=(4, M)).astype(np.float32)).cuda()
        sc = torch.empty((N, M), device="cuda")
        y = torch.empty((N, M), device="cuda")
        y2 = torch.empty((N, M), device="cuda")

        def two():
            st = torch.cuda.current_stream().cuda_stream
            _native.check(L.d3f_linear_bias_act_forward(x2.data_ptr(), w2.data_ptr(), N, K2, M, b[2].data_ptr(), None,
                                                        b[3].data_ptr(), 1.0, sc.data_ptr(), None, 0, st), "sc")
            _native.check(L.d3f_linear_bias_act_forward(x1.data_ptr(), w1.data_ptr(), N, K1, M, b[0].data_ptr(), sc.data_ptr(),
                                                        b[1].data_ptr(), 0.1, y.data_ptr(), None, 0, st), "u2")

        def one():
            st = torch.cuda.current_stream().cuda_stream
            _native.check(L.d3f_linear_pair_bias_act_forward(x1.data_ptr(), w1.data_ptr(), K1, x2.data_ptr(), w2.data_ptr(), K2,
                                                             N, M, b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(),
                                                             b[3].data_ptr(), 0.1, y2.data_ptr(), None, 0, st), "pair")
        t2, t1 = graph_timed(two), graph_timed(one)
        two()
        one()
        err = float((y - y2).abs().max() / y.abs().max())
        print("%-34s %10.1f %10.1f   rel diff %.1e" % ("%d x (%d | %d) -> %d" % (N, K1, K2, M), t2, t1, err))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "pair":
    pair()
if __name__ == "__main__" and len(sys.argv) == 1:
    main()
