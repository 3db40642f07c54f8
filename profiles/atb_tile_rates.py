"""Round 6: steady-state rate of the grouped weight-gradient kernel per TILE SHAPE -- one large problem (or a few copies)
of each kind, alone in a grouped launch, long enough (>= 30 GFLOP) that ramp-up and tail do not matter.
    python profiles/atb_tile_rates.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from d3feat_pytorch_amd import _native  # noqa: E402

L = _native.lib()
dev = torch.device("cuda:0")
PEAK = 157.3e12
CASES = [("64x64 tiles, split reduction   16384 x 1024 x 1024", [(16384, 1024, 1024)]),
         ("64x64 tiles, split reduction   6208 x 512 x 512 x8", [(6208, 512, 512)] * 8),
         ("64x64 tiles, direct            512 x 7680 x 512 x8", [(512, 7680, 512)] * 8),
         ("64x64 tiles, direct            1792 x 3840 x 256 x8", [(1792, 3840, 256)] * 8),
         ("32x64 tiles                    114624 x 32 x 384 x8", [(114624, 32, 384)] * 8),
         ("32x32 tiles                    114624 x 480 x 32 x8", [(114624, 480, 32)] * 8),
         ("64x32 tiles                    114624 x 128 x 32 x16", [(114624, 128, 32)] * 16),
         ("16x64 tiles                    114624 x 16 x 64 x32", [(114624, 16, 64)] * 32),
         ("64x64 tiles, 2 blocks          114624 x 128 x 64 x8", [(114624, 128, 64)] * 8),
         ("64x64 tiles, 15 blocks         23808 x 960 x 64 x8", [(23808, 960, 64)] * 8)]
pipe = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_native.set_tunables(atb_pipe=pipe)
print("# tunables().atb_pipe = %d (%s task body on the wide tiles)" % (pipe, "plain" if pipe else "software-pipelined"))
for label, shapes in CASES:
    ps = []
    g = torch.Generator(device=dev).manual_seed(1)
    cache = {}
    for (R, M, N) in shapes:
        if (R, M, N) not in cache:      # copies share their operands (the rate, not the footprint, is the question)
            cache[(R, M, N)] = (torch.randn(R, M, device=dev, generator=g), torch.randn(R, N, device=dev, generator=g))
        A, B = cache[(R, M, N)]
        ps.append((A, B, torch.empty(M, N, device=dev)))
    arr = (_native.AtbProblem * len(ps))()
    for q, (a, b, c) in zip(arr, ps):
        q.x, q.grad_out, q.grad_w = b.data_ptr(), a.data_ptr(), c.data_ptr()
        q.N, q.Cin, q.Cout, q.ldw = a.shape[0], b.shape[1], a.shape[1], b.shape[1]
    nb = L.d3f_linear_grad_weight_group_ws_bytes(arr, len(ps))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def fn():
        _native.check(L.d3f_linear_grad_weight_group(arr, len(ps), ws.data_ptr(), nb,
                                                     torch.cuda.current_stream().cuda_stream), "group")
    fn()
    torch.cuda.synchronize()
    a, b, c = ps[-1]
    err = float((c.double() - a.double().t() @ b.double()).abs().max() / (a.shape[0] ** 0.5))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    fl = sum(2.0 * R * M * N for R, M, N in shapes)
    by = sum(4.0 * R * (M + N) for R, M, N in shapes)
    print("%-52s %8.1f us %6.1f TFLOP/s = %.3f of peak   %5.2f TB/s operand stream   slabs %6.1f MB  err %.1e" % (
        label, best * 1e3, fl / best / 1e9, fl / best / 1e9 / (PEAK / 1e12), by / best / 1e9, nb / 1e6, err))
