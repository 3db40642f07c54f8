"""Round 6: the GROUPED weight-gradient launch (csrc/linear.hip: atb_grouped_kernel + atb_grouped_reduce_kernel) on the
28 problems of a 3-pair stack's training step -- all of them in one launch pair, against one launch pair per problem
with the same kernels and against the first (direct-load) form one problem at a time (rounds 1-5's schedule).

    python profiles/atb_group_bench.py [task_us ...]

Every variant: results checked against float64, bit-reproducibility checked, then the whole set captured in a hipGraph
and replayed (HIP events around 10 replays).  The library's tunables go through d3f_set_tunables (no environment)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from d3feat_pytorch_amd import _native  # noqa: E402

# (rows R, M = Cout, N = Cin) of C [M, N] = A^T B as bench.py's per-launch table lists them for --lanes 4 --stack 3
SHAPES = [(114688, 32, 384), (23872, 256, 256), (6208, 512, 512), (6208, 128, 512), (6208, 512, 128), (6208, 512, 256),
          (6208, 128, 256), (6208, 256, 64), (6208, 960, 64), (23872, 64, 256), (23872, 256, 64), (23872, 960, 64),
          (23872, 256, 128), (23872, 64, 128), (23872, 128, 32), (23872, 480, 32), (114688, 32, 128), (114688, 128, 32),
          (114688, 480, 32), (114688, 128, 64), (114688, 32, 64), (114688, 16, 64), (6208, 1920, 128)]
COUNT = {(6208, 128, 512): 2, (6208, 512, 128): 2, (23872, 64, 256): 2, (23872, 256, 64): 2, (23872, 960, 64): 2}
PEAK = 157.3e12

L = _native.lib()
dev = torch.device("cuda:0")


def shapes_of_a_step(Q=3):
    """(R, M, N) of every problem a 3-pair stack's training step queues (ops.WeightGradGroup), in queue order -- the 28
    many-row problems above plus the few-row weight gradients of the bottom levels that join the group."""
    import numpy as np
    from d3feat_pytorch_amd import config as cfgmod, ops, synthetic
    from d3feat_pytorch_amd.datasets import dataloader as dl
    from d3feat_pytorch_amd.train import TrainStep
    cfg = cfgmod.default_config()

    def sub(p, l, d):
        a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
        return a.cpu().numpy(), b.cpu().numpy()
    stack = tuple(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in synthetic.make_pair(2 * q + 1, 2 * q + 2, sub))
                  for q in range(Q))
    ts = TrainStep(cfg, [42] * 5, dev, seed=0)
    sizes = [[sum(int(ts.build_batch(it)['points'][l].shape[0]) for it in stack) for l in range(5)]]
    ts.enable_graph(TrainStep.capacities_for(sizes, slack=1.0), num_corr=int(stack[0][4].shape[0]), stack=Q)
    seen = []
    flush = ops.WeightGradGroup.flush

    def spy(self):
        seen.append([(t[3], t[5], t[4]) for t in self.problems])    # (N, Cout, Cin) = (R, M, N)
        return flush(self)
    ops.WeightGradGroup.flush = spy
    try:
        ts._static_step(stack)
        torch.cuda.synchronize()
    finally:
        ops.WeightGradGroup.flush = flush
    return seen[-1]


probs = []
g = torch.Generator(device=dev).manual_seed(1)
ALL = []
if "--step" in sys.argv:
    sys.argv.remove("--step")
    ALL = shapes_of_a_step()
    print("shapes of a 3-pair stack's step:", ALL)
else:
    for shp in SHAPES:
        ALL += [shp] * COUNT.get(shp, 1)
for shp in ALL:
    for _ in range(1):
        R, M, N = shp
        A = torch.randn(R, M, device=dev, generator=g)      # grad_out [R, Cout = M]
        B = torch.randn(R, N, device=dev, generator=g)      # x [R, Cin = N]
        probs.append((A, B, torch.empty(M, N, device=dev)))
flops = sum(2.0 * a.shape[0] * a.shape[1] * b.shape[1] for a, b, _ in probs)
byts = sum(4.0 * a.shape[0] * (a.shape[1] + b.shape[1]) + 4.0 * a.shape[1] * b.shape[1] for a, b, _ in probs)
refs = [(a.double().t() @ b.double()) for a, b, _ in probs]


def descr(ps):
    arr = (_native.AtbProblem * len(ps))()
    for q, (a, b, c) in zip(arr, ps):
        q.x, q.grad_out, q.grad_w = b.data_ptr(), a.data_ptr(), c.data_ptr()
        q.N, q.Cin, q.Cout, q.ldw = a.shape[0], b.shape[1], a.shape[1], b.shape[1]
    return arr


def grouped(ps):
    arr = descr(ps)
    nb = L.d3f_linear_grad_weight_group_ws_bytes(arr, len(ps))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def fn():
        st = torch.cuda.current_stream().cuda_stream
        _native.check(L.d3f_linear_grad_weight_group(arr, len(ps), ws.data_ptr(), nb, st), "group")
    return fn, (arr, ws), nb


def one_by_one(ps):
    fns = [grouped([p]) for p in ps]

    def fn():
        for f, _, _ in fns:
            f()
    return fn, fns, sum(n for _, _, n in fns)


def single_api(ps):
    wss = []
    for a, b, c in ps:
        nb = L.d3f_linear_grad_weight_ws_bytes(a.shape[0], b.shape[1], a.shape[1])
        wss.append((torch.empty(nb, dtype=torch.uint8, device=dev), nb))

    def fn():
        st = torch.cuda.current_stream().cuda_stream
        for (a, b, c), (ws, nb) in zip(ps, wss):
            _native.check(L.d3f_linear_grad_weight(b.data_ptr(), a.data_ptr(), a.shape[0], b.shape[1], a.shape[1],
                                                   c.data_ptr(), ws.data_ptr(), nb, st), "single")
    return fn, wss, sum(n for _, n in wss)


def check():
    worst = 0.0
    for (a, b, c), r in zip(probs, refs):
        worst = max(worst, float((c.double() - r).abs().max() / r.abs().max()))
    return worst


def timed(fn, label, ws_bytes):
    for _, _, c in probs:
        c.fill_(float("nan"))
    fn()
    torch.cuda.synchronize()
    err = check()
    first = [c.clone() for _, _, c in probs]
    fn()
    torch.cuda.synchronize()
    same = all(torch.equal(c, f) for (_, _, c), f in zip(probs, first))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record(s)
            for _ in range(10):
                gr.replay()
            e1.record(s)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
    print("%-44s %8.1f us  %6.1f TFLOP/s = %.3f of f32 MFMA peak  %5.2f TB/s  slabs %6.1f MB  err %.1e  %s" % (
        label, best * 1e3, flops / (best * 1e-3) / 1e12, flops / (best * 1e-3) / PEAK, byts / (best * 1e-3) / 1e12,
        ws_bytes / 1e6, err, "bit-reproducible" if same else "NOT REPRODUCIBLE"))
    return best


print("%d problems, %.2f GFLOP, %.1f MB algorithmic" % (len(probs), flops * 1e-9, byts * 1e-6))
us_list = [int(a) for a in sys.argv[1:]] or [0, 10, 15, 30, 40]
old = _native.set_tunables(atb_form=1)
fn, keep, nb = single_api(probs)
timed(fn, "first form, one launch pair per problem", nb)
_native.set_tunables(atb_form=0)
fn, keep, nb = single_api(probs)
timed(fn, "one-problem API (by size: first form / grouped)", nb)
for us in us_list:
    _native.set_tunables(atb_task_us=us)
    fn, keep, nb = one_by_one(probs)
    timed(fn, "grouped kernels, one problem per pair, us=%d" % us, nb)
    fn, keep, nb = grouped(probs)
    timed(fn, "ONE grouped launch pair, task_us=%d" % us, nb)
_native.set_tunables(**old)
