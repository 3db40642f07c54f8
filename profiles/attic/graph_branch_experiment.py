"""Do two independent small-grid kernels captured on two streams inside ONE hipGraph overlap on replay?
python profiles/graph_branch_experiment.py"""
import torch
dev = torch.device("cuda:0")
g = torch.randn(159, 512, device=dev)
W = torch.randn(512, 7680, device=dev)           # gwf = g @ W        [159, 7680]
wf = torch.randn(159, 7680, device=dev)          # dW  = wf^T @ g     [7680, 512]
o1 = torch.empty(159, 7680, device=dev)
o2 = torch.empty(7680, 512, device=dev)


def seq():
    torch.mm(g, W, out=o1)
    torch.mm(wf.t(), g, out=o2)


side = torch.cuda.Stream(device=dev)


def par():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        torch.mm(wf.t(), g, out=o2)
    torch.mm(g, W, out=o1)
    cur.wait_stream(side)


def bench(fn, reps=20):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


def only(which):
    if which == 0:
        return lambda: torch.mm(g, W, out=o1)
    return lambda: torch.mm(wf.t(), g, out=o2)


print("gemm A alone %.1f us, gemm B alone %.1f us" % (bench(only(0)), bench(only(1))))
print("sequential in one graph: %.1f us per pair" % bench(seq))
print("two streams in one graph: %.1f us per pair" % bench(par))
