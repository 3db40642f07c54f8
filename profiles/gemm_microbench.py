"""csrc/gemm.hip (register-direct split-K form) against torch.mm (+ the separate epilogue launch it replaces) on the
GEMM shapes of the training step's levels 1-4 (static capacities 8000 / 2112 / 640 / 192 rows), hipGraph-replayed
(what the training step does).  Sweeps the decomposition (fragments per wave along M, reduction split) through
d3f_debug_set_gemm_plan and checks every configuration against torch.mm.
    python profiles/gemm_microbench.py [--quick] [--json out.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3feat_pytorch_amd as d3f
from d3feat_pytorch_amd import _native, ops

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
ap.add_argument("--json", default=None)
args = ap.parse_args()

d3f.enable_tuned_gemms()
dev = torch.device("cuda:0")
L1, L2, L3, L4 = 8000, 2112, 640, 192
# (M, N, K, a_ks, b_ks, label)
CASES = [
    # forward x W^T
    (L1, 256, 128, 0, 0, "L1 fwd 128->256"), (L1, 64, 256, 0, 0, "L1 fwd 256->64"), (L1, 256, 256, 0, 0, "L1 dec 256->256"),
    (L2, 128, 256, 0, 0, "L2 fwd 256->128"), (L2, 512, 128, 0, 0, "L2 fwd 128->512"), (L2, 512, 256, 0, 0, "L2 sc 256->512"),
    (L2, 128, 512, 0, 0, "L2 fwd 512->128"), (L2, 512, 512, 0, 0, "L2 dec 512->512"),
    (L3, 256, 512, 0, 0, "L3 fwd 512->256"), (L3, 1024, 256, 0, 0, "L3 fwd 256->1024"), (L3, 1024, 512, 0, 0, "L3 sc 512->1024"),
    (L3, 256, 1024, 0, 0, "L3 fwd 1024->256"), (L3, 1024, 1024, 0, 0, "L3 dec 1024->1024"), (L3, 512, 1024, 0, 0, "L3 dec-c 1024->512"),
    (L4, 512, 1024, 0, 0, "L4 fwd 1024->512"), (L4, 2048, 512, 0, 0, "L4 fwd 512->2048"), (L4, 2048, 1024, 0, 0, "L4 sc 1024->2048"),
    (L4, 512, 2048, 0, 0, "L4 fwd 2048->512"), (L4, 1024, 2048, 0, 0, "L4 dec-c 2048->1024"),
    # KPConv contraction wf W
    (L2, 128, 1920, 0, 1, "L2 kpconv wf.W 1920->128"), (L3, 256, 3840, 0, 1, "L3 kpconv wf.W 3840->256"),
    (L4, 512, 7680, 0, 1, "L4 kpconv wf.W 7680->512"), (L4, 256, 3840, 0, 1, "L4 strided wf.W 3840->256"),
    # grad_x = g W
    (L2, 256, 128, 0, 1, "L2 dx g[.,128].W[128,256]"), (L2, 128, 512, 0, 1, "L2 dx g[.,512].W[512,128]"),
    (L3, 512, 256, 0, 1, "L3 dx g[.,256].W[256,512]"), (L3, 256, 1024, 0, 1, "L3 dx g[.,1024].W[1024,256]"),
    (L4, 1024, 512, 0, 1, "L4 dx g[.,512].W[512,1024]"), (L4, 512, 2048, 0, 1, "L4 dx g[.,2048].W[2048,512]"),
    # KPConv gW = (g/nn) W^T
    (L2, 1920, 128, 0, 0, "L2 kpconv gW 128->1920"), (L3, 3840, 256, 0, 0, "L3 kpconv gW 256->3840"),
    (L4, 7680, 512, 0, 0, "L4 kpconv gW 512->7680"),
    # weight gradients g^T x
    (256, 512, L3, 1, 1, "L3 dW [256x512] over 640"), (1024, 256, L3, 1, 1, "L3 dW [1024x256] over 640"),
    (2048, 512, L4, 1, 1, "L4 dW [2048x512] over 192"), (512, 2048, L4, 1, 1, "L4 dW [512x2048] over 192"),
    (3840, 256, L3, 1, 1, "L3 kpconv dW [3840x256] over 640"), (7680, 512, L4, 1, 1, "L4 kpconv dW [7680x512] over 192"),
    (1920, 128, L2, 1, 1, "L2 kpconv dW [1920x128] over 2112"), (512, 128, L2, 1, 1, "L2 dW [512x128] over 2112"),
    (64, 256, L1, 1, 1, "L1 dW [64x256] over 8000"),
]
if args.quick:
    CASES = CASES[::4]


def graph_time(fn, reps=10, replays=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


lib = _native.lib()
rows = []
print("%-36s %8s %8s | %8s | %8s  %s" % ("case", "mm_us", "mm+epi", "own_heur", "own_best", "best plan (fa,fb,kw,split)   all: fa/fb/kw/split=us"))
for M, N, K, aks, bks, label in CASES:
    A = torch.randn((K, M) if aks else (M, K), device=dev)
    B = torch.randn((K, N) if bks else (N, K), device=dev)
    bias = torch.randn(N, device=dev)
    Am = A.t() if aks else A
    Bm = B if bks else B.t()
    ref = torch.mm(Am.double(), Bm.double())
    scale = float(ref.abs().max())
    mm = lambda: torch.mm(Am, Bm)   # noqa: E731
    mme = lambda: ops.bias_act(torch.mm(Am, Bm), bias, slope=0.1)   # noqa: E731
    t_mm, t_mme = graph_time(mm), graph_time(mme)
    own = lambda: ops.gemm(A, B, a_ks=bool(aks), b_ks=bool(bks), bias1=bias, slope=0.1)   # noqa: E731
    plain = lambda: ops.gemm(A, B, a_ks=bool(aks), b_ks=bool(bks))   # noqa: E731
    results = {}
    chunks = (K + 15) // 16
    fas = [0] if aks else [1, 2]
    fbs = [0] if bks else [4, 2]
    cfgs = []
    for fa in fas:
        for fb in fbs:
            for kw in (1, 2, 4, 8):
                if kw > 1 and chunks // kw < 2:
                    continue
                cfgs.append((fa, fb, kw, 1))
            for sp in (2, 4, 8, 16):
                if chunks // (8 * sp) >= 2:
                    cfgs.append((fa, fb, 8, sp))
    lib.d3f_debug_set_gemm_plan(0, 0, 0, 0)
    err = float((plain().double() - ref).abs().max()) / scale
    assert err < 2e-5, (label, "heuristic", err)
    t_heur = graph_time(own)
    for cfg in cfgs:
        lib.d3f_debug_set_gemm_plan(*cfg)
        err = float((plain().double() - ref).abs().max()) / scale
        assert err < 2e-5, (label, cfg, err)
        results[cfg] = graph_time(own)
    lib.d3f_debug_set_gemm_plan(0, 0, 0, 0)
    best = min(results, key=results.get)
    allr = " ".join("%d/%d/%d/%d=%.1f" % (k + (t,)) for k, t in sorted(results.items()))
    print("%-36s %8.2f %8.2f | %8.2f | %8.2f  %s   %s" % (label, t_mm, t_mme, t_heur, results[best], best, allr))
    sys.stdout.flush()
    rows.append({"label": label, "M": M, "N": N, "K": K, "a_ks": aks, "b_ks": bks, "mm_us": t_mm, "mm_epi_us": t_mme,
                 "own_heuristic_us": t_heur, "own_best_us": results[best], "best": list(best),
                 "sweep": {"%d/%d/%d/%d" % k: v for k, v in results.items()},
                 "tflops_best": 2.0 * M * N * K / results[best] / 1e6})
tot_lib = sum(r["mm_epi_us"] for r in rows)
tot_own = sum(r["own_best_us"] for r in rows)
tot_heur = sum(r["own_heuristic_us"] for r in rows)
print("sum over cases: library+epilogue %.1f us, own (heuristic plan) %.1f us, own (best plan) %.1f us" % (tot_lib, tot_heur, tot_own))
if args.json:
    with open(args.json, "w") as f:
        json.dump(rows, f, indent=1)
