"""d3f_gemm_epilogue (csrc/gemm_epilogue.hip) against the library GEMM (+ the separate epilogue launch it needs) on the
shapes of a 3-pair stacked training step (static capacities 114624 / 23808 / 6208 / 1792 / 512 rows).

    python profiles/gemm_epilogue_bench.py [--reps 30] [--variants]

Per shape: library GEMM alone, library GEMM + d3f_bias_act_forward (forward shapes), own kernel on the built-in plan and
(--variants) with 32- / 64-row blocks x undivided / 8 / 16 partitions.  Times are CUDA-event averages over back-to-back
launches on one stream (operands warm in L2 / MALL for both sides)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from d3feat_pytorch_amd import _native, ops  # noqa: E402

NT, NN = ops.GEMM_NT, ops.GEMM_NN
# (label, R, K, N, mode, kblock, epilogue)   epilogue: 'f' = forward (row_div for KPConv + bias + LeakyReLU), 'a' = addend, '' = none
SHAPES = [
    # KPConv forward: wf [R, 15 Cin] @ W [15 Cin, Cout]
    ("kpconv fwd L1", 23808, 960, 64, NN, 0, 'fk'), ("kpconv fwd L2s", 6208, 960, 64, NN, 0, 'fk'),
    ("kpconv fwd L2", 6208, 1920, 128, NN, 0, 'fk'), ("kpconv fwd L3s", 1792, 1920, 128, NN, 0, 'fk'),
    ("kpconv fwd L3", 1792, 3840, 256, NN, 0, 'fk'), ("kpconv fwd L4s", 512, 3840, 256, NN, 0, 'fk'),
    ("kpconv fwd L4", 512, 7680, 512, NN, 0, 'fk'),
    # unary forward: x @ W^T
    ("unary L0 128>32", 114624, 128, 32, NT, 0, 'f'), ("unary L1 128>64", 23808, 128, 64, NT, 0, 'f'),
    ("unary L1 128>256", 23808, 128, 256, NT, 0, 'f'), ("unary L1 256>64", 23808, 256, 64, NT, 0, 'f'),
    ("unary L2 256>128", 6208, 256, 128, NT, 0, 'f'), ("unary L2 256>512", 6208, 256, 512, NT, 0, 'f'),
    ("unary L2 128>512", 6208, 128, 512, NT, 0, 'f'), ("unary L2 512>128", 6208, 512, 128, NT, 0, 'f'),
    ("unary L3 128>512", 1792, 128, 512, NT, 0, 'f'), ("unary L3 512>256", 1792, 512, 256, NT, 0, 'f'),
    ("unary L3 512>1024", 1792, 512, 1024, NT, 0, 'f'), ("unary L3 256>1024", 1792, 256, 1024, NT, 0, 'f'),
    ("unary L3 1024>256", 1792, 1024, 256, NT, 0, 'f'), ("unary L4 256>1024", 512, 256, 1024, NT, 0, 'f'),
    ("unary L4 1024>512", 512, 1024, 512, NT, 0, 'f'), ("unary L4 1024>2048", 512, 1024, 2048, NT, 0, 'f'),
    ("unary L4 512>2048", 512, 512, 2048, NT, 0, 'f'), ("unary L4 2048>512", 512, 2048, 512, NT, 0, 'f'),
    ("dec L4>3 coarse", 512, 2048, 1024, NT, 0, ''), ("dec L3 skip", 1792, 1024, 1024, NT, 0, ''),
    ("dec L3>2 coarse", 1792, 1024, 512, NT, 0, ''), ("dec L2 skip", 6208, 512, 512, NT, 0, ''),
    ("dec L2>1 coarse", 6208, 512, 256, NT, 0, ''), ("dec L1 skip", 23808, 256, 256, NT, 0, ''),
    ("last unary L0", 114624, 384, 32, NT, 0, 'f'),
    # grad-input of the unary blocks: g [R, Cout] @ W [Cout, Cin]
    ("dgrad L4 512<2048", 512, 2048, 512, NN, 0, 'a'), ("dgrad L4 2048<512", 512, 512, 2048, NN, 0, ''),
    ("dgrad L4 1024<2048", 512, 2048, 1024, NN, 0, 'a'), ("dgrad L3 256<1024", 1792, 1024, 256, NN, 0, 'a'),
    ("dgrad L3 1024<256", 1792, 256, 1024, NN, 0, ''), ("dgrad L3 512<1024", 1792, 1024, 512, NN, 0, 'a'),
    ("dgrad L2 128<512", 6208, 512, 128, NN, 0, 'a'), ("dgrad L2 512<128", 6208, 128, 512, NN, 0, ''),
    ("dgrad L2 256<512", 6208, 512, 256, NN, 0, 'a'), ("dgrad L1 128<256", 23808, 256, 128, NN, 0, 'a'),
    ("dgrad L1 256<64", 23808, 64, 256, NN, 0, ''),
    # KPConv grad-input, transposed aggregation: A [Ns, 15 Cout] @ W'
    ("kpconv dx L1", 23808, 960, 64, NT, 64, ''), ("kpconv dx L2", 6208, 1920, 128, NT, 128, ''),
    ("kpconv dx L3", 1792, 3840, 256, NT, 256, ''),
    # KPConv grad-input of the few-point layers: gW = (g / nn) [Nq, Cout] @ W^T -> [Nq, 15 Cin]
    ("kpconv gW L4", 512, 512, 7680, NT, 0, ''), ("kpconv gW L4s", 512, 256, 3840, NT, 0, ''),
]


def timed(fn, reps):
    """`reps` calls captured in one hipGraph and replayed: GPU time without the host's launch gaps."""
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--variants", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-table", action="store_true", help="library GEMMs on their default picks (no TunableOp table)")
    args = ap.parse_args()
    if not args.no_table:
        import d3feat_pytorch_amd
        print("# TunableOp table loaded:", d3feat_pytorch_amd.enable_tuned_gemms())
    L = _native.lib()
    rng = np.random.default_rng(3)
    lines = []
    variants = [(0, 0)] + ([(2, 1), (4, 1), (2, 8), (4, 8), (2, 16), (4, 16)] if args.variants else [])
    head = "%-20s %7s %5s %5s %2s | %8s %8s | " % ("shape", "R", "K", "N", "m", "lib", "lib+epi") + " ".join(
        "%9s" % ("own" if v == (0, 0) else "r%d/s%d" % v) for v in variants) + " |  TF(own)  best"
    print(head)
    lines.append(head)
    tot_lib = tot_libepi = tot_own = 0.0
    for label, R, K, N, mode, kblock, epi in SHAPES:
        x = torch.from_numpy(rng.normal(size=(R, K)).astype(np.float32)).cuda()
        if kblock:
            Kp = K // kblock
            W3 = torch.from_numpy(rng.normal(size=(Kp, N, kblock)).astype(np.float32)).cuda()   # [k][c][o]
            w = W3
            b_lib = W3.permute(0, 2, 1).reshape(K, N).contiguous()
            lib_mm = lambda: torch.mm(x, W3.permute(0, 2, 1).contiguous().view(K, N))   # (what the step does today: copy + mm)
        elif mode == NT:
            w = torch.from_numpy(rng.normal(size=(N, K)).astype(np.float32)).cuda()
            b_lib = w.t()
            lib_mm = lambda: torch.mm(x, w.t())
        else:
            w = torch.from_numpy(rng.normal(size=(K, N)).astype(np.float32)).cuda()
            b_lib = w
            lib_mm = lambda: torch.mm(x, w)
        row_div = torch.from_numpy(rng.integers(1, 40, size=R).astype(np.float32)).cuda() if 'k' in epi else None
        bias = torch.from_numpy(rng.normal(size=N).astype(np.float32)).cuda() if 'f' in epi else None
        addend = torch.from_numpy(rng.normal(size=(R, N)).astype(np.float32)).cuda() if 'a' in epi else None
        slope = 0.1 if 'f' in epi else 1.0
        out = torch.empty((R, N), device="cuda")
        def lib_epi():
            if addend is not None:
                return torch.addmm(addend, x, b_lib)
            raw = lib_mm()
            if 'f' in epi:
                _native.check(L.d3f_bias_act_forward(raw.data_ptr(), bias.data_ptr(), None, None, slope, R, N, out.data_ptr(), None,
                                                     0, row_div.data_ptr() if row_div is not None else None, None, 0, 0,
                                                     torch.cuda.current_stream().cuda_stream), "ba")
            return raw

        def own():
            ops.gemm_epilogue(x, w, mode, R, K, N, kblock, None, None, row_div, bias, addend, None, slope, None, out)

        t_lib = timed(lib_mm, args.reps)
        t_libepi = timed(lib_epi, args.reps)
        ts = []
        for v in variants:
            old = _native.set_tunables(xw_rows=v[0], xw_split=v[1])
            try:
                ts.append(timed(own, args.reps))
            finally:
                _native.set_tunables(**old)
        # check
        ref = x.double() @ b_lib.double()
        if row_div is not None:
            ref = ref / row_div.double()[:, None]
        if bias is not None:
            ref = ref + bias.double()
        if addend is not None:
            ref = ref + addend.double()
        ref = torch.where(ref > 0, ref, ref * slope)
        own()
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        flop = 2.0 * R * K * N
        best = int(np.argmin(ts))
        line = "%-20s %7d %5d %5d %2d | %8.1f %8.1f | " % (label, R, K, N, mode, t_lib, t_libepi) + " ".join(
            "%9.1f" % t for t in ts) + " | %6.1f  %s  err %.1e" % (flop / ts[0] * 1e-6, "own" if best == 0 else "r%d/s%d" % variants[best], err)
        print(line, flush=True)
        lines.append(line)
        tot_lib += t_lib
        tot_libepi += t_libepi
        tot_own += ts[0]
    tail = "# totals (each shape once): library %.1f us, library + epilogue %.1f us, own %.1f us" % (tot_lib, tot_libepi, tot_own)
    print(tail)
    lines.append(tail)
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
