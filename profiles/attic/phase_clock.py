"""Where do the waves of the fused KPConv forward / gather grad-input kernels spend their cycles?  The kernels stamp
s_memtime at their phase boundaries when d3f_debug_set_phase_clock is armed; this script runs the layers of the S1
pair one at a time and prints, per layer, the average shader cycles per wave and phase.
    python profiles/phase_clock.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from d3feat_pytorch_amd import _native, config as cfgmod, ops, synthetic
from d3feat_pytorch_amd.datasets import dataloader as dl

dev = torch.device("cuda:0")
cfg = cfgmod.default_config()
lib = _native.lib()


def sub(p, l, d):
    a, b = dl.batch_grid_subsampling_kpconv(torch.as_tensor(p).to(dev), torch.as_tensor(l).to(dev), sampleDl=d)
    return a.cpu().numpy(), b.cpu().numpy()


it = synthetic.make_pair(1, 2, sub)
batch = dl.collate_fn_descriptor([it], cfg, [42] * 5, device=dev, exact_width=False, reverse_tables=True)
pts, nb, pools = batch['points'], batch['neighbors'], batch['pools']
CAP = 1 << 16
clk = torch.zeros(8 + 8 * CAP, dtype=torch.int64, device=dev)
FWD = ["prologue", "phase A (4 queries)", "barrier wait", "wf_save", "phase B", "barrier + epilogue"]
GAT = ["prologue", "index + position loads", "rows (membership, gathers, MFMAs)", "barrier wait", "phase B", "store"]


def run(label, q, s, tab, cin, cout, extent, names_f=FWD, names_g=GAT):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand((s.shape[0], cin), device=dev, generator=g).requires_grad_(True)
    w = (torch.randn((15, cin, cout), device=dev, generator=g) * 0.05).requires_grad_(True)
    kp = torch.randn((15, 3), device=dev, generator=g) * extent * 0.5
    for which, names in (("fwd", names_f), ("gather", names_g)):
        res = None
        for rep in range(3):
            y = ops.kpconv(q, s, tab, x, kp, w, extent)
            go = torch.ones_like(y)
            torch.cuda.synchronize()
            if which == "fwd":
                clk.zero_(); clk[1] = CAP; lib.d3f_debug_set_phase_clock(clk.data_ptr())
                y2 = ops.kpconv(q, s, tab, x, kp, w, extent)
                torch.cuda.synchronize(); lib.d3f_debug_set_phase_clock(None)
            else:
                clk.zero_(); clk[1] = CAP; lib.d3f_debug_set_phase_clock(clk.data_ptr())
                y.backward(go)
                torch.cuda.synchronize(); lib.d3f_debug_set_phase_clock(None)
                x.grad = None; w.grad = None
            raw = clk.cpu().numpy()
            nrec = int(min(raw[0], CAP))
            res = raw[8:8 + 8 * nrec].reshape(nrec, 8)[:, :6].astype(np.float64)
        tot = res.sum(1).mean()
        print("%-28s %-7s waves %6d  cycles/wave %8.0f :: " % (label, which, nrec, tot) +
              "  ".join("%s %.0f" % (n, res[:, i].mean()) for i, n in enumerate(names)))


r = cfg.first_subsampling_dl * cfg.conv_radius
ext = lambda l: r * 2 ** l * cfg.KP_extent / cfg.conv_radius   # noqa: E731
run("enc1  L0 conv 32->32", pts[0], pts[0], nb[0], 32, 32, ext(0))
run("enc2  L0->L1 strided 32->32", pts[1], pts[0], pools[0], 32, 32, ext(0))
run("enc3  L1 conv 64->64", pts[1], pts[1], nb[1], 64, 64, ext(1))
run("enc5  L1->L2 strided 64->64", pts[2], pts[1], pools[1], 64, 64, ext(1))
run("enc6  L2 conv 128->128", pts[2], pts[2], nb[2], 128, 128, ext(2))
